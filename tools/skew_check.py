#!/usr/bin/env python3
"""Development check of the skewed-tile form (Semiglobal(form="skew")) against the paired form,
bit for bit, on a list of shapes; prints where the first mismatches are.  Needs a GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np
import torch
import stereomatch
dev = torch.device("cuda:0")
shapes = [(8, 8, 1), (16, 16, 1), (24, 40, 1), (40, 24, 1), (40, 90, 2), (90, 40, 2), (33, 129, 1), (1, 9, 1), (9, 1, 1),
          (13, 30, 3), (95, 151, 2), (120, 200, 4), (375, 450, 2)]
D = int(os.environ.get("D", "128"))
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
bad = 0
for H, W, B in shapes:
    rng = np.random.default_rng(H * 1000 + W)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(dev)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(dev)
    ref_op = stereomatch.aggregation.Semiglobal(form="paired")
    ref = ref_op(C, I).clone()
    rdisp = ref_op.aggregate_and_reduce(C, I)[1].clone()
    op = stereomatch.aggregation.Semiglobal(form="skew")
    t0 = time.time()
    got = op(C, I)
    torch.cuda.synchronize()
    st = op.status()
    same = torch.equal(got.view(torch.int32), ref.view(torch.int32))
    disp = op.aggregate_and_reduce(C, I)[1]
    torch.cuda.synchronize()
    same_d = torch.equal(disp, rdisp)
    vol2, disp2 = op.aggregate_and_reduce(C, I, keep_sum=True)
    same_k = torch.equal(vol2.view(torch.int32), ref.view(torch.int32)) and torch.equal(disp2, rdisp)
    print("%4d x %4d x %d B=%d form=%d status=%d sum %s disp %s keep %s  (%.2f s)" % (
        H, W, D, B, op._last_form, st, "ok" if same else "MISMATCH", "ok" if same_d else "MISMATCH",
        "ok" if same_k else "MISMATCH", time.time() - t0), flush=True)
    if not same:
        bad += 1
        diff = (got.view(torch.int32) != ref.view(torch.int32)).any(dim=-1)  # [B,H,W]
        idx = diff.nonzero()
        print("   mismatching pixels: %d of %d; first: %s" % (idx.shape[0], diff.numel(), idx[:6].tolist()))
        if H <= 40 and W <= 48:
            for b in range(min(B, 1)):
                for y in range(H):
                    print("   " + "".join("X" if diff[b, y, x] else "." for x in range(W)))
        b, y, x = idx[0].tolist()
        print("   got", got[b, y, x, :6].tolist(), "\n   ref", ref[b, y, x, :6].tolist())
    elif not (same_d and same_k):
        bad += 1
print("failures:", bad)
sys.exit(1 if bad else 0)
