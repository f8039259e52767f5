#!/usr/bin/env python3
"""Repeated runs of the sweep-fed tiled aggregation at D = 128 on shapes with several column
strips (the strips hand diagonal states to each other through mailboxes, a timing-dependent
path), each compared bit for bit with the paired form.  REPS (environment) runs per case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import torch, stereomatch
dev = torch.device("cuda:0")
reps = int(os.environ.get("REPS", "10"))
D = int(os.environ.get("D", "128"))
cases = [(40, 90, 1), (40, 90, 5), (375, 450, 1), (375, 450, 4), (64, 200, 3), (33, 129, 8), (100, 300, 2),
         (17, 65, 16), (200, 131, 3), (9, 1000, 2)]
if os.environ.get("BIG"):  # a full launch group: 8 strips x 32 images
    cases.append((375, 450, 32))
bad = 0; t0 = time.time()
for (H, W, B) in cases:
    g = torch.Generator().manual_seed(H * 1000 + W)
    C = (torch.rand((B, H, W, D), generator=g) * 50 + 10)
    C[..., D - 1] = torch.rand((B, H, W), generator=g)
    C = C.to(dev); img = torch.randint(0, 5, (B, H, W), generator=g).float().to(dev)
    ref = stereomatch.aggregation.Semiglobal(form="paired", penalty1=0.5, penalty2=3.0)(C, img)
    for r in range(reps):
        sg = stereomatch.aggregation.Semiglobal(form="tiled", penalty1=0.5, penalty2=3.0)
        out = sg(C, img)
        n = int((out.view(torch.int32) != ref.view(torch.int32)).sum())
        if n or sg.status():
            bad += 1
            print("FAIL %dx%dx%d B=%d rep %d: %d entries differ, status %d" % (H, W, D, B, r, n, sg.status()), flush=True)
print("%d cases x %d runs in %.0f s, failures: %d" % (len(cases), reps, time.time() - t0, bad))
sys.exit(1 if bad else 0)
