#!/usr/bin/env python3
"""Soak of the skewed-tile form's band hand-overs: many repetitions of shapes with many bands,
more units than workgroups, both disparity counts, against the paired form's result, bit for bit;
then the fuzz of tests/test_gpu_sgm_skew.py over many seeds.  N (repetitions per shape, default 60),
SEEDS (fuzz seeds, default 20).  Needs a GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import stereomatch
import test_gpu_sgm_skew as T
dev = torch.device("cuda:0")
n, seeds = int(os.environ.get("N", "60")), int(os.environ.get("SEEDS", "20"))
bad = 0
t0 = time.time()
for (H, W, D, B) in [(375, 450, 128, 24), (260, 90, 128, 64), (90, 900, 128, 16), (1080, 240, 128, 4), (200, 330, 256, 12), (64, 64, 128, 300)]:
    rng = np.random.default_rng(H + W + B)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(dev)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(dev)
    ref = stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1].clone()
    sgm = stereomatch.aggregation.Semiglobal(form="skew")
    fails = 0
    for rep in range(n):
        got = sgm.aggregate_and_reduce(C, I)[1]
        if sgm.status() != 0 or not torch.equal(got, ref):
            fails += 1
    bad += fails
    print("%4d x %4d x %d B=%-3d  %d runs, failures %d  (%.0f s)" % (H, W, D, B, n, fails, time.time() - t0), flush=True)
    del C, I, ref, sgm
    torch.cuda.empty_cache()
for seed in range(4000, 4000 + seeds):
    for fn in (T.test_skew_fuzz_against_paired, T.test_skew_d256_fuzz_against_paired):
        try:
            fn(dev, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, fn.__name__, str(e)[:300], flush=True)
print("fuzz seeds 4000..%d (25 + 20 cases each): done (%.0f s); failures in all: %d" % (4000 + seeds - 1, time.time() - t0, bad))
sys.exit(1 if bad else 0)
