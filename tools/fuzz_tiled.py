#!/usr/bin/env python3
"""Long run of tests/test_gpu_sgm_tiled.py::test_tiled_fuzz_against_paired's comparison (tiled
against paired form, bit for bit) over many seeds.  N seeds from SEED (environment).  Needs a GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_sgm_tiled as T
dev = torch.device("cuda:0")
n, seed0 = int(os.environ.get("N", "40")), int(os.environ.get("SEED", "1000"))
t0 = time.time(); bad = 0
for seed in range(seed0, seed0 + n):
    for form in ("tiled", "walks"):
        try:
            T.test_tiled_fuzz_against_paired(dev, seed, form)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, form, str(e)[:300], flush=True)
print("seeds %d..%d (25 cases each, two forms) in %.0f s, failures: %d" % (seed0, seed0 + n - 1, time.time() - t0, bad))
