#!/usr/bin/env python3
"""Runs tests/test_gpu_fuzz.py::test_fuzz_all_ops over many more seeded shapes than the suite
does (N, SEED from the environment), plus shapes with paths long enough for full segments at
every D.  Development tool; needs a GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import test_gpu_fuzz as F
dev = torch.device("cuda:0")
shapes = F._shapes(int(os.environ.get("N", "300")), seed=int(os.environ.get("SEED", "777")))
# add shapes with longer paths so that full 16/32-pixel segments occur in every family
rng = np.random.default_rng(5)
for D in (64, 128, 256, 192, 512, 32):
    for _ in range(4):
        shapes.append((int(rng.integers(40, 130)), int(rng.integers(40, 160)), D))
t0 = time.time(); bad = 0
for i, (H, W, D) in enumerate(shapes):
    if H * W * D > 2500000: continue
    try:
        F.test_fuzz_all_ops(dev, oracle, H, W, D)
    except AssertionError as e:
        bad += 1
        print("FAIL", (H, W, D), str(e)[:200])
    except Exception as e:
        print("ERR", (H, W, D), repr(e)[:200]); bad += 1
print("checked %d shapes in %.0f s, failures: %d" % (len(shapes), time.time() - t0, bad))
