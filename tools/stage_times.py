#!/usr/bin/env python3
"""Per-op timings on one GPU (HIP events), for DESIGN.md's tables.  Not part of bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np
import torch
import stereomatch
from bench import synthetic_pair

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return float(np.median(out))


for name, H, W, D, B in (("teddy", 375, 450, 128, 16), ("teddy", 375, 450, 128, 1),
                         ("1080p", 1080, 1920, 256, 8), ("1080p", 1080, 1920, 256, 2),
                         ("1080p", 1080, 1920, 256, 1)):
    pairs = [synthetic_pair(H, W, s) for s in range(B)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    Lf, Rf = L / 255.0, R / 255.0
    N = H * W * D * B
    res = {}
    ssd_auto = stereomatch.cost.SSD(D)
    ssd_ord = stereomatch.cost.SSD(D, mode="ordered")
    bt = stereomatch.cost.Birchfield(D)
    sgm = stereomatch.aggregation.Semiglobal()
    sgm1 = stereomatch.aggregation.Semiglobal(single_passes=True)
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    C = ssd_auto(L, R)
    res["ssd_auto_int"] = timeit(lambda: ssd_auto(L, R, cost_volume=C))
    res["ssd_auto_float(ordered)"] = timeit(lambda: ssd_auto(Lf, Rf, cost_volume=C), reps=2)
    C = ssd_auto(L, R, cost_volume=C)
    Bv = bt(L, R)
    res["birchfield"] = timeit(lambda: bt(L, R, cost_volume=Bv))
    S = sgm(C, L)
    res["sgm_paired"] = timeit(lambda: sgm(C, L, S))
    res["sgm_single_passes"] = timeit(lambda: sgm1(C, L, S), reps=3)
    S = sgm(C, L, S)
    d = wta(S)
    res["wta"] = timeit(lambda: wta(S, d))
    res["sgm_wta_fused"] = timeit(lambda: sgm.aggregate_and_reduce(C, L, S, d))
    S = sgm(C, L, S)
    res["dp"] = timeit(lambda: dp(S, d))
    gb = lambda k: k * N * 4 / 1e9
    print("%s %dx%dx%d B=%d  (one volume = %.2f GB)" % (name, H, W, D, B, gb(1)))
    for k, v in res.items():
        print("   %-26s %9.3f ms  %8.3f ms/pair" % (k, v, v / B))
