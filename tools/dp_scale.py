#!/usr/bin/env python3
"""DP reduce on random 1080p x 256 volumes for a range of batch sizes (HIP-event times):
how its throughput follows the number of resident workgroups.  python tools/dp_scale.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import stereomatch
dev = torch.device("cuda:0")
H, W, D = 1080, 1920, 256
dp = stereomatch.disparity_reduce.DynamicProgramming()
for B in (1, 2, 4, 6, 8, 12, 16):
    S = torch.rand((B, H, W, D), device=dev) * 50
    d = dp(S)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dp(S, d); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("B=%d  %.3f ms  %.3f ms/pair" % (B, np.median(ts), np.median(ts) / B), flush=True)
    del S
