#!/usr/bin/env python3
"""DP reduce on random volumes for a range of batch sizes (HIP-event times), with the forward
pass's two wavefront forms side by side (STM_DP_LANES=16: a DPP row of 16 lanes per image row;
64: the whole wave): how its throughput follows the number of resident waves.
python tools/dp_scale.py [1080p|teddy|HxWxD[:batch,batch,...]]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import stereomatch
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
if "x" in which:  # HxWxD[:b1,b2,...]
    shape, _, bl = which.partition(":")
    H, W, D = (int(v) for v in shape.split("x"))
    batches = tuple(int(v) for v in bl.split(",")) if bl else (1, 2, 4, 8)
else:
    H, W, D = (1080, 1920, 256) if which == "1080p" else (375, 450, 128)
    batches = (1, 2, 3, 4, 6, 8, 12, 16) if which == "1080p" else (1, 2, 4, 8, 16, 32, 64)
dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected" if D > H else "reference")
for B in batches:
    S = torch.rand((B, H, W, D), device=dev) * 50
    line = "B=%-3d" % B
    for lanes in ("16", "64", None):
        if lanes is None:
            os.environ.pop("STM_DP_LANES", None)
        else:
            os.environ["STM_DP_LANES"] = lanes
        d = dp(S)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dp(S, d); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        line += "  %s: %.3f ms (%.3f /pair)" % (lanes or "auto", np.median(ts), np.median(ts) / B)
    print(line, flush=True)
    del S
