import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np, torch, stereomatch
from bench import synthetic_pair
dev = torch.device("cuda:0")
for H, W, D in ((375, 450, 128), (1080, 1920, 256)):
    Ln, Rn = synthetic_pair(H, W, 0)
    L = torch.from_numpy(Ln).to(dev); R = torch.from_numpy(Rn).to(dev)
    C = stereomatch.cost.Birchfield(D)(L, R)
    S = stereomatch.aggregation.Semiglobal()(C, L)
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    for _ in range(3):
        d = dp(S)
    torch.cuda.synchronize()
    print(H, W, D, float((d == 7).float().mean()))
