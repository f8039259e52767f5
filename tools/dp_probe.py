import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np, torch, stereomatch
from bench import synthetic_pair
dev = torch.device("cuda:0")
for H, W, D, B in ((375, 450, 128, 1), (375, 450, 128, 16), (1080, 1920, 256, 1), (1080, 1920, 256, 2)):
    pairs = [synthetic_pair(H, W, s) for s in range(B)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    C = stereomatch.cost.Birchfield(D)(L, R)
    S = stereomatch.aggregation.Semiglobal()(C, L)
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    for _ in range(2):
        d = dp(S)
    torch.cuda.synchronize()
    print(H, W, D, B, float((d == 7).float().mean()))
