import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import stereomatch
dev = torch.device("cuda:0")
B = int(sys.argv[1])
S = torch.rand((B, 1080, 1920, 256), device=dev) * 50
dp = stereomatch.disparity_reduce.DynamicProgramming()
d = dp(S)
for _ in range(3): dp(S, d)
torch.cuda.synchronize()
