import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np, torch, stereomatch
from bench import synthetic_pair
dev = torch.device("cuda:0")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(b))
    return float(np.median(out))
for H, W, D, B in ((375, 450, 128, 16), (375, 450, 256, 16), (375, 1920, 128, 4), (1080, 1920, 128, 2), (1080, 1920, 256, 2)):
    pairs = [synthetic_pair(H, W, s) for s in range(B)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    C = torch.empty(B, H, W, D, device=dev)
    for mode, imgs in (("ordered", (L, R)), ("auto", (L / 255, R / 255))):
        op = stereomatch.cost.SSD(D, mode=mode)
        t = timeit(lambda: op(imgs[0], imgs[1], cost_volume=C))
        print("%dx%dx%d B=%d %-8s %8.3f ms  %.3f ns/output" % (H, W, D, B, mode, t, t * 1e6 / (B * H * W * D)))
