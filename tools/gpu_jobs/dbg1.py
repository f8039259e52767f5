import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-matching_amd"), os.path.join(ROOT, "tests")]
import oracle, stereomatch
from test_gpu_parity import _random_volume
dev = torch.device("cuda:0")
for (H, W, D) in [(9, 21, 256), (16, 16, 256), (8, 24, 256), (8, 17, 256), (35, 50, 256), (24, 65, 256)]:
    rng = np.random.default_rng(13 * H + 7 * W + D)
    C = _random_volume(rng, H, W, D, "float")
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    for segs in ("1", "2"):
        os.environ["STM_TILE_SEGS"] = segs
        sgm = stereomatch.aggregation.Semiglobal(0.5, 3.0, form="tiled")
        got = sgm(torch.from_numpy(C).to(dev), torch.from_numpy(image).to(dev)).cpu().numpy()
        bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
        ys, xs = np.nonzero(bad)
        print(H, W, D, "segs", segs, "status", sgm.status(), "bad pixels", bad.sum(), "cols", sorted(set(xs.tolist()))[:20], "rows", sorted(set(ys.tolist()))[:20], flush=True)
