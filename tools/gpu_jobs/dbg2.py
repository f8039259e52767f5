import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-matching_amd"), os.path.join(ROOT, "tests")]
import oracle, stereomatch
from test_gpu_parity import _random_volume
dev = torch.device("cuda:0")
H, W, D = 8, 17, 256
rng = np.random.default_rng(13 * H + 7 * W + D)
C = _random_volume(rng, H, W, D, "float")
image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
def geom(H, W, D, B):
    nTR, nTC = (H + 7) // 8, (W + 7) // 8
    strips = (W + 16 * (8 // (D // 64)) - 1) // (16 * (8 // (D // 64)))
    off = 64
    mbox = off; off += 2 * B * strips * 2 * 4 * D * 2
    flag = off; off += (B * nTR * 4 + 3) // 4 * 4
    rowb = off; rowb_img = 6 * nTR * W * D; off += rowb_img * B
    colb = off; colb_img = 6 * nTC * H * D
    return dict(nTR=nTR, nTC=nTC, flag=flag, rowb=rowb, colb=colb)
g = geom(H, W, D, 1)
def colb_line(ws, k, tc):
    f = ws.view(torch.float32)
    base = g["colb"] + ((k * g["nTC"] + tc) * H) * D
    return f[base:base + H * D].view(H, D).cpu().numpy()
res = {}
for form, segs in (("walks", "2"), ("tiled", "2"), ("tiled", "1")):
    os.environ["STM_TILE_SEGS"] = segs
    sgm = stereomatch.aggregation.Semiglobal(0.5, 3.0, form=form)
    got = sgm(torch.from_numpy(C).to(dev), torch.from_numpy(image).to(dev))
    torch.cuda.synchronize()
    ws = sgm._workspace
    res[(form, segs)] = (colb_line(ws, 0, 1).copy(), colb_line(ws, 0, 2).copy(), ws.view(torch.int32)[g["flag"]:g["flag"] + 4].cpu().numpy().copy(), ws.view(torch.int32)[:2].cpu().numpy().copy())
a = res[("walks", "2")][0]; b = res[("tiled", "2")][1]
print("flags", res[("tiled", "2")][2], "header", res[("tiled", "2")][3])
for y in range(H):
    eq = (a[y].view(np.uint32) == b[y].view(np.uint32))
    print("row", y, "equal", eq.all(), "first diffs", np.nonzero(~eq)[0][:8], a[y][:4], b[y][:4])
