#!/usr/bin/env python3
"""Development: wait statistics of a -DSTM_SKEW_X_STATS build (tools/_probe/libstm_stats.so).
python tools/skew_stats.py H W D B"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
from stereomatch import _native
lib = ctypes.CDLL(os.path.join(ROOT, "tools/_probe/libstm_stats.so"))
for name, (restype, argtypes) in _native.PROTOTYPES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
H, W, D, B = [int(a) for a in sys.argv[1:5]]
flags = _native.STM_SGM_FORMS["skew"]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
cost = (torch.rand((B, H, W, D), generator=g) * 50).to(dev)
image = torch.randint(0, 255, (B, H, W), generator=g).float().to(dev)
disp = torch.empty((B, H, W), dtype=torch.int32, device=dev)
scratch = torch.empty_like(cost)
n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
ws = torch.empty(n, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = lib.stm_sgm_wta_f32(cost.data_ptr(), image.data_ptr(), scratch.data_ptr(), disp.data_ptr(), H, W, D, 0.1, 0.2, flags, ws.data_ptr(), n, B, s)
    b.record(); torch.cuda.synchronize()
    assert rc == 0
    hdr = ws[:256].view(torch.int32).cpu().tolist()
    nb, nt = (H + 7) // 8, (W + 14) // 8
    print("%d x %d x %d B=%d: %.2f ms; bands %d tiles %d; looping waits %d (of %d tile starts in both passes), polls %d, waited %.1f ms in all = %.1f us per looping wait"
          % (H, W, D, B, a.elapsed_time(b), nb, nt, hdr[8], 2 * B * nb * nt, hdr[9], hdr[10] / 100e3, hdr[10] / 100.0 / max(hdr[8], 1)))
    n16 = max(hdr[16], 1)
    print("   forward, wave 0, per tile (us): stage+arrival %.1f, progress wait %.1f, top barrier %.1f; pair-1 drain %.1f, pair-1 barrier %.1f   (%d tiles)"
          % (hdr[13] / 100.0 / n16, hdr[14] / 100.0 / n16, hdr[15] / 100.0 / n16, hdr[11] / 100.0 / n16, hdr[12] / 100.0 / n16, hdr[16]))

import numpy as np
buf = np.zeros(8192 + 768, dtype=np.uint64)
fn = getattr(lib, "stm_debug_skew_stamps_d%d" % D)
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
st = buf[:8192].reshape(1024, 8)[:nb].astype(np.float64) / 100.0  # us
t0 = st[0, 0]
print("forward pass, image 0: start (us after band 0's first tile) of tiles 0, 1, 2, 4, 8, 32, 128, last; per band")
for i in list(range(0, min(nb, 6))) + list(range(8, nb, max(1, nb // 12))):
    print("  band %3d: " % i + " ".join("%9.1f" % (v - t0) for v in st[i]))

pub = buf[8192:8192 + 256].astype(np.float64) / 100.0
see = buf[8192 + 256:].astype(np.float64).reshape(256, 2)
print("band 1 (forward): tile, start time, counter value seen, when band 0 published (tile + 3), when band 0 published the value seen")
for j in list(range(0, 12)) + list(range(16, min(nt - 4, 240), 16)):
    v = int(see[j, 1])
    print("  tile %3d  start %9.1f  seen %3d  pub[j+3] %9.1f  pub[seen] %9.1f" % (j, see[j, 0] / 100.0 - t0, v, pub[min(j + 3, 255)] - t0, pub[min(v, 255)] - t0))
