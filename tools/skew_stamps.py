#!/usr/bin/env python3
"""Reads the s_memtime stamps of the diagnostic build of the skewed-tile forward pass
(tools/_probe/libstm_skstamps.so: tools/build_skew_variants.sh skstamps "-DSTM_SKEW_STAMPS") and
prints where a tile's time goes: median ticks between stamps per wave.  Development tool.

    python tools/skew_stamps.py [H W D B band]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
from stereomatch import _native  # noqa: E402

NAMES = ["stage cost (LDS stores) + progress wait", "top barrier, entries requested, penalties", "pair 0",
         "pair 1 first half", "drain of the stores", "barrier 1", "pair 1 second half + update",
         "pair 2 first half", "barrier 2", "pair 2 second half + update", "prefetch, pair 3 first half",
         "barrier 3", "pair 3 second half + update", "end barrier", "winner takes all"]


def main():
    H, W, D, B, band = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (375, 450, 128, 256, 20)
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libstm_skstamps.so"))
    for name, (restype, argtypes) in _native.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    setter = lib.stm_debug_set_skew_stamps_d128 if D == 128 else lib.stm_debug_set_skew_stamps_d256
    setter.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cost = (torch.rand((B, H, W, D), generator=g) * 50).to(dev)
    image = torch.randint(0, 255, (B, H, W), generator=g).float().to(dev)
    disp = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    scratch = torch.empty((1, H, W, D), device=dev)
    flags = _native.STM_SGM_FORMS["skew"]
    n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    nw = 4 if D == 128 else 8
    stamps = torch.zeros((64, nw, 16), dtype=torch.int64, device=dev)
    setter(stamps.data_ptr(), band)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        rc = lib.stm_sgm_wta_f32(cost.data_ptr(), image.data_ptr(), scratch.data_ptr(), disp.data_ptr(),
                                 H, W, D, 0.1, 0.2, flags, ws.data_ptr(), n, B, s)
        assert rc == 0, lib.stm_last_error()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.int64)
    ntc = min(64, (W + 7 + 7) // 8)
    st = st[2:ntc - 2]  # interior tiles
    d = np.diff(st, axis=2)  # [tile][wave][15]
    nxt = st[1:, :, 0] - st[:-1, :, 15]  # end of the tile -> top of the next
    period = np.median(st[1:, 0, 0] - st[:-1, 0, 0])
    print("tile period (wave 0): median %d ticks; per wave medians of each stretch:" % period)
    for k in range(15):
        print("  %2d -> %2d %-44s %s" % (k, k + 1, NAMES[k],
                                        " ".join("%6d" % v for v in np.median(d[:, :, k], axis=0))))
    print("  15 ->  0 %-44s %s" % ("loop back", " ".join("%6d" % v for v in np.median(nxt, axis=0))))
    wait = d[:, :, [5, 8, 11, 13]].sum(axis=2)
    print("barrier waits per tile, median per wave: %s of %d" % (np.median(wait, axis=0).astype(int).tolist(), period))


if __name__ == "__main__":
    main()
