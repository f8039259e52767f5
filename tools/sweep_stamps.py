#!/usr/bin/env python3
"""s_memtime stamps of the diagnostic build of the row-sweep kernel: where a row's time goes.
Development tool (tools/_probe/libstm_stamps.so, -DSTM_TILE_STAMPS)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
from stereomatch import _native  # noqa: E402

NAMES = ["row start -> leaving states published", "interior batches", "wait for arriving states",
         "last batch", "border-state stores + peek issue", "barrier"]


def main():
    H, W, D, B = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (1080, 1920, 256, 8)
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libstm_stamps.so"))
    for name, (restype, argtypes) in _native.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    lib.stm_debug_set_sweep_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cost = (torch.rand((B, H, W, D), generator=g) * 50).to(dev)
    image = torch.randint(0, 255, (B, H, W), generator=g).float().to(dev)
    disp = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    scratch = torch.empty_like(cost)
    flags = _native.STM_SGM_FORMS["tiled"]
    n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    NW = 16
    stamps = torch.zeros((8, NW, 16, 8), dtype=torch.int64, device=dev)
    lib.stm_debug_set_sweep_stamps(stamps.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        rc = lib.stm_sgm_wta_f32(cost.data_ptr(), image.data_ptr(), scratch.data_ptr(), disp.data_ptr(),
                                 H, W, D, 0.1, 0.2, flags, ws.data_ptr(), n, B, s)
        assert rc == 0, lib.stm_last_error()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.int64)  # [strip][wave][row][k]
    d = np.diff(st[..., :7], axis=-1)
    print("median cycles per segment, by wave class (edge = wave 0 / NW-1)")
    for label, sel in (("edge waves", [0, NW - 1]), ("inner waves", list(range(1, NW - 1)))):
        m = np.median(d[1:7, sel, 2:14, :], axis=(0, 1, 2))
        print(label)
        for k, nm in enumerate(NAMES):
            print("   %-45s %8d" % (nm, m[k]))
    period = np.diff(st[1:7, :, 2:14, 0], axis=2)
    print("row period (start to start): median %d cycles, p10 %d, p90 %d"
          % (np.median(period), np.percentile(period, 10), np.percentile(period, 90)))
    one = st[3, :, 5, :7] - st[3, :, 5, 0].min()
    print("strip 3, row 505, stamps per wave (cycles from the first wave's row start):")
    for w in range(NW):
        print("   wave %2d: %s" % (w, one[w].tolist()))


if __name__ == "__main__":
    main()
