#!/usr/bin/env python3
"""Reads the s_memtime stamps of the diagnostic build of the tile kernel
(tools/_probe/libstm_stamps.so, built with -DSTM_TILE_STAMPS) and prints where a tile's time
goes: median cycles between stamps, per wave group.  Development tool.

    python tools/tile_stamps.py [H W D B]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
from stereomatch import _native  # noqa: E402

NAMES0 = ["loads issued, C stored to LDS", "barrier B0", "chains 0", "add 0 + B1",
          "chains 2", "B2 + add 2", "B3"]
NAMES1 = ["loads issued, C stored to LDS", "barrier B0", "chains 1", "B1 + add 1",
          "B2 + chains 3", "B3", "add 3 (+ output)"]


def main():
    H, W, D, B = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (1080, 1920, 256, 8)
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libstm_stamps.so"))
    for name, (restype, argtypes) in _native.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    lib.stm_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cost = (torch.rand((B, H, W, D), generator=g) * 50).to(dev)
    image = torch.randint(0, 255, (B, H, W), generator=g).float().to(dev)
    disp = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    scratch = torch.empty_like(cost)
    flags = _native.STM_SGM_FORMS["tiled"]
    n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    stamps = torch.zeros((4096, 16, 8), dtype=torch.int64, device=dev)
    lib.stm_debug_set_tile_stamps(stamps.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        rc = lib.stm_sgm_wta_f32(cost.data_ptr(), image.data_ptr(), scratch.data_ptr(), disp.data_ptr(),
                                 H, W, D, 0.1, 0.2, flags, ws.data_ptr(), n, B, s)
        assert rc == 0, lib.stm_last_error()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.int64)
    ntiles = min(4096, ((H + 7) // 8) * ((W + 7) // 8))
    st = st[:ntiles]
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    rel = st - t0[:, :, None]
    if D == 128:  # sgm_tile_h2_kernel: four waves, all alike
        names = ["C stored to LDS", "barrier", "loads issued + pairs 0, 1, 2", "pair 3 chains", "barrier (pair 3)",
                 "pair 3 update, result, winner", "end barrier"]
        r = rel[:, 0:4, :]
        ok = (st[:, 0:4, 7] > 0).all(axis=1)  # tiles on the stamped (fast) path
        r = r[ok]
        print("stamped tiles: %d of %d" % (ok.sum(), ntiles))
        print("stamp medians %s" % np.median(r, axis=(0, 1)).astype(int).tolist())
        med = np.median(np.diff(r, axis=2), axis=(0, 1))
        for k, nm in enumerate(names):
            print("   %-40s %8d" % (nm, med[k]))
        s0 = np.sort(st[ok][:, 0, 0])
        # the same workgroup's consecutive tiles are `grid` ids apart: estimate the period from
        # stamps of ids id and id + grid when both are in range
        grid = int(os.environ.get("STM_GRID", "512"))
        per = [st[i + grid, 0, 0] - st[i, 0, 0] for i in range(0, ntiles - grid) if ok[i] and ok[i + grid]]
        if per:
            print("tile period of a workgroup: median %d ticks (pair 3 + end barrier = period - %d)"
                  % (np.median(per), np.median(r[:, :, 7])))
        return
    for grp, names in ((0, NAMES0), (1, NAMES1)):
        r = rel[:, grp * 8:(grp + 1) * 8, :]
        print("group %d: stamp medians %s" % (grp, np.median(r, axis=(0, 1)).astype(int).tolist()))
        d = np.diff(r, axis=2)
        med = np.median(d, axis=(0, 1))
        for k, nm in enumerate(names):
            print("   %-40s %8d" % (nm, med[k]))
    total = (st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1))
    print("tile lifetime (first wave start -> last wave end): median %d, p10 %d, p90 %d ticks"
          % (np.median(total), np.percentile(total, 10), np.percentile(total, 90)))
    # tiles started per unit time: the spread of start stamps over the sampled tiles
    starts = np.sort(st[:, :, 0].min(axis=1))
    print("sampled tiles: %d, first..last start spread %d ticks" % (ntiles, starts[-1] - starts[0]))


if __name__ == "__main__":
    main()
