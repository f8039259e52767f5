#!/usr/bin/env python3
"""Runs the fused aggregation of a given build of the library (for rocprofv3 kernel stats of
diagnostic / probe builds).  python tools/run_lib.py <lib.so> H W D B [tiled|paired] [extra flags]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
from stereomatch import _native  # noqa: E402

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, (restype, argtypes) in _native.PROTOTYPES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = restype, argtypes
H, W, D, B = [int(a) for a in sys.argv[2:6]]
flags = _native.STM_SGM_FORMS[sys.argv[6] if len(sys.argv) > 6 else "tiled"]
if len(sys.argv) > 7:
    flags |= int(sys.argv[7])
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
cost = (torch.rand((B, H, W, D), generator=g) * 50).to(dev)
image = torch.randint(0, 255, (B, H, W), generator=g).float().to(dev)
disp = torch.empty((B, H, W), dtype=torch.int32, device=dev)
scratch = torch.empty_like(cost)
n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
ws = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = lib.stm_sgm_wta_f32(cost.data_ptr(), image.data_ptr(), scratch.data_ptr(), disp.data_ptr(),
                             H, W, D, 0.1, 0.2, flags, ws.data_ptr(), n, B, s)
    assert rc == 0, lib.stm_last_error()
torch.cuda.synchronize()
print("ok")
