"""ctypes binding of ``libstereomatch_hip.so`` (C ABI: ``include/stereomatch_hip.h``).

This is the only place the Python package touches native code.  The library is
loaded from this directory (built in-tree by ``csrc/Makefile``); if it is missing the
import of any op fails loudly -- there is no fallback implementation.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (STM_LIB_PATH: another build of the same library, for development probes such as tools/_probe)
LIB_PATH = os.environ.get("STM_LIB_PATH") or os.path.join(_HERE, "libstereomatch_hip.so")

ABI_VERSION = 1

STM_SSD_MODES = {"auto": 0, "ordered": 1, "box": 2}
STM_SGM_ACCUMULATE = 1
STM_SGM_SINGLE_PASSES = 2
STM_SGM_KEEP_SUM = 4
STM_SGM_FULL_MIN = 8
STM_SGM_TRUE_PATHS = 16
STM_SGM_EXACT_NONFINITE = 32
STM_SGM_FORMS = {"auto": 0, "paired": 64, "tiled": 128, "walks": 512, "skew": 1024}
STM_SGM_COOPERATIVE = 256
STM_SGM_FORM_PAIRED, STM_SGM_FORM_SWEEPS, STM_SGM_FORM_WALKS, STM_SGM_FORM_SKEW = 0, 1, 2, 3
STM_DP_MODES = {"reference": 0, "corrected": 1}
STM_BT_INDEXING = {"reference": 0, "padded": 1}

# name -> (restype, argtypes); every symbol include/stereomatch_hip.h declares.
_vp, _sz, _i, _f = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float
PROTOTYPES = {
    "stm_abi_version": (_i, []),
    "stm_last_error": (ctypes.c_char_p, []),
    "stm_ssd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "stm_ssd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "stm_birchfield_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "stm_birchfield_ex_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "stm_sgm_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "stm_sgm_status": (ctypes.c_uint, [_vp]),
    "stm_sgm_uses_tiles": (_i, [_i, _i, _i, _i, _i]),
    "stm_sgm_form": (_i, [_i, _i, _i, _i, _i]),
    "stm_sgm_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _i, _vp]),
    "stm_sgm_wta_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _i, _vp]),
    "stm_wta_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "stm_dp_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "stereomatch: %s is missing -- build it with `make -C stereo-matching_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no "
                "fallback implementation" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.stm_abi_version() != ABI_VERSION:
            raise RuntimeError("stereomatch: ABI version mismatch (library %d, binding %d)"
                               % (handle.stm_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc: int) -> None:
    """Raise RuntimeError carrying stm_last_error() for a non-zero return code (the
    reference surfaces std::runtime_error the same way, check.hpp:12-19)."""
    if rc != 0:
        msg = lib().stm_last_error()
        raise RuntimeError(msg.decode("utf-8", "replace") if msg else "stereomatch: error %d" % rc)


def require_gpu(reference: torch.Tensor, *others: torch.Tensor) -> None:
    """All tensors on one HIP device (counterpart of STM_CHECK_DEVICE, check.hpp:27-43).
    CPU tensors are refused: this package has no CPU code path."""
    if not reference.is_cuda:
        raise RuntimeError("Check failed: Expected a gpu tensor (stereomatch for MI355X has no "
                           "CPU code path)")
    for t in others:
        if t.device != reference.device:
            raise RuntimeError("Check failed: Expected a %s tensor"
                               % ("gpu" if reference.is_cuda else "cpu"))


def require_float32(volume: torch.Tensor, op: str, reference_dispatch: str) -> None:
    """The device path computes in float32 only.  The reference additionally dispatches
    float64 (``AT_DISPATCH_FLOATING_TYPES``) for this op; that is declined here with a message
    that says so -- never served by a CPU or reduced-precision fallback."""
    if volume.dtype != torch.float32:
        raise RuntimeError(
            "%s: %s tensors are not supported on the MI355X path, which implements float32 only "
            "(the reference also dispatches float64 here: %s); convert with .float()"
            % (op, str(volume.dtype).replace("torch.", ""), reference_dispatch))


def stream_ptr(device: torch.device) -> int:
    """hipStream_t of torch's current stream on ``device``."""
    return torch.cuda.current_stream(device).cuda_stream


def split_batch(t: torch.Tensor, image_dims: int):
    """(batch, trailing shape) for a tensor that is either one item or a batch."""
    if t.dim() == image_dims:
        return 1, tuple(t.shape)
    if t.dim() == image_dims + 1:
        return int(t.shape[0]), tuple(t.shape[1:])
    raise RuntimeError("stereomatch: expected a %d-D tensor (or %d-D batch), got %d-D"
                       % (image_dims, image_dims + 1, t.dim()))
