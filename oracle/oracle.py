"""ctypes front-end of ``libstm_oracle.so`` (numpy in / numpy out).

TEST INFRASTRUCTURE ONLY -- see ``stm_oracle.c``.  Every wrapper mirrors one
reference entry point:

* ``ssd``        -> ``CostOps::ComputeSSD``              (src/ssd.cu:64-81)
* ``birchfield`` -> ``CostOps::ComputeBirchfield``       (src/birchfield_cost.cu:153-181)
* ``sgm``        -> ``AggregationOps::RunSemiglobal``    (src/semiglobal.cpp:167-197)
* ``wta``        -> ``DisparityReduceOps::RunWinnersTakeAll`` (src/winners_take_all.cu:89-114)
* ``dp``         -> ``DisparityReduceOps::RunDynamicProgramming`` (src/dynamic_programming.cu:184-225)
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstm_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i8p = ctypes.POINTER(ctypes.c_int8)


def build(force=False):
    """Compile the C restatement with gcc (a no-op when up to date)."""
    src = os.path.join(_HERE, "stm_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libstm_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_int, c_float = ctypes.c_int, ctypes.c_float
        L.orc_ssd_f32.argtypes = [_f32p, _f32p, _f32p, c_int, c_int, c_int, c_int]
        L.orc_ssd_entry_f32.argtypes = [_f32p, _f32p, c_int, c_int, c_int, c_int, c_int, c_int]
        L.orc_ssd_entry_f32.restype = c_float
        L.orc_birchfield_f32.argtypes = [_f32p, _f32p, _f32p, c_int, c_int, c_int]
        L.orc_birchfield_ex_f32.argtypes = [_f32p, _f32p, _f32p, c_int, c_int, c_int, c_int]
        L.orc_sgm_f32.argtypes = [_f32p, _f32p, _f32p, c_int, c_int, c_int, c_float, c_float]
        L.orc_sgm_ex_f32.argtypes = [_f32p, _f32p, _f32p, c_int, c_int, c_int, c_float, c_float, c_int]
        L.orc_sgm_num_paths.argtypes = [c_int, c_int]
        L.orc_sgm_num_paths.restype = ctypes.c_size_t
        L.orc_sgm_paths.argtypes = [c_int, c_int, _i16p]
        L.orc_wta_f32.argtypes = [_f32p, _i32p, c_int, c_int, c_int]
        L.orc_dp_f32.argtypes = [_f32p, _i8p, _f32p, _i32p, c_int, c_int, c_int, c_int]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (what, rc))


def ssd(left, right, max_disparity, kernel_size=7):
    left, lp = _f32(left)
    right, rp = _f32(right)
    H, W = left.shape
    out = np.empty((H, W, max_disparity), dtype=np.float32)
    _check(lib().orc_ssd_f32(lp, rp, out.ctypes.data_as(_f32p), H, W,
                             max_disparity, kernel_size), "ssd")
    return out


def ssd_entries(left, right, kernel_size, rows, cols, disps):
    """SSD values at the sampled (row, col, disp) triples."""
    left, lp = _f32(left)
    right, rp = _f32(right)
    H, W = left.shape
    fn = lib().orc_ssd_entry_f32
    return np.array([fn(lp, rp, H, W, kernel_size, int(r), int(c), int(d))
                     for r, c, d in zip(rows, cols, disps)], dtype=np.float32)


def birchfield(left, right, max_disparity, out=None, padded=False):
    """Entries d > x are left untouched (zero when ``out`` is None).  ``padded`` selects the
    CUDA functor's sample indexing (an opt-in extension of the HIP library)."""
    left, lp = _f32(left)
    right, rp = _f32(right)
    H, W = left.shape
    if out is None:
        out = np.zeros((H, W, max_disparity), dtype=np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    _check(lib().orc_birchfield_ex_f32(lp, rp, out.ctypes.data_as(_f32p), H, W,
                                       max_disparity, 1 if padded else 0), "birchfield")
    return out


def sgm(cost_volume, left, penalty1=0.1, penalty2=0.2, full_min=False, true_paths=False, out=None):
    """``full_min`` / ``true_paths`` select the corrected-algorithm variants (extensions of
    the HIP library); the defaults are the reference's behaviour.  The reference accumulates
    into its output tensor (src/semiglobal.cpp:150-156; the wrapper zero-fills it first,
    aggregation.py:51): ``out`` = the contents to accumulate on top of (default zeros)."""
    cost_volume, cp = _f32(cost_volume)
    left, ip = _f32(left)
    H, W, D = cost_volume.shape
    if out is None:
        out = np.zeros((H, W, D), dtype=np.float32)
    else:
        out = np.array(out, dtype=np.float32, order="C", copy=True)
        assert out.shape == (H, W, D)
    flags = (1 if full_min else 0) | (2 if true_paths else 0)
    _check(lib().orc_sgm_ex_f32(cp, ip, out.ctypes.data_as(_f32p), H, W, D,
                                penalty1, penalty2, flags), "sgm")
    return out


def sgm_paths(W, H):
    """The reference path table as an [n, 7] int16 array
    (sx, sy, ex, ey, dx, dy, size)."""
    n = lib().orc_sgm_num_paths(W, H)
    out = np.zeros((n, 7), dtype=np.int16)
    _check(lib().orc_sgm_paths(W, H, out.ctypes.data_as(_i16p)), "paths")
    return out


def wta(volume):
    volume, vp = _f32(volume)
    H, W, D = volume.shape
    out = np.empty((H, W), dtype=np.int32)
    _check(lib().orc_wta_f32(vp, out.ctypes.data_as(_i32p), H, W, D), "wta")
    return out


def dp(volume, mode=0, return_aux=False):
    volume, vp = _f32(volume)
    H, W, D = volume.shape
    path = np.zeros((H, W, D), dtype=np.int8)
    row_final = np.empty((H, D), dtype=np.float32)
    out = np.empty((H, W), dtype=np.int32)
    _check(lib().orc_dp_f32(vp, path.ctypes.data_as(_i8p),
                            row_final.ctypes.data_as(_f32p),
                            out.ctypes.data_as(_i32p), H, W, D, mode), "dp")
    if return_aux:
        return out, path, row_final
    return out


def synthetic_pair(H, W, seed=0, shift=7, kind="u8"):
    """SURVEY 8(d) generator: integer-valued 0..255 float32 images with a
    constant true disparity ``shift``; ``kind="unit"`` gives the /255-style
    non-integer variant used for robustness runs."""
    rng = np.random.default_rng(seed)
    if kind == "u8":
        base = rng.integers(0, 256, size=(H, W + 64)).astype(np.float32)
    elif kind == "unit":
        base = rng.random(size=(H, W + 64), dtype=np.float32)
    else:
        raise ValueError(kind)
    left = np.ascontiguousarray(base[:, 32:32 + W])
    right = np.ascontiguousarray(base[:, 32 + shift:32 + shift + W])
    return left, right


def digest16(a):
    """First 16 hex digits of sha256 over the raw little-endian bytes."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
