"""The C-ABI library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "stereomatch_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    assert declared_symbols() == sorted([
        "stm_abi_version", "stm_last_error", "stm_ssd_workspace_bytes", "stm_ssd_f32",
        "stm_birchfield_f32", "stm_birchfield_ex_f32", "stm_sgm_workspace_bytes", "stm_sgm_uses_tiles", "stm_sgm_form",
        "stm_sgm_status", "stm_sgm_f32", "stm_sgm_wta_f32", "stm_wta_f32", "stm_dp_f32"])


def test_library_exports_every_declared_symbol():
    from stereomatch import _native
    assert os.path.exists(_native.LIB_PATH), "build with __graft_entry__.build()"
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name
    assert sorted(_native.PROTOTYPES) == declared_symbols()
    assert _native.lib().stm_abi_version() == 1


def test_argument_validation_needs_no_gpu():
    """Bad arguments are refused before any HIP call, with a message."""
    from stereomatch import _native
    lib = _native.lib()
    assert lib.stm_wta_f32(None, None, 4, 4, 4, 1, None) == 1            # STM_EINVAL
    assert b"null pointer" in lib.stm_last_error()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.stm_wta_f32(p, p, 0, 4, 4, 1, None) == 1
    assert b"non-positive" in lib.stm_last_error()
    assert lib.stm_sgm_f32(p, p, p, 4, 4, 4, 0.1, 0.2, 0, None, 0, 1, None) == 1  # cost aliases out
    assert lib.stm_ssd_f32(p, p, p, 4, 4, 2048, 7, 1, None, 0, 1, None) == 2      # STM_EUNSUPPORTED
    assert lib.stm_ssd_f32(p, p, p, 4, 4, 4, 7, 0, None, 0, 1, None) == 4         # AUTO without workspace
    assert lib.stm_dp_f32(p, p, p, p, 4, 2, 8, 0, 1, None) == 1                   # needs D <= H <= W
    assert b"D <= H <= W" in lib.stm_last_error()
    with pytest.raises(RuntimeError, match="D <= H <= W"):
        _native.check(1)


def test_workspace_sizes():
    from stereomatch import _native
    lib = _native.lib()
    assert lib.stm_ssd_workspace_bytes(375, 450, 128, 7, 1) == 16
    single = 2  # STM_SGM_SINGLE_PASSES: only the spill rows of a re-walked diagonal
    assert lib.stm_sgm_workspace_bytes(300, 300, 128, 1, single) == 0          # square: no re-walks
    assert lib.stm_sgm_workspace_bytes(301, 300, 128, 1, single) == 0          # H = W+1: none either
    assert lib.stm_sgm_workspace_bytes(375, 450, 128, 2, single) == 2 * 375 * 128 * 4
    assert lib.stm_sgm_workspace_bytes(450, 375, 128, 1, single) == 375 * 128 * 4
    # Paired form: + checkpoints every K = 32 / (D / 64) pixels for the four forward families.
    H, W, D, K = 375, 450, 128, 16
    seg = lambda n: (n + K - 1) // K - 1
    ckpt = H * seg(W) + W * seg(H) + 2 * (W + H - 1) * seg(min(H, W))
    paired = 64  # STM_SGM_PAIRED
    assert lib.stm_sgm_workspace_bytes(H, W, D, 1, paired) == (H + ckpt) * D * 4
    # A single pair at D = 128 takes the tiles fed by walks (needs no device to decide): header +
    # mailboxes + row-segment flags + 6 / 8 floats of border states per element each for row and
    # column lines.
    assert lib.stm_sgm_form(H, W, D, 1, 0) == 2 and lib.stm_sgm_form(H, W, 64, 1, 0) == 0
    assert lib.stm_sgm_form(H, W, D, 1, paired) == 0 and lib.stm_sgm_form(H, W, D, 1, 512) == 2
    nTR, nTC, strips = (H + 7) // 8, (W + 7) // 8, (W + 63) // 64
    tiled = (64 + 2 * strips * 2 * 4 * D * 2 + (nTR * 4 + 3) // 4 * 4 + 6 * nTR * W * D + 6 * nTC * H * D) * 4
    assert lib.stm_sgm_workspace_bytes(H, W, D, 1, 512) == tiled
    assert lib.stm_sgm_workspace_bytes(H, W, D, 1, 0) == max(tiled, 256 + (H + ckpt) * D * 4)
