"""The skewed-tile form of the semiglobal aggregation (stm_sgm_skew.hip: two passes over sheared
8 x 8 tiles, four directions carried per pass, bands handed over through progress counters)
against the CPU oracle and against the paired form, bit for bit, through the C ABI.

Shapes cover: images narrower than one tile and with a single band, widths / heights that are
not multiples of the tile (clipped parallelograms on every side), wide and tall images (the
reference's re-walked and skipped inverse diagonals), many bands (hand-overs
between workgroups), batches (units by ticket, images interleaved), the corrected-algorithm
options and the fused winner-takes-all."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _gpu, _random_volume, assert_same_bits
from test_gpu_sgm_tiled import _fuzz_cases

pytestmark = pytest.mark.gpu

D = 128  # (the D = 256 build of the same kernels: test_skew_d256_*)
SKEW_CASES = [
    (1, 1), (2, 2), (1, 9), (9, 1), (8, 8), (7, 9), (9, 7), (16, 16), (17, 15), (13, 30), (30, 13),
    (24, 40), (40, 24), (40, 90), (90, 40), (33, 129), (129, 33), (64, 65), (65, 64), (100, 101),
    (101, 100), (3, 200), (200, 3), (95, 151),
    (20, 300), (45, 530), (9, 1030),  # very wide: many tiles per band
]


def _skew(**kw):
    import stereomatch
    return stereomatch.aggregation.Semiglobal(form="skew", **kw)


def _is_skew(op):
    from stereomatch import _native
    return op._last_form == _native.STM_SGM_FORM_SKEW


@pytest.mark.parametrize("H,W", SKEW_CASES)
@pytest.mark.parametrize("kind", ["float", "last"])
def test_skew_sgm_matches_oracle(device, oracle, H, W, kind):
    rng = np.random.default_rng(13 * H + 7 * W + D)
    C = _random_volume(rng, H, W, D, kind)
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    sgm = _skew(penalty1=0.5, penalty2=3.0)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert _is_skew(sgm) and sgm.status() == 0
    assert_same_bits(got.cpu().numpy(), want, "skew sgm " + kind)
    _, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
    assert sgm.status() == 0
    assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "skew fused wta")
    vol, disp2 = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device), keep_sum=True)
    assert_same_bits(vol.cpu().numpy(), want, "skew keep_sum volume")
    assert_same_bits(disp2.cpu().numpy(), oracle.wta(want), "skew keep_sum disparity")


@pytest.mark.parametrize("H,W", [(24, 40), (40, 24), (33, 70), (70, 33)])
@pytest.mark.parametrize("paths", ["reference", "hirschmuller"])
@pytest.mark.parametrize("full_min", [False, True])
def test_skew_sgm_corrected_modes(device, oracle, H, W, paths, full_min):
    rng = np.random.default_rng(H + W + D)
    C = _random_volume(rng, H, W, D, "ties")
    image = rng.integers(0, 4, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.25, 2.0, true_paths=paths == "hirschmuller", full_min=full_min)
    sgm = _skew(penalty1=0.25, penalty2=2.0, paths=paths, full_min=full_min)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert _is_skew(sgm)
    assert_same_bits(got.cpu().numpy(), want, "skew sgm %s full_min=%s" % (paths, full_min))


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_skew_fuzz_against_paired(device, seed):
    """Random shapes, batches and options: the skewed-tile form against the paired form (which
    the other tests hold against the oracle), sum volume and fused disparities, bit for bit."""
    import stereomatch
    for H, W, _, B, kind, paths, full_min in _fuzz_cases(25, seed):
        rng = np.random.default_rng(H * 7919 + W * 31 + D + B)
        C = np.stack([_random_volume(rng, H, W, D, kind) for _ in range(B)])
        image = rng.integers(0, 6, size=(B, H, W)).astype(np.float32)
        Cg, Ig = _gpu(C, device), _gpu(image, device)
        kw = dict(penalty1=0.25, penalty2=2.0, paths=paths, full_min=full_min)
        t = _skew(**kw)
        p = stereomatch.aggregation.Semiglobal(form="paired", **kw)
        what = "H=%d W=%d D=%d B=%d %s %s full_min=%s" % (H, W, D, B, kind, paths, full_min)
        St = t(Cg, Ig)
        Sp = p(Cg, Ig)
        assert _is_skew(t) and t.status() == 0, what
        assert torch.equal(St.view(torch.int32), Sp.view(torch.int32)), "sum volume: " + what
        _, dt = t.aggregate_and_reduce(Cg, Ig)
        _, dp_ = p.aggregate_and_reduce(Cg, Ig)
        assert torch.equal(dt, dp_), "fused disparities: " + what


@pytest.mark.parametrize("H,W,B", [(260, 90, 24), (90, 420, 40), (375, 450, 12)])
def test_band_handovers_hold_over_repeated_runs(device, H, W, B):
    """Many bands and more units than workgroups: every band waits for its neighbour's progress
    counter and reads its row states with sc1 loads.  Repeated runs (the workspace keeps the
    previous run's states: a stale read would be a plausible value) against the paired form."""
    import stereomatch
    rng = np.random.default_rng(H + W + B)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(device)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(device)
    ref = stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1].clone()
    sgm = _skew()
    for rep in range(12):
        if rep % 3 == 1:  # other contents for the states of the next run to be told from
            C2 = C * np.float32(1.5)
            sgm.aggregate_and_reduce(C2, I)
        got = sgm.aggregate_and_reduce(C, I)[1]
        assert _is_skew(sgm) and sgm.status() == 0
        assert torch.equal(got, ref), "run %d" % rep


def test_auto_form_takes_skewed_tiles_for_large_d128_batches(device, oracle):
    """What `auto` selects: a large D = 128 batch takes the skewed-tile form, a small one does not;
    ACCUMULATE / other D are outside it and fall back to the forms that cover them."""
    import stereomatch
    from stereomatch import _native
    lib = _native.lib()
    assert lib.stm_sgm_form(375, 450, 128, 256, 0) == _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(375, 450, 128, 64, 0) == _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(375, 450, 128, 8, 0) != _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(375, 450, 128, 256, _native.STM_SGM_FORMS["tiled"]) == _native.STM_SGM_FORM_SWEEPS
    assert lib.stm_sgm_form(375, 450, 128, 256, _native.STM_SGM_FORMS["paired"]) == _native.STM_SGM_FORM_PAIRED
    assert lib.stm_sgm_form(375, 450, 128, 256, _native.STM_SGM_ACCUMULATE) != _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(375, 450, 64, 256, _native.STM_SGM_FORMS["skew"]) != _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(1080, 1920, 256, 8, 0) == _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(1080, 1920, 256, 4, 0) == _native.STM_SGM_FORM_SWEEPS
    assert lib.stm_sgm_form(1080, 1920, 256, 16, 0) == _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(1080, 1920, 256, 2, _native.STM_SGM_FORMS["skew"]) == _native.STM_SGM_FORM_SKEW
    assert lib.stm_sgm_form(40, 90, 128, 2, _native.STM_SGM_FORMS["skew"]) == _native.STM_SGM_FORM_SKEW
    # ... and through the op: a forced skew call on a shape it covers runs it, bit-exact
    H, W = 40, 72
    rng = np.random.default_rng(9)
    C = _random_volume(rng, H, W, D, "float")
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    sgm = _skew()
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert _is_skew(sgm)
    assert_same_bits(got.cpu().numpy(), oracle.sgm(C, image, 0.1, 0.2), "forced skew")


@pytest.mark.parametrize("H,W", [(1, 9), (8, 8), (17, 15), (24, 40), (40, 24), (33, 129), (129, 33), (64, 65), (20, 300)])
def test_skew_d256_matches_oracle(device, oracle, H, W):
    """D = 256: one chain per wave, eight waves per tile (the same source compiled for the other
    disparity count): against the oracle, sum volume and fused winners."""
    rng = np.random.default_rng(5 * H + 3 * W)
    C = _random_volume(rng, H, W, 256, "float")
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    sgm = _skew(penalty1=0.5, penalty2=3.0)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert _is_skew(sgm) and sgm.status() == 0
    assert_same_bits(got.cpu().numpy(), want, "skew sgm D=256")
    _, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
    assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "skew fused wta D=256")


@pytest.mark.parametrize("seed", [31, 32])
def test_skew_d256_fuzz_against_paired(device, seed):
    import stereomatch
    for H, W, _, B, kind, paths, full_min in _fuzz_cases(20, seed):
        rng = np.random.default_rng(H * 7919 + W * 31 + 256 + B)
        C = np.stack([_random_volume(rng, H, W, 256, kind) for _ in range(B)])
        image = rng.integers(0, 6, size=(B, H, W)).astype(np.float32)
        Cg, Ig = _gpu(C, device), _gpu(image, device)
        kw = dict(penalty1=0.25, penalty2=2.0, paths=paths, full_min=full_min)
        t = _skew(**kw)
        p = stereomatch.aggregation.Semiglobal(form="paired", **kw)
        what = "H=%d W=%d D=256 B=%d %s %s full_min=%s" % (H, W, B, kind, paths, full_min)
        St, Sp = t(Cg, Ig), p(Cg, Ig)
        assert _is_skew(t) and t.status() == 0, what
        assert torch.equal(St.view(torch.int32), Sp.view(torch.int32)), "sum volume: " + what
        assert torch.equal(t.aggregate_and_reduce(Cg, Ig)[1], p.aggregate_and_reduce(Cg, Ig)[1]), "fused: " + what


def test_rewalked_diagonals_are_deferred_and_exact(device, oracle):
    """Wide and tall images whose re-walked inverse diagonals (added 1 + |W - H| times)
    cross many bands: those pixels are finished off the band pipeline (finish_deferred) -- every
    element of the sum and every winner against the oracle, D = 128 and 256."""
    for H, W, DD in [(40, 200, 128), (200, 40, 128), (48, 130, 256), (130, 48, 256), (41, 40, 128), (40, 42, 128)]:
        rng = np.random.default_rng(H + W)
        C = _random_volume(rng, H, W, DD, "float")
        image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
        want = oracle.sgm(C, image, 0.1, 0.2)
        sgm = _skew()
        vol, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device), keep_sum=True)
        assert _is_skew(sgm)
        assert_same_bits(vol.cpu().numpy(), want, "deferred diagonals, sum %dx%dx%d" % (H, W, DD))
        assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "deferred diagonals, winners %dx%dx%d" % (H, W, DD))
