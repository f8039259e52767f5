"""Seeded random-shape sweep of every op against the oracle (bit-exact).  Shapes are drawn
to hit the kernels' code-path boundaries: D around multiples of 64 and the DP tile sizes,
images narrower/wider than tiles, windows and disparity ranges."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _gpu, _random_volume, assert_same_bits

pytestmark = pytest.mark.gpu


def _shapes(n, seed):
    rng = np.random.default_rng(seed)
    ds = [1, 2, 3, 4, 8, 12, 16, 24, 32, 48, 60, 64, 68, 96, 100, 128, 132, 192, 256, 320]
    out = []
    for _ in range(n):
        D = int(rng.choice(ds))
        H = int(rng.integers(1, 70))
        W = int(rng.integers(1, 150))
        if H * W * D > 600000:
            W = max(1, 600000 // (H * D))
        out.append((H, W, D))
    return out


def _long_path_shapes():
    """Paths of several 16/32-pixel segments in every family, at every lanes-per-value count."""
    rng = np.random.default_rng(5)
    return [(int(rng.integers(40, 130)), int(rng.integers(40, 160)), D)
            for D in (32, 64, 128, 192, 256, 512) for _ in range(2)]


@pytest.mark.parametrize("H,W,D", _shapes(120, seed=2026) + _long_path_shapes())
def test_fuzz_all_ops(device, oracle, H, W, D):
    import stereomatch
    rng = np.random.default_rng(H * 131 + W * 17 + D)
    kind = ["u8", "unit"][int(rng.integers(0, 2))]
    k = int(rng.choice([1, 2, 3, 5, 7, 7, 7, 8]))
    L, R = oracle.synthetic_pair(H, W, seed=int(rng.integers(0, 1000)), kind=kind)
    Lg, Rg = _gpu(L, device), _gpu(R, device)
    want_C = oracle.ssd(L, R, D, k)
    for mode in ("auto", "ordered"):
        got = stereomatch.cost.SSD(D, kernel_size=k, mode=mode)(Lg, Rg)
        assert_same_bits(got.cpu().numpy(), want_C, "ssd k=%d %s %s" % (k, kind, mode))
    bt = stereomatch.cost.Birchfield(D)(Lg, Rg)
    assert_same_bits(bt.cpu().numpy(), oracle.birchfield(L, R, D), "birchfield")
    vol_kind = ["float", "ties", "inf", "last"][int(rng.integers(0, 4))]
    vol = _random_volume(rng, H, W, D, vol_kind)
    vol_raw = vol.copy()  # with its all-+inf pixels: for the exact non-finite variant below
    # Like every volume the cost ops produce, d = 0 is finite at every pixel.  A pixel whose
    # costs are all +inf makes the reference's recurrence emit inf - inf: NaN and -inf then
    # spread along every path through it, and which of them lands where depends on how
    # std::min_element and the '<' chains happen to treat NaN operands.
    vol[..., 0] = np.where(np.isfinite(vol[..., 0]), vol[..., 0], np.float32(5.0))
    image = rng.integers(0, 4, size=(H, W)).astype(np.float32)
    p1, p2 = float(rng.choice([0.1, 0.5, 2.0])), float(rng.choice([0.2, 3.0, 40.0]))
    paths = ["reference", "hirschmuller"][int(rng.integers(0, 2))]
    full_min = bool(rng.integers(0, 2))
    want_S = oracle.sgm(vol, image, p1, p2, full_min=full_min, true_paths=(paths == "hirschmuller"))
    vg, ig = _gpu(vol, device), _gpu(image, device)
    for single in (False, True):
        S = stereomatch.aggregation.Semiglobal(p1, p2, single_passes=single, paths=paths,
                                               full_min=full_min)(vg, ig)
        assert_same_bits(S.cpu().numpy(), want_S, "sgm single=%s %s full_min=%s" % (single, paths, full_min))
    if vol_kind == "inf":
        # STM_SGM_EXACT_NONFINITE on the UNPATCHED volume (30 % +inf entries, an all-+inf pixel):
        # NaN, -inf and +inf land where the reference puts them, in both forms of the kernels.
        raw_g = _gpu(vol_raw, device)
        want_raw = oracle.sgm(vol_raw, image, p1, p2, full_min=full_min, true_paths=(paths == "hirschmuller"))
        for single in (False, True):
            S = stereomatch.aggregation.Semiglobal(p1, p2, single_passes=single, paths=paths,
                                                   full_min=full_min, exact_nonfinite=True)(raw_g, ig)
            assert_same_bits(S.cpu().numpy(), want_raw, "sgm exact non-finite single=%s" % single)
    # The reducers run on a NaN-free volume: NaN handling differs by design in SGM/DP.
    red = np.where(np.isnan(want_S), np.float32(np.inf), want_S).astype(np.float32)
    rg = _gpu(red, device)
    wta = stereomatch.disparity_reduce.WinnerTakesAll(require_power_of_two=False)
    assert_same_bits(wta(rg).cpu().numpy(), oracle.wta(red), "wta")
    if not np.isnan(want_S).any():
        _, fused = stereomatch.aggregation.Semiglobal(p1, p2, paths=paths, full_min=full_min
                                                      ).aggregate_and_reduce(vg, ig)
        assert_same_bits(fused.cpu().numpy(), oracle.wta(want_S), "fused wta")
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")
    assert_same_bits(dp(rg).cpu().numpy(), oracle.dp(red, mode=1), "dp corrected")
    if D <= H <= W:
        dpr = stereomatch.disparity_reduce.DynamicProgramming()
        assert_same_bits(dpr(rg).cpu().numpy(), oracle.dp(red, mode=0), "dp reference")
