"""HIP path vs CPU oracle on seeded inputs (bit-exact), through the stereomatch package
and hence the C ABI.  Mirrors the reference's CPU-vs-GPU equality tests
(tests/test_cost.py:41-57,135-152, tests/test_semiglobalmatch.py:39-55,
tests/test_winners_take_all.py:26-48) with the oracle in the CPU role."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same_bits(got, want, what=""):
    """Bit-for-bit equality (so +inf and signed zeros count).  A NaN equals any NaN: which
    sign and payload an inf - inf produces, and which operand's NaN an addition passes on,
    are properties of x86 and of gfx950, not of the algorithm."""
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(_bits(got), _bits(want)):
        differ = _bits(got) != _bits(want)
        if got.dtype == np.float32:
            differ &= ~(np.isnan(got) & np.isnan(want))
        if not differ.any():
            return
        bad = np.argwhere(differ)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d entries differ; first at %s: got %r want %r"
                             % (what, len(bad), got.size, i, got[i], want[i]))


def _gpu(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _random_volume(rng, H, W, D, kind):
    if kind == "float":
        return (rng.random((H, W, D), dtype=np.float32) * 100).astype(np.float32)
    if kind == "ties":  # few distinct integer values: many exact ties
        return rng.integers(0, 4, size=(H, W, D)).astype(np.float32)
    if kind == "inf":
        v = (rng.random((H, W, D), dtype=np.float32) * 10).astype(np.float32)
        v[rng.random((H, W, D)) < 0.3] = np.inf
        v[0, 0, :] = np.inf       # an all-inf pixel must answer 0
        return v
    if kind == "last":  # minimum at the last disparity
        v = (rng.random((H, W, D), dtype=np.float32) * 50 + 10).astype(np.float32)
        v[..., D - 1] = rng.random((H, W), dtype=np.float32)
        return v
    raise ValueError(kind)


# ---------------------------------------------------------------- winner takes all
@pytest.mark.parametrize("H,W,D", [(7, 9, 16), (5, 33, 64), (6, 11, 100), (9, 17, 128),
                                    (4, 13, 256), (3, 5, 512), (3, 7, 1), (5, 6, 3), (2, 3, 1024)])
@pytest.mark.parametrize("kind", ["float", "ties", "inf"])
def test_wta_matches_oracle(device, oracle, H, W, D, kind):
    import stereomatch
    rng = np.random.default_rng(H * 1000 + W * 10 + D)
    vol = _random_volume(rng, H, W, D, kind)
    got = stereomatch.disparity_reduce.WinnerTakesAll(require_power_of_two=False)(_gpu(vol, device))
    assert got.dtype == torch.int32
    assert_same_bits(got.cpu().numpy(), oracle.wta(vol), "wta %s" % kind)


def test_wta_nan_rules(device, oracle):
    import stereomatch
    rng = np.random.default_rng(5)
    vol = _random_volume(rng, 6, 8, 64, "float")
    vol[0, 0, 0] = np.nan      # NaN at d = 0 always wins
    vol[1, 1, 5] = np.nan      # NaN at d > 0 never wins
    vol[2, 2, 1:] = np.nan
    got = stereomatch.disparity_reduce.WinnerTakesAll()(_gpu(vol, device))
    assert_same_bits(got.cpu().numpy(), oracle.wta(vol), "wta nan")


def test_wta_reference_arange_case(device, oracle):
    """tests/test_winners_take_all.py:26-39 uses an arange volume."""
    import stereomatch
    vol = np.arange(60 * 70 * 128, dtype=np.float32).reshape(60, 70, 128)
    got = stereomatch.disparity_reduce.WinnerTakesAll()(_gpu(vol, device))
    assert_same_bits(got.cpu().numpy(), oracle.wta(vol), "wta arange")


def test_wta_non_power_of_two_raises(device):
    """tests/test_winners_take_all.py:42-48."""
    import stereomatch
    with pytest.raises(RuntimeError):
        stereomatch.disparity_reduce.WinnerTakesAll()(torch.zeros(30, 30, 100, device=device))


# ---------------------------------------------------------------------------- SSD
SSD_CASES = [(24, 40, 16, 7), (40, 24, 16, 7), (33, 70, 64, 7), (20, 150, 128, 7),
             (9, 300, 256, 7), (30, 45, 32, 3), (18, 40, 24, 1), (25, 140, 100, 5),
             (16, 30, 64, 7)]


@pytest.mark.parametrize("H,W,D,k", SSD_CASES)
@pytest.mark.parametrize("kind", ["u8", "unit"])
@pytest.mark.parametrize("mode", ["ordered", "auto"])
def test_ssd_matches_oracle(device, oracle, H, W, D, k, kind, mode):
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=H + W + D, kind=kind)
    want = oracle.ssd(L, R, D, k)
    got = stereomatch.cost.SSD(D, kernel_size=k, mode=mode)(_gpu(L, device), _gpu(R, device))
    assert_same_bits(got.cpu().numpy(), want, "ssd %s %s" % (kind, mode))


@pytest.mark.parametrize("H,W,D,k", SSD_CASES)
def test_ssd_box_exact_on_integer_images(device, oracle, H, W, D, k):
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=3, kind="u8")
    got = stereomatch.cost.SSD(D, kernel_size=k, mode="box")(_gpu(L, device), _gpu(R, device))
    assert_same_bits(got.cpu().numpy(), oracle.ssd(L, R, D, k), "ssd box")


def test_ssd_rejects_uint8(device):
    """byte images raise 'SSDKernel: type not supported'."""
    import stereomatch
    img = torch.zeros(8, 8, dtype=torch.uint8, device=device)
    with pytest.raises(RuntimeError, match="type not supported"):
        stereomatch.cost.SSD(8)(img, img)


# --------------------------------------------------------------------- Birchfield
@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (40, 24, 16), (12, 130, 64), (7, 200, 128),
                                    (5, 70, 256), (9, 50, 20)])
@pytest.mark.parametrize("kind", ["u8", "unit"])
def test_birchfield_matches_oracle(device, oracle, H, W, D, kind):
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=11 + D, kind=kind)
    got = stereomatch.cost.Birchfield(D)(_gpu(L, device), _gpu(R, device))
    assert_same_bits(got.cpu().numpy(), oracle.birchfield(L, R, D), "birchfield " + kind)


def test_birchfield_leaves_upper_entries_untouched(device, oracle):
    import stereomatch
    L, R = oracle.synthetic_pair(10, 30, seed=1)
    buf = torch.full((10, 30, 16), 123.0, device=device)
    out = stereomatch.cost.Birchfield(16)(_gpu(L, device), _gpu(R, device), cost_volume=buf)
    assert out is buf
    want = oracle.birchfield(L, R, 16, out=np.full((10, 30, 16), 123.0, dtype=np.float32))
    assert_same_bits(out.cpu().numpy(), want, "birchfield reuse")


# ---------------------------------------------------------------------------- SGM
SGM_CASES = [(24, 40, 16), (40, 24, 16), (20, 20, 16), (21, 20, 8), (20, 22, 64), (11, 37, 64),
             (37, 11, 64), (13, 30, 128), (30, 13, 128), (9, 21, 256), (8, 12, 512),
             (10, 14, 100), (7, 9, 2), (6, 8, 1), (1, 9, 16), (9, 1, 16), (2, 2, 64)]


SGM_CASES += [(70, 200, 64), (200, 70, 64), (40, 90, 128), (90, 40, 128), (35, 50, 256),
              (50, 35, 256), (100, 101, 64), (101, 100, 64), (102, 100, 64), (33, 34, 512)]


@pytest.mark.parametrize("H,W,D", SGM_CASES)
@pytest.mark.parametrize("single", [False, True])
def test_sgm_on_ssd_volume_matches_oracle(device, oracle, H, W, D, single):
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=D + H)
    C = oracle.ssd(L, R, D, 3)
    want = oracle.sgm(C, L, 0.1, 0.2)
    sgm = stereomatch.aggregation.Semiglobal(0.1, 0.2, single_passes=single)
    got = sgm(_gpu(C, device), _gpu(L, device))
    assert_same_bits(got.cpu().numpy(), want, "sgm ssd-volume")


@pytest.mark.parametrize("H,W,D", SGM_CASES[:12] + SGM_CASES[17:23])
@pytest.mark.parametrize("kind", ["float", "ties", "last"])
def test_sgm_on_random_volume_matches_oracle(device, oracle, H, W, D, kind):
    import stereomatch
    rng = np.random.default_rng(17 * H + W + D)
    C = _random_volume(rng, H, W, D, kind)
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)   # many zero gradients
    want = oracle.sgm(C, image, 0.5, 3.0)
    got = stereomatch.aggregation.Semiglobal(0.5, 3.0)(_gpu(C, device), _gpu(image, device))
    assert_same_bits(got.cpu().numpy(), want, "sgm " + kind)


@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (40, 24, 16), (20, 22, 64), (13, 30, 128),
                                    (9, 21, 256), (10, 14, 100), (7, 9, 2), (33, 34, 512),
                                    (70, 200, 64), (50, 35, 256)])
@pytest.mark.parametrize("single", [False, True])
def test_sgm_exact_nonfinite_on_inf_volumes(device, oracle, H, W, D, single):
    """STM_SGM_EXACT_NONFINITE: volumes with 30 % +inf entries and all-+inf pixels, NOT patched
    to keep d = 0 finite.  The recurrence then makes (C + inf) - inf; the reference's
    strict-'<' minima (src/semiglobal.cpp:11-16,130) decide where NaN, -inf and +inf land,
    and the flag reproduces that placement (a NaN equals any NaN; everything else bit for
    bit), in the sum and in the fused reduction."""
    import stereomatch
    rng = np.random.default_rng(31 * H + W + D)
    C = _random_volume(rng, H, W, D, "inf")
    for _ in range(4):  # a few more pixels without any finite cost, away from the corner
        C[int(rng.integers(0, H)), int(rng.integers(0, W)), :] = np.inf
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    assert np.isnan(want).any(), "the case is meant to produce NaNs"
    sgm = stereomatch.aggregation.Semiglobal(0.5, 3.0, single_passes=single, exact_nonfinite=True)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert_same_bits(got.cpu().numpy(), want, "sgm exact non-finite")
    if not single:
        _, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
        assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "fused wta on NaN-bearing sums")


def test_sgm_default_nonfinite_contract(device, oracle):
    """The documented default (include/stereomatch_hip.h, STM_SGM_EXACT_NONFINITE): with an
    all-+inf pixel only the pixels a path reaches AFTER it may differ from the reference; all
    other pixels' sums are the reference's bit for bit."""
    import stereomatch
    H, W, D = 21, 33, 64
    rng = np.random.default_rng(7)
    C = _random_volume(rng, H, W, D, "float")
    py, px = 9, 14
    C[py, px, :] = np.inf
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    got = stereomatch.aggregation.Semiglobal(0.5, 3.0)(_gpu(C, device), _gpu(image, device)).cpu().numpy()
    yy, xx = np.mgrid[0:H, 0:W]
    on_a_line = (yy == py) | (xx == px) | (yy - py == xx - px) | (yy - py == px - xx)
    assert_same_bits(got[~on_a_line], want[~on_a_line], "pixels off the eight lines")
    exact = stereomatch.aggregation.Semiglobal(0.5, 3.0, exact_nonfinite=True)(
        _gpu(C, device), _gpu(image, device)).cpu().numpy()
    assert_same_bits(exact, want, "exact flag, whole volume")


def test_sgm_reference_ones_arange_case(device, oracle):
    """tests/test_semiglobalmatch.py:39-55: ones image, arange volume."""
    import stereomatch
    H = W = 60
    image = np.ones((H, W), dtype=np.float32)
    C = np.arange(H * W * 128, dtype=np.float32).reshape(H, W, 128)
    got = stereomatch.aggregation.Semiglobal()(_gpu(C, device), _gpu(image, device))
    assert_same_bits(got.cpu().numpy(), oracle.sgm(C, image), "sgm ones/arange")


def test_sgm_reuses_output_buffer(device, oracle):
    import stereomatch
    L, R = oracle.synthetic_pair(16, 20, seed=2)
    C = oracle.ssd(L, R, 16, 3)
    sgm = stereomatch.aggregation.Semiglobal()
    buf = torch.full((16, 20, 16), 7.0, device=device)
    out = sgm(_gpu(C, device), _gpu(L, device), buf)
    assert out is buf
    assert_same_bits(out.cpu().numpy(), oracle.sgm(C, L), "sgm reuse")


# ----------------------------------------------------------------------------- DP
DP_CASES = [(20, 30, 16), (24, 40, 16), (64, 64, 64), (16, 16, 16), (130, 140, 128),
            (70, 90, 40), (8, 300, 4), (5, 9, 1), (260, 270, 256)]


@pytest.mark.parametrize("H,W,D", DP_CASES)
@pytest.mark.parametrize("kind", ["float", "ties", "inf"])
def test_dp_reference_mode_matches_oracle(device, oracle, H, W, D, kind):
    import stereomatch
    rng = np.random.default_rng(H + 3 * W + D)
    vol = _random_volume(rng, H, W, D, kind)
    want, wpath, wfinal = oracle.dp(vol, mode=0, return_aux=True)
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    got = dp(_gpu(vol, device))
    assert_same_bits(dp.row_final_costs.cpu().numpy(), wfinal, "dp row_final")
    assert_same_bits(dp.path_volume.cpu().numpy(), wpath, "dp path")
    assert_same_bits(got.cpu().numpy(), want, "dp disparity")


@pytest.mark.parametrize("H,W,D", [(10, 7, 16), (5, 40, 64), (30, 12, 128), (3, 2, 8)])
def test_dp_corrected_mode_matches_oracle(device, oracle, H, W, D):
    import stereomatch
    rng = np.random.default_rng(H * W + D)
    vol = _random_volume(rng, H, W, D, "float")
    want, wpath, wfinal = oracle.dp(vol, mode=1, return_aux=True)
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")
    got = dp(_gpu(vol, device))
    assert_same_bits(dp.path_volume.cpu().numpy(), wpath, "dp path")
    assert_same_bits(got.cpu().numpy(), want, "dp corrected")


def test_dp_reference_mode_needs_D_le_H_le_W(device):
    import stereomatch
    with pytest.raises(RuntimeError, match="D <= H <= W"):
        stereomatch.disparity_reduce.DynamicProgramming()(torch.zeros(8, 4, 16, device=device))


def test_dp_after_sgm_long_chains(device, oracle):
    """Smooth volumes make the in-place chain long; the fixed-point iteration must still
    reproduce it."""
    import stereomatch
    H, W, D = 70, 80, 64
    d = np.arange(D, dtype=np.float32)
    vol = np.broadcast_to(0.001 * d, (H, W, D)).astype(np.float32).copy()
    vol += np.random.default_rng(0).random((H, W, 1), dtype=np.float32)
    want, wpath, _ = oracle.dp(vol, mode=0, return_aux=True)
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    got = dp(_gpu(vol, device))
    assert_same_bits(dp.path_volume.cpu().numpy(), wpath, "dp path (chains)")
    assert_same_bits(got.cpu().numpy(), want, "dp chains")


# ------------------------------------------------------------------------ batching
def test_batched_calls_equal_per_pair_calls(device, oracle):
    import stereomatch
    H, W, D, B = 18, 26, 64, 3
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in range(B)]
    L = _gpu(np.stack([p[0] for p in pairs]), device)
    R = _gpu(np.stack([p[1] for p in pairs]), device)
    ssd = stereomatch.cost.SSD(D, kernel_size=3)
    sgm = stereomatch.aggregation.Semiglobal()
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")
    bt = stereomatch.cost.Birchfield(D)
    C = ssd(L, R)
    S = sgm(C, L)
    disp = wta(S)
    dpd = dp(S)
    B_ = bt(L, R)
    for b in range(B):
        Cb = oracle.ssd(pairs[b][0], pairs[b][1], D, 3)
        Sb = oracle.sgm(Cb, pairs[b][0])
        assert_same_bits(C[b].cpu().numpy(), Cb, "batched ssd")
        assert_same_bits(S[b].cpu().numpy(), Sb, "batched sgm")
        assert_same_bits(disp[b].cpu().numpy(), oracle.wta(Sb), "batched wta")
        assert_same_bits(dpd[b].cpu().numpy(), oracle.dp(Sb, mode=1), "batched dp")
        assert_same_bits(B_[b].cpu().numpy(), oracle.birchfield(pairs[b][0], pairs[b][1], D),
                         "batched birchfield")


# ------------------------------------------------------- committed golden fixtures
@pytest.mark.parametrize("name", ["tiny_wide", "tiny_tall", "dp_square"])
def test_gpu_matches_committed_golden_vectors(device, name):
    """tests/golden/*.npz hold full tensors made by make_golden.py from the pinned oracle;
    this comparison does not involve the oracle at run time."""
    import os
    import stereomatch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    D, k = int(z["D"]), int(z["k"])
    L, R = _gpu(z["left"], device), _gpu(z["right"], device)
    for mode in ("ordered", "box", "auto"):
        C = stereomatch.cost.SSD(D, kernel_size=k, mode=mode)(L, R)
        assert_same_bits(C.cpu().numpy(), z["ssd"], "golden ssd " + mode)
    assert_same_bits(stereomatch.cost.Birchfield(D)(L, R).cpu().numpy(), z["birchfield"], "golden bt")
    S = stereomatch.aggregation.Semiglobal()(C, L)
    assert_same_bits(S.cpu().numpy(), z["sgm"], "golden sgm")
    wta = stereomatch.disparity_reduce.WinnerTakesAll()(S)
    assert_same_bits(wta.cpu().numpy(), z["wta"], "golden wta")
    dpc = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")(S)
    assert_same_bits(dpc.cpu().numpy(), z["dp_corrected"], "golden dp corrected")
    if "dp" in z.files:
        dpr = stereomatch.disparity_reduce.DynamicProgramming()(S)
        assert_same_bits(dpr.cpu().numpy(), z["dp"], "golden dp")


# ------------------------------------------------------ fused aggregation + reduction
@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (40, 24, 16), (30, 30, 64), (20, 70, 128),
                                    (70, 20, 128), (12, 40, 256), (9, 11, 100), (31, 30, 64)])
@pytest.mark.parametrize("keep_sum", [False, True])
def test_fused_sgm_wta_equals_separate_calls(device, oracle, H, W, D, keep_sum):
    import stereomatch
    rng = np.random.default_rng(H + W + D)
    C = _random_volume(rng, H, W, D, "ties")   # plenty of exact ties for the index rule
    image = rng.integers(0, 6, size=(H, W)).astype(np.float32)
    want_S = oracle.sgm(C, image, 0.1, 0.2)
    want_d = oracle.wta(want_S)
    sgm = stereomatch.aggregation.Semiglobal()
    vol, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device), keep_sum=keep_sum)
    assert_same_bits(disp.cpu().numpy(), want_d, "fused disparity")
    if keep_sum:
        assert_same_bits(vol.cpu().numpy(), want_S, "fused kept sum")


def test_pipeline_fuses_by_default_and_matches_unfused(device, oracle):
    import stereomatch
    L, R = oracle.synthetic_pair(40, 64, seed=9)
    Lg, Rg = _gpu(L, device), _gpu(R, device)
    mk = lambda fuse: stereomatch.Pipeline(stereomatch.cost.SSD(32),
                                           stereomatch.disparity_reduce.WinnerTakesAll(),
                                           stereomatch.aggregation.Semiglobal(), fuse=fuse)
    a, b = mk(True), mk(False)
    da, db = a.estimate(Lg, Rg), b.estimate(Lg, Rg)
    assert torch.equal(da, db)
    assert_same_bits(db.cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(L, R, 32), L)), "pipeline")
    assert a.estimate(Lg, Rg) is da      # buffers are reused in the fused path too
    with pytest.raises(RuntimeError):    # the reducer's power-of-two rule still applies
        stereomatch.Pipeline(stereomatch.cost.SSD(24), stereomatch.disparity_reduce.WinnerTakesAll(),
                             stereomatch.aggregation.Semiglobal()).estimate(Lg, Rg)


# --------------------------------------------- stream ordering and graph capture
def _pipeline_ops(D):
    import stereomatch
    return (stereomatch.cost.SSD(D), stereomatch.cost.Birchfield(D),
            stereomatch.aggregation.Semiglobal(), stereomatch.disparity_reduce.WinnerTakesAll(),
            stereomatch.disparity_reduce.DynamicProgramming(mode="corrected"))


def test_ops_run_on_the_current_stream(device, oracle):
    """The library launches on the stream it is handed (torch's current stream), not on
    the legacy default stream like the reference."""
    import stereomatch
    L, R = oracle.synthetic_pair(30, 44, seed=5)
    side = torch.cuda.Stream(device=device)
    Lg, Rg = _gpu(L, device), _gpu(R, device)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        pipe = stereomatch.Pipeline(stereomatch.cost.SSD(32), stereomatch.disparity_reduce.WinnerTakesAll(),
                                    stereomatch.aggregation.Semiglobal())
        disp = pipe.estimate(Lg, Rg)
    side.synchronize()
    assert_same_bits(disp.cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(L, R, 32), L)), "side stream")


def test_pipeline_can_be_captured_into_a_graph(device, oracle):
    """No entry point allocates, copies or synchronises, so a whole estimate() can be
    captured once and replayed on new inputs (include/stereomatch_hip.h, conventions)."""
    import stereomatch
    H, W, D = 36, 52, 64
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in (1, 2)]
    Lg, Rg = _gpu(pairs[0][0], device), _gpu(pairs[0][1], device)
    pipe = stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal())
    pipe.estimate(Lg, Rg)                       # allocates the three buffers and workspaces
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pipe.estimate(Lg, Rg)
    for L, R in pairs[::-1]:
        Lg.copy_(_gpu(L, device))
        Rg.copy_(_gpu(R, device))
        graph.replay()
        torch.cuda.synchronize()
        assert_same_bits(out.cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(L, R, D), L)), "graph replay")


def test_non_contiguous_inputs_are_accepted(device, oracle):
    """The reference honours strides through accessors; the wrappers make a contiguous copy."""
    import stereomatch
    L, R = oracle.synthetic_pair(20, 64, seed=8)
    wide = _gpu(np.concatenate([L, R], axis=1), device)
    Lv, Rv = wide[:, :64], wide[:, 64:]
    assert not Lv.is_contiguous()
    C = stereomatch.cost.SSD(16, kernel_size=3)(Lv, Rv)
    assert_same_bits(C.cpu().numpy(), oracle.ssd(L, R, 16, 3), "strided ssd")
    S = stereomatch.aggregation.Semiglobal()(C, Lv)
    assert_same_bits(S.cpu().numpy(), oracle.sgm(oracle.ssd(L, R, 16, 3), L), "strided sgm")


# --------------------------------------------------------------- degenerate shapes
@pytest.mark.parametrize("H,W,D", [(1, 1, 64), (1, 1, 4), (5, 7, 64), (3, 70, 128), (2, 3, 256),
                                    (65, 2, 64), (4, 130, 192)])
def test_degenerate_shapes_full_chain(device, oracle, H, W, D):
    """Images narrower than the window or the disparity range, single pixels, D that is a
    multiple of 64 but not a power of two."""
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=H * W + D)
    Lg, Rg = _gpu(L, device), _gpu(R, device)
    for mode in ("auto", "ordered", "box"):
        C = stereomatch.cost.SSD(D, mode=mode)(Lg, Rg)
        assert_same_bits(C.cpu().numpy(), oracle.ssd(L, R, D, 7), "ssd " + mode)
    Bt = stereomatch.cost.Birchfield(D)(Lg, Rg)
    assert_same_bits(Bt.cpu().numpy(), oracle.birchfield(L, R, D), "birchfield")
    Ch = oracle.ssd(L, R, D, 7)
    for single in (False, True):
        S = stereomatch.aggregation.Semiglobal(single_passes=single)(C, Lg)
        assert_same_bits(S.cpu().numpy(), oracle.sgm(Ch, L), "sgm single=%s" % single)
    Sh = oracle.sgm(Ch, L)
    wta = stereomatch.disparity_reduce.WinnerTakesAll(require_power_of_two=False)
    assert_same_bits(wta(S).cpu().numpy(), oracle.wta(Sh), "wta")
    _, fused = stereomatch.aggregation.Semiglobal().aggregate_and_reduce(C, Lg)
    assert_same_bits(fused.cpu().numpy(), oracle.wta(Sh), "fused wta")
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")
    assert_same_bits(dp(S).cpu().numpy(), oracle.dp(Sh, mode=1), "dp corrected")


# ------------------------------------------- corrected-algorithm options (extensions)
@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (40, 24, 16), (30, 30, 64), (13, 50, 128), (50, 13, 128),
                                    (9, 21, 256), (6, 8, 1), (10, 14, 100)])
@pytest.mark.parametrize("paths,full_min", [("hirschmuller", False), ("reference", True),
                                             ("hirschmuller", True)])
@pytest.mark.parametrize("single", [False, True])
def test_sgm_corrected_modes_match_oracle(device, oracle, H, W, D, paths, full_min, single):
    import stereomatch
    rng = np.random.default_rng(H + 2 * W + D)
    C = _random_volume(rng, H, W, D, "last")     # minima at d = D-1 make full_min matter
    image = rng.integers(0, 9, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.1, 0.2, full_min=full_min, true_paths=(paths == "hirschmuller"))
    if D > 1 and (full_min or (paths == "hirschmuller" and H != W)):
        assert not np.array_equal(want, oracle.sgm(C, image, 0.1, 0.2))   # the option matters
    sgm = stereomatch.aggregation.Semiglobal(0.1, 0.2, single_passes=single, paths=paths,
                                             full_min=full_min)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert_same_bits(got.cpu().numpy(), want, "sgm %s full_min=%s" % (paths, full_min))
    if not single:
        _, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
        assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "fused, corrected modes")


def test_true_paths_visit_every_pixel_eight_times(device):
    """With unit costs in a 1-disparity volume every pass adds exactly 1 per visited pixel:
    the corrected path set gives 8 everywhere, the reference's does not on non-square images."""
    import stereomatch
    C = torch.ones(20, 33, 1, device=device)
    image = torch.zeros(20, 33, device=device)
    fixed = stereomatch.aggregation.Semiglobal(paths="hirschmuller")(C, image)
    assert torch.equal(fixed, torch.full_like(fixed, 8.0))
    ref = stereomatch.aggregation.Semiglobal()(C, image)
    assert float(ref.min()) == 6.0 and float(ref.max()) == 6.0 + 2 * (33 - 20 + 1)


def test_cli_image_end_to_end(device, oracle, tmp_path):
    """stm-image equivalent: 8-bit PNGs in, 16-bit disparity PNG out, same numbers as the
    oracle chain on the unscaled grey values (upstream cli_image.py:70-83)."""
    from PIL import Image
    from stereomatch import cli_image
    L, R = oracle.synthetic_pair(40, 64, seed=12)
    Image.fromarray(L.astype(np.uint8)).save(tmp_path / "l.png")
    Image.fromarray(R.astype(np.uint8)).save(tmp_path / "r.png")
    out = tmp_path / "d.png"
    assert cli_image.main([str(tmp_path / "l.png"), str(tmp_path / "r.png"), "32", str(out),
                           "-am", "sgm", "-c"]) == 0
    got = np.array(Image.open(out)).astype(np.int32)
    want = oracle.wta(oracle.sgm(oracle.ssd(L, R, 32), L))
    assert np.array_equal(got, want)
    cli_image.main([str(tmp_path / "l.png"), str(tmp_path / "r.png"), "16", str(tmp_path / "d.npy"),
                    "-cm", "birchfield", "-dm", "dyn"])
    assert np.load(tmp_path / "d.npy").shape == (40, 64)


def test_integration_stub_presents_the_reference_extension_api(device, oracle):
    """a module with the reference's three static-method classes
    (src/_cstereomatch.cpp:10-17) over the C ABI, called with the reference's argument lists
    and caller-allocated outputs (cost.py:44-46, aggregation.py:51-55, disparity_reduce.py:45,77-88)."""
    import cstereomatch_hip_stub as ext
    H, W, D = 24, 40, 16
    L, R = oracle.synthetic_pair(H, W, seed=0)
    Lg, Rg = _gpu(L, device), _gpu(R, device)
    cost = torch.empty(H, W, D, device=device)
    ext.CostOps.compute_ssd(Lg, Rg, cost, 7)
    assert_same_bits(cost.cpu().numpy(), oracle.ssd(L, R, D, 7), "stub ssd")
    bt = torch.zeros(H, W, D, device=device)
    ext.CostOps.compute_birchfield(Lg, Rg, bt, 4)
    assert_same_bits(bt.cpu().numpy(), oracle.birchfield(L, R, D), "stub birchfield")
    agg = torch.zeros_like(cost)                       # zeros_tensor_like, then "+=" semantics
    ext.AggregationOps.run_semiglobal(cost, Lg, 0.1, 0.2, agg)
    want = oracle.sgm(oracle.ssd(L, R, D, 7), L)
    assert_same_bits(agg.cpu().numpy(), want, "stub sgm")
    disp = torch.empty(H, W, dtype=torch.int32, device=device)
    ext.DisparityReduceOps.run_winners_take_all(agg, disp)
    assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "stub wta")
    path = torch.empty(H, W, D, dtype=torch.int8, device=device)      # torch.empty, as upstream
    final = torch.empty(H, D, device=device)
    ext.DisparityReduceOps.run_dynamic_programming(agg, path, final, disp)
    assert_same_bits(disp.cpu().numpy(), oracle.dp(want), "stub dp")
    with pytest.raises(RuntimeError):
        ext.DisparityReduceOps.run_dynamic_programming(torch.zeros(4, 2, 8, device=device), path, final, disp)


# ------------------------------------------------- widest disparity ranges, batched quirks
@pytest.mark.parametrize("H,W,D", [(6, 10, 1024), (5, 9, 768), (12, 7, 512)])
def test_largest_disparity_ranges(device, oracle, H, W, D):
    """D = 512..1024 exercise the 8- and 16-values-per-lane kernels (the 32/64-step DP tiles
    only with fewer than 16 columns: test_dp_wide_tiles covers them properly)."""
    import stereomatch
    rng = np.random.default_rng(D + H)
    C = _random_volume(rng, H, W, D, "float")
    image = rng.integers(0, 7, size=(H, W)).astype(np.float32)
    Cg, Ig = _gpu(C, device), _gpu(image, device)
    want = oracle.sgm(C, image)
    for single in (False, True):
        S = stereomatch.aggregation.Semiglobal(single_passes=single)(Cg, Ig)
        assert_same_bits(S.cpu().numpy(), want, "sgm D=%d single=%s" % (D, single))
    wta = stereomatch.disparity_reduce.WinnerTakesAll(require_power_of_two=False)
    assert_same_bits(wta(S).cpu().numpy(), oracle.wta(want), "wta")
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode="corrected")
    assert_same_bits(dp(S).cpu().numpy(), oracle.dp(want, mode=1), "dp")
    L, R = oracle.synthetic_pair(H, W, seed=D)
    got = stereomatch.cost.SSD(D)(_gpu(L, device), _gpu(R, device))
    assert_same_bits(got.cpu().numpy(), oracle.ssd(L, R, D, 7), "ssd")
    bt = stereomatch.cost.Birchfield(D)(_gpu(L, device), _gpu(R, device))
    assert_same_bits(bt.cpu().numpy(), oracle.birchfield(L, R, D), "birchfield")


@pytest.mark.parametrize("H,W,D,mode", [
    (40, 70, 576, "corrected"),    # O = 64: the lane's column wraps (16 columns per period)
    (30, 50, 384, "corrected"),    # O = 32
    (9, 100, 1024, "corrected"),   # O = 64, six periods
    (520, 530, 512, "reference"),  # O = 32 with the transposed initial read and the Q11b clobber
    (1030, 1040, 1024, "reference"),  # O = 64, the largest reference-mode shape (D <= H <= W)
])
def test_dp_wide_tiles(device, oracle, H, W, D, mode):
    """dp_tile_kernel<32> and <64> on rows longer than one 16-column period, in both modes:
    path volume, final costs and disparities bit for bit."""
    import stereomatch
    rng = np.random.default_rng(D + W)
    vol = rng.random((H, W, D), dtype=np.float32)
    vol *= np.float32(50.0)
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode=mode)
    got = dp(_gpu(vol, device))
    want, wpath, wfin = oracle.dp(vol, mode=1 if mode == "corrected" else 0, return_aux=True)
    assert_same_bits(got.cpu().numpy(), want, "dp %s O-tile" % mode)
    assert_same_bits(dp.row_final_costs.cpu().numpy(), wfin, "dp row_final")
    gpath = dp.path_volume.cpu().numpy()
    assert np.array_equal(gpath, wpath), "dp path volume"


@pytest.mark.parametrize("lanes", ["16", "64", None])
@pytest.mark.parametrize("B,H,W,D,mode", [
    (1, 65, 65, 64, "reference"),     # the whole-wave form's narrowest image (W = 65), D < its period of 256
    (2, 33, 131, 96, "corrected"),    # D not a power of two, rows not a multiple of four, a batch
    (1, 260, 300, 256, "reference"),  # every chunk of the period in use; the Q11a/b quirks
    (3, 5, 200, 128, "corrected"),    # three columns per lane and a bit
    (1, 1, 97, 32, "corrected"),      # one row, one line of disparities
    (1, 40, 130, 512, "corrected"),   # O = 8 on the whole wave
])
def test_dp_forward_forms(device, oracle, monkeypatch, lanes, B, H, W, D, mode):
    """The forward pass's two wavefront forms (a DPP row of 16 lanes per image row, four rows per
    wave / the whole wave on one row -- picked by the number of rows in the call, forced here
    through STM_DP_LANES) give the same path volume, final costs and disparities bit for bit."""
    import stereomatch
    if lanes is None:
        monkeypatch.delenv("STM_DP_LANES", raising=False)
    else:
        monkeypatch.setenv("STM_DP_LANES", lanes)
    rng = np.random.default_rng(B * 7 + H + W + D)
    vols = [_random_volume(rng, H, W, D, kind) for kind in ("float", "ties", "inf")[:B]]
    while len(vols) < B:
        vols.append(_random_volume(rng, H, W, D, "float"))
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode=mode)
    got = dp(_gpu(np.stack(vols) if B > 1 else vols[0], device))
    for b, vol in enumerate(vols):
        want, wpath, wfin = oracle.dp(vol, mode=1 if mode == "corrected" else 0, return_aux=True)
        g = got[b] if B > 1 else got
        gp = dp.path_volume[b] if B > 1 else dp.path_volume
        gf = dp.row_final_costs[b] if B > 1 else dp.row_final_costs
        assert_same_bits(gf.cpu().numpy(), wfin, "dp row_final (pair %d)" % b)
        assert np.array_equal(gp.cpu().numpy(), wpath), "dp path volume (pair %d)" % b
        assert_same_bits(g.cpu().numpy(), want, "dp disparity (pair %d)" % b)


@pytest.mark.parametrize("H,W", [(14, 31), (31, 14)])
def test_batched_sgm_with_rewalked_diagonals(device, oracle, H, W):
    """Wide and tall images in a batch: every pair has its own spill rows and checkpoints."""
    import stereomatch
    D, B = 64, 3
    rng = np.random.default_rng(H)
    C = np.stack([_random_volume(rng, H, W, D, "float") for _ in range(B)])
    image = rng.integers(0, 5, size=(B, H, W)).astype(np.float32)
    for single in (False, True):
        S = stereomatch.aggregation.Semiglobal(single_passes=single)(_gpu(C, device), _gpu(image, device))
        for b in range(B):
            assert_same_bits(S[b].cpu().numpy(), oracle.sgm(C[b], image[b]), "batched sgm %d" % b)
    _, disp = stereomatch.aggregation.Semiglobal().aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
    for b in range(B):
        assert_same_bits(disp[b].cpu().numpy(), oracle.wta(oracle.sgm(C[b], image[b])), "batched fused")


@pytest.mark.parametrize("H,W,D", [(24, 40, 16), (7, 200, 128), (5, 70, 256), (9, 50, 20), (3, 3, 64)])
@pytest.mark.parametrize("kind", ["u8", "unit"])
def test_birchfield_padded_indexing_matches_oracle(device, oracle, H, W, D, kind):
    """Opt-in correction: the CUDA functor's sample indexing (src/birchfield_cost.cu:92-139)."""
    import stereomatch
    L, R = oracle.synthetic_pair(H, W, seed=3 + D, kind=kind)
    want = oracle.birchfield(L, R, D, padded=True)
    got = stereomatch.cost.Birchfield(D, indexing="padded")(_gpu(L, device), _gpu(R, device))
    assert_same_bits(got.cpu().numpy(), want, "birchfield padded")
    if W > 8:
        assert not np.array_equal(want, oracle.birchfield(L, R, D))


def test_evaluation_over_a_synthetic_middlebury_tree(device, oracle, tmp_path):
    """The evaluation driver (upstream workflows/evaluation/workflow.py) on two synthetic
    scenes with constant true disparity 7: metrics equal those computed from the oracle's maps."""
    from PIL import Image
    from stereomatch.data import MiddleburyDataset, disparity_metrics, write_pfm
    from stereomatch.evaluation import evaluate_dataset, to_markdown
    H, W, ndisp = 40, 72, 30          # ndisp 30 -> D = 32; D <= H <= W for the reference DP
    for s in range(2):
        d = tmp_path / ("scene%d" % s)
        d.mkdir()
        L, R = oracle.synthetic_pair(H, W, seed=s)
        Image.fromarray(np.stack([L] * 3, axis=-1).astype(np.uint8)).save(d / "im0.png")
        Image.fromarray(np.stack([R] * 3, axis=-1).astype(np.uint8)).save(d / "im1.png")
        write_pfm(d / "disp0.pfm", np.full((H, W), 7.0, dtype=np.float32))
        (d / "calib.txt").write_text("width=%d\nheight=%d\nndisp=%d\n" % (W, H, ndisp))
    ds = MiddleburyDataset(str(tmp_path))
    rows = evaluate_dataset(ds, device=str(device))
    assert [r["pipeline"] for r in rows] == ["ssd-wta-None", "ssd-dyn-None", "ssd-dyn-sgm"]
    # Independent expectation for the first configuration from the oracle.
    want = 0.0
    for s in range(2):
        L, R = oracle.synthetic_pair(H, W, seed=s)
        grey = lambda a: (0.2989 * a + 0.587 * a + 0.114 * a).astype(np.float32)
        disp = oracle.wta(oracle.ssd(grey(L), grey(R), 32, 7))
        want += disparity_metrics(disp, np.full((H, W), 7.0), bad_thresholds=(2.0,))["rmse"]
    assert abs(rows[0]["rmse"] - want / 2) < 1e-9
    assert "ssd-dyn-sgm" in to_markdown(rows)


@pytest.mark.gpu
def test_graft_entry_smoke_runs(device):
    """The driver's round-end smoke() is exercised here so it cannot drift from the ops."""
    import __graft_entry__ as entry
    entry.smoke()


def test_float64_is_declined_with_the_reference_dispatch_named(device):
    """The reference dispatches float64 for SGM, WTA and DP (AT_DISPATCH_FLOATING_TYPES);
    the MI355X path is float32 only and says so -- it must not fall back to anything."""
    import stereomatch
    vol = torch.zeros(20, 24, 16, dtype=torch.float64, device=device)
    img = torch.zeros(20, 24, device=device)
    for call, cite in ((lambda: stereomatch.aggregation.Semiglobal()(vol, img), "semiglobal.cpp:182"),
                       (lambda: stereomatch.disparity_reduce.WinnerTakesAll()(vol), "winners_take_all.cu:99"),
                       (lambda: stereomatch.disparity_reduce.DynamicProgramming()(vol),
                        "dynamic_programming.cu:199")):
        with pytest.raises(RuntimeError) as e:
            call()
        assert "float64" in str(e.value) and cite in str(e.value)
    # Birchfield casts its images like the reference wrapper (cost.py:98-99): float64 in, float32 out.
    out = stereomatch.cost.Birchfield(16)(img.double(), img.double())
    assert out.dtype == torch.float32


@pytest.mark.parametrize("H,W,D,mode", [(40, 48, 32, "reference"), (40, 70, 128, "corrected"), (300, 320, 256, "reference")])
def test_dp_nan_contract(device, oracle, H, W, D, mode):
    """Non-finite inputs of the dynamic programming, as DESIGN.md section 2 documents them: +inf
    costs are the reference's bit for bit (tested above); a NaN is an operand the kernel's
    three-way minimum IGNORES (v_min_f32) where the reference's strict-'<' chain
    (src/dynamic_programming.cu:50-59) lets it decide -- so a row that holds a NaN may differ
    from the reference, and only such a row: rows are independent, every other row of the volume
    is the reference's in disparity, path record and final costs, and the affected rows still
    produce valid records (directions in {-1, 0, 1}, disparities in [0, D))."""
    import stereomatch
    rng = np.random.default_rng(H + W + D)
    C = (rng.random((H, W, D)) * 30).astype(np.float32)
    clean = C.copy()
    bad_rows = sorted(rng.choice(np.arange(D if mode == "reference" else 0, H), size=3, replace=False).tolist())
    for r in bad_rows:  # (reference mode reads cost[d][row][0] of rows < D for every row: keep those clean)
        C[r, rng.integers(1, W, size=5), rng.integers(0, D, size=5)] = np.nan
    dp = stereomatch.disparity_reduce.DynamicProgramming(mode=mode)
    got = dp(_gpu(C, device)).cpu().numpy()
    path = dp.path_volume.cpu().numpy()
    want, wpath, wfin = oracle.dp(clean, mode={"reference": 0, "corrected": 1}[mode], return_aux=True)
    good = np.array([r not in bad_rows for r in range(H)])
    assert np.array_equal(got[good], want[good]), "rows without a NaN equal the reference"
    assert got.min() >= 0 and got.max() < D
    assert np.array_equal(path[good], wpath[good])
    assert set(np.unique(path[~good]).tolist()) <= {-1, 0, 1}
    assert_same_bits(dp.row_final_costs.cpu().numpy()[good], wfin[good], "final costs of the rows without a NaN")
