"""Full-size parity on the GPU: Teddy-sized runs against the reference-output digests of
SURVEY.md Appendix C (committed in tests/golden/digests.json) and 1080p runs against
sampled / full oracle results."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "digests.json")) as f:
    GOLDEN = json.load(f)


def digest16(t):
    if isinstance(t, torch.Tensor):
        t = t.cpu().numpy()
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]


def _pair(oracle, H, W, device, seed=0):
    L, R = oracle.synthetic_pair(H, W, seed=seed)
    return L, R, torch.from_numpy(L).to(device), torch.from_numpy(R).to(device)


@pytest.mark.parametrize("case", ["tiny-wide 24x40x16", "tiny-tall 40x24x16",
                                  "teddy 375x450x64", "teddy 375x450x128"])
@pytest.mark.parametrize("mode", ["auto", "ordered"])
def test_pipeline_reproduces_reference_digests(device, oracle, case, mode):
    """Config 1/2: Pipeline(SSD(D), WTA[, SGM]) on the seed-0 pair, twice (buffer reuse)."""
    import stereomatch
    g = GOLDEN[case]
    H, W, D = g["H"], g["W"], g["D"]
    L, R, Lg, Rg = _pair(oracle, H, W, device)
    assert digest16(L) == g["L"] and digest16(R) == g["R"]
    plain = stereomatch.Pipeline(stereomatch.cost.SSD(D, mode=mode),
                                 stereomatch.disparity_reduce.WinnerTakesAll())
    full = stereomatch.Pipeline(stereomatch.cost.SSD(D, mode=mode),
                                stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal())
    fused = stereomatch.Pipeline(stereomatch.cost.SSD(D, mode=mode),
                                 stereomatch.disparity_reduce.WinnerTakesAll(),
                                 stereomatch.aggregation.Semiglobal())
    full.fuse = False
    assert digest16(fused.estimate(Lg, Rg)) == g["ssd_sgm_wta"], "fused SGM+WTA"
    for attempt in range(2):
        d1 = plain.estimate(Lg, Rg)
        assert digest16(plain._cost_volume) == g["ssd"], "SSD volume (attempt %d)" % attempt
        assert digest16(d1) == g["ssd_wta"]
        d2 = full.estimate(Lg, Rg)
        assert digest16(full._aggregation_volume) == g["sgm"], "SGM volume (attempt %d)" % attempt
        assert digest16(d2) == g["ssd_sgm_wta"]
        if attempt == 0:
            ids = (plain._cost_volume.data_ptr(), d1.data_ptr(), full._aggregation_volume.data_ptr())
        else:  # second call reuses the first call's buffers
            assert ids == (plain._cost_volume.data_ptr(), d1.data_ptr(),
                           full._aggregation_volume.data_ptr())


def test_config3_birchfield_sgm_dp_digests(device, oracle):
    """Config 3 in its pinned form: Birchfield(128) -> SGM -> DP."""
    import stereomatch
    g = GOLDEN["config3 375x450x128"]
    L, R, Lg, Rg = _pair(oracle, g["H"], g["W"], device)
    pipe = stereomatch.Pipeline(stereomatch.cost.Birchfield(g["D"]),
                                stereomatch.disparity_reduce.DynamicProgramming(),
                                stereomatch.aggregation.Semiglobal())
    disp = pipe.estimate(Lg, Rg)
    assert digest16(pipe._cost_volume) == g["birchfield"]
    assert digest16(pipe._aggregation_volume) == g["sgm"]
    assert digest16(disp) == g["dp"]
    assert abs(float((disp == 7).float().mean()) - 0.992) < 0.002


def test_1080p_ssd_sampled_and_wta_full(device, oracle):
    """Config 4 stages with local support: SSD checked on 20k sampled entries (both
    modes agree everywhere), WTA checked on every pixel."""
    import stereomatch
    H, W, D = 1080, 1920, 256
    L, R, Lg, Rg = _pair(oracle, H, W, device)
    C_auto = stereomatch.cost.SSD(D, mode="auto")(Lg, Rg)
    C_ord = stereomatch.cost.SSD(D, mode="ordered")(Lg, Rg)
    assert torch.equal(C_auto.view(torch.int32), C_ord.view(torch.int32))
    rng = np.random.default_rng(0)
    n = 20000
    rows = rng.integers(0, H, n)
    cols = rng.integers(0, W, n)
    disps = rng.integers(0, D, n)
    rows[:200] = rng.integers(0, 8, 200)          # top border
    rows[200:400] = rng.integers(H - 8, H, 200)   # bottom border
    cols[400:800] = rng.integers(0, 300, 400)     # left clip / +inf region
    cols[800:1000] = rng.integers(W - 8, W, 200)  # right border
    want = oracle.ssd_entries(L, R, 7, rows, cols, disps)
    got = C_auto[torch.from_numpy(rows).to(device), torch.from_numpy(cols).to(device),
                 torch.from_numpy(disps).to(device)].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    disp = stereomatch.disparity_reduce.WinnerTakesAll()(C_auto)
    assert np.array_equal(disp.cpu().numpy(), oracle.wta(C_auto.cpu().numpy()))


def test_1080p_sgm_full_volume_matches_oracle(device, oracle):
    """Config 4 aggregation, every element: the oracle takes ~15 s on one core."""
    import stereomatch
    H, W, D = 1080, 1920, 256
    L, R, Lg, Rg = _pair(oracle, H, W, device)
    C = stereomatch.cost.SSD(D)(Lg, Rg)
    S = stereomatch.aggregation.Semiglobal()(C, Lg)
    S1 = stereomatch.aggregation.Semiglobal(single_passes=True)(C, Lg)
    assert torch.equal(S.view(torch.int32), S1.view(torch.int32))   # paired == eight passes
    del S1
    disp = stereomatch.disparity_reduce.WinnerTakesAll()(S)
    _, fused_disp = stereomatch.aggregation.Semiglobal().aggregate_and_reduce(C, Lg)
    assert torch.equal(fused_disp, disp)                            # fused reduction == separate
    Ch = C.cpu().numpy()
    del C
    want = oracle.sgm(Ch, L, 0.1, 0.2)
    got = S.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(disp.cpu().numpy(), oracle.wta(want))
    assert float((disp == 7).float().mean()) > 0.9


def test_1080p_dynamic_programming_matches_oracle_in_both_forward_forms(device, oracle, monkeypatch):
    """The DP reduce of one 1080p x 256 volume (the call shape that takes the whole-wave forward
    pass) against the oracle -- path volume, final costs, disparities -- and the same volume
    through the narrow form (forced) bit for bit."""
    import stereomatch
    H, W, D = 1080, 1920, 256
    L, R, Lg, Rg = _pair(oracle, H, W, device)
    C = stereomatch.cost.Birchfield(D)(Lg, Rg)
    monkeypatch.delenv("STM_DP_LANES", raising=False)
    dp = stereomatch.disparity_reduce.DynamicProgramming()
    disp = dp(C)
    path, fin = dp.path_volume.clone(), dp.row_final_costs.clone()
    monkeypatch.setenv("STM_DP_LANES", "16")
    dp16 = stereomatch.disparity_reduce.DynamicProgramming()
    disp16 = dp16(C)
    assert torch.equal(disp16, disp) and torch.equal(dp16.path_volume, path)
    assert torch.equal(dp16.row_final_costs.view(torch.int32), fin.view(torch.int32))
    del dp16
    want, wpath, wfin = oracle.dp(C.cpu().numpy(), mode=0, return_aux=True)
    assert np.array_equal(disp.cpu().numpy(), want)
    assert np.array_equal(fin.cpu().numpy().view(np.uint32), wfin.view(np.uint32))
    assert np.array_equal(path.cpu().numpy(), wpath)


def test_teddy_batch_takes_the_streaming_variants_and_matches_single_pairs(device, oracle):
    """A four-pair Teddy batch is past the Infinity-Cache threshold, so SSD stores and the
    aggregation's segment loads/stores use their non-temporal variants: every pair must still
    equal its own single-pair run (cached variants), bit for bit, and pair 0 the reference
    digests."""
    import stereomatch
    g = GOLDEN["teddy 375x450x128"]
    H, W, D = g["H"], g["W"], g["D"]
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in range(4)]
    Lb = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
    Rb = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
    ssd = stereomatch.cost.SSD(D)
    sgm = stereomatch.aggregation.Semiglobal()
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    Cb = ssd(Lb, Rb)
    Sb = sgm(Cb, Lb)
    _, fused_b = sgm.aggregate_and_reduce(Cb, Lb)
    assert digest16(Cb[0]) == g["ssd"] and digest16(Sb[0]) == g["sgm"]
    assert digest16(fused_b[0]) == g["ssd_sgm_wta"]
    for b in range(4):
        C1 = ssd(Lb[b], Rb[b])
        assert torch.equal(C1.view(torch.int32), Cb[b].view(torch.int32)), "ssd pair %d" % b
        S1 = sgm(C1, Lb[b])
        assert torch.equal(S1.view(torch.int32), Sb[b].view(torch.int32)), "sgm pair %d" % b
        assert torch.equal(wta(S1), fused_b[b]), "fused pair %d" % b


with open(os.path.join(HERE, "golden", "oracle_digests.json")) as f:
    ORACLE_GOLDEN = json.load(f)


def test_config5_share_eight_1080p_pairs(device, oracle):
    """Config 5's per-GPU share: a batch of eight 1080p x 256 pairs through SSD -> SGM -> WTA
    (fused and unfused).  Pair 0 must hash to the oracle digests committed in
    tests/golden/oracle_digests.json (cost volume, aggregated volume, disparity map) and
    every pair must equal its own single-pair run bit for bit."""
    import stereomatch
    g = ORACLE_GOLDEN["1080p 1080x1920x256"]
    H, W, D, B = g["H"], g["W"], g["D"], 8
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in range(B)]
    assert digest16(pairs[0][0]) == g["L"] and digest16(pairs[0][1]) == g["R"]
    Lb = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
    Rb = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
    ssd = stereomatch.cost.SSD(D)
    sgm = stereomatch.aggregation.Semiglobal()
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    pipe = stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal())
    fused_b = pipe.estimate(Lb, Rb).clone()
    del pipe
    torch.cuda.empty_cache()
    Cb = ssd(Lb, Rb)
    Sb = sgm(Cb, Lb)
    disp_b = wta(Sb)
    assert torch.equal(disp_b, fused_b)
    assert digest16(Cb[0]) == g["ssd"], "cost volume of pair 0"
    assert digest16(Sb[0]) == g["sgm"], "aggregated volume of pair 0"
    assert digest16(fused_b[0]) == g["ssd_sgm_wta"], "disparity map of pair 0"
    assert abs(float((fused_b[0] == 7).float().mean()) - g["pixels_at_true_disparity"]) < 1e-6
    ssd1 = stereomatch.cost.SSD(D)
    sgm1 = stereomatch.aggregation.Semiglobal()
    C1 = S1 = None
    for b in range(B):
        C1 = ssd1(Lb[b], Rb[b], cost_volume=C1)
        assert torch.equal(C1.view(torch.int32), Cb[b].view(torch.int32)), "ssd pair %d" % b
        S1 = sgm1(C1, Lb[b], S1)
        assert torch.equal(S1.view(torch.int32), Sb[b].view(torch.int32)), "sgm pair %d" % b
        assert torch.equal(wta(S1), fused_b[b]), "disparity pair %d" % b


@pytest.mark.parametrize("form", ["auto", "tiled"])
def test_teddy_large_batch_takes_the_tiled_form_and_matches_single_pairs(device, oracle, form):
    """A 48-pair Teddy batch is what the bench's primary workload looks like to the library: the
    aggregation takes the skewed-tile form by default (two passes over sheared tiles, bands handed
    over between workgroups); forced, the sweep-fed tiled form (row sweeps + two-lines-per-wave
    tiles, workgroups owning tile rows).  Pair 0 must hash to the reference digests, and every pair
    must equal its own single-pair run (paired form), sum volume and fused disparity map, bit for
    bit."""
    import stereomatch
    from stereomatch import _native
    g = GOLDEN["teddy 375x450x128"]
    H, W, D, B = g["H"], g["W"], g["D"], 48
    want_form = _native.STM_SGM_FORM_SKEW if form == "auto" else _native.STM_SGM_FORM_SWEEPS
    assert _native.lib().stm_sgm_form(H, W, D, B, _native.STM_SGM_FORMS[form]) == want_form
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in range(B)]
    Lb = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
    Rb = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
    ssd = stereomatch.cost.SSD(D)
    sgm = stereomatch.aggregation.Semiglobal(form=form)
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    Cb = ssd(Lb, Rb)
    Sb = sgm(Cb, Lb)
    assert sgm.status() == 0 and sgm._last_form == want_form
    _, fused_b = sgm.aggregate_and_reduce(Cb, Lb)
    assert sgm.status() == 0
    assert digest16(Cb[0]) == g["ssd"] and digest16(Sb[0]) == g["sgm"]
    assert digest16(fused_b[0]) == g["ssd_sgm_wta"]
    sgm1 = stereomatch.aggregation.Semiglobal(form="paired")
    S1 = None
    for b in range(B):
        S1 = sgm1(Cb[b], Lb[b], S1)
        assert torch.equal(S1.view(torch.int32), Sb[b].view(torch.int32)), "sgm pair %d" % b
        assert torch.equal(wta(S1), fused_b[b]), "fused pair %d" % b
