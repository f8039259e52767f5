"""The tiled form of the semiglobal aggregation (row sweeps + 8x8 tiles, stm_sgm_tiled.hip)
against the CPU oracle and against the paired form, bit for bit, through the C ABI.

Shapes cover: images narrower / wider than one column strip (hand-offs between workgroups
through global mailboxes), widths and heights that are not multiples of the tile or of a
wave's column count, wide and tall images (the reference's re-walked and skipped inverse
diagonals), every supported disparity count, batches that need more than one
sweep launch, the corrected-algorithm options and the fused winner-takes-all."""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import _gpu, _random_volume, assert_same_bits

pytestmark = pytest.mark.gpu

TILED_CASES = [
    (20, 22, 64), (11, 37, 64), (37, 11, 64), (13, 30, 128), (30, 13, 128), (9, 21, 256),
    (2, 2, 64), (1, 9, 64), (9, 1, 64), (8, 8, 128), (16, 16, 256), (17, 15, 64),
    (70, 200, 64), (200, 70, 64), (40, 90, 128), (90, 40, 128), (35, 50, 256), (50, 35, 256),
    (100, 101, 64), (101, 100, 64), (102, 100, 64), (33, 129, 128), (24, 65, 256),
    (31, 130, 256), (64, 257, 64),
]


def _tiled(form="tiled", **kw):
    import stereomatch
    return stereomatch.aggregation.Semiglobal(form=form, **kw)


# form "tiled": border states from row sweeps; "walks": from per-path walks (stm_sgm_bwalk.hip)
@pytest.mark.parametrize("H,W,D", TILED_CASES)
@pytest.mark.parametrize("kind", ["float", "last"])
@pytest.mark.parametrize("form", ["tiled", "walks"])
def test_tiled_sgm_matches_oracle(device, oracle, H, W, D, kind, form):
    rng = np.random.default_rng(13 * H + 7 * W + D)
    C = _random_volume(rng, H, W, D, kind)
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    sgm = _tiled(form, penalty1=0.5, penalty2=3.0)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert sgm.status() == 0
    assert_same_bits(got.cpu().numpy(), want, "tiled sgm " + kind)
    _, disp = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device))
    assert sgm.status() == 0
    assert_same_bits(disp.cpu().numpy(), oracle.wta(want), "tiled fused wta")
    vol, disp2 = sgm.aggregate_and_reduce(_gpu(C, device), _gpu(image, device), keep_sum=True)
    assert_same_bits(vol.cpu().numpy(), want, "tiled keep_sum volume")
    assert_same_bits(disp2.cpu().numpy(), oracle.wta(want), "tiled keep_sum disparity")


@pytest.mark.parametrize("H,W,D", [(24, 40, 64), (40, 24, 64), (33, 70, 128), (70, 33, 256)])
@pytest.mark.parametrize("paths", ["reference", "hirschmuller"])
@pytest.mark.parametrize("full_min", [False, True])
@pytest.mark.parametrize("form", ["tiled", "walks"])
def test_tiled_sgm_corrected_modes(device, oracle, H, W, D, paths, full_min, form):
    rng = np.random.default_rng(H + W + D)
    C = _random_volume(rng, H, W, D, "ties")
    image = rng.integers(0, 4, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.1, 0.2, full_min=full_min, true_paths=(paths == "hirschmuller"))
    sgm = _tiled(form, paths=paths, full_min=full_min)
    got = sgm(_gpu(C, device), _gpu(image, device))
    assert sgm.status() == 0
    assert_same_bits(got.cpu().numpy(), want, "tiled sgm %s full_min=%s" % (paths, full_min))


@pytest.mark.parametrize("H,W,D,B", [(14, 31, 64, 3), (31, 14, 64, 3), (40, 300, 64, 70), (20, 500, 256, 20),
                                     (40, 90, 128, 5), (100, 60, 128, 50), (60, 100, 128, 9)])
@pytest.mark.parametrize("form", ["tiled", "walks"])
def test_tiled_sgm_batches(device, oracle, H, W, D, B, form):
    """Batches: per-image mailboxes and border states; (40, 300) and (20, 500) need several sweep
    launches (more column strips than the chip has CUs); the D = 128 cases run the tile kernel
    whose workgroups own whole tile rows (more rows than workgroups in (100, 60, 128, 50))."""
    import stereomatch
    rng = np.random.default_rng(B + H)
    C = np.stack([_random_volume(rng, H, W, D, "float") for _ in range(B)])
    image = rng.integers(0, 5, size=(B, H, W)).astype(np.float32)
    Cg, Ig = _gpu(C, device), _gpu(image, device)
    sgm = _tiled(form)
    S = sgm(Cg, Ig)
    assert sgm.status() == 0
    ref = stereomatch.aggregation.Semiglobal(form="paired")(Cg, Ig)
    assert torch.equal(S.view(torch.int32), ref.view(torch.int32)), "tiled == paired on the batch"
    for b in range(0, B, max(1, B // 4)):
        assert_same_bits(S[b].cpu().numpy(), oracle.sgm(C[b], image[b]), "batched tiled sgm %d" % b)
    _, disp = sgm.aggregate_and_reduce(Cg, Ig)
    wta = stereomatch.disparity_reduce.WinnerTakesAll()
    assert torch.equal(disp, wta(ref))


def test_tiled_form_is_refused_where_it_does_not_apply(device):
    """form="tiled" on a shape the tiled form does not cover silently takes the paired form
    (same results); the flag only ever changes speed."""
    import stereomatch
    rng = np.random.default_rng(3)
    C = _random_volume(rng, 12, 20, 48, "float")
    image = rng.integers(0, 5, size=(12, 20)).astype(np.float32)
    a = stereomatch.aggregation.Semiglobal(form="tiled")(_gpu(C, device), _gpu(image, device))
    b = stereomatch.aggregation.Semiglobal(form="paired")(_gpu(C, device), _gpu(image, device))
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        D = int(rng.choice([64, 128, 128, 128, 256]))
        H, W = int(rng.integers(1, 120)), int(rng.integers(1, 180))
        u = rng.random()
        if u < 0.3:    # enough interior tiles for the two-pixels-per-wave forms
            H, W = int(rng.integers(24, 120)), int(rng.integers(24, 180))
        elif u < 0.4:  # very tall
            H, W = int(rng.integers(100, 260)), int(rng.integers(1, 40))
        elif u < 0.5:  # very wide
            H, W = int(rng.integers(1, 40)), int(rng.integers(100, 400))
        elif u < 0.6:  # one to three rows or columns
            if rng.random() < 0.5:
                H = int(rng.integers(1, 4))
            else:
                W = int(rng.integers(1, 4))
        B = int(rng.integers(1, 5))
        cases.append((H, W, D, B, str(rng.choice(["float", "ties", "last"])),
                      str(rng.choice(["reference", "reference", "hirschmuller"])), bool(rng.random() < 0.3)))
    return cases


@pytest.mark.parametrize("seed,form", [(11, "tiled"), (12, "tiled"), (13, "tiled"), (14, "walks"), (15, "walks")])
def test_tiled_fuzz_against_paired(device, seed, form):
    """Random shapes, batches and options: the tiled form against the paired form (which the
    other tests hold against the oracle), sum volume and fused disparities, bit for bit."""
    import stereomatch
    for H, W, D, B, kind, paths, full_min in _fuzz_cases(25, seed):
        rng = np.random.default_rng(H * 7919 + W * 31 + D + B)
        C = np.stack([_random_volume(rng, H, W, D, kind) for _ in range(B)])
        image = rng.integers(0, 6, size=(B, H, W)).astype(np.float32)
        Cg, Ig = _gpu(C, device), _gpu(image, device)
        kw = dict(penalty1=0.25, penalty2=2.0, paths=paths, full_min=full_min)
        t = stereomatch.aggregation.Semiglobal(form=form, **kw)
        p = stereomatch.aggregation.Semiglobal(form="paired", **kw)
        what = "H=%d W=%d D=%d B=%d %s %s full_min=%s" % (H, W, D, B, kind, paths, full_min)
        if os.environ.get("STM_FUZZ_VERBOSE"):
            print(what, flush=True)
        St = t(Cg, Ig)
        if os.environ.get("STM_FUZZ_VERBOSE"):
            torch.cuda.synchronize()
            print("  tiled sum done", flush=True)
        Sp = p(Cg, Ig)
        if os.environ.get("STM_FUZZ_VERBOSE"):
            torch.cuda.synchronize()
            print("  paired sum done", flush=True)
        assert t.status() == 0, what
        assert torch.equal(St.view(torch.int32), Sp.view(torch.int32)), "sum volume: " + what
        _, dt = t.aggregate_and_reduce(Cg, Ig)
        _, dp_ = p.aggregate_and_reduce(Cg, Ig)
        assert torch.equal(dt, dp_), "fused disparities: " + what


def test_auto_form_picks_walks_for_single_pairs_and_sweeps_for_batches(device):
    """What `auto` selects (stm_sgm_uses_tiles + the flags): a single Teddy- or 1080p-sized pair
    takes the tiled form (fed by walks: its status word exists and stays 0), D = 64 the paired
    form; forcing a form through the flags is honoured."""
    from stereomatch import _native
    lib = _native.lib()
    assert lib.stm_sgm_uses_tiles(375, 450, 128, 1, 0) == 1
    assert lib.stm_sgm_uses_tiles(1080, 1920, 256, 1, 0) == 1
    assert lib.stm_sgm_uses_tiles(375, 450, 128, 256, 0) == 1
    assert lib.stm_sgm_uses_tiles(375, 450, 64, 1, 0) == 0
    assert lib.stm_sgm_uses_tiles(375, 450, 128, 1, _native.STM_SGM_FORMS["paired"]) == 0
    assert lib.stm_sgm_uses_tiles(20, 22, 64, 1, _native.STM_SGM_FORMS["walks"]) == 1
    assert lib.stm_sgm_uses_tiles(12, 20, 48, 1, _native.STM_SGM_FORMS["walks"]) == 0


def test_single_pair_default_path_matches_oracle(device, oracle):
    """The reference's call shape -- one pair, defaults -- through Pipeline (auto form: tiles fed
    by walks at D = 128), against the oracle."""
    import stereomatch
    H, W, D = 96, 130, 128
    L, R = oracle.synthetic_pair(H, W, seed=5)
    pipe = stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal())
    got = pipe.estimate(_gpu(L, device), _gpu(R, device))
    assert pipe.status() == 0 and pipe.aggregation._last_tiled
    assert_same_bits(got.cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(L, R, D), L)), "single pair, auto form")
