"""Robustness of the tiled form of the semiglobal aggregation: the STM_SGM_ACCUMULATE flag
(what the reference-side stub of INTEGRATION.md passes), the fall-back to the paired form for
pointers the tiled form cannot take, graph capture, two tiled calls in flight on one device,
and the status word when something else holds the device's CUs -- a tiled call either
produces the paired form's bits or reports it, never a wrong map with status 0."""
import ctypes
import os
import subprocess
import time

import numpy as np
import pytest
import torch

from test_gpu_parity import _gpu, _random_volume, assert_same_bits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw_sgm(lib, native, C, I, out, flags, fused_disp=None, ws_slack=0):
    """stm_sgm_f32 / stm_sgm_wta_f32 through the C ABI on torch tensors; returns the workspace."""
    B = C.shape[0] if C.dim() == 4 else 1
    H, W, D = C.shape[-3:]
    n = int(lib.stm_sgm_workspace_bytes(H, W, D, B, flags))
    ws = torch.empty(max(n, 16) + ws_slack, dtype=torch.uint8, device=C.device)
    s = torch.cuda.current_stream().cuda_stream
    if fused_disp is None:
        rc = lib.stm_sgm_f32(C.data_ptr(), I.data_ptr(), out.data_ptr(), H, W, D, 0.5, 3.0, flags,
                             ws.data_ptr(), n, B, s)
    else:
        rc = lib.stm_sgm_wta_f32(C.data_ptr(), I.data_ptr(), out.data_ptr(), fused_disp.data_ptr(), H, W, D,
                                 0.5, 3.0, flags, ws.data_ptr(), n, B, s)
    native.check(rc)
    return ws


@pytest.mark.parametrize("H,W,D", [(20, 22, 64), (40, 90, 128), (90, 40, 128), (35, 50, 256), (17, 15, 64),
                                   (33, 129, 128), (64, 80, 128)])
@pytest.mark.parametrize("form", ["tiled", "walks", "paired"])
def test_accumulate_matches_reference_semantics(device, oracle, H, W, D, form):
    """STM_SGM_ACCUMULATE: out = (((out0 + L>) + L<) + ...) on top of the caller's contents
    (the reference's `+=` into its output tensor), in both forms, sum and fused reduction."""
    from stereomatch import _native
    lib = _native.lib()
    rng = np.random.default_rng(H * 31 + W + D)
    C = _random_volume(rng, H, W, D, "float")
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    init = (rng.random((H, W, D)) * 40).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0, out=init)
    flags = _native.STM_SGM_ACCUMULATE | _native.STM_SGM_FORMS[form]
    assert lib.stm_sgm_uses_tiles(H, W, D, 1, flags) == (0 if form == "paired" else 1)
    Cg, Ig = _gpu(C, device), _gpu(image, device)
    out = _gpu(init, device)
    _raw_sgm(lib, _native, Cg, Ig, out, flags)
    assert_same_bits(out.cpu().numpy(), want, form + " accumulate")
    # pre-zeroed output + ACCUMULATE = what the reference wrapper does (aggregation.py:51)
    out = torch.zeros_like(Cg)
    _raw_sgm(lib, _native, Cg, Ig, out, flags)
    assert_same_bits(out.cpu().numpy(), oracle.sgm(C, image, 0.5, 3.0), form + " accumulate on zeros")
    # fused reduction on top of the caller's contents, sum kept and not kept
    for keep in (_native.STM_SGM_KEEP_SUM, 0):
        out = _gpu(init, device)
        disp = torch.empty((H, W), dtype=torch.int32, device=device)
        _raw_sgm(lib, _native, Cg, Ig, out, flags | keep, fused_disp=disp)
        assert_same_bits(disp.cpu().numpy(), oracle.wta(want), form + " accumulate, fused")
        if keep:
            assert_same_bits(out.cpu().numpy(), want, form + " accumulate, fused, kept sum")


def test_auto_form_takes_misaligned_pointers(device, oracle):
    """The library's own choice of the tiled form must not turn a call the paired form accepts
    into an error: an output (or workspace) that is not 16-byte aligned falls back to the
    paired form behind the same workspace header; forcing STM_SGM_TILED keeps the error."""
    from stereomatch import _native
    lib = _native.lib()
    H, W, D, B = 64, 320, 128, 40
    assert lib.stm_sgm_uses_tiles(H, W, D, B, 0) == 1, "shape no longer selects the tiled form by itself"
    rng = np.random.default_rng(5)
    C = np.stack([_random_volume(rng, H, W, D, "float") for _ in range(B)])
    image = rng.integers(0, 5, size=(B, H, W)).astype(np.float32)
    Cg, Ig = _gpu(C, device), _gpu(image, device)
    want = torch.empty_like(Cg)
    _raw_sgm(lib, _native, Cg, Ig, want, _native.STM_SGM_FORMS["paired"])
    backing = torch.empty(Cg.numel() + 1, dtype=torch.float32, device=device)
    out = backing[1:].view(Cg.shape)  # 4 bytes off a 16-byte boundary
    assert out.data_ptr() % 16 == 4
    ws = _raw_sgm(lib, _native, Cg, Ig, out, 0)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), want.view(torch.int32))
    assert int(ws[:4].view(torch.int32).item()) == 0, "status word behind the fallback"
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        _raw_sgm(lib, _native, Cg, Ig, out, _native.STM_SGM_FORMS["tiled"])
    # the same through the package: an offset view as the reuse tensor
    import stereomatch
    sgm = stereomatch.aggregation.Semiglobal(0.5, 3.0)
    got = sgm(Cg, Ig, out)
    assert got.data_ptr() == out.data_ptr() and sgm.status() == 0
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_tiled_pipeline_can_be_captured_into_a_graph(device, oracle):
    """The tiled form inside a captured graph (its sweeps then use the plain launch), replayed
    on new inputs; the status word is read through status()."""
    import stereomatch
    H, W, D, B = 48, 72, 128, 3
    pipe = stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal(form="tiled"))
    pairs = [oracle.synthetic_pair(H, W, seed=s) for s in range(2 * B)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs[:B]])).to(device)
    R = torch.from_numpy(np.stack([p[1] for p in pairs[:B]])).to(device)
    pipe.estimate(L, R)  # allocates outputs and workspace outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pipe.estimate(L, R)
    L.copy_(torch.from_numpy(np.stack([p[0] for p in pairs[B:]])))
    R.copy_(torch.from_numpy(np.stack([p[1] for p in pairs[B:]])))
    graph.replay()
    torch.cuda.synchronize()
    assert pipe.status() == 0
    for b in range(B):
        l, r = pairs[B + b]
        assert_same_bits(out[b].cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(l, r, D), l)), "graph replay %d" % b)


def test_single_pair_default_pipeline_can_be_captured_into_a_graph(device, oracle):
    """The reference's call shape with the defaults (D = 128: tiles fed by walks) inside a captured
    graph, replayed on a new pair."""
    import stereomatch
    H, W, D = 64, 88, 128
    pipe = stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                stereomatch.aggregation.Semiglobal())
    l0, r0 = oracle.synthetic_pair(H, W, seed=1)
    l1, r1 = oracle.synthetic_pair(H, W, seed=2)
    L, R = torch.from_numpy(l0).to(device), torch.from_numpy(r0).to(device)
    pipe.estimate(L, R)
    torch.cuda.synchronize()
    assert pipe.aggregation._last_form == 2
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pipe.estimate(L, R)
    L.copy_(torch.from_numpy(l1))
    R.copy_(torch.from_numpy(r1))
    graph.replay()
    torch.cuda.synchronize()
    assert pipe.status() == 0
    assert_same_bits(out.cpu().numpy(), oracle.wta(oracle.sgm(oracle.ssd(l1, r1, D), l1)), "graph replay, walks form")


def _two_stream_case(device, form, **kw):
    import stereomatch
    H, W, D, B = 120, 300, 128, 24
    rng = np.random.default_rng(17)
    Cs = [torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(device) for _ in range(2)]
    Is = [torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(device) for _ in range(2)]
    ref = [stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1] for C, I in zip(Cs, Is)]
    torch.cuda.synchronize()
    ops = [stereomatch.aggregation.Semiglobal(form=form, **kw) for _ in range(2)]
    streams = [torch.cuda.Stream(device) for _ in range(2)]
    outs = [None, None]
    for rep in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k] = ops[k].aggregate_and_reduce(Cs[k], Is[k])[1]
    torch.cuda.synchronize()
    return ops, outs, ref


def test_two_cooperative_tiled_calls_in_flight(device):
    """Two tiled aggregations on two streams of one device with cooperative=True
    (STM_SGM_COOPERATIVE): the runtime runs cooperative launches one at a time, both succeed."""
    ops, outs, ref = _two_stream_case(device, "tiled", cooperative=True)
    for k in range(2):
        assert ops[k].status() == 0
        assert torch.equal(outs[k], ref[k]), "stream %d" % k


def test_two_plain_tiled_calls_in_flight_never_give_a_wrong_map_silently(device):
    """The same without the cooperative launch -- outside the residency contract: the two calls
    may each hold a part of the CUs and wait for the other's.  Whatever happens, each result
    either equals the paired form's or its status word is set and check() raises."""
    t0 = time.time()
    ops, outs, ref = _two_stream_case(device, "tiled")
    for k in range(2):
        st = ops[k].status()
        if st == 0:
            assert torch.equal(outs[k], ref[k]), "status 0 but a wrong disparity map (stream %d)" % k
        else:
            with pytest.raises(RuntimeError, match="invalid"):
                ops[k].check()
    assert time.time() - t0 < 120


def _hold_lib():
    """tests/helpers/libstm_hold.so: a kernel that occupies CUs for a given time (test helper,
    built by __graft_entry__.build(); not part of the product library)."""
    path = os.path.join(ROOT, "tests", "helpers", "libstm_hold.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "helpers")])
    lib = ctypes.CDLL(path)
    lib.stm_test_hold_cus.restype = ctypes.c_int
    lib.stm_test_hold_cus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("hold_ms,fraction", [(300, 1.0), (300, 0.5), (3500, 0.5)])
def test_tiled_call_beside_a_kernel_that_holds_cus(device, hold_ms, fraction):
    """A long-running kernel on a second stream takes the LDS of all / half of the CUs while a
    tiled aggregation is launched.  Held for 0.3 s the sweeps simply start late (status 0,
    right bits); held for 3.5 s -- longer than the sweeps' bounded waits -- the call reports
    it: status word set, check() raises, the next call raises.  Never a wrong map with status 0."""
    import stereomatch
    hold = _hold_lib()
    H, W, D, B = 96, 640, 128, 26  # 10 strips x 26 images: one sweep launch on (almost) every CU
    rng = np.random.default_rng(23)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(device)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(device)
    ref = stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1]
    sgm = stereomatch.aggregation.Semiglobal(form="tiled")
    warm = sgm.aggregate_and_reduce(C, I)[1]
    assert sgm.status() == 0 and torch.equal(warm, ref)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    side = torch.cuda.Stream(device)
    torch.cuda.synchronize()
    assert hold.stm_test_hold_cus(int(cus * fraction), hold_ms, side.cuda_stream) == 0
    time.sleep(0.05)  # the holder is running before the aggregation is launched
    out = sgm.aggregate_and_reduce(C, I)[1].clone()
    st = sgm.status()
    torch.cuda.synchronize()
    if st == 0:
        assert torch.equal(out, ref), "status 0 but a wrong disparity map"
    else:
        assert st & 1
        with pytest.raises(RuntimeError, match="invalid"):
            sgm.check()
    if hold_ms < 1000:
        assert st == 0, "a 0.3 s hold is far inside the sweeps' 2 s bounded waits"
    # recovery as documented: the paired form
    again = stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1]
    assert torch.equal(again, ref)


@pytest.mark.parametrize("hold_ms,fraction", [(300, 1.0), (3500, 0.5)])
def test_skewed_tiles_beside_a_kernel_that_holds_cus(device, hold_ms, fraction):
    """The skewed-tile form has no residency requirement: its units are taken by ticket, so
    whatever a workgroup waits for is held by a workgroup that runs.  A kernel that holds all CUs
    for 0.3 s, or half of them for 3.5 s (longer than the bounded waits), only delays it: status 0
    and the right map either way; and two such calls on two streams run side by side."""
    import stereomatch
    hold = _hold_lib()
    H, W, D, B = 96, 640, 128, 26
    rng = np.random.default_rng(29)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(device)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(device)
    ref = stereomatch.aggregation.Semiglobal(form="paired").aggregate_and_reduce(C, I)[1]
    sgm = stereomatch.aggregation.Semiglobal(form="skew")
    warm = sgm.aggregate_and_reduce(C, I)[1]
    assert sgm._last_form == stereomatch._native.STM_SGM_FORM_SKEW and sgm.status() == 0 and torch.equal(warm, ref)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    side = torch.cuda.Stream(device)
    torch.cuda.synchronize()
    assert hold.stm_test_hold_cus(int(cus * fraction), hold_ms, side.cuda_stream) == 0
    time.sleep(0.05)
    out = sgm.aggregate_and_reduce(C, I)[1].clone()
    assert sgm.status() == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # two calls in flight on two streams (each with its own op, i.e. its own workspace)
    other = stereomatch.aggregation.Semiglobal(form="skew")
    with torch.cuda.stream(side):
        a = other.aggregate_and_reduce(C, I)[1]
    b = sgm.aggregate_and_reduce(C, I)[1]
    torch.cuda.synchronize()
    assert other.status() == 0 and sgm.status() == 0 and torch.equal(a, ref) and torch.equal(b, ref)


def test_failed_call_is_reported_by_the_next_call(device):
    """The deferred report: a set status word raises at the start of the next call on the same
    op (no blocking), with the recovery named."""
    import stereomatch
    H, W, D, B = 40, 90, 128, 2
    rng = np.random.default_rng(3)
    C = torch.from_numpy((rng.random((B, H, W, D)) * 50).astype(np.float32)).to(device)
    I = torch.from_numpy(rng.integers(0, 6, size=(B, H, W)).astype(np.float32)).to(device)
    sgm = stereomatch.aggregation.Semiglobal(form="tiled")
    sgm(C, I)
    assert sgm.status() == 0
    sgm(C, I)
    torch.cuda.synchronize()
    sgm._status_host.fill_(1)  # what a timed-out sweep would have left in the word
    with pytest.raises(RuntimeError, match="form='paired'"):
        sgm(C, I)
    sgm(C, I)  # the report is delivered once
    assert sgm.status() == 0


def test_sharded_estimate_recovers_from_a_reported_failure(device):
    """estimate_sharded's recovery.  A tiled call whose status word says "invalid" is
    re-run with the paired form on the same op; reading the word consumes the report, so the
    re-run is not refused, the maps are the paired form's, and nothing raises."""
    import stereomatch
    from stereomatch.distributed import estimate_sharded
    H, W, D, B = 40, 90, 128, 2
    rng = np.random.default_rng(5)
    L = torch.from_numpy(rng.integers(0, 256, size=(B, H, W)).astype(np.float32)).to(device)
    R = torch.roll(L, -3, dims=2)

    def pipe(form):
        return stereomatch.Pipeline(stereomatch.cost.SSD(D), stereomatch.disparity_reduce.WinnerTakesAll(),
                                    stereomatch.aggregation.Semiglobal(form=form))
    want = pipe("paired").estimate(L, R).clone()
    p = pipe("tiled")
    sgm = p.aggregation
    post = sgm._post_status
    calls = []

    def failing_post(dev):  # the first call's word arrives as 1: what a timed-out sweep leaves
        post(dev)
        calls.append(sgm._last_form)
        if len(calls) == 1:
            sgm._status_event.synchronize()
            sgm._status_host.fill_(1)
    sgm._post_status = failing_post
    with pytest.warns(UserWarning, match="paired form"):
        got = estimate_sharded(p, L, R)
    assert len(calls) == 2 and calls[1] == stereomatch._native.STM_SGM_FORM_PAIRED
    assert torch.equal(got, want)
    assert sgm.form == "tiled" and p.status() == 0
    # without the collective context's status() the same failure raises at the next call
    sgm._post_status = post
    p.estimate(L, R)
    torch.cuda.synchronize()
    sgm._status_host.fill_(1)
    assert sgm.status() == 1 and sgm.status() == 1  # consumed, remembered
    p.estimate(L, R)  # not refused
    assert p.status() == 0
    sgm.raise_deferred = False
    p.estimate(L, R)
    torch.cuda.synchronize()
    sgm._status_host.fill_(1)
    p.estimate(L, R)  # no implicit raise when the caller has taken the verdict over


def test_tiled_forms_fall_back_where_they_do_not_apply(device, oracle):
    """Where the tiled forms stop (include/stereomatch_hip.h): D other than 64 / 128 / 256 (e.g.
    the reference limit's D = 512, src/semiglobal_gpu.cu:109-120), SINGLE_PASSES and
    EXACT_NONFINITE.  form="tiled" / "walks" then silently take the paired form -- same results,
    the flag only ever changes speed -- and stm_sgm_form says so."""
    import stereomatch
    from stereomatch import _native
    lib = _native.lib()
    for D in (192, 512, 1024):
        for form in ("tiled", "walks"):
            assert lib.stm_sgm_form(64, 96, D, 4, _native.STM_SGM_FORMS[form]) == _native.STM_SGM_FORM_PAIRED
    for extra in (_native.STM_SGM_SINGLE_PASSES, _native.STM_SGM_EXACT_NONFINITE):
        assert lib.stm_sgm_form(64, 96, 128, 4, _native.STM_SGM_FORMS["tiled"] | extra) == _native.STM_SGM_FORM_PAIRED
    assert lib.stm_sgm_form(64, 96, 128, 4, _native.STM_SGM_FORMS["tiled"] | _native.STM_SGM_ACCUMULATE) == _native.STM_SGM_FORM_SWEEPS
    rng = np.random.default_rng(9)
    H, W, D = 20, 36, 512
    C = _random_volume(rng, H, W, D, "float")
    image = rng.integers(0, 5, size=(H, W)).astype(np.float32)
    want = oracle.sgm(C, image, 0.5, 3.0)
    for form in ("tiled", "walks", "auto"):
        sgm = stereomatch.aggregation.Semiglobal(0.5, 3.0, form=form)
        got = sgm(_gpu(C, device), _gpu(image, device))
        assert sgm.status() == 0 and not sgm._last_tiled
        assert_same_bits(got.cpu().numpy(), want, "D = 512, form=" + form)


def test_bench_starts_its_own_ranks(device):
    """`python bench.py --gpus 2` without a launcher's environment starts two ranks itself (children
    under torch.distributed.run; here gloo, both on the one GPU of the box) and prints exactly one
    JSON line on stdout with n_gpus = 2, both workloads, the parity gates of all ranks passed and
    the aggregation's status word clear."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--shared-device", "--steps", "1", "--warmup", "1", "--batch", "4", "--also-batch", "1",
                          "--also-steps", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout, got %d:\n%s" % (len(lines), out.stdout[-1000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["unit"] == "pairs/s" and line["sgm_status"] == 0
    assert line["config"]["pairs_per_gpu_per_step"] == 4 and line["value"] > 0
    assert line["also"]["n_gpus"] == 2 and line["also"]["value"] > 0
    assert "roofline" in line and line["cpu_baseline"] is None  # (the CPU baseline is a rank-0, N = 1 item)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_strip_handoffs_hold_over_repeated_runs(device, D):
    """The sweep strips hand diagonal states to each other through mailboxes -- a timing-dependent
    path (round 3 met rows whose tags matched and whose values did not).  Shapes
    with several strips and several images, each run five times against the paired form
    (tools/stress_sweeps.py is the long version)."""
    import stereomatch
    for (H, W, B) in [(40, 90, 5), (64, 200, 3), (33, 129, 8), (17, 65, 16), (9, 1000, 2), (100, 300, 2)]:
        g = torch.Generator().manual_seed(H * 1000 + W + D)
        C = torch.rand((B, H, W, D), generator=g) * 50 + 10
        C[..., D - 1] = torch.rand((B, H, W), generator=g)
        C = C.to(device)
        img = torch.randint(0, 5, (B, H, W), generator=g).float().to(device)
        ref = stereomatch.aggregation.Semiglobal(form="paired", penalty1=0.5, penalty2=3.0)(C, img)
        for rep in range(5):
            sg = stereomatch.aggregation.Semiglobal(form="tiled", penalty1=0.5, penalty2=3.0)
            out = sg(C, img)
            assert sg.status() == 0
            n = int((out.view(torch.int32) != ref.view(torch.int32)).sum())
            assert n == 0, "%dx%dx%d B=%d run %d: %d entries differ from the paired form" % (H, W, D, B, rep, n)
