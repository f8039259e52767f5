#!/usr/bin/env python3
"""Benchmark of the stereo cost-volume hot path on MI355X.

One *step* = SSD cost fill -> 8-path semiglobal aggregation -> winner-takes-all over one
batch of synthetic stereo pairs that is already resident in HBM (through the
``stereomatch`` package, i.e. the C ABI of libstereomatch_hip.so).  The metric is
BASELINE.json's: stereo pairs per second, whole job.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          # N > 1 without a launcher: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --backend gloo --shared-device   # rehearsal of the N-rank flow on a 1-GPU box

Workloads (BASELINE.json ``configs``):
  teddy  = configs[1]: 375x450, SSD(128, k=7) + SGM(0.1, 0.2) + WTA      (primary, `value`)
  1080p  = configs[3]: 1080x1920, SSD(256) + SGM + WTA                    (reported under
           "also", with its own roofline: the HBM-roofline configuration)
Multi-GPU: independent pairs are sharded over ranks (weak scaling: every rank runs the
same batch size) and their disparity maps stay on the GPU that computed them: no collective
on the data path.  The last step's maps are all-gathered once over RCCL inside the timed
region (--gather last, the default; --gather every does it after every step, overlapped with
the next step's kernels; --gather none leaves it out).

Rank 0 prints ONE JSON line.  ``roofline`` is for the semiglobal aggregation stage (the
dominant kernels: one sgm_ckpt_kernel + four sgm_pair_kernel launches per step, the last
with the winner-takes-all fused in): achieved = 8 bytes x H*W*D x pairs (SURVEY 8d's
aggregation figure; the fused stage also does the reduction and skips the final write)
/ time of those launches, measured with HIP events on the launch stream inside the timed
region.  ``cpu_baseline`` times the CPU oracle (1 thread,
the reference as shipped) on a bounded sample of the primary workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "stereo-matching_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FORCE_GATHER = False  # --rehearse-nccl: run the gather code path in a 1-rank RCCL group
GATHER_MODE = "last"  # --gather: every | last | none
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (H, W, D, default pairs per GPU per step, digest of pair 0's disparity map)
    # digests: SURVEY.md Appendix C, SSD+SGM+WTA on the seed-0 pair (reference CPU output).
    "teddy": dict(H=375, W=450, D=128, batch=256, digest="9693be8fb6aa9b99",
                  config="Teddy 450x375, SSD + 8-path SGM + WTA, D=128"),
    # digest: the oracle's (tests/golden/oracle_digests.json; the survey did not run the
    # reference at this size).  batch 8 = config 5's per-GPU share (64 pairs over 8 GPUs).
    "1080p": dict(H=1080, W=1920, D=256, batch=8, digest="63d42685337550c1",
                  config="1920x1080 synthetic pair, SSD + SGM + WTA, D=256"),
}


def synthetic_pair(H, W, seed, shift=7):
    """integer-valued 0..255 float32 images, constant true disparity 7."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H, W + 64)).astype(np.float32)
    return (np.ascontiguousarray(base[:, 32:32 + W]),
            np.ascontiguousarray(base[:, 32 + shift:32 + shift + W]))


def digest16(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]


class _TimedCost:
    """The cost callable of the pipeline, recording a HIP event when its work is queued."""

    def __init__(self, op):
        self.op, self.event = op, None

    def __call__(self, left, right, cost_volume=None):
        out = self.op(left, right, cost_volume=cost_volume)
        if self.event is not None:
            self.event.record()
        return out


class Job:
    """Buffers and ops of one workload on one GPU: stereomatch.Pipeline(SSD, WTA, Semiglobal)
    on a [B, H, W] stack of pairs.  With this package's own ops the pipeline runs the
    aggregation and the winner-takes-all as one fused call (stm_sgm_wta_f32)."""

    def __init__(self, name, batch, rank, device, form="auto"):
        import stereomatch
        w = WORKLOADS[name]
        self.name, self.w, self.batch, self.device = name, w, batch, device
        H, W, D = w["H"], w["W"], w["D"]
        pairs = [synthetic_pair(H, W, seed=rank * batch + b) for b in range(batch)]
        self.left = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
        self.right = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
        self.cost_op = _TimedCost(stereomatch.cost.SSD(D, kernel_size=7))
        self.sgm = stereomatch.aggregation.Semiglobal(0.1, 0.2, form=form)
        # In a multi-rank run a failed call is reported by check_status() on ALL ranks at a common
        # point (all-reduced); the op's own deferred raise would fire on one rank alone, mid-loop,
        # and leave the others in the next barrier.
        self.sgm.raise_deferred = False
        self.pipe = stereomatch.Pipeline(self.cost_op, stereomatch.disparity_reduce.WinnerTakesAll(),
                                         self.sgm)
        self.disp = None
        self.gathered = None
        self.stage = None
        self.pending = None
        self.n_elems = H * W * D
        self.D = D

    def step(self, events=None, world=1, gather=None):
        ev = events
        if ev is not None:
            ev[0].record()
        self.cost_op.event = ev[1] if ev is not None else None
        self.disp = self.pipe.estimate(self.left, self.right)
        if ev is not None:
            ev[2].record()
        if gather is None:
            gather = GATHER_MODE == "every"
        if gather and (world > 1 or FORCE_GATHER):
            # The only collective of the job: every rank ends up with all disparity maps.
            import torch.distributed as dist
            if dist.get_backend() == "nccl":
                # RCCL all-gather on its own stream, overlapped with the next step's kernels:
                # the maps are staged out of the pipeline's (reused) output buffer first.
                if self.gathered is None:
                    self.stage = torch.empty_like(self.disp)
                    self.gathered = torch.empty((world,) + tuple(self.disp.shape),
                                                dtype=torch.int32, device=self.device)
                self.finish_gather()
                self.stage.copy_(self.disp)
                self.pending = dist.all_gather_into_tensor(self.gathered, self.stage, async_op=True)
            else:  # rehearsal backends
                from stereomatch.distributed import gather_disparities
                self.gathered = gather_disparities(self.disp, n_pairs=self.batch * world)
        return self.disp

    def finish_gather(self):
        if getattr(self, "pending", None) is not None:
            self.pending.wait()
            self.pending = None

    def unfused_sgm_ms(self, reps=3):
        """stm_sgm_f32 alone (aggregated volume written out) on the resident cost volume."""
        out = None
        self.sgm(self.pipe._cost_volume, self.left, out)
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self.sgm(self.pipe._cost_volume, self.left, out)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        return float(np.median(ms))


def timed_run(job, steps, warmup, world):
    import torch.distributed as dist
    for k in range(warmup):
        # The last warm-up step rehearses the gather (RCCL sets its communicator up on first
        # use) and finishes it before anything else is launched; the others do not gather.
        job.step(world=world, gather=(GATHER_MODE != "none" and k == warmup - 1))
        job.finish_gather()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        # (the sweeps of the tiled aggregation need every CU for their resident workgroups: a
        # collective running beside every step would take CUs from them, so by default only the
        # last step's maps are gathered)
        job.step(events[k], world=world, gather=(GATHER_MODE == "every" or
                                                  (GATHER_MODE == "last" and k == steps - 1)))
    job.finish_gather()      # the gather is part of the timed job
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=job.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in events])) for i in range(2)]
    return elapsed, stage_ms


def roofline(job, sgm_ms, traffic_and_note=(None, None)):
    traffic, note = traffic_and_note[:2]
    algorithmic = 8.0 * job.n_elems * job.batch  # read cost once + write sum once
    achieved = algorithmic / (sgm_ms * 1e-3) / 1e9
    form = int(getattr(job.sgm, "_last_form", 0))  # 0 paired, 1 tiles behind row sweeps, 2 tiles behind walks
    form_name = {0: "paired", 1: "tiled (sweeps)", 2: "tiled (walks)", 3: "skewed tiles"}[form]
    if traffic is not None and len(traffic_and_note) > 2 and traffic_and_note[2] not in (None, form_name):
        # (e.g. the shared-device rehearsal, which forces the paired form)
        traffic, note = None, "profiles/traffic.json was measured on the form '%s', this run took '%s'" % (
            traffic_and_note[2], form_name)
    tile = "sgm_tile_h2_kernel" if getattr(job, "D", 0) == 128 else "sgm_tile_kernel"
    kernels = {0: "paired form: sgm_ckpt_kernel + 4 x sgm_pair_kernel",
               1: "tiled form: sgm_hwalk_kernel + sgm_sweep_kernel (down, up) + " + tile,
               2: "tiled form behind walks: sgm_bwalk_kernel + " + tile,
               3: "skewed-tile form: sgm_skew_rev_kernel + sgm_skew_fwd_kernel"}[form]
    return {"bound": "hbm", "traffic_note": note,
            "kernel": "semiglobal aggregation with fused WTA (stm_sgm_wta_f32), " + kernels,
            "form": form_name,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes": algorithmic, "ms": round(sgm_ms, 4),
            # the rate the counted HBM traffic moves at (not part of the contract's fields)
            "traffic_rate": None if traffic is None else round(traffic / (sgm_ms * 1e-3) / 1e9, 1)}


def csrc_sha16():
    """Hash of the kernel sources the loaded library was built from (they travel with it)."""
    import hashlib
    d = os.path.join(ROOT, "stereo-matching_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) or fn == "Makefile":
            with open(os.path.join(d, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def load_traffic(name, batch):
    """HBM bytes the aggregation stage moves per step: the per-pair figure of the committed
    PMC summary (profiles/traffic.json) times the pairs per step.  The summary is stamped
    with the hash of the kernel sources it was measured on (profiles/summarize.py); when
    that differs from the sources in this tree the figure is stale and None is returned
    (with the reason), so an old PMC number cannot ride on new kernels."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    per_pair = doc.get(name + "_per_pair")
    if per_pair is None:
        return None, "profiles/traffic.json has no %s_per_pair" % name
    if doc.get(name + "_pairs_per_step") not in (None, batch):
        # the library picks the aggregation's form by batch size: a figure measured at another
        # batch may belong to other kernels
        return None, "profiles/traffic.json was measured at %s pairs per step, this run uses %d" % (
            doc.get(name + "_pairs_per_step"), batch)
    have, want = doc.get("csrc_sha16"), csrc_sha16()
    if have != want:
        return None, "profiles/traffic.json was measured on kernel sources %s, this tree is %s" % (have, want)
    return int(per_pair) * batch, None, doc.get(name + "_form")


def cpu_baseline(budget_s=20.0):
    """The CPU oracle (bit-exact restatement of the reference CPU path), one thread, on
    Teddy-sized pairs of the primary workload until ~budget_s seconds are spent."""
    import oracle
    oracle.build()
    w = WORKLOADS["teddy"]
    done, t0 = 0, time.perf_counter()
    stages = [0.0, 0.0, 0.0]
    while True:
        L, R = synthetic_pair(w["H"], w["W"], seed=done)
        a = time.perf_counter()
        C = oracle.ssd(L, R, w["D"], 7)
        b = time.perf_counter()
        S = oracle.sgm(C, L, 0.1, 0.2)
        c = time.perf_counter()
        disp = oracle.wta(S)
        d = time.perf_counter()
        stages[0] += b - a
        stages[1] += c - b
        stages[2] += d - c
        if done == 0:
            assert digest16(disp) == w["digest"], "oracle digest mismatch"
        done += 1
        if time.perf_counter() - t0 > budget_s * 0.75:
            break
    total = time.perf_counter() - t0
    return {"value": round(done / total, 4), "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d Teddy-sized pairs (375x450, D=128), oracle SSD+SGM+WTA, 1 thread of %d "
                      "host cores; per pair: SSD %.2f s, SGM %.2f s, WTA %.3f s"
                      % (done, os.cpu_count() or 0, stages[0] / done, stages[1] / done,
                         stages[2] / done)}


def latency_b1(device, reps=20):
    """The reference's call shape: ONE pair per Pipeline.estimate call (pipeline.py:69-94 has
    no batch axis).  Milliseconds per call, HIP events around `reps` back-to-back calls after
    two warm-up calls; inputs resident.  `unit_images_ordered` is the same Teddy pipeline on
    /255-scaled images (the reference's own fixtures, tests/conftest.py:24-29), which take the
    ORDERED SSD path: exact for any input and ~10x the integer-image fill."""
    import stereomatch
    out = {}
    for key, (H, W, D, scale) in {"teddy": (375, 450, 128, 1.0), "1080p": (1080, 1920, 256, 1.0),
                                  "teddy_unit_images_ordered": (375, 450, 128, 1.0 / 255.0)}.items():
        L, R = synthetic_pair(H, W, seed=0)
        Lg = torch.from_numpy(L * np.float32(scale)).to(device)
        Rg = torch.from_numpy(R * np.float32(scale)).to(device)
        cost = _TimedCost(stereomatch.cost.SSD(D, kernel_size=7))
        pipe = stereomatch.Pipeline(cost, stereomatch.disparity_reduce.WinnerTakesAll(),
                                    stereomatch.aggregation.Semiglobal(0.1, 0.2))
        for _ in range(2):
            pipe.estimate(Lg, Rg)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
        torch.cuda.synchronize()
        for k in range(reps):
            ev[k][0].record()
            cost.event = ev[k][1]
            pipe.estimate(Lg, Rg)
            ev[k][2].record()
        torch.cuda.synchronize()
        ssd = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
        agg = float(np.median([e[1].elapsed_time(e[2]) for e in ev]))
        out[key] = {"ms_per_call": round(ssd + agg, 4), "ssd_ms": round(ssd, 4),
                    "sgm_wta_ms": round(agg, 4), "pairs_per_s": round(1e3 / (ssd + agg), 1)}
        # The same call captured once into a HIP graph and replayed (what a caller with a fixed
        # image size can do: one graph launch instead of the call's five to seven kernel and
        # memset launches).  Extra information; a failure here never fails the run.
        try:
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                raise RuntimeError("a process group is active: no stream capture beside its watchdog")
            plain = stereomatch.Pipeline(stereomatch.cost.SSD(D, kernel_size=7),
                                         stereomatch.disparity_reduce.WinnerTakesAll(),
                                         stereomatch.aggregation.Semiglobal(0.1, 0.2))
            want = plain.estimate(Lg, Rg).clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                got = plain.estimate(Lg, Rg)
            for _ in range(2):
                graph.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            ok = bool(torch.equal(got, want)) and plain.status() == 0
            out[key]["graph_replay_ms_per_call"] = round(a.elapsed_time(b) / reps, 4) if ok else None
            del graph, plain, got, want
        except Exception as e:  # noqa: BLE001
            print("latency_b1: graph replay of %s not measured (%r)" % (key, e), file=sys.stderr, flush=True)
            out[key]["graph_replay_ms_per_call"] = None
        del pipe, cost
        torch.cuda.empty_cache()
    return out


def config3_line(device, steps, batch=64):
    """BASELINE config 3: Teddy 450x375, Birchfield + SGM + dynamic-programming reduce, D=128
    (reported beside the primary workload; parity gate: the reference's DP digest)."""
    import stereomatch
    H, W, D = 375, 450, 128
    pairs = [synthetic_pair(H, W, seed=b) for b in range(batch)]
    left = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
    right = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
    pipe = stereomatch.Pipeline(stereomatch.cost.Birchfield(D),
                                stereomatch.disparity_reduce.DynamicProgramming(),
                                stereomatch.aggregation.Semiglobal())
    disp = pipe.estimate(left, right)
    torch.cuda.synchronize()
    assert digest16(disp[0].cpu().numpy()) == "0922d012b07edd03", "config 3 parity gate failed"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.estimate(left, right)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "Teddy 450x375, Birchfield + SGM + dynamic-programming reduce, D=128",
            "value": round(batch * steps / dt, 2), "unit": "pairs/s", "pairs_per_step": batch,
            "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4)}


def self_launch(n):
    """--gpus N > 1 without a launcher's environment: start the N ranks as children (the same
    command line under torch.distributed.run) and relay their output.  Called before anything
    in this process has touched the GPU; this process only waits and exits with the children's
    code -- it is never replaced by another program."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="teddy", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = default)")
    ap.add_argument("--also", default="1080p", help="second workload reported under 'also' ('' = none)")
    ap.add_argument("--also-steps", type=int, default=5)
    ap.add_argument("--also-batch", type=int, default=0, help="pairs per step of the second workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the Birchfield+SGM+DP extra line")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 per-call latencies")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--gather", default="last", choices=["every", "last", "none"],
                    help="N > 1: all_gather of the disparity maps after every step / the last step / never")
    ap.add_argument("--rehearse-nccl", action="store_true",
                    help="rehearsal only: 1-rank RCCL group, gather code path exercised")
    ap.add_argument("--shared-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo) on a 1-GPU box")
    ap.add_argument("--form", default=None, choices=["auto", "paired", "tiled", "walks", "skew"],
                    help="form of the aggregation (default: auto; paired with --shared-device, "
                         "since ranks sharing one GPU must not run tiled aggregations side by side)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # Exactly ONE line on stdout: everything libraries print there while the job runs (RCCL's
    # version banner, gloo's connection notes) goes to stderr instead; the JSON line is written to
    # the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    form = args.form or ("paired" if args.shared_device else "auto")

    global GATHER_MODE
    GATHER_MODE = args.gather
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.rehearse_nccl and world == 1:
        import torch.distributed as dist
        global FORCE_GATHER
        FORCE_GATHER = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus, "WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus)

    from stereomatch import _native
    _native.lib()  # no HIP library -> fail loudly, there is no fallback

    def gate(job, label):
        """Parity gate before any timing: rank 0's pair 0 (seed 0) must reproduce the pinned
        disparity map.  The verdict is shared, so that every rank stops together instead of
        the others waiting in a collective for a rank that has raised."""
        disp = job.step(world=1)
        torch.cuda.synchronize()
        ok, got = 1, None
        if rank == 0 and job.w["digest"]:
            got = digest16(disp[0].cpu().numpy())
            ok = int(got == job.w["digest"])
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([ok], dtype=torch.int32,
                             device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if not ok:
            if rank == 0:
                print("parity gate failed (%s): %s != %s" % (label, got, job.w["digest"]),
                      file=sys.stderr, flush=True)
            if world > 1:
                import torch.distributed as dist
                dist.destroy_process_group()
            sys.exit(1)
        check_status(job, label + " (gate)")

    def check_status(job, label):
        """The tiled aggregation's status word, all ranks: a set bit (its row sweeps waited for
        workgroups that were not resident) means invalid results -- no metric is printed."""
        st = int(job.sgm.status())
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([st], dtype=torch.int32,
                             device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            st = int(t.item())
        if st:
            if rank == 0:
                print("aggregation status word %d after %s: results invalid, no metric" % (st, label),
                      file=sys.stderr, flush=True)
            if world > 1:
                import torch.distributed as dist
                dist.destroy_process_group()
            sys.exit(2)
        return st

    name = args.workload
    batch = args.batch or WORKLOADS[name]["batch"]
    job = Job(name, batch, rank, device, form)
    gate(job, name)
    elapsed, stage_ms = timed_run(job, args.steps, args.warmup, world)
    sgm_status = check_status(job, name + " (timed loop)")
    pairs = batch * world * args.steps
    value = pairs / elapsed
    w = WORKLOADS[name]
    out = {
        "metric": "stereo pairs/sec (SSD+SGM+WTA)", "value": round(value, 2), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": w["config"], "H": w["H"], "W": w["W"], "D": w["D"],
                   "pairs_per_gpu_per_step": batch, "ssd_mode": "auto", "kernel_size": 7,
                   "penalties": [0.1, 0.2],
                   "sharding": "pairs over ranks, no collective on the data path; all_gather of the disparity maps: " + args.gather},
        "stage_ms": {"ssd": round(stage_ms[0], 4), "sgm_wta_fused": round(stage_ms[1], 4),
                     "sgm_alone_unfused": round(job.unfused_sgm_ms(), 4)},
        "roofline": roofline(job, stage_ms[1], load_traffic(name, batch)),
        "csrc_sha16": csrc_sha16(),
        "sgm_status": sgm_status,  # status word of the tiled aggregation after the timed loop (0 = ok)
    }
    del job
    torch.cuda.empty_cache()

    if args.also and args.also != name:
        # Second workload, on every rank count: at N = 8 the 1080p leg with 8 pairs per GPU
        # is BASELINE config 5 (64 pairs sharded over 8 GPUs, disparity maps gathered).
        w2 = WORKLOADS[args.also]
        b2 = args.also_batch or w2["batch"]
        job2 = Job(args.also, b2, rank, device, form)
        gate(job2, args.also)
        e2, s2 = timed_run(job2, args.also_steps, 1, world)
        check_status(job2, args.also + " (timed loop)")
        d7 = float((job2.disp == 7).float().mean())
        out["also"] = {
            "workload": w2["config"] + (" x %d pairs sharded over %d GPUs" % (b2 * world, world)
                                        if world > 1 else ""),
            "value": round(b2 * world * args.also_steps / e2, 3),
            "unit": "pairs/s", "n_gpus": world, "pairs_per_gpu_per_step": b2, "steps": args.also_steps,
            "ms_per_step": round(e2 / args.also_steps * 1e3, 3),
            "stage_ms": {"ssd": round(s2[0], 4), "sgm_wta_fused": round(s2[1], 4),
                         "sgm_alone_unfused": round(job2.unfused_sgm_ms(), 4)},
            "roofline": roofline(job2, s2[1], load_traffic(args.also, b2)),
            "pixels_at_true_disparity": round(d7, 4), "parity_gate": w2["digest"],
        }
        del job2
        torch.cuda.empty_cache()

    if world == 1 and not args.no_latency:
        out["latency_b1"] = latency_b1(device)

    if world == 1 and not args.no_config3:
        out["also_config3"] = config3_line(device, args.also_steps)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        out["cpu_baseline"] = None
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():  # also the 1-rank RCCL rehearsal group
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
