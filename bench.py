#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: NLP solves/s (N=12, 6-state bicycle).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg2_filtered|cfg3|cfg4|cfg5|lmpc|races|game|overtake] [--scaling weak|strong]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.

Headline (`metric`/`value`, every N): BASELINE.json configs[1] -- MPC-CBF NLP (control/control.py:476-607 of the
reference), 1 obstacle, batch 256 per GPU, N = 12, drawn exactly as SURVEY.md section 8d prescribes (no scenario filter
since round 2: crash states are part of the batch and end in the restoration verdict).  The batch shards by problem, no
data-path collective: weak scaling, `value` = problems of all ranks / max-over-ranks time.

Without --workload the same run also measures the other single-GPU BASELINE configs and reports them in `configs`
(each with its own roofline object): cfg3 (1024 planner scenarios x 4 region QPs + selection), cfg4 (16384 tracking NLPs,
N = 20, 3 obstacles), lmpc (learning-MPC QPs) and cfg5 -- configs[4], the Monte-Carlo sweep
of overtake-planner scenarios sharded by scenario over the ranks, raw scenario -> Bezier/bounds prep -> region QPs ->
selection -> ONE all-gather of the winners (crx.pipeline.PlannerSweep), both weak (16384 scenarios per GPU) and strong
(131072 scenarios in total); the collective is timed separately (`allgather_ms`).

`--gpus N` with N > 1 and no torch.distributed environment re-launches itself under torch.distributed.run (one process per
GPU, rendezvous on 127.0.0.1).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

# The HIP runtime multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue serialise.  The closed-loop workloads use up to six streams (two sub-batches, each with its two branch streams): with 4
# queues their overlap is partly lost (overtake 3.97 ms per step), with 8 it is there (3.25 ms; tools/gpu_round3_i.sh).  Must be set
# before the runtime initialises; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "car-racing_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_GFLOPS = 78600.0  # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (vector FP64, spec)
METRIC = "NLP solves/sec (N=12, 6-state bicycle); p50 per-step solve latency"


def kernel_source_hash():
    """sha256 over everything the solver kernels are built from -- every kernel source and header, the Makefile's flags, and the COMPILER's version
    string: guards the rocprofv3 numbers kept under profiles/ (HBM traffic per launch) against a kernel or a toolchain that changed since."""
    import hashlib
    import subprocess
    h = hashlib.sha256()
    d = os.path.join(ROOT, "car-racing_amd", "csrc")
    for f in ("crx_kernels.hip", "crx_kernels_obs.hip", "crx_kernels_gen.hip", "crx_lmpc.hip", "crx_prep.hip", "crx_lmpcprep.hip", "crx_kparams.h", "crx_wave.h",
              "Makefile"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    try:    # a toolchain bump re-schedules the kernels without touching a source line
        ver = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True, timeout=60).stdout
        ver = "\n".join(ln for ln in ver.splitlines() if "InstalledDir" not in ln)
    except Exception:
        ver = "hipcc: unavailable"
    h.update(ver.encode())
    return h.hexdigest()[:16]


def algorithmic_bytes(kind, N, n_obs):
    """doubles in + doubles out per solve, times 8."""
    if kind == "planner":
        return (11 * N + 18) * 8                                   # a1: 1200 B at N = 12
    if kind == "lmpc":                                              # n_obs carries M here
        return (8 + 54 * N + 7 * n_obs + 1 + 6 * (N + 1) + 2 * N + n_obs + 3) * 8   # 8912 B at N = 12, M = 44
    tgt = (N + 1) if kind == "cbf_tracking" else 0                  # per-stage ey target
    d_in = 6 + 6 + 2 * n_obs * (N + 1) + n_obs + tgt
    d_out = 6 * (N + 1) + 2 * N + n_obs * (N + 1) + 3
    return (d_in + d_out) * 8                                       # a5: 1256 B (N=12, 1 obs); a6: 3152 B


class Ctx:
    """Process-wide state of one bench invocation."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev = self.device()

    def device(self):
        return torch.device("cuda", self.local)

    def dsync(self):
        torch.cuda.synchronize()

    def to_dev(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        return t.to(dtype) if dtype is not None else t

    def sync_all(self):
        if self.world > 1:
            dist.barrier()
        self.dsync()


class Workload:
    """One measurable configuration: step() = one pass of the hot path, solve() = the dominant kernel alone (for the
    HIP-event kernel time), ws = the solver's status/iters/kkt outputs, host_call() = one control step issued with host
    arrays, cpu = (descriptor, host arrays) for the CPU baseline."""
    name = key = kernel = ""
    baseline_config = None
    kind = "cbf"
    N = n_obs = units = batch = 0
    desc = ws = cpu = None
    host_call = None
    gather = None          # the collective alone (cfg5)
    rewind = None          # closed-loop workloads: put the races back to the stated lap phase (outside every timed region)
    extra = None
    scaling = "weak"

    def step(self):
        raise NotImplementedError

    def solve(self):
        raise NotImplementedError


class CatWs:
    """status / iters / kkt of K concurrent sub-batches as one (crx.montecarlo.Concurrent)."""

    def __init__(self, conc, get):
        self.conc, self.get = conc, get

    status = property(lambda self: self.conc.cat(lambda p: self.get(p).status))
    iters = property(lambda self: self.conc.cat(lambda p: self.get(p).iters))
    kkt = property(lambda self: self.conc.cat(lambda p: self.get(p).kkt))


def sub_batches(n, k):
    """k contiguous slices of range(n), sizes differing by at most one."""
    k = max(1, min(int(k), n))
    base, rem = divmod(n, k)
    out, lo = [], 0
    for i in range(k):
        hi = lo + base + (1 if i < rem else 0)
        out.append(slice(lo, hi))
        lo = hi
    return out


def make_cbf(cx, key, args, batch=None, filtered=False):
    from crx import abi, synth, torch_api
    A, B = synth.load_AB()
    w = Workload()
    seed_shift = 1000 * cx.rank
    if key.startswith("cfg2"):
        w.batch = batch or 256
        p = synth.cfg2_mpccbf(w.batch, N=12, seed=2 + seed_shift, safe_start=filtered)
        w.desc = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
        w.kind, w.baseline_config = "cbf", 1
        w.name = "MPC-CBF NLP (control.py:476-607), l_shape model, 1 obstacle, N=12, batch %d/GPU, SURVEY 8d draw%s" % (
            w.batch, " restricted to safe, non-doomed starts (round-1 scenario filter)" if filtered else "")
    else:
        w.batch = batch or 16384
        p = synth.cfg4_tracking_cbf(w.batch, N=20, seed=4 + seed_shift, safe_start=filtered)
        w.desc = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
        w.kind, w.baseline_config = "cbf_tracking", 3
        w.name = "tracking NLP with CBF rows (control.py:251-473 form), 3 obstacles, N=20, batch %d/GPU" % w.batch
    w.key, w.N, w.n_obs, w.units = key, w.desc.N, w.desc.n_obs_max, w.batch
    w.kernel = "crx_solve_kernel<%d>" % w.n_obs
    w.extra = {"scenario_filter": bool(filtered)}
    t_in = [cx.to_dev(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [cx.to_dev(p["n_obs"], torch.int32)]
    w.ws = torch_api.CbfWorkspace(w.desc, w.batch, cx.dev)
    # dispatch order (include/crx.h): computed inside the timed step, one more launch -- from the iteration counts of the previous
    # step (what a receding-horizon loop has) or, with no previous solve, from the start barrier of every problem
    obuf = torch.empty(w.batch, dtype=torch.int32, device=cx.dev)
    mode = args.dispatch
    if mode == "auto":    # more problems than resident slots (256 CUs x 4): the launch has a tail worth ordering; else all start at once
        mode = "start_barrier" if w.batch > 1024 else "index"
    # (longest_first on a STATIC batch re-solves the same problems: the previous iteration counts are those of the very launch being
    # ordered -- an oracle order, an upper bound of what a receding-horizon loop gets; labelled so)
    w.extra["dispatch"] = "longest_first(oracle order on a static batch: upper bound)" if mode == "longest_first" else mode
    order = {"index": lambda: None, "longest_first": lambda: torch_api.longest_first(w.ws.iters, out=obuf),
             "start_barrier": lambda: torch_api.cbf_order_dev(w.desc, *t_in, out=obuf)}[mode]
    w.step = w.solve = lambda: torch_api.cbf_solve_dev(w.desc, *t_in, ws=w.ws, order=order())
    w.step_is_one_launch = mode == "index"
    keys = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
    w.cpu = ("cbf", w.desc, {k: p[k] for k in keys})
    if key.startswith("cfg2"):   # all-cores CPU figure: 4096 problems of the same generator and seed (the first 256 are the batch)
        pc = synth.cfg2_mpccbf(4096, N=12, seed=2 + seed_shift, safe_start=filtered)
        w.cpu = ("cbf", w.desc, {k: pc[k] for k in keys})
    import crx
    hb = crx.binding()
    a1 = tuple(p[k][:1] for k in keys)
    w.host_call = lambda: hb.cbf_solve(w.desc, *a1)
    return w


def make_cbf_streams(cx, args, k=4):
    """configs[1] once more, as a Monte-Carlo sweep would issue it: K INDEPENDENT batches of 256 NLPs (other seeds), one launch each per step on K
    HIP streams that sit on different hardware queues (crx_streams_create).  A 256-problem launch occupies 256 of the chip's 1024 SIMDs and ends with
    its slowest problem (36 iterations against a median of 11); independent batches fill the idle SIMDs.  NOT the headline: a receding-horizon loop
    cannot overlap its own consecutive steps.  units = K x 256 per step."""
    from crx import abi, synth, torch_api
    A, B = synth.load_AB()
    w = Workload()
    streams, n_conc = torch_api.new_streams(k, cx.dev)
    parts = []
    for i in range(k):
        p = synth.cfg2_mpccbf(256, N=12, seed=2 + 17 * i + 1000 * cx.rank, safe_start=False)
        d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
        t_in = [cx.to_dev(p[q]) for q in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [cx.to_dev(p["n_obs"], torch.int32)]
        parts.append((d, t_in, torch_api.CbfWorkspace(d, 256, cx.dev)))
    cur = torch.cuda.current_stream(cx.dev)
    for st in streams:
        st.wait_stream(cur)

    def step():
        for (d, t_in, ws), st in zip(parts, streams):
            with torch.cuda.stream(st):
                torch_api.cbf_solve_dev(d, *t_in, ws=ws)

    class Cat:
        status = property(lambda self: torch.cat([q[2].status for q in parts]))
        iters = property(lambda self: torch.cat([q[2].iters for q in parts]))
        kkt = property(lambda self: torch.cat([q[2].kkt for q in parts]))

    w.key, w.kind, w.baseline_config, w.N, w.n_obs, w.desc, w.ws = "cfg2_x%d_streams" % k, "cbf", 1, 12, 1, parts[0][0], Cat()
    w.batch, w.units = 256, 256 * k
    w.kernel = "crx_solve_kernel<1>"
    w.step = step
    w.solve_parts = [(lambda q=q: torch_api.cbf_solve_dev(q[0], *q[1], ws=q[2])) for q in parts]
    w.name = "MPC-CBF NLP as configs[1], %d independent batches of 256 in flight on %d HIP streams (%d on distinct hardware queues): what a Monte-Carlo sweep gets from the chip at batch 256" % (k, k, n_conc)
    w.extra = {"batches_in_flight": k, "streams_on_distinct_queues": int(n_conc), "kernel_ms_is": "sum of the K launches of one step, each timed alone (in the step they overlap)"}
    return w


def make_planner(cx, args, n_scen=None):
    import crx
    from crx import abi, synth, torch_api
    A, B = synth.load_AB()
    w = Workload()
    n_scen = n_scen or 1024
    p = synth.cfg3_planner(n_scen, N=12, seed=3 + 1000 * cx.rank)
    N, V = 12, p["V"]
    w.key, w.kind, w.baseline_config, w.N, w.n_obs = "cfg3", "planner", 2, N, 0
    w.desc = abi.planner_desc(N, A, B)
    sdesc = abi.select_desc(N, V, p["lap_length"])
    t_in = [cx.to_dev(p[k]) for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
    t_sel = [cx.to_dev(p["n_veh"], torch.int32), cx.to_dev(p["obs_s"]), cx.to_dev(p["obs_ey"]), cx.to_dev(p["old_flag"], torch.int32)]
    w.batch = w.units = n_scen * (V + 1)
    w.ws = torch_api.PlannerWorkspace(w.desc, w.batch, cx.dev)
    sws = torch_api.SelectWorkspace(sdesc, n_scen, cx.dev)
    w.kernel = "crx_solve_kernel<0>"
    w.solve = lambda: torch_api.planner_solve_dev(w.desc, *t_in, ws=w.ws)

    def step():
        torch_api.planner_solve_dev(w.desc, *t_in, ws=w.ws)
        torch_api.select_dev(sdesc, t_sel[0], w.ws.X.view(n_scen, V + 1, N + 1, 6), t_sel[1], t_sel[2], t_sel[3], ws=sws)

    w.step = step
    w.name = "overtake planner: %d scenarios x %d region QPs (overtake_traj_planner.py:248-379) + selection (:205-246), N=12, per GPU" % (n_scen, V + 1)
    keys = ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")
    w.cpu = ("planner", w.desc, {k: p[k] for k in keys})
    hb = crx.binding()
    R1 = V + 1
    a1 = tuple(p[k][:R1] for k in keys) + tuple(p[k][:1] for k in ("n_veh", "obs_s", "obs_ey", "old_flag"))
    w.host_call = lambda: hb.planner_plan(w.desc, sdesc, *a1)
    return w


def make_sweep(cx, args, scaling):
    """BASELINE configs[4]: Monte-Carlo sweep of overtake-planner scenarios sharded by scenario over the ranks."""
    from crx import dist as cdist
    from crx import pipeline, synth
    A, B = synth.load_AB()
    w = Workload()
    if scaling == "strong":
        n_total = args.sweep_total
        lo, hi = cdist.shard_bounds(n_total, cx.rank, cx.world)
        raw_all = synth.cfg3_raw(n_total, N=12, seed=5)                       # same global draw on every rank, sliced
        raw = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == n_total else v) for k, v in raw_all.items()}
    else:
        n_local = args.sweep_per_gpu
        n_total = n_local * cx.world
        raw = synth.cfg3_raw(n_local, N=12, seed=5 + 1000 * cx.rank)
    sw = pipeline.PlannerSweep(raw, A, B, n_total, cx.dev, backend=SWEEP_BACKEND() if SWEEP_BACKEND else None)
    w.key, w.kind, w.baseline_config, w.N, w.n_obs, w.scaling = "cfg5_" + scaling, "planner", 4, 12, 0, scaling
    w.desc, w.ws = sw.desc, sw.ws
    w.batch = w.units = sw.n_local * (sw.V + 1)
    w.kernel = "crx_solve_kernel<0>"
    w.step, w.gather = sw.step, sw.gather
    w.solve = lambda: sw.be.planner_solve_dev(sw.desc, sw.pws.x0, sw.pws.bez_s, sw.pws.bez_ey, sw.pws.ey_lb, sw.pws.ey_ub, ws=sw.ws)
    w.name = ("Monte-Carlo overtake-planner sweep, %s scaling: %d scenarios in total, %d on this rank x %d region QPs; raw scenario -> "
              "Bezier/bounds prep -> QPs -> selection on the device, then ONE all-gather of the winners over %d rank(s)" % (
                  scaling, n_total, sw.n_local, sw.V + 1, cx.world))
    w.extra = {"scenarios_total": int(n_total), "scenarios_this_rank": int(sw.n_local), "winner_record_bytes": int(sw.exchange.rec * 8),
               "allgather_bytes_per_rank_out": int(sw.exchange.recv.numel() * 8)}
    return w


def make_lmpc(cx, args, batch=None):
    import crx
    from crx import abi, torch_api
    w = Workload()
    g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
    ok = np.nonzero(g["lmpc_success"])[0]
    w.batch = w.units = batch or 4096
    idx = ok[(np.arange(w.batch) + 7 * cx.rank) % len(ok)]
    N, M = g["lmpc/A"].shape[1], g["lmpc/ss"].shape[2]
    w.key, w.kind, w.N, w.n_obs = "lmpc", "lmpc", N, M
    w.desc = abi.lmpc_desc(N=N, n_ss_max=M)
    p = dict(x0=g["lmpc/x"][idx], u_old=g["lmpc/u_old"][idx], A=g["lmpc/A"][idx].reshape(w.batch, N, 36),
             B=g["lmpc/B"][idx].reshape(w.batch, N, 12), C=g["lmpc/C"][idx], ss=g["lmpc/ss"][idx], qfun=g["lmpc/qfun"][idx],
             n_ss=np.full(w.batch, M, dtype=np.int32))
    keys = ("x0", "u_old", "A", "B", "C", "ss", "qfun")
    t_in = [cx.to_dev(p[k]) for k in keys] + [cx.to_dev(p["n_ss"], torch.int32)]
    w.ws = torch_api.LmpcWorkspace(w.desc, w.batch, cx.dev)
    w.kernel = "crx_lmpc_kernel"
    obuf = torch.empty(w.batch, dtype=torch.int32, device=cx.dev)
    order = (lambda: torch_api.longest_first(w.ws.iters, out=obuf)) if args.dispatch == "longest_first" else (lambda: None)
    w.extra = {"dispatch": "longest_first(oracle order on a static batch: upper bound)" if args.dispatch == "longest_first" else "index"}
    w.step = w.solve = lambda: torch_api.lmpc_solve_dev(w.desc, *t_in, ws=w.ws, order=order())
    w.step_is_one_launch = args.dispatch != "longest_first"
    w.name = "learning-MPC QP (control.py:610-730), N=%d, %d safe-set points, LTV models and safe sets recorded from the reference's LMPC lap, batch %d/GPU" % (N, M, w.batch)
    w.cpu = ("lmpc", w.desc, {k: p[k] for k in keys + ("n_ss",)})
    hb = crx.binding()
    a1 = tuple(p[k][:1] for k in keys + ("n_ss",))
    w.host_call = lambda: hb.lmpc_solve(w.desc, *a1)
    return w


LAP_WINDOW = 60


def to_lap_phase(cx, w, conc, phase, what):
    """Closed-loop workloads: `phase` untimed control steps after construction, a snapshot there, and a rewind hook that measure()
    calls after its warm-up -- the timed steps always cover control steps [phase, phase + steps) of the lap, so the record does not
    depend on --warmup and two runs with the same --steps report the same lap phase (the share of converged /
    relaxed learning-MPC QPs changes through the lap)."""
    from crx import montecarlo
    for _ in range(phase):
        conc.step()
    cx.dsync()
    snap = montecarlo.Snapshot(conc)

    def rewind():
        cx.dsync()
        snap.restore()
        cx.dsync()

    w.rewind = rewind
    # [r6] a FIXED window of the lap, whatever --steps is: the timed steps are control steps phase .. phase + WINDOW - 1, whole passes of them (--steps
    # is rounded up to a multiple of WINDOW and reported as such; rewind between passes, outside the timed region); the status / iteration fields are
    # taken over one untimed pass through the whole window.  A 20-step request and a 60-step record describe the same 60 control steps.
    w.window = LAP_WINDOW
    w.extra["lap_phase"] = ("timed steps = whole passes over control steps [%d, %d) of %s (--steps rounded up to a multiple of %d); "
                            "the status / iteration fields are taken over all %d steps of the window") % (phase, phase + LAP_WINDOW, what, LAP_WINDOW, LAP_WINDOW)


def make_races(cx, args, batch=None):
    from crx import montecarlo, synth
    from utils import racing_env
    A, B = synth.load_AB()
    w = Workload()
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    w.batch = w.units = batch or 4096
    rng = np.random.default_rng(50 + cx.rank)
    s0 = np.sort(rng.uniform(3.0, 17.0, (w.batch, 2)), axis=1)
    s0[:, 1] = np.maximum(s0[:, 1], s0[:, 0] + 2.0)
    cv, ce = rng.uniform(0.1, 0.4, (w.batch, 2)), rng.choice([-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], (w.batch, 2))
    parts = [montecarlo.MpccbfRaces(track.point_and_tangent, track.lap_length, track.width, A, B, np.zeros((sl.stop - sl.start, 6)),
                                    np.zeros((sl.stop - sl.start, 6)), s0[sl], cv[sl], ce[sl], vt=0.8, N=10, device=cx.dev)
             for sl in sub_batches(w.batch, args.race_streams)]
    conc = montecarlo.Concurrent(parts, cx.dev)
    from crx import torch_api
    w.key, w.kind, w.N, w.n_obs, w.desc, w.ws = "races", "cbf", 10, 2, parts[0].desc, CatWs(conc, lambda p: p.ws)
    w.kernel = "crx_solve_kernel<2>"
    w.step = conc.step
    for r in parts:
        r.dispatch = args.dispatch
    w.solve_parts = [(lambda r=r: torch_api.cbf_solve_dev(r.desc, r.xc_next, r.xt, r.obs_s, r.obs_e, r.lap_off, r.n_obs, ws=r.ws,
                                                          order=montecarlo._order(r, r.ws.iters))) for r in parts]
    w.name = ("closed-loop MPC-CBF races (tests/auto_mpccbf_test.py scenario family): %d races per GPU, one control step of every race per "
              "step (predictions, window filter, NLP N=10 with 2 scripted cars, plant); %d sub-batches on %d HIP streams" % (w.batch, len(parts), len(parts)))
    w.extra = {"race_streams": len(parts), "kernel_ms_is": "sum of the sub-batch launches of one step, each timed alone (in the step they overlap)", "dispatch": "longest_first" if args.dispatch == "longest_first" else "index"}
    return w


def make_game(cx, args, batch=None):
    """SURVEY.md section 8f rows 1 + 4: B learning-MPC laps of the racing game, device-resident (regression + safe-set
    selection, QP, add_point, plant per control step); every race starts from the reference's recorded safe set."""
    from crx import abi, montecarlo
    from utils import racing_env
    w = Workload()
    g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    w.batch = w.units = batch or 4096
    Bn, N = w.batch, 12
    ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1)); us = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
    qf = np.ascontiguousarray(g["ss/Qfun0"].T); time_ss = g["ss/time_ss"].astype(np.int32)
    rng = np.random.default_rng(60 + cx.rank)
    x0 = np.tile(g["lmpc/x"][0], (Bn, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (Bn, 1))
    x0[:, 0] += rng.uniform(-0.03, 0.03, Bn); x0[:, 5] += rng.uniform(-0.05, 0.05, Bn); xg0[:, 0] = x0[:, 0]
    def part(sl):
        n = sl.stop - sl.start
        return montecarlo.LmpcLaps(track.point_and_tangent, track.lap_length, track.width, np.tile(ss[None], (n, 1, 1, 1)),
                                   np.tile(us[None], (n, 1, 1, 1)), np.tile(qf[None], (n, 1, 1)), np.tile(time_ss[None], (n, 1)),
                                   np.full(n, 2, dtype=np.int32), x0[sl], xg0[sl], np.tile(ss[0, 1:N + 2][None], (n, 1, 1)),
                                   np.tile(us[0, 1:N + 1][None], (n, 1, 1)), N=N, device=cx.dev)

    parts = [part(sl) for sl in sub_batches(Bn, args.race_streams)]
    conc = montecarlo.Concurrent(parts, cx.dev)
    from crx import torch_api
    w.key, w.kind, w.N, w.n_obs, w.desc, w.ws = "game", "lmpc", N, parts[0].desc.n_ss_max, parts[0].desc, CatWs(conc, lambda p: p.ws)
    w.kernel = "crx_lmpc_kernel"
    w.step = conc.step
    for l in parts:
        l.dispatch = args.dispatch
    w.solve_parts = [(lambda l=l: torch_api.lmpc_solve_dev(l.desc, l.xc_next, l.u_prev, l.pws.A, l.pws.B, l.pws.C, l.pws.ss, l.pws.qfun, l.n_ss, ws=l.ws,
                                                           order=montecarlo._order(l, l.ws.iters))) for l in parts]
    w.name = ("learning-MPC laps of the racing game (tests/auto_racing_game_test.py lap 3): %d races per GPU from the reference's recorded safe "
              "set, one control step of every race per step (12 local regressions + safe-set selection, LMPC QP N=12 / 44 points, add_point, plant)" % Bn)
    w.extra = {"note": "races run lap after lap (crx_lmpc_addtraj_dev hands every completed lap over to the safe set) until the four laps of storage are full",
               "race_streams": len(parts), "kernel_ms_is": "sum of the sub-batch launches of one step, each timed alone (in the step they overlap)", "dispatch": "longest_first" if args.dispatch == "longest_first" else "index"}
    to_lap_phase(cx, w, conc, args.lap_phase, "the first learning-MPC lap after the two recorded ones")
    return w


def make_overtake(cx, args, batch=None):
    """SURVEY.md section 8f row 4, the row's stated purpose: closed-loop Monte-Carlo of the racing game WITH traffic -- B laps,
    every race against two scripted cars at random gaps / speeds / lanes, both branches of LMPCRacingGame.calc_input (masked launches) on the
    device (crx.montecarlo.GameLaps)."""
    from crx import montecarlo, synth
    from utils import racing_env
    A, B = synth.load_AB()
    w = Workload()
    g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    opt = np.genfromtxt(os.path.join(ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    w.batch = w.units = batch or 4096
    Bn, N = w.batch, 12
    ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1)); us = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
    qf = np.ascontiguousarray(g["ss/Qfun0"].T); time_ss = g["ss/time_ss"].astype(np.int32)
    rng = np.random.default_rng(70 + cx.rank)
    x0 = np.tile(g["lmpc/x"][0], (Bn, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (Bn, 1))
    s0, v, ey = synth.multi_tests_traffic(Bn, 3, seed=70 + cx.rank)      # the reference's own random traffic (--multi-tests), 3 cars
    tile = lambda a: np.tile(a[None], (Bn,) + (1,) * a.ndim)   # noqa: E731
    def part(sl):
        n = sl.stop - sl.start
        tl = lambda a: np.tile(a[None], (n,) + (1,) * a.ndim)   # noqa: E731
        return montecarlo.GameLaps(track.point_and_tangent, track.lap_length, track.width, A, B, opt, tl(ss), tl(us), tl(qf), tl(time_ss),
                                   np.full(n, 2, dtype=np.int32), x0[sl], xg0[sl], tl(ss[0, 1:N + 2]), tl(us[0, 1:N + 1]), s0[sl], v[sl], ey[sl], device=cx.dev)

    parts = [part(sl) for sl in sub_batches(Bn, args.race_streams)]
    conc = montecarlo.Concurrent(parts, cx.dev)
    from crx import torch_api
    w.key, w.kind, w.N, w.n_obs, w.desc, w.ws = "overtake", "cbf_tracking", 10, 3, parts[0].track_desc, CatWs(conc, lambda p: p.tws)
    w.kernel = "crx_solve_kernel<3>"
    w.step = conc.step
    # the tracking-NLP launch of the last step(): the state it was built for (the plant swapped xc / xc_next since) and its mask
    for g in parts:
        g.dispatch = args.dispatch
    w.solve_parts = [(lambda g=g: torch_api.cbf_solve_dev(g.track_desc, g.lm.xc_next, g.xt, g.obs_s, g.obs_e, g.lap_off, g.n_obs, ws=g.tws,
                                                          active=g.overtake.to(torch.int32), order=montecarlo._order(g, g.tws.iters, g.overtake)))
                     for g in parts]
    w.name = ("racing game with traffic (tests/auto_racing_game_test.py lap 4 / overtake_planner_test.py --multi-tests): %d races per GPU against three "
              "scripted cars each (the reference's random traffic), one control step of every race per step: scene, Bezier/bounds, 4 region QPs + "
              "selection, tracking NLP (N=10, CBF rows), "
              "12 regressions + LMPC QP, add_point, plant -- masked launches, every race runs its own branch" % Bn)
    w.extra = {"note": "status / iteration fields describe the tracking NLP of the overtake branch", "race_streams": len(parts), "kernel_ms_is": "sum of the sub-batch launches of one step, each timed alone (in the step they overlap)",
               "dispatch": "longest_first" if args.dispatch == "longest_first" else "index"}
    # the scene stage keeps at most CRX_MAX_OBS vehicles of interest per race (the nearest): how often were there more?
    w.post = lambda: {"scene_overflow_races": int(sum((g.overflow_seen > 0).sum().item() for g in parts))}
    to_lap_phase(cx, w, conc, args.lap_phase, "the racing-game lap with traffic")
    return w


def kernel_ms_samples(cx, w, reps):
    """Device time of the dominant kernel, HIP events recorded by libcrx on the stream the launch goes to (`crx_timer_*`,
    include/crx.h).  solve() re-issues the dominant launch of the last step() with the same inputs and the same mask (closed-loop
    workloads keep the state the launch was built for); concurrent sub-batches: the launches of one step, one after the other.
    Returns (train, single): `train` = ONE event pair around `reps` back-to-back passes, divided by reps -- the launch as it runs in
    the timed loop (an event pair costs ~5 us of its own: bracketing every launch made the per-launch figure exceed ms_per_step on
    the 0.4 ms headline); `single` = the mean of per-pass event pairs (the lone launch, event overhead included)."""
    from crx import torch_api
    tm = torch_api.Timer()
    fs = getattr(w, "solve_parts", None) or [w.solve]
    kms = []
    for _ in range(reps):
        t_ms = 0.0
        for f in fs:
            tm.begin()
            f()
            tm.end()
            t_ms += tm.ms()
        kms.append(t_ms)
    if len(fs) == 1:
        tm.begin()
        for _ in range(reps):
            fs[0]()
        tm.end()
        train = tm.ms() / reps
    else:   # sub-batches on several streams: no single stream carries the train; the per-pass sum stands
        train = float(np.mean(kms))
    return train, float(np.mean(kms))


def occupancy(w):
    """(LDS bytes of one problem = one single-wave workgroup, problems resident per CU as the runtime computes it: min(LDS, registers))."""
    import crx
    L = crx.lib()
    lk = 1 if w.kind == "lmpc" else 0
    L.crx_debug_lds_bytes.restype = C.c_long
    return int(L.crx_debug_lds_bytes(lk, int(w.N), int(w.n_obs))), int(L.crx_debug_resident_per_cu(lk, int(w.N), int(w.n_obs)))


def measure(cx, w, steps, warmup, with_latency=True):
    """W untimed steps, then exactly `steps` timed steps bracketed by barrier + synchronize, MAX over ranks."""
    for _ in range(warmup):
        w.step()
    if w.rewind is not None:     # closed loops: the timed steps start at the stated lap phase whatever --warmup was
        w.rewind()
    cx.sync_all()
    # a step that IS one launch of the dominant kernel (cfg2, lmpc at index dispatch): HIP events on the launch stream around the timed steps
    # themselves -- the kernel's average duration over the timed region, inside the wall-clock bracket, so that it cannot exceed ms_per_step
    region_tm = None
    if getattr(w, "step_is_one_launch", False):
        from crx import torch_api
        region_tm = torch_api.Timer()
    window = getattr(w, "window", 0)
    if window:
        # closed loops: WHOLE passes over the fixed window (to_lap_phase), one barrier + synchronize bracket per pass -- the sub-batches on their streams
        # pipeline across consecutive control steps, which a bracket per step would destroy (+25 % on `game`).  `steps` is rounded up to a multiple of
        # the window and reported as such: a 20-step request and a 60-step request time the same 60 control steps.
        n_pass = max(1, -(-steps // window))
        elapsed = 0.0
        for _ in range(n_pass):
            w.rewind()
            cx.sync_all()
            t0 = time.perf_counter()
            for _ in range(window):
                w.step()
            cx.sync_all()
            elapsed += time.perf_counter() - t0
        steps = n_pass * window
    else:
        t0 = time.perf_counter()
        if region_tm is not None:
            region_tm.begin()
        for _ in range(steps):
            w.step()
        if region_tm is not None:
            region_tm.end()
        cx.sync_all()
        elapsed = time.perf_counter() - t0
    region_ms = region_tm.ms() / steps if region_tm is not None else None
    if cx.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cx.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units_all = w.units
    if cx.world > 1:                                   # shards may be ragged (strong scaling): sum the units of all ranks
        t = torch.tensor([w.units], dtype=torch.int64, device=cx.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        units_all = int(t.item())
    launched_rate = units_all * steps / elapsed
    allgather_ms = None
    if w.gather is not None:                           # the collective alone, same barrier / synchronize bracket
        reps = min(50, max(5, steps))
        w.gather()
        cx.sync_all()
        t1 = time.perf_counter()
        for _ in range(reps):
            w.gather()
        cx.sync_all()
        allgather_ms = (time.perf_counter() - t1) / reps * 1e3
    lat, hlat = [], []
    if with_latency:
        # latency of one synchronous step (what a 10 Hz controller sees): p50 over blocking calls
        for _ in range(min(100, max(10, steps))):
            t1 = time.perf_counter()
            w.step()
            cx.dsync()
            lat.append((time.perf_counter() - t1) * 1e3)
        # ONE control step as the reference's class surface issues it: host arrays in, host arrays out -- PCIe-inclusive
        if w.host_call is not None:
            w.host_call()
            for _ in range(50):
                t1 = time.perf_counter()
                w.host_call()
                hlat.append((time.perf_counter() - t1) * 1e3)
    # ---- status / iteration fields: read after one more step(), i.e. they describe a launch the workload really issues
    win = None
    if window:     # closed loops: one untimed pass through the whole window, every step's status / iteration counts kept
        w.rewind()
        acc = []
        for _ in range(window):
            w.step()
            cx.dsync()
            acc.append((w.ws.status.cpu().numpy().copy(), w.ws.iters.cpu().numpy().copy()))
        win = (np.concatenate([a[0] for a in acc]), np.concatenate([a[1] for a in acc]))
    else:
        w.step()
        cx.dsync()
    st, it, kkt = w.ws.status.cpu().numpy(), w.ws.iters.cpu().numpy(), w.ws.kkt.cpu().numpy()
    k_ms, k_ms_single = kernel_ms_samples(cx, w, min(50, max(5, steps)))
    k_ms_train = k_ms
    if region_ms is not None:
        k_ms = region_ms
    kkt_unscaled, kkt_parts, n_beyond = None, {}, None
    if w.kind in ("cbf", "cbf_tracking", "planner") and not getattr(w, "solve_parts", None):
        # the UNSCALED KKT quantities of the converged problems (libcrx diagnostics, crx_debug_kkt_unscaled(mode): the same launch once more with kkt[] = 2: the
        # reduced Lagrangian gradient, 3: the constraint violation in the reference's row units, 4: the complementarity -- no s_d, no row scaling); outside
        # every timed region.  Since libcrx 0.4.0 these are what the termination test itself bounds (IPOPT's dual_inf_tol 1, constr_viol_tol 1e-4,
        # compl_inf_tol 1e-4): `value` counts a problem only if its status is 0 AND its violation and complementarity are <= 1e-4 (north_star's figure).
        import crx
        L = crx.lib()
        beyond = np.zeros(len(st), dtype=bool)
        try:
            for mode, name in ((2, "dual"), (3, "viol"), (4, "compl")):
                L.crx_debug_kkt_unscaled(mode)
                w.solve()
                cx.dsync()
                ku, su = w.ws.kkt.cpu().numpy(), w.ws.status.cpu().numpy()
                kkt_parts[name] = float(ku[su == 0].max()) if (su == 0).any() else None
                if name != "dual":
                    beyond |= (su == 0) & ~(ku <= 1e-4)
                else:
                    kkt_parts["dual_above_1e-4"] = int(((su == 0) & ~(ku <= 1e-4)).sum())
        finally:
            L.crx_debug_kkt_unscaled(0)
            w.solve()
            cx.dsync()
        n_beyond = int(beyond.sum())
        vals = [v for k, v in kkt_parts.items() if k in ("dual", "viol", "compl") and v is not None]
        kkt_unscaled = max(vals) if vals else None
    conv = st == 0
    if n_beyond:                # status 0 with an unscaled violation / complementarity beyond 1e-4: not counted (cannot happen at default options)
        conv = conv & ~beyond
    opt = st == 0               # converged at tol (a proved-infeasible planner QP is ANSWERED, not converged: its kkt is +inf by definition)
    if w.kind == "planner":     # a region QP PROVED infeasible (screen / certificate) is an answered problem: the planner consumes the verdict
        conv = conv | (st == 2)
    ran = st != 4
    N, n_obs = w.N, w.n_obs
    abytes = algorithmic_bytes(w.kind, N, n_obs)
    launched = int(ran.sum())                          # masked launches: the problems whose wavefront did not return at once
    conv_of_launched = float(conv[ran].mean()) if ran.any() else 0.0
    achieved = abytes * launched / (k_ms * 1e-3) / 1e9
    # analytic FP64 work: Riccati factor + solves per interior-point iteration
    nx, nu = 6 + n_obs, 2 + n_obs
    nz = nx + nu
    flop_iter = N * (2 * nx * nx * nz + 2 * nx * nz * nz + nu ** 3 / 3 + 2 * nu * nu * (nx + 1) + 2 * nu * nx * (nx + 1)) \
        + N * (4 * nx * nz + 2 * nu * nx) + 40 * N * (8 + 2 * n_obs)
    if w.kind == "lmpc":   # Cholesky of K_u (2N) with 7 carried right-hand sides and its assembly; G = D + T T' in product form
        nu2, M = 2 * N, n_obs
        flop_iter = nu2 ** 3 / 3 + 2 * 7 * nu2 * nu2 / 2 + (45 * 6 + 6 * 12 + 24 * 4) * M + 4 * (N - 1) * nu2 * nu2 / 2 \
            + 2 * nu2 * nu2 + 12 * (N - 1) * nu2
    gflops = float(it.sum()) * flop_iter / (k_ms * 1e-3) / 1e9
    lds, resident = occupancy(w)
    traffic, tsrc = None, "none"     # none: no rocprofv3 --pmc summary under profiles/ for this workload and batch
    try:  # PMC-measured HBM bytes per launch of this workload at this batch (profiles/, rocprofv3 --pmc, calibrated)
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
        e = pm.get(w.key.replace("cfg2_filtered", "cfg2").replace("cfg5_weak", "cfg5"))
        if e and e["batch"] == w.batch:
            if pm.get("kernel_source_sha256") == kernel_source_hash():
                traffic, tsrc = e["traffic_bytes"], "kept"      # kept rocprofv3 --pmc summary of this command on these kernel sources; not measured in this run
            else:
                tsrc = "stale"                                   # the kept summary was measured on other kernel sources
    except Exception:
        traffic = None
    if win is not None:      # fractions and iteration statistics over the window (the roofline figures above stay per launch: the window's last step)
        st, it = win
        conv = st == 0
        ran = st != 4
        conv_of_launched = float(conv[ran].mean()) if ran.any() else 0.0
    cfg = {"workload": w.name, "baseline_config": w.baseline_config, "batch_per_gpu": int(w.batch), "problems_launched": launched,
           "horizon": int(N), "n_obs": 0 if w.kind == "lmpc" else int(n_obs), "n_ss": int(n_obs) if w.kind == "lmpc" else 0,
           "tol": w.desc.opts.tol,
           "status_frac": {"converged": float(conv.mean()), "max_iter": float((st == 1).mean()),
                           "infeasible": float((st == 2).mean()), "restored": float((st == 3).mean()),
                           "skipped_masked": float((st == 4).mean()), "stalled": float((st == 5).mean())},
           "converged_frac": float(conv.mean()), "converged_frac_of_launched": conv_of_launched if ran.any() else None,
           "kkt_max_converged": float(kkt[opt].max()) if opt.any() else None, "kkt_unscaled_max": kkt_unscaled,
           "kkt_unscaled_dual_max": kkt_parts.get("dual"), "kkt_unscaled_viol_max": kkt_parts.get("viol"), "kkt_unscaled_compl_max": kkt_parts.get("compl"),
           "converged_beyond_1e-4_viol_or_compl": n_beyond, "converged_with_dual_above_1e-4": kkt_parts.get("dual_above_1e-4"),
           "iters_p50": float(np.median(it[st != 4])) if (st != 4).any() else 0.0,
           "iters_p90": float(np.percentile(it[st != 4], 90)) if (st != 4).any() else 0.0, "iters_max": int(it.max())}
    if with_latency:
        cfg.update({"p50_step_latency_ms": float(np.median(lat)), "p99_step_latency_ms": float(np.percentile(lat, 99)),
                    "p50_host_call_one_control_step_ms": float(np.median(hlat)) if hlat else None})
    if w.extra:
        cfg.update(w.extra)
    if getattr(w, "post", None):
        cfg.update(w.post())
    # `value` leads with CONVERGED solves (status 0 at tol; planner QPs: or PROVED infeasible, the verdict being the answer the planner
    # consumes): a problem the solver gave up on is not a solve; `value_launched` = every problem the launch worked on
    rec = {"key": w.key, "value": launched_rate * conv_of_launched, "value_launched": launched_rate, "unit": "solves/s", "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": w.scaling, "config": cfg,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "kernel": w.kernel, "kernel_ms": k_ms, "kernel_ms_single_launch": k_ms_single, "kernel_ms_train_after_the_run": k_ms_train,
                        "kernel_ms_is": ("HIP events around the timed steps themselves (one launch per step)" if region_ms is not None else
                                         "HIP events around a train of re-issued launches after the timed steps"), "algorithmic_bytes_per_solve": abytes,
                        "note": "serial-dependency/FP64-issue bound, not HBM bound (DESIGN.md section 5)",
                        "fp64_gflops": gflops, "fp64_frac_of_valu_peak": gflops / FP64_VALU_PEAK_GFLOPS,
                        "lds_bytes_per_problem": lds, "resident_problems_per_cu": resident, "lds_limit_per_cu": int((160 * 1024) // max(lds, 1))}}
    if allgather_ms is not None:
        rec["allgather_ms"] = allgather_ms
        rec["world_size"] = dist.get_world_size() if dist.is_initialized() else 1
    return rec


def r4(x):
    """four significant digits: the stdout line is for the driver's parser, the full precision goes to the full record"""
    if isinstance(x, float) and x == x and abs(x) != float("inf"):
        return float("%.4g" % x)
    return x


def summary8(rec):
    """One sub-config in EIGHT numbers for the stdout line: converged solves/s, launched solves/s, ms per step, ms of
    the dominant kernel, converged share of the launched problems, median and maximum iteration count, HBM-roofline fraction."""
    c, r = rec["config"], rec["roofline"]
    out = {"value": r4(rec["value"]), "value_launched": r4(rec["value_launched"]), "ms_per_step": r4(rec["ms_per_step"]), "kernel_ms": r4(r["kernel_ms"]),
           "converged": r4(c["converged_frac_of_launched"]), "iters_p50": c["iters_p50"], "iters_max": c["iters_max"], "roofline_frac": r4(r["frac"])}
    if "allgather_ms" in rec:
        out["allgather_ms"] = r4(rec["allgather_ms"])
    return out


def stdout_line(full):
    """The ONE line rank 0 prints: headline scalars, `config`, `roofline`, `cpu_baseline`, and `summary` = eight numbers per
    sub-config.  Short on purpose (< 6 KB: the round-3 line had grown to 24 KB and the driver could not parse it); the full
    record -- every sub-config with its own config / roofline object -- is written to `full_record` (a file)."""
    c, r = full["config"], full["roofline"]
    line = {k: (r4(full[k]) if k in ("value", "value_launched", "ms_per_step") else full[k]) for k in
            ("metric", "value", "value_launched", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    ck = ("workload", "baseline_config", "batch_per_gpu", "problems_launched", "horizon", "n_obs", "tol", "converged_frac", "kkt_max_converged", "kkt_unscaled_max",
          "kkt_unscaled_dual_max", "kkt_unscaled_viol_max", "kkt_unscaled_compl_max", "converged_beyond_1e-4_viol_or_compl", "converged_with_dual_above_1e-4",
          "iters_p50", "iters_p90", "iters_max", "p50_step_latency_ms", "p99_step_latency_ms", "p50_host_call_one_control_step_ms", "dispatch")
    line["config"] = {k: r4(c[k]) for k in ck if c.get(k) is not None}
    line["config"]["workload"] = str(c["workload"])[:160]
    line["config"]["status_frac"] = {k: r4(v) for k, v in c["status_frac"].items() if v}
    line["roofline"] = {k: r4(r[k]) for k in r if k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "kernel_ms_single_launch",
                                                "algorithmic_bytes_per_solve", "fp64_gflops", "fp64_frac_of_valu_peak", "lds_bytes_per_problem",
                                                "resident_problems_per_cu")}
    for k in ("allgather_ms", "world_size", "collective"):
        if k in full:
            line[k] = r4(full[k])
    b = full.get("cpu_baseline")
    if b:
        line["cpu_baseline"] = {"value": r4(b["value"]), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "sample": b["sample"][:120],
                                "one_thread": r4(b["one_thread"]["value"]), "scipy_slsqp_one_thread": r4(b.get("scipy_slsqp", {}).get("value")),
                                "reference_casadi": "unavailable" if "unavailable" in b.get("reference", "") else "importable, not timed"}
    line["summary"] = {rec["key"]: summary8(rec) for rec in [full["headline"]] + full.get("configs", [])}
    line["full_record"] = full.get("full_record")
    return line


def cpu_baseline(w):
    """B1/B2 of SURVEY.md section 8d on this box's host cores, bounded to ~25 s: the oracle (a C port of the same
    iteration with a condensed dense Cholesky) on ONE thread and on all cores (OpenMP over problems), and scipy SLSQP on a
    64-problem subsample.  B3, the reference's own CasADi/IPOPT path, is timed only if `import casadi` works here (it does
    not in this image; nothing is substituted)."""
    import oracle

    orc = oracle.load()
    kind, desc, p = w.cpu
    host_threads = oracle.threads()                  # omp_get_max_threads(): the HOST's cores, not what this container may use
    quota = None
    try:   # cgroup v2 CPU quota of the box, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        quota = None
    affinity = len(os.sched_getaffinity(0))
    cores = max(1, min(host_threads, affinity, int(np.ceil(quota)) if quota else host_threads))   # threads actually used

    def run(n):
        a = tuple(v[:n] for v in p.values())
        return {"cbf": orc.cbf_solve, "planner": orc.planner_solve, "lmpc": orc.lmpc_solve}[kind](desc, *a)

    def timed(n, budget):
        run(min(n, 8))
        reps, t0 = 0, time.perf_counter()
        while True:
            run(n)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget or reps >= 2000:
                return n * reps / el, reps, el

    n = min(len(next(iter(p.values()))), 4096)
    oracle.set_threads(1)
    v1, r1, e1 = timed(min(n, 256), 8.0)
    oracle.set_threads(cores)
    vall, rall, eall = timed(n, 10.0)
    out = {"value": vall, "unit": "solves/s", "cores": cores, "cpu_quota_cores": quota, "affinity_cores": affinity, "host_threads": host_threads,
           "speedup_vs_one_thread": vall / v1, "kind": "port",
           "sample": "%d repetitions of %d problems of the same generator and seed%s, OpenMP over problems, %.1f s wall" % (
               rall, n, " (the first %d are the batch)" % w.batch if n > w.batch else "", eall),
           "one_thread": {"value": v1, "cores": 1, "sample": "%d repetitions of the first %d problems, %.1f s wall" % (r1, min(n, 256), e1)}}
    if kind == "cbf" and not desc.per_stage_target:
        from oracle import slsqp_baseline
        rate, costs, viol = slsqp_baseline.time_batch(desc, p, 64)
        ro = run(64)
        rel = np.abs(costs - ro["cost"]) / np.maximum(1.0, np.abs(ro["cost"]))
        feas = viol >= -1e-6
        out["scipy_slsqp"] = {"value": rate, "cores": 1, "sample": "first 64 problems, zero start, analytic gradients, maxiter 300",
                              "feasible_frac": float(feas.mean()), "reached_port_cost_frac": float((feas & (rel <= 1e-6)).mean()),
                              "feasible_and_below_port_cost": int((feas & (ro["status"] == 0) & (costs < ro["cost"] - 1e-6 * np.maximum(1, np.abs(ro["cost"])))).sum())}
    try:
        import casadi  # noqa: F401
        out["reference"] = "casadi importable: reference path NOT timed in this round"
    except Exception:
        out["reference"] = "casadi: unavailable (not installed; no network) -> reference CasADi/IPOPT path not timed"
    return out


def self_spawn(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # re-launch THIS program (sys.argv[0]: bench.py, or the harness of tests/test_bench_plumbing.py that drives it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


SWEEP_BACKEND = None   # solver back-end of the cfg5 sweep (None = crx.torch_api = libcrx; the CPU harness in tests/ puts its stand-in here)


def init_backend(cx, args):
    """GPU, process group (nccl = RCCL) and libcrx for this rank."""
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; libcrx has no CPU path", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(cx.local)
    if cx.world > 1 or (args.force_collective and args.collective == "torch"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=cx.dev, rank=cx.rank, world_size=cx.world)
    import crx
    from crx import abi
    crx.init(cx.local)
    # solver options that differ from the defaults travel in crx_ipm_opts of every descriptor this run builds (ABI 0.2: no process-global switches)
    if args.no_reach_screen:
        abi.OPTS_OVERRIDE["reach_screen"] = 0
    if args.slack_start is not None:
        abi.OPTS_OVERRIDE["slack_start"] = int(args.slack_start)
    if args.qp_method is not None:
        abi.OPTS_OVERRIDE["qp_method"] = int(args.qp_method)


def shutdown_backend(cx, args):
    from crx import dist as cdist
    cdist.CrxComm.destroy()


def headline_workload(cx, args, make):
    return make[args.workload or "cfg2"]()


def sub_configs(cx, args):
    """every other single-GPU BASELINE config in the same driver-timed run: (key, maker, steps, warmup, with_latency); slow configs get fewer steps (stated)"""
    if args.workload is not None or args.no_sub_configs:
        return []
    return [("cfg2_filtered", lambda: make_cbf(cx, "cfg2_filtered", args, None, filtered=True), args.steps, args.warmup, False),
            ("cfg2_x4_streams", lambda: make_cbf_streams(cx, args, 4), args.steps, args.warmup, False),
            ("cfg3", lambda: make_planner(cx, args), args.steps, args.warmup, True),
            ("cfg4", lambda: make_cbf(cx, "cfg4", args), min(args.steps, 30), min(args.warmup, 3), True),
            ("lmpc", lambda: make_lmpc(cx, args), min(args.steps, 100), min(args.warmup, 5), True),
            ("cfg5_weak", lambda: make_sweep(cx, args, "weak"), min(args.steps, 40), min(args.warmup, 3), False),
            ("cfg5_strong", lambda: make_sweep(cx, args, "strong"), min(args.steps, 10), min(args.warmup, 2), False),
            ("game", lambda: make_game(cx, args), min(args.steps, 60), min(args.warmup, 5), False),
            ("overtake", lambda: make_overtake(cx, args), min(args.steps, 60), min(args.warmup, 5), False),
            ("races", lambda: make_races(cx, args), min(args.steps, 30), min(args.warmup, 3), False)]


def write_full_record(full, path):
    """The full record (every sub-config with its config / roofline objects) goes to a FILE, not to stdout.  Default: gpurun_out/ (the
    scratch directory that travels back from the GPU box); copy what is to be kept into profiles/."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        print("bench.py: full record not written (%s)" % e, file=sys.stderr)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg2_filtered", "cfg2_streams", "cfg3", "cfg4", "cfg5", "lmpc", "races", "game", "overtake"],
                    help="measure only this workload (default: headline cfg2 + every other single-GPU config in `summary`)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="cfg5 only")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU problems (cfg2/cfg4/lmpc/races) or scenarios (cfg3); 0 = BASELINE size")
    ap.add_argument("--sweep-per-gpu", type=int, default=16384, help="cfg5 weak: scenarios per GPU")
    ap.add_argument("--sweep-total", type=int, default=131072, help="cfg5 strong: scenarios in total")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-configs", action="store_true", help="headline only")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full_last.json"),
                    help="file that receives the full record (stdout carries the short line only)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the nccl (= RCCL) process group and issue the winners' all-gather even at world size 1 (a 1-GPU box can exercise the collective)")
    ap.add_argument("--race-streams", type=int, default=4,
                    help="closed-loop workloads (races, game, overtake): independent sub-batches of the races on this many HIP streams (1 = one batch)")
    ap.add_argument("--dispatch", default="auto", choices=["auto", "index", "longest_first", "start_barrier"],
                    help="workgroup -> problem mapping of the solver launches (crx_*_solve_ordered_dev): index = launch order; longest_first = "
                         "the problems whose previous solve took most iterations first (crx_order_longest_first_dev: what a closed loop has; on the "
                         "STATIC batches cfg2 / cfg4 / lmpc the previous solve is the same problem, i.e. an oracle order: an upper bound); "
                         "start_barrier (cfg2 / cfg4) = from the problem's own inputs, no previous solve needed (crx_cbf_order_dev); auto = "
                         "start_barrier for a CBF batch larger than the resident slots, else index.  The order kernel is part of the timed step")
    ap.add_argument("--lap-phase", type=int, default=40,
                    help="closed-loop workloads game / overtake: untimed control steps before the measurement; the timed steps start there whatever --warmup is")
    ap.add_argument("--no-reach-screen", action="store_true",
                    help="planner QPs: reachability screen off (crx_ipm_opts.reach_screen = 0): every region goes through the interior-point iteration")
    ap.add_argument("--qp-method", type=int, default=None, choices=[0, 1],
                    help="the convex rows (planner / path / learning-MPC QPs), crx_ipm_opts.qp_method: 0 = Mehrotra's predictor-corrector (default), "
                         "1 = IPOPT's monotone barrier + filter line search (libcrx <= 0.3)")
    ap.add_argument("--slack-start", type=int, default=None, choices=[0, 1, 2, 3],
                    help="CBF NLPs, crx_ipm_opts.slack_start: 0 = IPOPT's sigma = 0 start only, 1 = start the slacks at their provable lower bounds, "
                         "2 (default) = the crash path (provable crash states start from a feasible interior point, a stalled solve restarts from it), "
                         "3 = the eager crash path (every violated zero start takes that point: faster, further from the reference's local minimum)")
    ap.add_argument("--collective", default="torch", choices=["torch", "crx"],
                    help="cfg5's all-gather: torch.distributed (nccl = RCCL) or libcrx's own crx_allgather_winners_dev (RCCL through the C ABI)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    cx = Ctx()
    if args.gpus != cx.world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, cx.world), file=sys.stderr)
        sys.exit(2)
    init_backend(cx, args)
    from crx import dist as cdist
    cdist.COLLECTIVE = args.collective
    if args.force_collective:
        cdist.FORCE_COLLECTIVE = True
    b = args.batch or None
    make = {"cfg2": lambda: make_cbf(cx, "cfg2", args, b), "cfg2_streams": lambda: make_cbf_streams(cx, args, 4), "cfg2_filtered": lambda: make_cbf(cx, "cfg2_filtered", args, b, filtered=True),
            "cfg3": lambda: make_planner(cx, args, b), "cfg4": lambda: make_cbf(cx, "cfg4", args, b),
            "cfg5": lambda: make_sweep(cx, args, args.scaling), "lmpc": lambda: make_lmpc(cx, args, b), "races": lambda: make_races(cx, args, b),
            "game": lambda: make_game(cx, args, b), "overtake": lambda: make_overtake(cx, args, b)}
    head = headline_workload(cx, args, make)
    rec = measure(cx, head, args.steps, args.warmup, with_latency=head.host_call is not None)
    full = {"metric": METRIC, "value": rec["value"], "value_launched": rec["value_launched"], "unit": "solves/s", "n_gpus": cx.world,
            "steps": rec["steps"], "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": rec["scaling"],   # (rec["steps"] == --steps but for the closed loops: whole passes of their lap window)
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": rec["config"], "roofline": rec["roofline"], "headline": rec,
            "kernel_source_sha256": kernel_source_hash()}
    for k in ("allgather_ms", "world_size"):
        if k in rec:
            full[k] = rec[k]
    if cx.rank == 0 and cx.world == 1 and not args.no_cpu_baseline and head.cpu is not None:   # N = 1 only: the other ranks would wait for it
        full["cpu_baseline"] = cpu_baseline(head)
    full["configs"] = []
    for key, mk, st, wu, wl in sub_configs(cx, args):
        w = mk()
        full["configs"].append(measure(cx, w, st, wu, with_latency=wl))
        del w
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    if cx.world > 1:
        # the headline (cfg2, weak) shards by problem and has NO collective; the path's one exchange step is the winners' all-gather of the
        # planner sweep -- its time over all ranks goes to the top level of the line so that a scaling record shows RCCL saw N ranks
        for r_ in full["configs"]:
            if r_["key"] == "cfg5_weak" and "allgather_ms" in r_:
                full["allgather_ms"], full["world_size"] = r_["allgather_ms"], r_["world_size"]
                full["collective"] = ("ONE all-gather of %d-byte winner records {int32 flag; int32 status; double X[13][6]} of the cfg5 weak sweep (%d "
                                      "scenarios per rank), %s over %d ranks; the headline itself has no collective" % (
                                          r_["config"]["winner_record_bytes"], r_["config"]["scenarios_this_rank"],
                                          "torch.distributed nccl (= RCCL)" if args.collective == "torch" else "crx_allgather_winners_dev (RCCL)", r_["world_size"]))
    shutdown_backend(cx, args)
    if dist.is_initialized():
        dist.destroy_process_group()
    if cx.rank == 0:
        full["full_record"] = write_full_record(full, args.full_out)
        # the JSON line must be the LAST line on stdout: RCCL prints its version banner through C stdio, which (on a pipe)
        # is flushed only at exit, i.e. after anything Python printed -- flush C stdio first, then print
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(stdout_line(full)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
