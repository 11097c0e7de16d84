#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: NLP solves/s (N=12, 6-state bicycle).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|lmpc]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM.  Default workload = BASELINE.json configs[1]: MPC-CBF NLP (control/control.py:476-607 of
the reference), 1 obstacle, batch 256, N = 12 -- per GPU (weak scaling: the batch shards by problem,
no data-path collective).  `--workload cfg3` = 1024 planner scenarios x 4 region QPs + region
selection per GPU, followed by ONE all-gather of the winning trajectories (the only exchange step
the path has); cfg4 = 16384 tracking NLPs, N = 20, 3 obstacles; lmpc (not a
BASELINE config) = 4096 learning-MPC QPs (control.py:610-730): the certified instances recorded from the
reference's LMPC lap (tests/golden/racing_game.npz), tiled.

Rank 0 prints ONE JSON line.  `value` counts every problem handed to the solver per second of the
timed region, whole job (all GPUs); converged fraction and KKT bound are reported beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "car-racing_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_GFLOPS = 78600.0  # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (vector FP64, spec)


def algorithmic_bytes(workload, N, n_obs):
    """doubles in + doubles out per solve, times 8."""
    if workload == "cfg3":
        return (11 * N + 18) * 8                                   # a1: 1200 B at N = 12
    if workload == "lmpc":                                          # n_obs carries M here
        return (8 + 54 * N + 7 * n_obs + 1 + 6 * (N + 1) + 2 * N + n_obs + 3) * 8   # 8912 B at N = 12, M = 44
    tgt = (N + 1) if workload == "cfg4" else 0                      # per-stage ey target
    d_in = 6 + 6 + 2 * n_obs * (N + 1) + n_obs + tgt
    d_out = 6 * (N + 1) + 2 * N + n_obs * (N + 1) + 3
    return (d_in + d_out) * 8                                       # a5: 1256 B (N=12, 1 obs); a6: 3152 B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "lmpc", "races"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU problems (cfg2/cfg4) or scenarios (cfg3); 0 = BASELINE size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-prep", action="store_true",
                    help="cfg3: start each step from the raw scenarios (crx_planner_prep on the device) instead of prepared QP arrays")
    ap.add_argument("--no-scenario-filter", action="store_true",
                    help="keep doomed / unsafe-start scenarios in the synthetic batch (DESIGN.md section 6)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d needs torch.distributed.run (one process per GPU)" % args.gpus, file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; libcrx has no CPU path", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import crx
    from crx import abi, synth, torch_api

    crx.init(local)
    A, B = synth.load_AB()
    wl = args.workload
    seed_shift = 1000 * rank

    def to_dev(a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return t.to(dtype) if dtype is not None else t

    races = None
    if wl == "races":
        # B closed-loop MPC-CBF races, one control step of all of them per bench step
        from crx import montecarlo
        from utils import racing_env
        track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
        batch = args.batch or 4096
        rng = np.random.default_rng(50 + rank)
        s0 = np.sort(rng.uniform(3.0, 17.0, (batch, 2)), axis=1)
        s0[:, 1] = np.maximum(s0[:, 1], s0[:, 0] + 2.0)
        races = montecarlo.MpccbfRaces(track.point_and_tangent, track.lap_length, track.width, A, B, np.zeros((batch, 6)),
                                       np.zeros((batch, 6)), s0, rng.uniform(0.1, 0.4, (batch, 2)),
                                       rng.choice([-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], (batch, 2)), vt=0.8, N=10, device=dev)
        desc, ws, N, n_obs, units = races.desc, races.ws, 10, 2, batch
        p = None
        step = races.step
        name = "closed-loop MPC-CBF races (tests/auto_mpccbf_test.py scenario family): %d races per GPU, one control step of every race per step (predictions, window filter, NLP N=10 with 2 scripted cars, plant)" % batch
    elif wl == "lmpc":
        g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
        ok = np.nonzero(g["lmpc_success"])[0]
        batch = args.batch or 4096
        idx = ok[(np.arange(batch) + 7 * rank) % len(ok)]
        N, M = g["lmpc/A"].shape[1], g["lmpc/ss"].shape[2]
        desc = abi.lmpc_desc(N=N, n_ss_max=M)
        p = dict(x0=g["lmpc/x"][idx], u_old=g["lmpc/u_old"][idx], A=g["lmpc/A"][idx].reshape(batch, N, 36),
                 B=g["lmpc/B"][idx].reshape(batch, N, 12), C=g["lmpc/C"][idx], ss=g["lmpc/ss"][idx], qfun=g["lmpc/qfun"][idx],
                 n_ss=np.full(batch, M, dtype=np.int32))
        n_obs = M
        t_in = [to_dev(p[k]) for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun")] + [to_dev(p["n_ss"], torch.int32)]
        ws = torch_api.LmpcWorkspace(desc, batch, dev)
        units = batch

        def step():
            torch_api.lmpc_solve_dev(desc, *t_in, ws=ws)

        name = "learning-MPC QP (control.py:610-730), N=%d, %d safe-set points, LTV models and safe sets recorded from the reference's LMPC lap, batch %d/GPU" % (N, M, batch)
    elif wl in ("cfg2", "cfg4"):
        if wl == "cfg2":
            batch = args.batch or 256
            p = synth.cfg2_mpccbf(batch, N=12, seed=2 + seed_shift, safe_start=not args.no_scenario_filter)
            desc = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
        else:
            batch = args.batch or 16384
            p = synth.cfg4_tracking_cbf(batch, N=20, seed=4 + seed_shift, safe_start=not args.no_scenario_filter)
            desc = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
        N, n_obs = desc.N, desc.n_obs_max
        t_in = [to_dev(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [to_dev(p["n_obs"], torch.int32)]
        ws = torch_api.CbfWorkspace(desc, batch, dev)
        units = batch

        def step():
            torch_api.cbf_solve_dev(desc, *t_in, ws=ws)

        name = ("MPC-CBF NLP (control.py:476-607), l_shape model, 1 obstacle, N=12, batch 256/GPU" if wl == "cfg2"
                else "tracking NLP with CBF rows (control.py:251-473 form), 3 obstacles, N=20, batch %d/GPU" % batch)
    else:
        n_scen = args.batch or 1024
        p = synth.cfg3_planner(n_scen, N=12, seed=3 + seed_shift)
        N, n_obs, V = 12, 0, p["V"]
        desc = abi.planner_desc(N, A, B)
        sdesc = abi.select_desc(N, V, p["lap_length"])
        t_in = [to_dev(p[k]) for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
        t_sel = [to_dev(p["n_veh"], torch.int32), to_dev(p["obs_s"]), to_dev(p["obs_ey"]), to_dev(p["old_flag"], torch.int32)]
        batch = n_scen * (V + 1)
        ws = torch_api.PlannerWorkspace(desc, batch, dev)
        sws = torch_api.SelectWorkspace(sdesc, n_scen, dev)
        gathered = torch.empty((world * n_scen, N + 1, 6), dtype=torch.float64, device=dev) if world > 1 else None
        units = batch
        if args.device_prep:
            w = p["raw"]
            pdesc = abi.prep_desc(N, V, len(w["opt_s"]), w["track_width"], w["lap_length"])
            t_raw = [to_dev(w["x"]), to_dev(w["x"]), to_dev(w["n_veh"], torch.int32), to_dev(w["veh_info"]), to_dev(w["max_dv"]),
                     to_dev(w["obs_s"]), to_dev(w["obs_ey"]), to_dev(w["opt_s"]), to_dev(w["opt_ey"])]
            pws = torch_api.PrepWorkspace(pdesc, n_scen, dev)
            t_in = [pws.x0, pws.bez_s, pws.bez_ey, pws.ey_lb, pws.ey_ub]

        def step():
            if args.device_prep:
                torch_api.planner_prep_dev(pdesc, *t_raw, ws=pws)
            torch_api.planner_solve_dev(desc, *t_in, ws=ws)
            torch_api.select_dev(sdesc, t_sel[0], ws.X.view(n_scen, V + 1, N + 1, 6), t_sel[1], t_sel[2], t_sel[3], ws=sws)
            if world > 1:  # the path's only exchange: winners to every rank (RCCL all-gather over xGMI)
                dist.all_gather_into_tensor(gathered, sws.best_X)

        name = "overtake planner: %d scenarios x %d region QPs (overtake_traj_planner.py:248-379) + selection (:205-246), N=12, per GPU%s" % (
            n_scen, V + 1, "; Bezier references and ey bounds built on the device from the raw scenarios" if args.device_prep else "")

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = units * world * args.steps / elapsed

    # ---- per-launch kernel time, measured with HIP events on the launch stream -----------------------
    L = crx.lib()
    L.crx_set_timing(1)
    kms = []
    for _ in range(min(50, max(5, args.steps))):
        if wl == "cfg3":
            torch_api.planner_solve_dev(desc, *t_in, ws=ws)
        elif wl == "lmpc":
            torch_api.lmpc_solve_dev(desc, *t_in, ws=ws)
        elif wl == "races":
            torch_api.cbf_solve_dev(desc, races.xc, races.xt, races.obs_s, races.obs_e, races.lap_off, races.n_obs, ws=ws)
        else:
            torch_api.cbf_solve_dev(desc, *t_in, ws=ws)
        kms.append(L.crx_last_kernel_ms())
    L.crx_set_timing(0)
    k_ms = float(np.mean(kms))
    # latency of one synchronous control step (what a 10 Hz controller sees): p50 over blocking calls
    lat = []
    for _ in range(min(100, max(10, args.steps))):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    # ONE control step as the reference's class surface issues it: host arrays in, host arrays out, the batch
    # one step sees (1 NLP / 1 QP, or the V+1 region QPs + selection of one planner call) -- PCIe-inclusive
    hb = crx.binding()
    if wl == "races":
        one = lambda: None  # noqa: E731  (the class-surface step of this scenario is the cfg2-type call)
    elif wl == "cfg3":
        R1 = V + 1
        a1 = tuple(p[k][:R1] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")) + tuple(p[k][:1] for k in ("n_veh", "obs_s", "obs_ey", "old_flag"))
        one = lambda: hb.planner_plan(desc, sdesc, *a1)  # noqa: E731
    elif wl == "lmpc":
        a1 = tuple(p[k][:1] for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun", "n_ss"))
        one = lambda: hb.lmpc_solve(desc, *a1)  # noqa: E731
    else:
        a1 = tuple(p[k][:1] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs"))
        one = lambda: hb.cbf_solve(desc, *a1)  # noqa: E731
    one()
    hlat = []
    for _ in range(50):
        t1 = time.perf_counter()
        one()
        hlat.append((time.perf_counter() - t1) * 1e3)
    st = ws.status.cpu().numpy()
    it = ws.iters.cpu().numpy()
    kkt = ws.kkt.cpu().numpy()
    conv = st == 0
    abytes = algorithmic_bytes(wl, N, n_obs)
    achieved = abytes * batch / (k_ms * 1e-3) / 1e9
    # analytic FP64 work: Riccati factor + solves per interior-point iteration
    nx, nu = 6 + n_obs, 2 + n_obs
    nz = nx + nu
    flop_iter = N * (2 * nx * nx * nz + 2 * nx * nz * nz + nu ** 3 / 3 + 2 * nu * nu * (nx + 1) + 2 * nu * nx * (nx + 1)) \
        + N * (4 * nx * nz + 2 * nu * nx) + 40 * N * (8 + 2 * n_obs)
    if wl == "lmpc":   # Cholesky of K_u (2N) with 7 carried right-hand sides and its assembly; G = D + T T' in product form:
        # 21 + 24 wave scans (6 adds per element) and ~12 element-wise operations per rank-one factor and solve step
        nu2, M = 2 * N, n_obs
        flop_iter = nu2 ** 3 / 3 + 2 * 7 * nu2 * nu2 / 2 + (45 * 6 + 6 * 12 + 24 * 4) * M + 4 * (N - 1) * nu2 * nu2 / 2 \
            + 2 * nu2 * nu2 + 12 * (N - 1) * nu2
    gflops = float(it.sum()) * flop_iter / (k_ms * 1e-3) / 1e9

    L.crx_debug_lds_bytes.restype = C.c_long
    lds = int(L.crx_debug_lds_bytes(1 if wl == "lmpc" else 0, int(N), int(n_obs)))
    resident = int(L.crx_debug_resident_per_cu(1 if wl == "lmpc" else 0, int(N), int(n_obs)))   # runtime: min(LDS, registers)
    traffic = None
    try:  # PMC-measured HBM bytes per launch of this workload at this batch (profiles/, collected with rocprofv3 --pmc)
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
        if wl in pm and pm[wl]["batch"] == batch:
            traffic = pm[wl]["traffic_bytes"]
    except Exception:
        traffic = None
    out = {
        "metric": "NLP solves/sec (N=12, 6-state bicycle); p50 per-step solve latency",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": name, "baseline_config": {"cfg2": 1, "cfg3": 2, "cfg4": 3, "lmpc": None, "races": None}[wl], "batch_per_gpu": int(batch),
                   "horizon": int(N), "n_obs": 0 if wl == "lmpc" else int(n_obs), "n_ss": int(n_obs) if wl == "lmpc" else 0,
                   "tol": desc.opts.tol,
                   "scenario_filter": not args.no_scenario_filter,
                   "converged_frac": float(conv.mean()), "kkt_max_converged": float(kkt[conv].max()) if conv.any() else None,
                   "iters_p50": float(np.median(it)), "iters_max": int(it.max()),
                   "p50_step_latency_ms": float(np.median(lat)), "p99_step_latency_ms": float(np.percentile(lat, 99)),
                   "p50_host_call_one_control_step_ms": None if wl == "races" else float(np.median(hlat))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "crx_lmpc_kernel" if wl == "lmpc" else "crx_solve_kernel<%d>" % n_obs, "kernel_ms": k_ms, "algorithmic_bytes_per_solve": abytes,
                     "note": "serial-dependency/FP64-latency bound, not HBM bound (DESIGN.md section 5)",
                     "fp64_gflops": gflops, "fp64_frac_of_valu_peak": gflops / FP64_VALU_PEAK_GFLOPS,
                     "lds_bytes_per_problem": lds, "resident_problems_per_cu": resident, "lds_limit_per_cu": int((160 * 1024) // lds)},
    }
    if rank == 0 and not args.no_cpu_baseline and wl != "races":
        out["cpu_baseline"] = cpu_baseline(wl, desc, p, batch)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(wl, desc, p, batch):
    """The oracle (a C port of the same iteration, condensed dense Cholesky) on the host cores of this
    box, on a bounded sample of the same workload.  The reference's own CasADi/IPOPT path is timed
    only if `import casadi` works here (it does not in this image; nothing is substituted)."""
    import oracle

    orc = oracle.load()
    cores = oracle.threads()
    n = min(batch, 1024)
    if wl == "cfg3":
        a = (p["x0"][:n], p["bez_s"][:n], p["bez_ey"][:n], p["ey_lb"][:n], p["ey_ub"][:n])
        fn = lambda: orc.planner_solve(desc, *a)  # noqa: E731
    elif wl == "lmpc":
        a = tuple(p[k][:n] for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun", "n_ss"))
        fn = lambda: orc.lmpc_solve(desc, *a)  # noqa: E731
    else:
        a = (p["x0"][:n], p["xt"][:n], p["obs_s"][:n], p["obs_ey"][:n], p["lap_off"][:n], p["n_obs"][:n])
        fn = lambda: orc.cbf_solve(desc, *a)  # noqa: E731
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el > 8.0 or reps >= 400:
            break
    try:
        import casadi  # noqa: F401
        ref = "casadi importable: reference path NOT timed in this round"
    except Exception:
        ref = "casadi: unavailable (not installed; no network) -> reference CasADi/IPOPT path not timed"
    return {"value": n * reps / el, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d repetitions of the first %d problems of the same batch, OpenMP over problems, %.1f s wall" % (reps, n, el),
            "reference": ref}


if __name__ == "__main__":
    main()
