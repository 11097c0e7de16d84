"""Batched closed-loop races, device-resident from the first step to the last.

`mpccbf_races` is the simulation loop of the reference's MPC-CBF racing scenario
(tests/auto_mpccbf_test.py:9-46 -> racing/offboard.py:114-131 -> utils/base.py:780-794) for B independent
races at once: per control step, obstacle predictions of the scripted cars (utils/base.py:879-886), the
window filter and lap offsets of control.mpccbf (control/control.py:499-523,538-540), crx_cbf_prep_dev), ONE crx_cbf_solve_dev over all races, ONE crx_plant_step_wrap_dev incl. the lap
bookkeeping (utils/base.py:795-819).  Three libcrx launches per control step, nothing returns to the host
inside the loop; torch only owns the device memory.

The reference runs such sweeps one race at a time (car_racing/tests/overtake_planner_test.py
--multi-tests); this is the same experiment with the race index as the batch dimension.
"""
import numpy as np
import torch

from . import abi, torch_api

_LAP_TRUNC = torch.trunc


class _Noise:
    """The plant's process noise (utils/base.py:929-939) for the batched loops: standard-normal draws from a torch generator
    on the device, one [B, 3] tensor per control step; clipping and scaling happen in crx_plant_step_noise_dev.  seed None =
    zero noise (the `--zero-noise` runs of the reference's scripts; the reference's default is noise ON)."""

    def __init__(self, seed, device):
        self.gen = None
        if seed is not None:
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(int(seed))
        self.device = device

    def draw(self, batch):
        if self.gen is None:
            return None
        return torch.randn((batch, 3), generator=self.gen, dtype=torch.float64, device=self.device)


def _order(obj, iters, active=None):
    """Dispatch order of the next solver launch (include/crx.h, "Dispatch order"): obj.dispatch = "longest_first" lists the races
    whose previous solve took most iterations first (and the masked-out ones last); "index" (default) = launch order."""
    if getattr(obj, "dispatch", "index") != "longest_first":
        return None
    return torch_api.longest_first(iters, active)


class Snapshot:
    """The state of a closed-loop object (MpccbfRaces / LmpcLaps / GameLaps, or a Concurrent of them) at this moment: every tensor
    and every plain number reachable through its attributes (workspaces and nested loops included) is copied; restore() puts the
    values AND the attribute bindings back (step() swaps xc / xc_next); tensors held in lists, tuples and dicts and the state of torch.Generator
    objects (the plant-noise stream) are captured too [r6]; objects without a __dict__ hold nothing that is captured.  bench.py uses it to start the timed steps of the closed-loop
    workloads at a stated lap phase whatever --steps / --warmup are.  The caller synchronises the device around both calls."""

    def __init__(self, obj):
        import ctypes
        self.tensors, self.scalars, self.items, self.gens = [], [], [], []
        seen = set()
        skip = (ctypes.Structure, torch.cuda.Stream, type)

        def visit(holder, key, v, is_item):
            """one value reachable from `obj`: holder.key (attribute) or holder[key] (list / dict item)"""
            if torch.is_tensor(v):
                (self.items if is_item else self.tensors).append((holder, key, v, v.clone()))
            elif isinstance(v, torch.Generator):
                self.gens.append((v, v.get_state()))          # the plant-noise stream restarts where it was
            elif isinstance(v, (bool, int, float)):
                if not is_item:
                    self.scalars.append((holder, key, v))
            elif isinstance(v, (list, tuple)):
                if id(v) not in seen:
                    seen.add(id(v))
                    for i, e in enumerate(v):
                        visit(v, i, e, True)
            elif isinstance(v, dict):
                if id(v) not in seen:
                    seen.add(id(v))
                    for k2, e in list(v.items()):
                        visit(v, k2, e, True)
            elif hasattr(v, "__dict__") and not isinstance(v, skip):
                walk(v)

        def walk(o):
            if id(o) in seen or not hasattr(o, "__dict__"):      # (objects with __slots__ only: nothing to capture)
                return
            seen.add(id(o))
            for k, v in list(vars(o).items()):
                visit(o, k, v, False)

        walk(obj)

    def restore(self):
        for o, k, t, c in self.tensors:
            t.copy_(c)
            setattr(o, k, t)
        for holder, k, t, c in self.items:             # tensors held in lists / dicts: values back; a mutable holder also gets its binding back
            t.copy_(c)
            if not isinstance(holder, tuple):
                holder[k] = t
        for o, k, v in self.scalars:
            setattr(o, k, v)
        for gen, state in self.gens:
            gen.set_state(state)


class Concurrent:
    """K independent sub-batches of races stepping on K HIP streams.  Races do not interact, so a batch can be cut anywhere; what
    the cut buys is OVERLAP: the streams free-run (no join per step), a sub-batch's plant (100 serial Euler sub-steps per vehicle:
    ~0.2 ms on 64 wavefronts, 6 % of the chip) and the straggler tail of its solver launch run while the next sub-batch's solver
    launch occupies the other CUs.  Every race computes exactly what it computes in one big batch (bit-identical).  `parts` are
    MpccbfRaces / LmpcLaps / GameLaps objects over disjoint slices of the races; step() issues one control step of every part and
    returns without waiting; call torch.cuda.synchronize() (or sync()) before reading results.
    Streams only overlap when they sit on different HARDWARE queues: the HIP runtime multiplexes a process's streams onto
    GPU_MAX_HW_QUEUES of them (default 4) and streams sharing one serialise -- two sub-batches on one queue are SLOWER than one
    batch.  Which streams share is not ours to choose and, with torch's stream pool, depended on how many streams the process
    had handed out before (3.24 .. 5.6 ms per racing-game step: tools/stream_offset_probe.py), so the streams are libcrx's:
    crx_streams_create measures which of its candidates overlap and returns a set that does (include/crx.h "Streams").  Two
    sub-batches of GameLaps want six (one each + the two branch streams of each): set GPU_MAX_HW_QUEUES=8 in the environment
    before the runtime initialises (bench.py does); with fewer queues than that the parts run their branches one after the
    other on their sub-batch stream.  Measured, 4096 races per step (tools/gpu_round3_i.sh): races 0.784 -> 0.687 ms,
    learning-MPC laps 3.82 -> 3.31 ms, racing game 3.56 -> 3.25 ms with two sub-batches and 8 queues."""

    def __init__(self, parts, device=None, streams=None):
        self.parts = list(parts)
        dev = torch.device(device if device is not None else "cuda")
        self.device = dev
        K = len(self.parts)
        inner = [p for p in self.parts if hasattr(p, "set_branch_streams")]
        if streams is not None:            # the caller's streams (K, or K + 2 per GameLaps part), taken as they are
            self.n_concurrent = len(streams)
        else:
            streams, self.n_concurrent = torch_api.new_streams(K + 2 * len(inner), dev)   # the first n_concurrent overlap pairwise
        self.streams = streams[:K]
        for i, part in enumerate(inner):
            if K == 1 or self.n_concurrent >= K + 2 * len(inner):
                part.set_branch_streams(streams[K + 2 * i], streams[K + 2 * i + 1])
            else:                          # not enough hardware queues for nested streams: branches in sequence on the part's stream
                part.overlap = False
        cur = torch.cuda.current_stream(dev)
        for st in self.streams:            # whatever set the parts up on the caller's stream comes first
            st.wait_stream(cur)

    def step(self):
        for part, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                part.step()

    def sync(self):
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def cat(self, get):
        """Concatenate a per-race tensor of every part (after sync()): get(part) -> tensor [B_part, ...]."""
        self.sync()
        return torch.cat([get(p) for p in self.parts], dim=0)


class MpccbfRaces:
    """State of B races on the device; step() advances all of them by one control step."""

    def __init__(self, track_table, lap_length, track_width, A, B, xcurv0, xglob0, car_s0, car_v, car_ey,
                 vt=0.8, eyt=0.0, N=10, alpha=0.8, timestep=0.1, device=None, noise_seed=None):
        dev = torch.device(device if device is not None else "cuda")
        f64 = dict(dtype=torch.float64, device=dev)
        self.noise = _Noise(noise_seed, dev)
        self.xc = torch.as_tensor(np.ascontiguousarray(xcurv0), **f64).clone()
        self.xg = torch.as_tensor(np.ascontiguousarray(xglob0), **f64).clone()
        self.s0, self.v, self.ey = (torch.as_tensor(np.ascontiguousarray(a), **f64) for a in (car_s0, car_v, car_ey))
        Bn, V = self.s0.shape
        if V > abi.CRX_MAX_OBS:
            raise ValueError("at most %d scripted cars" % abi.CRX_MAX_OBS)
        self.N, self.V, self.batch, self.lap_length, self.timestep = N, V, Bn, lap_length, timestep
        self.tab = torch.as_tensor(np.ascontiguousarray(track_table), **f64)
        self.desc = abi.cbf_desc(N, V, A, B, alpha=alpha, margin=0.2, ey_max=track_width)
        self.pdesc = abi.plant_desc(self.tab.shape[0], lap_length, timestep=timestep)
        self.xt = torch.tensor([vt, 0, 0, 0, 0, eyt], **f64).repeat(Bn, 1).contiguous()
        self.ws = torch_api.CbfWorkspace(self.desc, Bn, dev)
        self.jdt = torch.arange(N + 1, **f64) * timestep
        self.laps = torch.zeros(Bn, dtype=torch.int32, device=dev)
        self.ar = torch.arange(V, device=dev)
        self.obs_s = torch.empty((Bn, V, N + 1), **f64)
        self.obs_e = torch.empty((Bn, V, N + 1), **f64)
        self.lap_off = torch.empty((Bn, V), **f64)
        self.n_obs = torch.empty((Bn,), dtype=torch.int32, device=dev)
        self.xg_next, self.xc_next = torch.empty_like(self.xg), torch.empty_like(self.xc)
        self.t = 0.0   # every vehicle's own clock, advanced by `+= timestep` like the reference's (base.py:889,941)
        self.u = None

    def step(self):
        """One control step of every race: three libcrx launches, nothing else."""
        N = self.N
        torch_api.cbf_prep_dev(N, self.lap_length, self.t, self.timestep, self.xc, self.s0, self.v, self.ey,
                               self.obs_s, self.obs_e, self.lap_off, self.n_obs)
        torch_api.cbf_solve_dev(self.desc, self.xc, self.xt, self.obs_s, self.obs_e, self.lap_off, self.n_obs, ws=self.ws,
                                order=_order(self, self.ws.iters))
        # the plant reads u_0 of every race straight out of the solver's U [B][N][2]
        torch_api.plant_step_wrap_dev(self.pdesc, self.tab, self.xg, self.xc, self.ws.U, 2 * N, self.xg_next, self.xc_next, self.laps,
                                      noise_z=self.noise.draw(self.batch))
        self.xg, self.xg_next = self.xg_next, self.xg
        self.xc, self.xc_next = self.xc_next, self.xc
        self.u = self.ws.U[:, 0, :]
        self.t += self.timestep

    def step_glue(self):
        """The same control step with the prediction / window filter / packing / lap wrap written as element-wise
        torch ops (the first version; kept as an independent restatement for the tests)."""
        N, L, xc = self.N, self.lap_length, self.xc
        tt = self.t + self.jdt                                                       # [N+1]
        obs_s = self.v[:, :, None] * tt[None, None, :] + self.s0[:, :, None]         # [B,V,N+1]
        obs_e = self.ey[:, :, None] + 0.0 * tt[None, None, :]
        margin = 2.0 * xc[:, 0:1]
        nce = _LAP_TRUNC(xc[:, 4:5] / L)
        dist_ego = xc[:, 4:5] - nce * L
        nco = _LAP_TRUNC(obs_s[:, :, 0] / L)
        dist_obs = obs_s[:, :, 0] - nco * L
        keep = (dist_ego > dist_obs - margin) & (dist_ego < dist_obs + margin)       # [B,V]
        lap_off = (nce - nco) * L
        order = torch.sort((~keep).to(torch.int8), dim=1, stable=True).indices
        n_obs = keep.sum(dim=1).to(torch.int32)
        live = (self.ar[None, :] < n_obs[:, None])
        ps = torch.gather(obs_s, 1, order[:, :, None].expand(-1, -1, N + 1)) * live[:, :, None]
        pe = torch.gather(obs_e, 1, order[:, :, None].expand(-1, -1, N + 1)) * live[:, :, None]
        po = torch.gather(lap_off, 1, order) * live
        torch_api.cbf_solve_dev(self.desc, xc, self.xt, ps.contiguous(), pe.contiguous(), po.contiguous(), n_obs.contiguous(),
                                ws=self.ws)
        self.u = self.ws.U[:, 0, :].contiguous()
        self.xg, xc = torch_api.plant_step_dev(self.pdesc, self.tab, self.xg, xc, self.u)
        crossed = xc[:, 4] > L
        xc[:, 4] = torch.where(crossed, xc[:, 4] - L, xc[:, 4])
        self.laps += crossed.to(torch.int32)
        self.xc = xc
        self.t += self.timestep


def mpccbf_races(track_table, lap_length, track_width, A, B, xcurv0, xglob0, car_s0, car_v, car_ey, steps,
                 vt=0.8, eyt=0.0, N=10, alpha=0.8, timestep=0.1, device=None, log_every=1, glue=False):
    """xcurv0, xglob0 [B,6]; car_s0, car_v, car_ey [B,V] with V <= 3: scripted cars s(t) = v t + s0, ey(t) = ey
    (NoDynamicsModel, utils/base.py:847-890).  Returns host arrays: xcurv [T+1,B,6] (T = steps/log_every), u [T,B,2],
    status [T,B], laps [B]."""
    r = MpccbfRaces(track_table, lap_length, track_width, A, B, xcurv0, xglob0, car_s0, car_v, car_ey, vt=vt, eyt=eyt,
                    N=N, alpha=alpha, timestep=timestep, device=device)
    log_x, log_u, log_st = [r.xc.clone()], [], []
    for k in range(steps):
        (r.step_glue if glue else r.step)()
        if (k + 1) % log_every == 0:
            log_x.append(r.xc.clone())
            log_u.append(r.u.clone())
            log_st.append(r.ws.status.clone())
    return dict(xcurv=torch.stack(log_x).cpu().numpy(), u=torch.stack(log_u).cpu().numpy(),
                status=torch.stack(log_st).cpu().numpy(), laps=r.laps.cpu().numpy())


class LmpcLaps:
    """B learning-MPC laps at once, device-resident (SURVEY.md section 8f rows 1 + 4): the lap of the reference's racing
    game in which LMPCRacingGame drives alone (tests/auto_racing_game_test.py:60-66 -> utils/base.py:468-517), with the
    race index as the batch dimension.  Per control step and without touching the host:

        crx_lmpc_prep_dev      N stage models from the two previous laps + safe-set selection   (estimate_ABC, control.lmpc :625-639)
        crx_lmpc_solve_dev     the learning-MPC QP                                               (control.lmpc :640-730)
        crx_lmpc_addpoint_dev  the applied (x, u) extends the previous lap's safe set            (add_point)
        crx_plant_step_wrap_dev  plant + lap bookkeeping                                         (forward_dynamics, update_memory)
        crx_lmpc_addtraj_dev   a race that crossed the line: its logged lap becomes a safe-set lap (add_trajectory)

    Five libcrx launches per step; torch owns the memory, copies u_old and appends the applied (x, u) to the race's lap log.
    Every race carries its own safe set (ss_xcurv [B, L, P, 6], u_ss [B, L, P, 2], qfun [B, L, P], time_ss [B, L]: the
    reference's arrays stored lap-major) and runs lap after lap: the lap hand-over (the reference's test script calls
    add_trajectory between laps, tests/auto_racing_game_test.py) happens per race on the device, until the race's L laps
    of safe-set storage are full (it then keeps racing on its last two laps)."""

    def __init__(self, track_table, lap_length, track_width, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input,
                 N=12, timestep=0.1, device=None, noise_seed=None):
        dev = torch.device(device if device is not None else "cuda")
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.noise = _Noise(noise_seed, dev)

        def t(a, kw):
            return torch.as_tensor(np.ascontiguousarray(a), **kw).clone()

        self.ss, self.us, self.qf = t(ss_xcurv, f64), t(u_ss, f64), t(qfun, f64)
        self.time_ss, self.it = t(time_ss, i32), t(it, i32)
        Bn, L, P, _ = self.ss.shape
        self.batch, self.N, self.lap_length, self.timestep = Bn, N, lap_length, timestep
        self.xc, self.xg = t(xcurv0, f64), t(xglob0, f64)
        self.lin_points, self.lin_input = t(lin_points, f64), t(lin_input, f64)
        self.tab = t(track_table, f64)
        self.pdesc = abi.lmpcprep_desc(N, P, L, self.tab.shape[0], timestep, lap_length)
        self.desc = abi.lmpc_desc(N=N, n_ss_max=self.pdesc.n_ss_per_lap * self.pdesc.n_ss_laps, ey_max=track_width)
        self.plant = abi.plant_desc(self.tab.shape[0], lap_length, timestep=timestep)
        self.pws = torch_api.LmpcPrepWorkspace(self.pdesc, Bn, dev)
        self.ws = torch_api.LmpcWorkspace(self.desc, Bn, dev)
        self.n_ss = torch.full((Bn,), self.desc.n_ss_max, **i32)
        self.u_old = torch.zeros((Bn, 2), **f64)
        self.u_prev = torch.zeros((Bn, 2), **f64)     # u_old of the last QP launch (bench.py re-issues that launch to time it)
        self.step_no = torch.zeros((Bn,), **i32)
        self.laps = torch.zeros((Bn,), **i32)
        self.xg_next, self.xc_next = torch.empty_like(self.xg), torch.empty_like(self.xc)
        self.k = 0
        # the running lap as the simulator logs it (update_memory): states incl. the crossing one, inputs
        self.log_x = torch.zeros((Bn, P, 6), **f64)
        self.log_u = torch.zeros((Bn, P, 2), **f64)
        self.log_x[:, 0] = self.xc
        self.n_log = torch.ones((Bn,), **i32)
        self.laps_prev = torch.zeros((Bn,), **i32)
        self.traj_status = torch.zeros((Bn,), **i32)
        self._ar = torch.arange(Bn, device=dev)
        self._P = P
        self.u = torch.zeros((Bn, 2), **f64)              # the input applied in the last step
        self.addpoint_step = torch.zeros((Bn,), **i32)
        self.crossed = torch.zeros((Bn,), **i32)

    def log_and_handover(self, u):
        """After the plant step: append (new state, applied input) to every race's lap log -- the crossing state with its s
        unwrapped, like update_memory -- (crx_game_log_dev) and hand the completed laps over to the safe set
        (crx_lmpc_addtraj_dev)."""
        torch_api.game_log_dev(self.lap_length, self.xc, u, self.laps, self.laps_prev, self.log_x, self.log_u, self.n_log, self.crossed)
        torch_api.lmpc_addtraj_dev(self.pdesc, self.crossed, self.log_x, self.log_u, self.n_log, self.ss, self.us, self.qf, self.time_ss, self.it,
                                   self.step_no, self.xc, self.traj_status)

    def log_and_handover_torch(self, u):
        """log_and_handover written as element-wise torch ops (round 2; kept as an independent restatement for the tests)."""
        crossed = (self.laps > self.laps_prev).to(torch.int32)
        self.laps_prev.copy_(self.laps)
        xu = self.xc.clone()
        xu[:, 4] += self.lap_length * crossed.to(torch.float64)       # (an int32 tensor times a Python float would be float32)
        n = self.n_log.long()
        self.log_x[self._ar, torch.clamp(n, max=self._P - 1)] = xu
        self.log_u[self._ar, torch.clamp(n - 1, min=0, max=self._P - 1)] = u
        self.n_log += 1
        torch_api.lmpc_addtraj_dev(self.pdesc, crossed, self.log_x, self.log_u, self.n_log, self.ss, self.us, self.qf, self.time_ss, self.it,
                                   self.step_no, self.xc, self.traj_status)

    def step(self, torch_glue=False):
        """One control step of every race: six libcrx launches (regression, QP, commit, add_point, plant, log) + the lap hand-over.
        torch_glue = True: the bookkeeping as element-wise torch ops instead of crx_game_commit_dev / crx_game_log_dev (round 2's
        formulation, kept for the tests: both must produce the same bits)."""
        N = self.N
        if self.k == 0:     # first call of the lap: linearisation points handed over by the previous controller (utils/base.py:651-653)
            torch_api.lmpc_prep_dev(self.pdesc, self.ss, self.us, self.qf, self.time_ss, self.it, self.xc, self.lin_points, self.lin_input,
                                    self.tab, False, ws=self.pws)
        else:               # afterwards: the previous plan, shifted by one stage inside the kernel (control.py:726-728)
            torch_api.lmpc_prep_dev(self.pdesc, self.ss, self.us, self.qf, self.time_ss, self.it, self.xc, self.ws.X, self.ws.U,
                                    self.tab, True, ws=self.pws)
        torch_api.lmpc_solve_dev(self.desc, self.xc, self.u_old, self.pws.A, self.pws.B, self.pws.C, self.pws.ss, self.pws.qfun, self.n_ss,
                                 ws=self.ws, order=_order(self, self.ws.iters))
        if torch_glue:
            torch_api.lmpc_addpoint_dev(self.pdesc, self.ss, self.us, self.time_ss, self.it, self.step_no, self.xc, self.ws.U, 2 * N)
            torch_api.plant_step_wrap_dev(self.plant, self.tab, self.xg, self.xc, self.ws.U, 2 * N, self.xg_next, self.xc_next, self.laps,
                                          noise_z=self.noise.draw(self.batch))
            self.xg, self.xg_next = self.xg_next, self.xg
            self.xc, self.xc_next = self.xc_next, self.xc
            self.u_prev.copy_(self.u_old)
            self.u_old.copy_(self.ws.U[:, 0, :])
            self.u.copy_(self.u_old)
            self.step_no += 1
            self.k += 1
            self.log_and_handover_torch(self.u_old)
            return
        # applied input, plan hand-over (u_old; lin_points / lin_input are the shifted plan), step counter: one launch
        torch_api.game_commit_dev(N, N, None, None, self.ws.X, self.ws.U, None, self.u, self.u_old, self.u_prev, self.lin_points, self.lin_input,
                                  self.step_no, self.addpoint_step, None)
        torch_api.lmpc_addpoint_dev(self.pdesc, self.ss, self.us, self.time_ss, self.it, self.addpoint_step, self.xc, self.u, 2)
        torch_api.plant_step_wrap_dev(self.plant, self.tab, self.xg, self.xc, self.u, 2, self.xg_next, self.xc_next, self.laps,
                                      noise_z=self.noise.draw(self.batch))
        self.xg, self.xg_next = self.xg_next, self.xg
        self.xc, self.xc_next = self.xc_next, self.xc
        self.k += 1
        self.log_and_handover(self.u)


def lmpc_laps(track_table, lap_length, track_width, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input, steps,
              N=12, timestep=0.1, device=None, noise_seed=None):
    """Run `steps` control steps of B learning-MPC laps; returns host logs xcurv [steps+1, B, 6], u [steps, B, 2], status [steps, B]
    (QP status), prep_status [steps, B] (singular regression: the stage keeps its previous / zero model rows), laps [B],
    traj_status [B] (lap hand-over: 1 = safe-set storage full)."""
    r = LmpcLaps(track_table, lap_length, track_width, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input, N=N,
                 timestep=timestep, device=device, noise_seed=noise_seed)
    log_x, log_u, log_st, log_ps = [r.xc.clone()], [], [], []
    for _ in range(steps):
        r.step()
        log_x.append(r.xc.clone())
        log_u.append(r.u_old.clone())
        log_st.append(r.ws.status.clone())
        log_ps.append(r.pws.status.clone())
    return dict(xcurv=torch.stack(log_x).cpu().numpy(), u=torch.stack(log_u).cpu().numpy(), status=torch.stack(log_st).cpu().numpy(),
                prep_status=torch.stack(log_ps).cpu().numpy(), laps=r.laps.cpu().numpy(),
                traj_status=r.traj_status.cpu().numpy())   # 1: the race's n_laps of safe-set storage were full at its last hand-over


class GameLaps:
    """B laps of the racing game WITH traffic at once, device-resident (the row's stated purpose:
    the closed-loop Monte-Carlo of the overtaking game, car_racing/tests/overtake_planner_test.py:307-309): per control step,
    LMPCRacingGame.calc_input (utils/base.py:456-583) for every race --

        scripted cars' states and predictions (NoDynamicsModel, utils/base.py:847-890; element-wise torch ops)
        crx_planner_scene_dev      vehicles of interest, partial sort, veh_infos                 (get_overtake_flag, get_local_traj :64-92)
      overtake branch (:518-582)
        crx_planner_prep_dev       Bezier references, ey bounds                                   (:94-117, 277-324)
        crx_planner_plan_dev       the region QPs + selection                                     (solve_optimization_problem)
        crx_track_prep_dev         per-stage targets, obstacle window / packing                   (control.mpc_multi_agents :293-382)
        crx_cbf_solve_dev          the tracking NLP with CBF rows                                 (:270-473)
      learning-MPC branch (:468-517)
        crx_lmpc_prep_dev, crx_lmpc_solve_dev, crx_lmpc_addpoint_dev
      crx_plant_step_wrap_dev

    Every launch covers the whole batch; the heavy kernels of a branch (tracking NLP; regression + learning-MPC QP) are
    MASKED launches (crx_*_masked_dev): the wavefront of a race that is in the other branch returns at once -- the region
    QPs of the planner included; the cheap kernels around them (scene, prep, selection, tracking prep) run for every race.  The applied input, the
    plan hand-over (u_old, linearisation points), add_point and the direction flag are taken from the branch the race is
    in; no host round trip, no compaction.  Nine libcrx launches per step."""

    def __init__(self, track_table, lap_length, track_width, A, B, opt_xcurv, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0,
                 lin_points, lin_input, car_s0, car_v, car_ey, N=12, N_plan=10, timestep=0.1, device=None, noise_seed=None):
        self.lm = LmpcLaps(track_table, lap_length, track_width, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input,
                           N=N, timestep=timestep, device=device, noise_seed=noise_seed)
        lm = self.lm
        dev = lm.xc.device
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        Bn = lm.batch
        self.s0, self.v, self.ey = (torch.as_tensor(np.ascontiguousarray(a), **f64) for a in (car_s0, car_v, car_ey))
        VA = self.s0.shape[1]
        V = min(VA, abi.CRX_MAX_VEH)
        self.Np, self.V, self.VA, self.L = N_plan, V, VA, lap_length
        self.scene = abi.scene_desc(N_plan, VA, V, lap_length)
        self.prep = abi.prep_desc(N_plan, V, opt_xcurv.shape[0], track_width, lap_length)
        self.plan, self.sel = abi.planner_desc(N_plan, A, B), abi.select_desc(N_plan, V, lap_length)
        self.track_desc = abi.cbf_desc(N_plan, V, A, B, Q=(10.0, 0, 0, 5.0, 0, 50.0), R=(0.1, 0.1), alpha=0.6, margin=0.15,
                                       ey_max=track_width, per_stage_target=True, l_sum=0.4, w_sum=0.2)
        self.opt_s = torch.as_tensor(np.ascontiguousarray(opt_xcurv[:, 4]), **f64)
        self.opt_ey = torch.as_tensor(np.ascontiguousarray(opt_xcurv[:, 5]), **f64)
        self.sws = torch_api.SceneWorkspace(self.scene, Bn, dev)
        self.pws = torch_api.PrepWorkspace(self.prep, Bn, dev)
        self.qws = torch_api.PlannerWorkspace(self.plan, Bn * (V + 1), dev)
        self.selws = torch_api.SelectWorkspace(self.sel, Bn, dev)
        self.tws = torch_api.CbfWorkspace(self.track_desc, Bn, dev)
        self.xt = torch.empty((Bn, N_plan + 1, 6), **f64)
        self.obs_s, self.obs_e = torch.empty((Bn, V, N_plan + 1), **f64), torch.empty((Bn, V, N_plan + 1), **f64)
        self.lap_off, self.n_obs = torch.empty((Bn, V), **f64), torch.empty((Bn,), **i32)
        self.n_all = torch.full((Bn,), VA, **i32)
        self.old_flag = torch.full((Bn,), -1, **i32)
        self.jdt = torch.arange(N_plan + 1, **f64) * timestep
        self.veh = torch.zeros((Bn, VA, 6), **f64)
        self.u = torch.zeros((Bn, 2), **f64)
        self.lin_points, self.lin_input = lm.lin_points.clone(), lm.lin_input.clone()
        self.neg = torch.full((Bn,), -(1 << 20), **i32)
        self.overtake = torch.zeros((Bn,), dtype=torch.bool, device=dev)
        self.m_ot, self.m_lm = torch.zeros((Bn,), **i32), torch.zeros((Bn,), **i32)
        self.pred_s, self.pred_e = torch.zeros((Bn, VA, N_plan + 1), **f64), torch.zeros((Bn, VA, N_plan + 1), **f64)
        self.overflow_seen = torch.zeros((Bn,), **i32)    # scene overflow (more vehicles of interest than slots), accumulated
        # the two branches of a step are independent until the commit: their kernels go to two HIP streams and overlap (the
        # masked launches of a branch leave part of the chip idle: a third of the races plan and track, the rest regress and solve)
        self.s_ot = self.s_lm = None   # set_branch_streams, or two of libcrx's at the first step
        self.overlap = True            # False: both branches on the caller's stream, one after the other
        self.t = 0.0

    def step(self):
        """One control step of every race: thirteen libcrx launches (traffic, scene, masks, prep, plan = QPs + selection, track
        prep, tracking NLP, regression, learning-MPC QP, commit, add_point, plant, log) + the lap hand-over."""
        lm, L, N = self.lm, self.L, self.lm.N
        torch_api.game_traffic_dev(self.Np, L, self.t, lm.timestep, self.s0, self.v, self.ey, self.veh, self.pred_s, self.pred_e)
        torch_api.planner_scene_dev(self.scene, lm.xc, self.n_all, self.veh, self.pred_s, self.pred_e, ws=self.sws)
        torch_api.game_masks_dev(self.sws.n_veh, self.m_ot, self.m_lm, overflow=self.sws.overflow, overflow_seen=self.overflow_seen)
        self.overtake = self.m_ot                                  # int32 mask; bool(overtake[b]) = race b is in the overtake branch
        self._branches(self.m_ot, self.m_lm, overlap=self.overlap)
        torch_api.game_commit_dev(N, self.Np, self.m_ot, self.tws.U, lm.ws.X, lm.ws.U, self.selws.flag, self.u, lm.u_old, lm.u_prev,
                                  self.lin_points, self.lin_input, lm.step_no, lm.addpoint_step, self.old_flag)
        torch_api.lmpc_addpoint_dev(lm.pdesc, lm.ss, lm.us, lm.time_ss, lm.it, lm.addpoint_step, lm.xc, self.u, 2)
        torch_api.plant_step_wrap_dev(lm.plant, lm.tab, lm.xg, lm.xc, self.u, 2, lm.xg_next, lm.xc_next, lm.laps, noise_z=lm.noise.draw(lm.batch))
        lm.xg, lm.xg_next = lm.xg_next, lm.xg
        lm.xc, lm.xc_next = lm.xc_next, lm.xc
        self.t += lm.timestep
        lm.log_and_handover(self.u)

    def set_branch_streams(self, s_ot, s_lm):
        """The two streams the branches of a step run on (crx.montecarlo.Concurrent hands out streams it measured to overlap)."""
        self.s_ot, self.s_lm = s_ot, s_lm

    def _branches(self, m_ot, m_lm, overlap=False):
        """The solver launches of both branches (masked: every race runs its own branch's kernels only); overlap: each branch
        on its own stream, forked from and joined to the current one."""
        if overlap and self.s_ot is None:
            (self.s_ot, self.s_lm), nc = torch_api.new_streams(2, self.lm.xc.device)
            if nc < 2:
                overlap = self.overlap = False
        if overlap:
            cur = torch.cuda.current_stream(self.lm.xc.device)
            self.s_ot.wait_stream(cur); self.s_lm.wait_stream(cur)
            with torch.cuda.stream(self.s_ot):
                self._branch_overtake(m_ot)
            with torch.cuda.stream(self.s_lm):
                self._branch_lmpc(m_lm)
            cur.wait_stream(self.s_ot); cur.wait_stream(self.s_lm)
        else:
            self._branch_overtake(m_ot)
            self._branch_lmpc(m_lm)

    def _branch_overtake(self, m_ot):
        lm, L = self.lm, self.L
        torch_api.planner_prep_dev(self.prep, lm.xc, lm.xc, self.sws.n_veh, self.sws.veh_info, self.sws.max_dv, self.sws.obs_s, self.sws.obs_ey,
                                   self.opt_s, self.opt_ey, ws=self.pws)
        torch_api.planner_plan_dev(self.plan, self.sel, self.pws.x0, self.pws.bez_s, self.pws.bez_ey, self.pws.ey_lb, self.pws.ey_ub,
                                   self.sws.n_veh, self.sws.obs_s, self.sws.obs_ey, self.old_flag, self.qws, self.selws, active=m_ot)
        torch_api.track_prep_dev(self.Np, self.V, L, lm.xc, self.sws.n_veh, self.sws.obs_s, self.sws.obs_ey, self.selws.best_X, self.xt,
                                 self.obs_s, self.obs_e, self.lap_off, self.n_obs)
        torch_api.cbf_solve_dev(self.track_desc, lm.xc, self.xt, self.obs_s, self.obs_e, self.lap_off, self.n_obs, ws=self.tws, active=m_ot,
                                order=_order(self, self.tws.iters, m_ot))

    def _branch_lmpc(self, m_lm):
        lm = self.lm
        torch_api.lmpc_prep_dev(lm.pdesc, lm.ss, lm.us, lm.qf, lm.time_ss, lm.it, lm.xc, self.lin_points, self.lin_input, lm.tab, False, ws=lm.pws,
                                active=m_lm)
        torch_api.lmpc_solve_dev(lm.desc, lm.xc, lm.u_old, lm.pws.A, lm.pws.B, lm.pws.C, lm.pws.ss, lm.pws.qfun, lm.n_ss, ws=lm.ws, active=m_lm,
                                 order=_order(self, lm.ws.iters, m_lm))

    def step_torch(self):
        """The same control step with the bookkeeping written as element-wise torch ops (round 2's formulation, ~40 small launches;
        kept as an independent restatement for the tests: step() must produce the same bits)."""
        lm, L, N = self.lm, self.L, self.lm.N
        # scripted cars at their own clock t: s = v t + s0 (wrapped once past the line, update_memory), predictions unwrapped (quirk Q6)
        s_now = self.v * self.t + self.s0
        self.veh[:, :, 0] = self.v
        self.veh[:, :, 4] = s_now - L * torch.clamp(torch.ceil(s_now / L) - 1.0, min=0.0)   # update_memory: wrapped whenever s > L, lap after lap
        self.veh[:, :, 5] = self.ey
        pred_s = (self.v[:, :, None] * (self.t + self.jdt)[None, None, :] + self.s0[:, :, None]).contiguous()
        pred_e = (self.ey[:, :, None] + 0.0 * self.jdt[None, None, :]).contiguous()
        torch_api.planner_scene_dev(self.scene, lm.xc, self.n_all, self.veh, pred_s, pred_e, ws=self.sws)
        self.overtake = self.sws.n_veh > 0
        m_ot = self.overtake.to(torch.int32)                      # masked launches: every race runs its own branch's kernels only
        m_lm = 1 - m_ot
        self._branches(m_ot, m_lm)
        # ---- the branch each race is in
        ot = self.overtake
        self.u.copy_(torch.where(ot[:, None], self.tws.U[:, 0, :], lm.ws.U[:, 0, :]))
        torch_api.lmpc_addpoint_dev(lm.pdesc, lm.ss, lm.us, lm.time_ss, lm.it, torch.where(ot, self.neg, lm.step_no), lm.xc, self.u, 2)
        lm.u_prev.copy_(lm.u_old)
        lm.u_old.copy_(torch.where(ot[:, None], lm.u_old, lm.ws.U[:, 0, :]))
        X, U = lm.ws.X, lm.ws.U
        self.lin_points.copy_(torch.where(ot[:, None, None], self.lin_points, torch.cat((X[:, 1:], X[:, -1:]), dim=1)))
        self.lin_input.copy_(torch.where(ot[:, None, None], self.lin_input, torch.cat((U[:, 1:], U[:, -1:]), dim=1)))
        lm.step_no += (~ot).to(torch.int32)
        self.old_flag.copy_(torch.where(ot, self.selws.flag, torch.full_like(self.old_flag, -1)))
        torch_api.plant_step_wrap_dev(lm.plant, lm.tab, lm.xg, lm.xc, self.u, 2, lm.xg_next, lm.xc_next, lm.laps, noise_z=lm.noise.draw(lm.batch))
        lm.xg, lm.xg_next = lm.xg_next, lm.xg
        lm.xc, lm.xc_next = lm.xc_next, lm.xc
        self.t += lm.timestep
        lm.log_and_handover_torch(self.u)


def game_laps(track_table, lap_length, track_width, A, B, opt_xcurv, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input,
              car_s0, car_v, car_ey, steps, device=None, noise_seed=None):
    """Run `steps` control steps of B racing-game laps with scripted traffic.  Host logs: xcurv [steps+1, B, 6], u [steps, B, 2],
    overtake [steps, B] (which branch), flag [steps, B] (direction flag, -1 in the LMPC branch), cars_s [steps, B, V], laps [B]."""
    r = GameLaps(track_table, lap_length, track_width, A, B, opt_xcurv, ss_xcurv, u_ss, qfun, time_ss, it, xcurv0, xglob0, lin_points, lin_input,
                 car_s0, car_v, car_ey, device=device, noise_seed=noise_seed)
    lx, lu, lo, lf, lc = [r.lm.xc.clone()], [], [], [], []
    for _ in range(steps):
        lc.append((r.v * r.t + r.s0).clone())
        r.step()
        lx.append(r.lm.xc.clone()); lu.append(r.u.clone()); lo.append(r.overtake.clone().bool()); lf.append(r.old_flag.clone())
    return dict(xcurv=torch.stack(lx).cpu().numpy(), u=torch.stack(lu).cpu().numpy(), overtake=torch.stack(lo).cpu().numpy(),
                flag=torch.stack(lf).cpu().numpy(), cars_s=torch.stack(lc).cpu().numpy(), laps=r.lm.laps.cpu().numpy(),
                traj_status=r.lm.traj_status.cpu().numpy(), scene_overflow=r.overflow_seen.cpu().numpy())
