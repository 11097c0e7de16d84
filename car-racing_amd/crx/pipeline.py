"""The overtake planner's hot path for a whole shard of scenarios, device-resident end to end
(BASELINE.json configs[4]):

    raw scenarios --crx_planner_prep_dev--> region QP arrays --crx_planner_solve_dev--> X of every region
                  --crx_select_dev--> winners --ONE all-gather (RCCL over xGMI)--> winners of every rank

i.e. OvertakeTrajPlanner.get_local_traj -> solve_optimization_problem (planning/overtake_traj_planner.py:44-246)
for n_local scenarios per rank.  Four launches + one collective per step; nothing returns to the host.

The solver back-end is an argument (default: crx.torch_api = libcrx) so that the multi-rank control flow --
sharding, buffers, the collective, trimming -- runs unchanged in the world_size-2 gloo tests with a stand-in
back-end on CPU tensors.  There is no CPU solver in the product: without libcrx the default back-end raises."""
import numpy as np
import torch

from . import abi
from . import dist as cdist


class PlannerSweep:
    def __init__(self, raw, A, B, n_total, device, backend=None, lo=0):
        """raw: dict as crx.synth.cfg3_raw for THIS rank's scenarios [lo, lo + n_local) of n_total."""
        if backend is None:
            from . import torch_api as backend
        self.be = backend
        N, V = int(raw["N"]), int(raw["V"])
        self.N, self.V, self.n_local, self.n_total = N, V, int(raw["x"].shape[0]), int(n_total)
        self.desc = abi.planner_desc(N, A, B)
        self.sdesc = abi.select_desc(N, V, float(raw["lap_length"]))
        self.pdesc = abi.prep_desc(N, V, len(raw["opt_s"]), float(raw["track_width"]), float(raw["lap_length"]))

        def dev(a, dtype=torch.float64):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)

        self.x = dev(raw["x"])
        self.n_veh = dev(raw["n_veh"], torch.int32)
        self.veh_info, self.max_dv = dev(raw["veh_info"]), dev(raw["max_dv"])
        self.obs_s, self.obs_ey = dev(raw["obs_s"]), dev(raw["obs_ey"])
        self.opt_s, self.opt_ey = dev(raw["opt_s"]), dev(raw["opt_ey"])
        self.old_flag = dev(raw["old_flag"], torch.int32)
        S, R = self.n_local, V + 1
        self.pws = backend.PrepWorkspace(self.pdesc, S, device)
        self.ws = backend.PlannerWorkspace(self.desc, S * R, device)
        self.sws = backend.SelectWorkspace(self.sdesc, S, device)
        self.exchange = cdist.WinnerExchange(n_total, N, device)
        self.flag_all = self.best_all = self.status_all = None

    def solve_local(self):
        """prep -> region QPs -> selection for this rank's scenarios (no communication)."""
        be, S, R, N = self.be, self.n_local, self.V + 1, self.N
        be.planner_prep_dev(self.pdesc, self.x, self.x, self.n_veh, self.veh_info, self.max_dv, self.obs_s, self.obs_ey,
                            self.opt_s, self.opt_ey, ws=self.pws)
        be.planner_solve_dev(self.desc, self.pws.x0, self.pws.bez_s, self.pws.bez_ey, self.pws.ey_lb, self.pws.ey_ub, ws=self.ws)
        be.select_dev(self.sdesc, self.n_veh, self.ws.X.view(S, R, N + 1, 6), self.obs_s, self.obs_ey, self.old_flag, ws=self.sws)

    def gather(self):
        # status of the WINNING region's QP (SURVEY 8e's record carries it: a fall-back winner must be recognisable on every rank)
        S, R = self.n_local, self.V + 1
        st = self.ws.status.view(S, R).gather(1, self.sws.flag.long().clamp_(0, R - 1).unsqueeze(1)).squeeze(1)
        self.flag_all, self.best_all, self.status_all = self.exchange(self.sws.flag, self.sws.best_X, st)

    def step(self):
        self.solve_local()
        self.gather()
        return self.flag_all, self.best_all, self.status_all
