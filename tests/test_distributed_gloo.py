"""The N>1 path on CPU: two gloo ranks shard a scenario batch contiguously, run the WHOLE per-step pipeline that
bench.py's cfg5 workload runs (crx.pipeline.PlannerSweep: prep -> region QPs -> selection -> ONE all-gather of the
winners) and must arrive at the single-process answer on every rank.

The solve itself needs a GPU; here the solver back-end is a stand-in on CPU tensors whose 'winners' are a
deterministic function of each scenario's own data -- exactly what makes sharding / ordering / padding / trimming
mistakes visible.  Everything else (sharding, buffers, the collective, trimming) is the product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StubBackend:
    """Same call surface as crx.torch_api (workspaces included); arithmetic replaced by a per-scenario function."""

    def __init__(self):
        from crx import torch_api

        self.PrepWorkspace, self.PlannerWorkspace, self.SelectWorkspace = (
            torch_api.PrepWorkspace, torch_api.PlannerWorkspace, torch_api.SelectWorkspace)
        self.calls = []

    def planner_prep_dev(self, desc, xw, xr, n_veh, veh_info, max_dv, obs_s, obs_ey, opt_s, opt_ey, ws=None):
        R = desc.n_veh_max + 1
        ws.x0.copy_(xr.repeat_interleave(R, dim=0))
        ws.bez_s.copy_((veh_info[:, :, 0].sum(1)).repeat_interleave(R)[:, None] + torch.arange(desc.N + 1, dtype=torch.float64)[None])
        self.calls.append("prep")

    def planner_solve_dev(self, desc, x0, bez_s, bez_ey, ey_lb, ey_ub, ws=None):
        B = x0.shape[0]
        reg = (torch.arange(B) % 4).to(torch.float64)
        ws.X.copy_(x0[:, None, :] + bez_s[:, :, None] / 64.0 + reg[:, None, None])
        # a per-problem "solver status" (0 converged / 2 proved infeasible = fall-back winner) that depends on the scenario's own data
        ws.status.copy_((torch.floor(bez_s[:, 0] * 7.0).to(torch.int64) % 3 == 0).to(torch.int32) * 2)
        self.calls.append("solve")

    def select_dev(self, desc, n_veh, X, obs_s, obs_ey, old_flag, ws=None):
        S = X.shape[0]
        flag = (torch.floor(X[:, 0, 0, 4] * 1000.0).to(torch.int64) % (desc.n_veh_max + 1)).to(torch.int32)
        ws.flag.copy_(flag)
        ws.best_X.copy_(X[torch.arange(S), flag.long()])
        self.calls.append("select")


def _sweep(raw_all, A, B, rank, world):
    from crx import dist as cd
    from crx import pipeline

    n_total = raw_all["x"].shape[0]
    lo, hi = cd.shard_bounds(n_total, rank, world)
    raw = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == n_total else v) for k, v in raw_all.items()}
    be = StubBackend()
    sw = pipeline.PlannerSweep(raw, A, B, n_total, torch.device("cpu"), backend=be, lo=lo)
    out = sw.step()
    assert be.calls == ["prep", "solve", "select"]
    return out


def _worker(rank, world, port, n_total, out_dir):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crx import synth

    A, B = synth.load_AB()
    fa, Xa, sa = _sweep(synth.cfg3_raw(n_total, N=12, seed=5), A, B, rank, world)
    torch.save((fa.clone(), Xa.clone(), sa.clone()), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7])  # even and ragged shards
def test_two_rank_planner_sweep(tmp_path, n_total):
    from crx import synth

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    A, B = synth.load_AB()
    f_ref, X_ref, s_ref = _sweep(synth.cfg3_raw(n_total, N=12, seed=5), A, B, 0, 1)   # single process, no process group
    assert f_ref.shape == (n_total,) and X_ref.shape == (n_total, 13, 6) and s_ref.shape == (n_total,)
    assert s_ref.dtype == torch.int32 and set(s_ref.tolist()) <= {0, 2} and len(set(s_ref.tolist())) == 2   # both verdicts occur among the winners
    for r in range(world):
        fa, Xa, sa = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        assert torch.equal(fa, f_ref)
        assert torch.equal(Xa, X_ref)
        assert torch.equal(sa, s_ref)            # the record carries the winner's status


def test_winner_record_is_survey_8e_layout():
    """{int32 flag; int32 status; double X[N+1][6]}: 632 B at N = 12, word 0 = the two int32 bit for bit."""
    from crx import dist as cd

    ex = cd.WinnerExchange(5, 12, torch.device("cpu"))
    assert ex.rec * 8 == 632
    flag = torch.tensor([3, 0, 1, 2, 3], dtype=torch.int32)
    st = torch.tensor([0, 2, 0, 5, 3], dtype=torch.int32)
    X = torch.randn((5, 13, 6), dtype=torch.float64)
    f2, X2, s2 = ex(flag, X, st)
    assert torch.equal(f2, flag) and torch.equal(X2, X) and torch.equal(s2, st)
    raw = ex.send.numpy().view(np.int32).reshape(5, -1)
    assert np.array_equal(raw[:, 0], flag.numpy()) and np.array_equal(raw[:, 1], st.numpy())
    f3, _, s3 = ex(flag, X)                      # no status given: the field is 0
    assert torch.equal(f3, flag) and not s3.any()


def test_shard_bounds_cover_everything():
    from crx import dist as cd

    for n in (0, 1, 5, 8, 131072):
        for world in (1, 2, 3, 8):
            spans = [cd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == cd.shard_sizes(n, world)
