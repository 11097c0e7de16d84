"""The N>1 path on CPU: two gloo ranks shard a scenario batch contiguously, produce winner records
and all-gather them; the result must equal the single-process answer.  (The solve itself needs a
GPU; here each rank's 'winners' are a deterministic function of the global scenario index, which is
exactly what makes ordering / padding / trimming mistakes visible.)"""
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


def _winners(lo, hi, N):
    idx = torch.arange(lo, hi, dtype=torch.float64)
    flag = (torch.arange(lo, hi) % 4).to(torch.int32)
    X = idx[:, None, None] + torch.arange((N + 1) * 6, dtype=torch.float64).reshape(1, N + 1, 6) / 1000.0
    return flag, X


def _worker(rank, world, port, n_total, N, out_dir):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crx import dist as cd

    lo, hi = cd.shard_bounds(n_total, rank, world)
    flag, X = _winners(lo, hi, N)
    fa, Xa = cd.allgather_winners(flag, X, n_total)
    torch.save((fa, Xa), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7])  # even and ragged shards
def test_two_rank_shard_and_allgather(tmp_path, n_total):
    N, world = 12, 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, N, str(tmp_path)), nprocs=world, join=True)
    f_ref, X_ref = _winners(0, n_total, N)
    for r in range(world):
        fa, Xa = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        assert torch.equal(fa, f_ref)
        assert torch.equal(Xa, X_ref)


def test_shard_bounds_cover_everything():
    from crx import dist as cd

    for n in (0, 1, 5, 8, 131072):
        for world in (1, 2, 3, 8):
            spans = [cd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
