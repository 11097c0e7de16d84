"""Multi-GPU sharding of a scenario batch: one process per GPU, contiguous scenario blocks, and the
path's single exchange step -- an all-gather of the fixed-size winner records.

Backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; all V+1 regions of a scenario stay on one rank, so
    the region arg-min (overtake_traj_planner.py:244) needs no communication."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_winners(flag, best_X, n_total=None):
    """Gather per-scenario winners from every rank in rank order.

    flag [n_local] int32, best_X [n_local, N+1, 6] float64 (same device).  Ragged shards are padded
    to the largest shard for the collective and trimmed afterwards.  Returns (flag_all, best_X_all)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flag, best_X
    world = dist.get_world_size()
    n_local = torch.tensor([flag.shape[0]], dtype=torch.int64, device=flag.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    rec = best_X.shape[1] * best_X.shape[2]
    # one fixed-size record per scenario: [flag, X...] as float64
    buf = torch.zeros((n_max, 1 + rec), dtype=torch.float64, device=flag.device)
    buf[: flag.shape[0], 0] = flag.to(torch.float64)
    buf[: flag.shape[0], 1:] = best_X.reshape(flag.shape[0], rec)
    out = torch.empty((world * n_max, 1 + rec), dtype=torch.float64, device=flag.device)
    dist.all_gather_into_tensor(out, buf)  # ONE collective: concatenation along dim 0, rank order
    out = out.view(world, n_max, 1 + rec)
    parts = [out[r, : sizes[r]] for r in range(world)]
    allrec = torch.cat(parts, dim=0)
    if n_total is not None:
        assert allrec.shape[0] == n_total
    return allrec[:, 0].to(torch.int32), allrec[:, 1:].reshape(-1, best_X.shape[1], best_X.shape[2])
