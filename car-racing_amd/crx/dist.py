"""Multi-GPU sharding of a scenario batch: one process per GPU, contiguous scenario blocks, and the
path's single exchange step -- ONE all-gather of the fixed-size winner records.

Backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU tests)."""
import torch
import torch.distributed as dist


# bench.py --force-collective: issue the all-gather through the backend even with one rank (RCCL plumbing check on a 1-GPU box)
FORCE_COLLECTIVE = False


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; all V+1 regions of a scenario stay on one rank, so
    the region arg-min (overtake_traj_planner.py:244) needs no communication."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_total, world):
    return [hi - lo for lo, hi in (shard_bounds(n_total, r, world) for r in range(world))]


class WinnerExchange:
    """The all-gather of the winners, with its buffers allocated once.

    A winner record is [flag, X (N+1)*6] as float64 (flags are small integers: exact).  Every rank knows the shard
    sizes from (n_total, world) alone -- shard_bounds is a pure function -- so no size exchange is needed: ragged
    shards are padded to the largest one and trimmed after the ONE collective."""

    def __init__(self, n_total, N, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.sizes = shard_sizes(n_total, self.world)
        self.n_total, self.n_max, self.rec = int(n_total), max(self.sizes), 1 + (N + 1) * 6
        self.shape_X = (N + 1, 6)
        self.even = min(self.sizes) == self.n_max
        self.send = torch.zeros((self.n_max, self.rec), dtype=torch.float64, device=device)
        self.recv = torch.empty((self.world * self.n_max, self.rec), dtype=torch.float64, device=device)

    def __call__(self, flag, best_X):
        n = self.sizes[self.rank]
        if flag.shape[0] != n or best_X.shape[0] != n:
            raise ValueError("rank %d holds %d winners, its shard of %d over %d ranks is %d" % (self.rank, flag.shape[0], self.n_total, self.world, n))
        self.send[:n, 0] = flag
        self.send[:n, 1:] = best_X.reshape(n, self.rec - 1)
        if self.world == 1 and not (dist.is_initialized() and FORCE_COLLECTIVE):
            allrec = self.send[:n]
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)   # the ONE collective of the path
            if self.even:
                allrec = self.recv
            else:
                out = self.recv.view(self.world, self.n_max, self.rec)
                allrec = torch.cat([out[r, : self.sizes[r]] for r in range(self.world)], dim=0)
        return allrec[:, 0].to(torch.int32), allrec[:, 1:].reshape(-1, *self.shape_X)


def allgather_winners(flag, best_X, n_total=None):
    """Gather per-scenario winners from every rank in rank order: flag [n_local] int32, best_X [n_local, N+1, 6] float64
    -> (flag_all [n_total], best_X_all [n_total, N+1, 6]).  One collective.  `n_total` = scenarios over all ranks
    (sharded by shard_bounds); None means every rank holds the same number."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flag, best_X
    if n_total is None:
        n_total = flag.shape[0] * dist.get_world_size()
    return WinnerExchange(n_total, best_X.shape[1] - 1, flag.device)(flag, best_X)
