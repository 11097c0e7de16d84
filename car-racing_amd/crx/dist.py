"""Multi-GPU sharding of a scenario batch: one process per GPU, contiguous scenario blocks, and the
path's single exchange step -- ONE all-gather of the fixed-size winner records.

Backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU tests)."""
import torch
import torch.distributed as dist


# bench.py --force-collective: issue the all-gather through the backend even with one rank (RCCL plumbing check on a 1-GPU box)
FORCE_COLLECTIVE = False
# "torch": torch.distributed's all_gather_into_tensor (backend nccl = RCCL on the GPU box, gloo in the CPU tests);
# "crx":   crx_allgather_winners_dev, the same exchange issued on RCCL by libcrx itself (include/crx.h) -- what a C caller
#          of the library uses; torch.distributed then only carries the 128-byte communicator id to the ranks
COLLECTIVE = "torch"


class CrxComm:
    """libcrx's own RCCL communicator (crx_comm_*), one per process.  The rendezvous is the caller's business: here the id
    travels through torch.distributed's object broadcast when there is a process group, and nowhere when world == 1."""
    _ready = False

    @classmethod
    def ensure(cls):
        import ctypes as C

        from . import binding, lib
        if cls._ready:
            return
        binding()
        L = lib()
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        buf = C.create_string_buffer(128)
        if rank == 0 and L.crx_comm_get_unique_id(buf) != 0:
            raise RuntimeError("crx_comm_get_unique_id: " + (L.crx_last_error() or b"").decode())
        ids = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        if L.crx_comm_init_rank(C.create_string_buffer(ids[0], 128), C.c_int(world), C.c_int(rank)) != 0:
            raise RuntimeError("crx_comm_init_rank: " + (L.crx_last_error() or b"").decode())
        cls._ready = True

    @classmethod
    def destroy(cls):
        from . import lib
        if cls._ready:
            lib().crx_comm_destroy()
            cls._ready = False


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

    A winner record is SURVEY.md section 8e's {int32 flag; int32 status; double X[N+1][6]} -- 632 B at N = 12 -- carried as
    1 + 6 (N + 1) 8-byte words (word 0 = the two int32, bit for bit; the collective only moves bytes).  `status` is the crx_status of
    the WINNING region's QP: a consumer on another rank can tell a fall-back winner (overtake_traj_planner.py:365-374) from a solved
    one.  Every rank knows the shard sizes from (n_total, world) alone -- shard_bounds is a pure function -- so no size exchange is
    needed: ragged shards are padded to the largest one and trimmed after the ONE collective.  Returns (flag, X, status)."""

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

    def _unpack(self, allrec):
        head = allrec[:, :1].contiguous().view(torch.int32)          # [n, 2]: flag, status
        return head[:, 0], allrec[:, 1:].reshape(-1, *self.shape_X), head[:, 1]

    def __call__(self, flag, best_X, status=None):
        n = self.sizes[self.rank]
        if flag.shape[0] != n or best_X.shape[0] != n:
            raise ValueError("rank %d holds %d winners, its shard of %d over %d ranks is %d" % (self.rank, flag.shape[0], self.n_total, self.world, n))
        if COLLECTIVE == "crx" and (self.world > 1 or FORCE_COLLECTIVE):
            # pack + ncclAllGather inside libcrx, on torch's current stream (the winners never pass through torch ops)
            if self.group is not None:
                # libcrx holds ONE communicator and it spans the world group: a sub-group exchange would deadlock or gather the wrong ranks
                raise ValueError("COLLECTIVE == 'crx' gathers over the world group only; pass group=None or use COLLECTIVE = 'torch' for a sub-group")
            import ctypes as C

            from . import lib
            from .torch_api import _ptr, _stream
            CrxComm.ensure()
            fl = flag.to(torch.int32).contiguous()
            bx = best_X.contiguous()
            stt = status.to(torch.int32).contiguous() if status is not None else None
            if lib().crx_allgather_winners_dev(C.c_int(n), C.c_int(self.n_max), C.c_int(self.shape_X[0] - 1), _ptr(fl), _ptr(stt) if stt is not None else None,
                                               _ptr(bx), _ptr(self.send), _ptr(self.recv), _stream()) != 0:
                raise RuntimeError("crx_allgather_winners_dev: " + (lib().crx_last_error() or b"").decode())
            if self.even:
                allrec = self.recv
            else:
                out = self.recv.view(self.world, self.n_max, self.rec)
                allrec = torch.cat([out[r, : self.sizes[r]] for r in range(self.world)], dim=0)
            return self._unpack(allrec)
        head = self.send.view(torch.int32)                            # [n_max, 2 * rec]: word 0 of a record = int32 columns 0 and 1
        head[:n, 0] = flag
        head[:n, 1] = status if status is not None else 0
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
        return self._unpack(allrec)


def allgather_winners(flag, best_X, n_total=None, status=None):
    """Gather per-scenario winners from every rank in rank order: flag [n_local] int32, best_X [n_local, N+1, 6] float64, status
    [n_local] int32 (crx_status of the winning region's QP; None: zeros) -> (flag_all [n_total], best_X_all [n_total, N+1, 6],
    status_all [n_total]).  One collective.  `n_total` = scenarios over all ranks (sharded by shard_bounds); None means every rank
    holds the same number."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flag, best_X, (status if status is not None else torch.zeros_like(flag))
    if n_total is None:
        n_total = flag.shape[0] * dist.get_world_size()
    return WinnerExchange(n_total, best_X.shape[1] - 1, flag.device)(flag, best_X, status)
