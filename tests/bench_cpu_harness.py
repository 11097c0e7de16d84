"""CPU harness around bench.py for tests/test_bench_plumbing.py: bench.py's multi-rank control flow -- self-spawn, rendezvous,
sharding, barriers, max-over-ranks timing, the collective, the full-record file and the ONE short JSON line -- on CPU tensors over gloo,
with stand-ins for everything that needs a GPU.  Nothing is measured; the point is that a typo in that path cannot surface first on
the driver's 8-GPU box.  The stand-ins live HERE (test scaffolding), not in the measurement tool: this file replaces bench.py's
device / back-end / kernel-timer hooks and then runs bench.main() unchanged."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))
for p in (HERE, ROOT, os.path.join(ROOT, "car-racing_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


class CpuCtx(bench.Ctx):
    def device(self):
        return torch.device("cpu")

    def dsync(self):
        pass


def init_backend(cx, args):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if cx.world > 1:
        dist.init_process_group("gloo", rank=cx.rank, world_size=cx.world)


def sweep_backend():
    """the stand-in solver back-end of tests/test_distributed_gloo.py; status / iteration outputs zeroed so that measure() can read them"""
    from test_distributed_gloo import StubBackend

    be = StubBackend()
    inner = be.PlannerWorkspace

    def zeroed(desc, batch, device):
        ws = inner(desc, batch, device)
        for k in ("status", "iters", "kkt", "cost", "X", "U"):
            getattr(ws, k).zero_()
        return ws

    be.PlannerWorkspace = zeroed
    return be


def headline(cx, args, make):
    """a stand-in for the headline workload (no solver on CPU): same measure() path, a trivial step"""
    from crx import abi, synth

    w = bench.Workload()
    w.key, w.kind, w.N, w.n_obs, w.batch, w.units, w.kernel, w.baseline_config = "stand_in", "cbf", 12, 1, 256, 256, "none", 1
    w.name = "CPU harness: no solver ran"
    A, B = synth.load_AB()
    w.desc = abi.cbf_desc(12, 1, A, B)
    w.ws = type("WS", (), dict(status=torch.zeros(256, dtype=torch.int32), iters=torch.zeros(256, dtype=torch.int32), kkt=torch.zeros(256, dtype=torch.float64)))()
    acc = torch.zeros(1)
    w.step = w.solve = lambda: acc.add_(1.0)
    return w


bench.Ctx = CpuCtx
bench.init_backend = init_backend
bench.shutdown_backend = lambda cx, args: None
bench.SWEEP_BACKEND = sweep_backend
bench.headline_workload = headline
bench.kernel_ms_samples = lambda cx, w, reps: (float("nan"), float("nan"))    # no kernel, no time
bench.occupancy = lambda w: (1, 0)
bench.sub_configs = lambda cx, args: [("cfg5_weak", lambda: bench.make_sweep(cx, args, "weak"), 2, 1, False),
                                      ("cfg5_strong", lambda: bench.make_sweep(cx, args, "strong"), 2, 1, False)]

if __name__ == "__main__":
    bench.main()
