"""bench.py's multi-rank control flow on CPU: `bench.py --gpus 2 --plumbing-check` re-launches itself under
torch.distributed.run exactly as it does on a GPU node (self-spawn, 127.0.0.1 rendezvous, one process per rank), shards the
cfg5 sweep weak and strong, runs the product's PlannerSweep.step() with the stand-in back-end of test_distributed_gloo.py over
gloo, goes through measure()'s barriers / max-over-ranks reductions / collective timing, and rank 0 prints ONE JSON line with
the fields the driver parses.  Nothing is measured; the point is that a typo in that path cannot surface first on the
driver's 8-GPU box."""
import json
import os
import subprocess
import sys

import conftest


def _run(gpus, extra=()):
    env = dict(os.environ)
    env["CRX_BENCH_BACKEND"] = "test_distributed_gloo:StubBackend"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(conftest.ROOT, "tests"), conftest.ROOT, conftest.PKG, env.get("PYTHONPATH", "")])
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--plumbing-check",
           "--sweep-per-gpu", "6", "--sweep-total", "13", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_self_spawn():
    out = _run(2)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["higher_is_better"] is True
    for k in ("metric", "value", "unit", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "summary"):
        assert k in out, k
    weak, strong = out["configs"]
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert weak["world_size"] == 2 and "allgather_ms" in weak and "allgather_ms" in strong
    assert weak["config"]["scenarios_total"] == 12 and strong["config"]["scenarios_total"] == 13      # ragged strong shards: 7 + 6
    # whole-job units: both ranks' region QPs per step
    assert abs(weak["value"] * weak["ms_per_step"] * 1e-3 - 12 * 4) < 1e-6
    assert abs(strong["value"] * strong["ms_per_step"] * 1e-3 - 13 * 4) < 1e-6
    assert set(out["summary"]) == {"plumbing", "cfg5_weak", "cfg5_strong"}


def test_bench_one_rank_no_process_group():
    out = _run(1)
    assert out["n_gpus"] == 1 and out["configs"][1]["config"]["scenarios_this_rank"] == 13
