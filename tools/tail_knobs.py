"""Crash-state tail experiments on the CPU oracle (round 3): iteration statistics of the BASELINE draws under
different restoration triggers.  python tools/tail_knobs.py"""
import sys, os
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "car-racing_amd")]
import numpy as np
import oracle
from crx import synth, abi

orc = oracle.load(); A, B = synth.load_AB()
KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
p2 = synth.cfg2_mpccbf(2048, safe_start=False)
d2 = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
p4 = synth.cfg4_tracking_cbf(1024, safe_start=False)
d4 = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
DEF = {0: 1e-3, 1: 5, 2: 100, 3: 0.0, 4: 0, 5: 2, 6: -1}


def run(knobs, base=None, slack_start=2):
    """knobs: oracle experiment knobs (crx_oracle.c g_knob); slack_start: crx_ipm_opts.slack_start of both descriptors"""
    for i, v in {**DEF, **knobs}.items():
        orc.lib.crx_oracle_set_knob(int(i), __import__("ctypes").c_double(float(v)))
    out = []
    for d in (d2, d4):
        d.opts.slack_start = int(slack_start)
    for name, p, d in (("cfg2", p2, d2), ("cfg4", p4, d4)):
        r = orc.cbf_solve(d, *[p[k] for k in KEYS])
        st, it = r["status"], r["iters"]
        s = "%s conv %.3f inf %.3f rest %.3f cap %.3f | it mean %.1f p50 %d p90 %d p99 %d max %d | max over 8 batches of 256: %s" % (
            name, (st == 0).mean(), (st == 2).mean(), (st == 3).mean(), (st == 1).mean(), it.mean(), *np.percentile(it, [50, 90, 99, 100]).astype(int),
            [int(it[i:i + 256].max()) for i in range(0, min(len(it), 2048), 256)])
        if base is not None:
            rb = base[name]
            both = (st == 0) & (rb["status"] == 0)
            rel = np.abs(r["cost"][both] - rb["cost"][both]) / np.maximum(1, np.abs(rb["cost"][both]))
            s += " | same-cost %.4f worse %d better %d newly-conv %d lost %d" % ((rel <= 1e-6).mean(), int(((r["cost"][both] - rb["cost"][both]) / np.maximum(1, np.abs(rb["cost"][both])) > 1e-6).sum()),
                  int(((rb["cost"][both] - r["cost"][both]) / np.maximum(1, np.abs(rb["cost"][both])) > 1e-6).sum()), int(((st == 0) & (rb["status"] != 0)).sum()), int(((st != 0) & (rb["status"] == 0)).sum()))
        out.append((name, r, s))
    return {n: r for n, r, _ in out}, [s for _, _, s in out]


if __name__ == "__main__":
    base, lines = run({}, slack_start=0)
    print("slack_start = 0 (libcrx 0.1.x: zero start, closed-form slack restoration)"); [print("  ", l) for l in lines]
    for ss in (1, 2):
        _, lines = run({}, base, slack_start=ss)
        print("slack_start = %d" % ss); [print("  ", l) for l in lines]
    trials = [{2: 20}, {3: 0.05, 4: 4}, {3: 0.05, 4: 6}, {3: 0.1, 4: 4}, {3: 0.1, 4: 6}, {3: 0.02, 4: 4}, {3: 0.1, 4: 8}, {3: 0.05, 4: 4, 5: 3}, {3: 0.2, 4: 5}]
    for t in trials:
        _, lines = run(t, base)
        print(t); [print("  ", l) for l in lines]
