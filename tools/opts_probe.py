"""Diagnostics: iteration counts of the planner QPs (convex: the solution does not depend on the barrier path) and of the
MPC-CBF NLPs under other barrier strategies than IPOPT's defaults (crx_ipm_opts); prints iterations, outcomes and the
distance of the solutions to the default ones."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd")
import numpy as np
import crx
from crx import abi, synth
gpu = crx.init()
A, B = synth.load_AB()
p3 = synth.cfg3_planner(256, N=12)
p2 = synth.cfg2_mpccbf(256, N=12, seed=2)
base3 = base2 = None
for name, kw in (("default", {}), ("mu_init 0.01", dict(mu_init=0.01)), ("mu_init 1", dict(mu_init=1.0)), ("kappa_mu 0.1", dict(kappa_mu=0.1)),
                 ("kappa_mu 0.05", dict(kappa_mu=0.05)), ("theta_mu 1.8", dict(theta_mu=1.8)), ("kappa_eps 30", dict(kappa_eps=30.0)),
                 ("kappa_mu .1 + theta 1.8", dict(kappa_mu=0.1, theta_mu=1.8)), ("mu .01 kappa .1 theta 1.8", dict(mu_init=0.01, kappa_mu=0.1, theta_mu=1.8))):
    o = abi.default_opts()
    for k, v in kw.items(): setattr(o, k, v)
    d3 = abi.planner_desc(12, A, B); d3.opts = o
    r3 = gpu.planner_solve(d3, p3["x0"], p3["bez_s"], p3["bez_ey"], p3["ey_lb"], p3["ey_ub"])
    d2 = abi.cbf_desc(12, 1, A, B, alpha=p2["alpha"], margin=p2["margin"]); d2.opts = o
    r2 = gpu.cbf_solve(d2, *[p2[k] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")])
    if base3 is None: base3, base2 = r3, r2
    ok3 = (r3["status"] == 0) & (base3["status"] == 0); ok2 = (r2["status"] == 0) & (base2["status"] == 0)
    print("%-28s planner: iters mean %.1f max %d, status %s, same outcome %.3f, max|dX| %.1e | cbf: iters mean %.1f max %d, status %s, max|dX| %.1e" % (
        name, r3["iters"].mean(), r3["iters"].max(), np.bincount(r3["status"], minlength=3), ((r3["status"] == 0) == (base3["status"] == 0)).mean(),
        np.abs(r3["X"][ok3] - base3["X"][ok3]).max(), r2["iters"].mean(), r2["iters"].max(), np.bincount(r2["status"], minlength=3),
        np.abs(r2["X"][ok2] - base2["X"][ok2]).max()))
