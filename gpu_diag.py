import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'car-racing_amd'); sys.path.insert(0, 'tests')
import numpy as np
import crx, oracle
from crx import abi, synth
import helpers
from conftest import Golden
np.set_printoptions(precision=4, linewidth=220)
gpu = crx.init(); orc = oracle.load()
A, B = synth.load_AB()
def rep(tag, rg, ro):
    sg, so = rg['status'], ro['status']
    both = (sg == 0) & (so == 0)
    print("==", tag, "n", len(sg), "gpu status", np.bincount(sg, minlength=3), "orc status", np.bincount(so, minlength=3), "verdict mismatch", int(((sg==0)!=(so==0)).sum()))
    if both.any():
        dX = np.abs(rg['X'][both]-ro['X'][both])
        rel = np.abs(rg['cost'][both]-ro['cost'][both])/np.maximum(1, np.abs(ro['cost'][both]))
        print("   both-conv %d  dXw %.2e dXall %.2e dU %.2e relcost %.2e  kkt gpu max %.2e  iters equal frac %.3f  gpu iters p50 %d orc p50 %d" % (both.sum(), dX[...,[0,4,5]].max(), dX.max(), np.abs(rg['U'][both]-ro['U'][both]).max(), rel.max(), rg['kkt'][both].max(), (rg['iters'][both]==ro['iters'][both]).mean(), np.median(rg['iters'][both]), np.median(ro['iters'][both])))
        worst = np.argsort(-dX.reshape(both.sum(), -1).max(axis=1))[:3]
        idx = np.nonzero(both)[0][worst]
        print("   worst idx", idx, "iters gpu", rg['iters'][idx], "orc", ro['iters'][idx], "dX", dX.reshape(both.sum(), -1).max(axis=1)[worst])
    mm = np.nonzero((sg==0)!=(so==0))[0]
    if len(mm):
        print("   mismatch idx", mm[:10], "gpu st", sg[mm][:10], "orc st", so[mm][:10], "gpu it", rg['iters'][mm][:10], "orc it", ro['iters'][mm][:10], "gpu kkt", rg['kkt'][mm][:10], "orc kkt", ro['kkt'][mm][:10])
g = Golden('tests/golden/mpccbf.npz')
for name in g.names:
    c = g.case(name); d, args = helpers.mpccbf_inputs(c, A, B)
    rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
    print(name, "gpu st %d it %d cost %.9f kkt %.1e | orc st %d it %d cost %.9f | golden %.9f  dX %.2e" % (rg['status'][0], rg['iters'][0], rg['cost'][0], rg['kkt'][0], ro['status'][0], ro['iters'][0], ro['cost'][0], c['cert'][0], np.abs(rg['X']-ro['X']).max()))
g = Golden('tests/golden/planner.npz')
for name in g.names:
    c = g.case(name)
    if not bool(c['overtake_flag']): continue
    d, args = helpers.planner_inputs(c, A, B)
    rg, ro = gpu.planner_solve(d, *args), orc.planner_solve(d, *args)
    print(name, "gpu st", rg['status'], "it", rg['iters'], "orc st", ro['status'], "it", ro['iters'], "golden ok", c['region_success'].astype(int), "dX %.2e" % np.abs(rg['X']-ro['X']).max(), "kkt", rg['kkt'])
p = synth.cfg2_mpccbf(256); d = abi.cbf_desc(p['N'], 1, A, B, alpha=p['alpha'], margin=p['margin'])
args = (p['x0'], p['xt'], p['obs_s'], p['obs_ey'], p['lap_off'], p['n_obs'])
rep("cfg2", gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args))
p = synth.cfg4_tracking_cbf(192); d = abi.cbf_desc(p['N'], 3, A, B, alpha=0.6, margin=0.15, Q=(10.,0,0,5.,0,50.), per_stage_target=True)
args = (p['x0'], p['xt'], p['obs_s'], p['obs_ey'], p['lap_off'], p['n_obs'])
rep("cfg4", gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args))
for N in (12, 20):
    p = synth.cfg3_planner(128, N=N); d = abi.planner_desc(N, A, B)
    args = (p['x0'], p['bez_s'], p['bez_ey'], p['ey_lb'], p['ey_ub'])
    rep("cfg3 N=%d" % N, gpu.planner_solve(d, *args), orc.planner_solve(d, *args))
