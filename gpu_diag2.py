import sys, os, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'car-racing_amd'); sys.path.insert(0, 'tests')
import numpy as np
import crx, oracle
from crx import abi, synth
np.set_printoptions(precision=3, linewidth=220)
gpu = crx.init(); orc = oracle.load(); L = crx.lib()
A, B = synth.load_AB()
N = 12
p = synth.cfg3_planner(128, N=N); d = abi.planner_desc(N, A, B)
args = (p['x0'], p['bez_s'], p['bez_ey'], p['ey_lb'], p['ey_ub'])
for idx in (0, 10):
    L.crx_trace_enable(idx, 40)
    rg = gpu.planner_solve(d, *args)
    tr = np.zeros((40, 8)); L.crx_trace_read(tr.ctypes.data_as(ctypes.c_void_p), 40)
    print("problem", idx, "status", rg['status'][idx], rg['iters'][idx], rg['kkt'][idx])
    print("   e_d       e_p       e_c       mu        alpha     a_d       dw      acc")
    for r in tr[:32]: print("  ", " ".join("%9.2e" % v for v in r))
    orc.lib.crx_oracle_set_verbose(1)
    sl = slice(idx, idx+1)
    ro = orc.planner_solve(d, *[a[sl] for a in args])
    orc.lib.crx_oracle_set_verbose(0)
