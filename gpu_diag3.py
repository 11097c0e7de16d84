import sys, os, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'car-racing_amd')
import numpy as np
import crx
from crx import abi, synth
np.set_printoptions(precision=3, linewidth=220)
gpu = crx.init(); L = crx.lib()
A, B = synth.load_AB()
p = synth.cfg2_mpccbf(256); d = abi.cbf_desc(12, 1, A, B)
args = (p['x0'], p['xt'], p['obs_s'], p['obs_ey'], p['lap_off'], p['n_obs'])
R = 20
L.crx_trace_enable(3, -R if len(sys.argv) > 1 else R)
rg = gpu.cbf_solve(d, *args)
tr = np.zeros((R, 16)); L.crx_trace_read(tr.ctypes.data_as(ctypes.c_void_p), R)
print("status", rg['status'][3], "iters", rg['iters'][3])
print("cycles per phase: rows/kkt  adjoint  mu-update  asm_newton  ric_back  ric_fwd  rowsteps  linesearch | alpha")
for r in tr[:rg['iters'][3]]: print("  ", " ".join("%9d" % v for v in r[8:16]), " total %9d" % r[8:16].sum(), " alpha %.2e" % r[4])
