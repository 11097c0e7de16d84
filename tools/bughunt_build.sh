#!/bin/bash
# Round 5: the failing source state of DESIGN.md section 8, observation (1), rebuilt under ONE changed condition per variant (what
# profiles/r05_wrong_result_hunt.txt reports).  Source = commit 81b5071 + profiles/r04_open_observation.patch, regenerated here into
# tools/ab/bugsrc/ (git-ignored); the obstacle translation unit is built from it and linked with the in-tree objects of everything else
# (same ABI; a stub stands in for crx_kernels_gen.o, the old unit holds its own general instantiations).  GPU side: tools/bughunt_run.sh TAG [variant]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/car-racing_amd/csrc; B=$R/tools/ab/bugsrc
mkdir -p $B; cd $R
git show 81b5071:car-racing_amd/csrc/crx_kernels.hip > $B/crx_kernels_81b5071.hip
git show 81b5071:car-racing_amd/csrc/crx_wave.h > $B/crx_wave.h
git show 81b5071:car-racing_amd/csrc/crx_kparams.h | sed 's#"../../include/crx.h"#"crx.h"#' > $B/crx_kparams.h
git show 81b5071:include/crx.h > $B/crx.h
patch -s -p0 -o $B/crx_kernels.hip $B/crx_kernels_81b5071.hip < profiles/r04_open_observation.patch
printf '#define CRX_TU_OBSTACLES 1\n#include "crx_kernels.hip"\n' > $B/crx_kernels_obs.hip
printf '#include "crx_kparams.h"\nhipError_t crx_launch_solve_general(const crx_kparams&, int, hipStream_t) { return hipErrorInvalidValue; }\nint crx_solve_resident_per_cu_general(int, int) { return -1; }\n' > $B/gen_stub.hip
for v in nop7 builtin wb dump; do mkdir -p $B/$v; cp $B/crx_kernels.hip $B/crx_kernels_obs.hip $B/crx_kparams.h $B/crx.h $B/crx_wave.h $B/$v/; done
sed -i 's/"s_nop 1\\n\\t"/"s_nop 7\\n\\ts_nop 7\\n\\t"/g' $B/nop7/crx_wave.h
sed -i 's|#define SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")|#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)|' $B/wb/crx_wave.h
python3 - $B <<'PY'
import sys
B = sys.argv[1]
# builtin: row_dot through __builtin_amdgcn_update_dpp instead of inline assembly
s = open(B + "/crx_wave.h").read(); a = s.index("#define CRX_DPPT(i)"); b = s.index("#define ROW_REDUCE")
new = '''template <int CNT, int FIRST>
__device__ __forceinline__ double row_dot(double x, const double* m, double acc) {
#define CRX_BT(i) if constexpr (CNT > i) acc = fma(__builtin_amdgcn_update_dpp(0.0, x, 0x150 + FIRST + i, 0xF, 0xF, true), m[i], acc);
    CRX_BT(0) CRX_BT(1) CRX_BT(2) CRX_BT(3) CRX_BT(4) CRX_BT(5) CRX_BT(6) CRX_BT(7) CRX_BT(8)
#undef CRX_BT
    return acc;
}

'''
open(B + "/builtin/crx_wave.h", "w").write(s[:a] + new + s[b:])
# dump: the Lagrangian-gradient array, the multipliers and the CBF Jacobians at the kernel's exit, rows 32.. of the trace buffer (tools/trace_fuzz.py CRX_TRACE_DUMP)
s = open(B + "/crx_kernels.hip").read()
old = "    if (lane == 0) { kp.status[b] = status; kp.kkt[b] = E0; kp.iters[b] = it; }"
assert s.count(old) == 1
s = s.replace(old, old + '''
    if (kp.trace && b == kp.trace_problem) {
        for (int e = threadIdx.x; e < 300; e += WAVE) kp.trace[512 + e] = LD(L::ga + e);
        for (int e = threadIdx.x; e < 100; e += WAVE) kp.trace[812 + e] = LD(L::rnu + e);
        for (int e = threadIdx.x; e < 100; e += WAVE) kp.trace[912 + e] = NOBS ? LD(L::Jc + e) : 0.0;
    }''')
open(B + "/dump/crx_kernels.hip", "w").write(s)
PY
make -C $S -s
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -disable-machine-licm"
ILP="-mllvm -amdgpu-sched-strategy=iterative-ilp"
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC --offload-arch=gfx950 -I$B -c $B/gen_stub.hip -o $B/gen_stub.o
one() {  # name srcdir flags...
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FL "$@" -c $d/crx_kernels_obs.hip -o $B/obs_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libcrx_bug_$n.so $S/crx_kernels.o $B/obs_$n.o $B/gen_stub.o $S/crx_lmpc.o $S/crx_prep.o $S/crx_lmpcprep.o $S/crx_api.o && echo built bug_$n
}
one base $B $ILP & one defsched $B & one ll1 $B $ILP -DCRX_SWEEP_LOCAL_LANE=1 & one nop7 $B/nop7 $ILP & wait
one noagpr $B $ILP -mllvm -amdgpu-spill-vgpr-to-agpr=0 & one nopostra $B $ILP -mllvm -disable-post-ra & one prealloc $B $ILP -mllvm -amdgpu-prealloc-sgpr-spill-vgprs & one builtin $B/builtin $ILP & wait
one noaa $B $ILP -mllvm -amdgpu-use-aa-in-codegen=0 & one nomisched $B $ILP -mllvm -enable-misched=0 & one wb $B/wb $ILP & one nossc $B $ILP -mllvm -disable-ssc & wait
one dump_ilp $B/dump $ILP & one dump_def $B/dump & wait
