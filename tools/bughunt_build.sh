#!/bin/bash
# Round 5: variants of the failing source state of DESIGN.md section 8, observation (1) (profiles/r04_open_observation.patch applied to 81b5071,
# kept under tools/ab/bugsrc/): the obstacle translation unit is built from it under one changed condition each and linked with the in-tree
# objects of everything else (same ABI).  On the GPU: tools/bughunt_run.sh
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/car-racing_amd/csrc; B=$R/tools/ab/bugsrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -disable-machine-licm"
ILP="-mllvm -amdgpu-sched-strategy=iterative-ilp"
one() {  # name srcdir flags...
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FL "$@" -c $d/crx_kernels_obs.hip -o $B/obs_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libcrx_bug_$n.so $S/crx_kernels.o $B/obs_$n.o $B/gen_stub.o $S/crx_lmpc.o $S/crx_prep.o $S/crx_lmpcprep.o $S/crx_api.o && echo built bug_$n
}
one base $B $ILP &
one nossc $B $ILP -mllvm -disable-ssc &
one prealloc $B $ILP -mllvm -amdgpu-prealloc-sgpr-spill-vgprs &
one nop7 $B/nop7 $ILP &
wait
one noagpr $B $ILP -mllvm -amdgpu-spill-vgpr-to-agpr=0 &
one builtin $B/builtin $ILP &
one nopostra $B $ILP -mllvm -disable-post-ra &
one defsched $B &
wait
one ll1 $B $ILP -DCRX_SWEEP_LOCAL_LANE=1 &   # positive control: the forward sweep's lane barrier alone passed in round 4
wait
