#!/bin/bash
# A/B builds of libcrx: tools/build_variant.sh NAME "EXTRA-FLAGS" [file.hip ...]
# compiles the named sources (default: the two tuned translation units of the solver, crx_kernels.hip + crx_kernels_obs.hip; the general
# one is crx_kernels_gen.hip) with the extra
# flags into tools/ab/NAME/ and links
# tools/ab/libcrx_NAME.so with the in-tree objects of the other sources.  Select at run time: CRX_LIB=tools/ab/libcrx_NAME.so.
# (*.so is git-ignored but travels to the GPU box with the snapshot.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2 || true
FILES=${@:-crx_kernels.hip crx_kernels_obs.hip}
PLAN_SCHED=${PLAN_SCHED--mllvm -amdgpu-sched-strategy=max-ilp}          # PLAN_SCHED= (empty): the planner unit with the default scheduler
OBS_SCHED=${OBS_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp}   # OBS_SCHED= (empty) builds the obstacle unit with the default scheduler
S=$R/car-racing_amd/csrc
D=$R/tools/ab/$NAME
mkdir -p $D
make -C $S -s
OBJS=""
for f in crx_kernels crx_kernels_obs crx_kernels_spec crx_kernels_gen crx_lmpc crx_prep crx_lmpcprep crx_api; do
  if echo " $FILES " | grep -q " $f.hip "; then
    NOLICM=${NOLICM--mllvm -disable-machine-licm}                         # NOLICM= (empty): MachineLICM left on
    LICM=""; [ $f = crx_kernels ] && LICM="$NOLICM $PLAN_SCHED"; [ $f = crx_kernels_obs ] && LICM="$NOLICM $OBS_SCHED"; [ $f = crx_kernels_spec ] && LICM="$NOLICM $OBS_SCHED"; [ $f = crx_kernels_gen ] && LICM="$NOLICM"; [ $f = crx_lmpc ] && LICM="$NOLICM ${LMPC_SCHED--mllvm -amdgpu-sched-strategy=max-ilp}"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $LICM $EXTRA -I$S -c $S/$f.hip -o $D/$f.o
    OBJS="$OBJS $D/$f.o"
  else
    OBJS="$OBJS $S/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libcrx_$NAME.so $OBJS
echo built tools/ab/libcrx_$NAME.so
