#!/bin/bash
# A/B builds of the solver kernels whose code generation differs most; every variant must pass the parity tests on the GPU
# (ADVICE r3 / VERDICT r4 item 1: two A/B builds of round 4 computed wrong numbers -- DESIGN.md section 8).  Run in the build container, then
#     gpurun -- 'bash tools/gpu_pass.sh TAG gen:g000 gen:g001 ... gen:g111 suite:fenceonly suite:slim0 suite:nfix0 suite:opaqueall suite:nodpp suite:ptreg0 suite:r5form bits:ptreg0 bits:r5form'
# [r5] The GENERAL instantiations (csrc/crx_kernels_gen.hip: run-time horizon / exponent, 4..6 obstacles) over the full 2^3 matrix of the
# three flags whose combination failed in round 4: gSDM with S = static LDS (+ no register floor: the AGPR-parking build), D = inline-assembly
# DPP dot products, M = masked sweeps (the shipped unit is g101: static LDS + AGPR copies, v_readlane, masked sweeps).  Every cell is first
# passed through tools/exec_prologue_check.py (a spill store in front of an EXEC restore: the compiler bug of DESIGN.md section 8); the S = 0 cells
# spill to scratch and are where it appears.
# Tuned units: fenceonly = SYNC() as up to 0.2.1 (fence without the wave barrier), slim0 / nfix0 / opaqueall / nodpp as in round 4.
cd "$(dirname "$0")/.."
rm -rf tools/ab/g[01][01][01] tools/ab/libcrx_g[01][01][01].so
for S in 0 1; do for D in 0 1; do
  for M in 0 1; do
    W=2; [ $S = 1 ] && W=0
    bash tools/build_variant.sh g$S$D$M "-DCRX_STATIC_LDS=$S -DCRX_GEN_WAVES=$W -DCRX_ROWDPP=$D -DCRX_SWEEP_MASK=$M" crx_kernels_gen.hip > /dev/null &
  done; wait
done; done
for c in 000 001 010 011 100 101 110 111; do
  S=${c:0:1}; D=${c:1:1}; M=${c:2:1}; W=2; [ $S = 1 ] && W=0
  echo "g$c: $(python tools/exec_prologue_check.py tools/ab/libcrx_g$c.so | tail -1)"     # [r6] on the code objects of the variant itself
done | tee tools/ab/exec_prologue_matrix.txt
bash tools/build_variant.sh fenceonly "-DCRX_SYNC_FENCE_ONLY" crx_kernels.hip crx_kernels_obs.hip crx_kernels_gen.hip crx_lmpc.hip crx_prep.hip crx_lmpcprep.hip > /dev/null &
bash tools/build_variant.sh slim0 "-DCRX_SLIM=0" crx_kernels_obs.hip > /dev/null &
wait
bash tools/build_variant.sh nfix0 "-DCRX_NFIX=0" crx_kernels.hip crx_kernels_obs.hip > /dev/null &
bash tools/build_variant.sh opaqueall "-DCRX_OPAQUE_LANE=2" crx_kernels_obs.hip > /dev/null &
bash tools/build_variant.sh nodpp "-DCRX_ROWDPP=0" crx_kernels.hip crx_kernels_obs.hip > /dev/null &
wait
# [r6] ptreg0 = the Riccati sweep with P through LDS (the form of rounds 1-5; the shipped one keeps (P | p) in registers: DESIGN.md 5.1) -- also the
# reference build of `gpu_pass.sh TAG bits:ptreg0`.  (The gS1M cells and nfix0 run the register form on run-time horizons.)
bash tools/build_variant.sh ptreg0 "-DCRX_PT_REG=0" > /dev/null &
# r5form = the Riccati sweep as rounds 1-5 had it, every round-6 switch off: P through LDS, mirrored H stores, select + add, the scheduler's load order
bash tools/build_variant.sh r5form "-DCRX_PT_REG=0 -DCRX_H_MIRROR=1 -DCRX_MASK_FMA=0 -DCRX_HUU_FIRST=0" crx_kernels.hip crx_kernels_obs.hip crx_kernels_gen.hip > /dev/null &
wait
ls -la tools/ab/*.so
