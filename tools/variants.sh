#!/bin/bash
# The A/B builds of the solver kernels whose code generation differs most (a semantically neutral barrier once produced a
# faulting kernel; every variant must pass the parity suite): run in the build container, then on the GPU box
#     gpurun -- 'bash tools/gpu_pass.sh TAG suite:slim0 suite:nfix0 suite:deg0 suite:opaque0 suite:opaqueall suite:opaqueall_slim0'
cd "$(dirname "$0")/.."
bash tools/build_variant.sh slim0 "-DCRX_SLIM=0" crx_kernels_obs.hip &
bash tools/build_variant.sh nfix0 "-DCRX_NFIX=0" &
bash tools/build_variant.sh deg0 "-DCRX_DEG6=0" crx_kernels_obs.hip &
wait
bash tools/build_variant.sh opaque0 "-DCRX_OPAQUE_LANE=0" crx_kernels.hip &
bash tools/build_variant.sh opaqueall "-DCRX_OPAQUE_LANE=2" crx_kernels_obs.hip &
bash tools/build_variant.sh opaqueall_slim0 "-DCRX_OPAQUE_LANE=2 -DCRX_SLIM=0" crx_kernels_obs.hip &
wait
ls -la tools/ab/*.so
