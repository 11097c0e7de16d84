# round-2 final measurements on the GPU box (one gpurun call): default bench line (with the CPU baselines), rocprofv3 kernel
# stats of every workload, PMC passes (HBM traffic; LDS / VALU counters), per-phase cycle traces (TRACE build).  Summaries
# land in gpurun_out/ and are copied to profiles/ after review.
R=$GRAFT_REPO_ROOT
cd $R
make -C oracle -s
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -c 600 gpurun_out/final/bench_default.json; echo
bash tools/gpu_prof_r2.sh > gpurun_out/final/prof.log 2>&1
bash tools/gpu_pmc_r2.sh > gpurun_out/final/pmc.log 2>&1
bash tools/gpu_pmc_lds.sh > gpurun_out/final/pmc_lds.log 2>&1
python3 profiles/summarize_pmc_lds.py r02 >> gpurun_out/final/pmc_lds.log 2>&1
cp profiles/r02_pmc_hbm_traffic.txt profiles/r02_pmc_lds_valu.txt profiles/pmc_hbm_traffic.json gpurun_out/final/ 2>/dev/null
make -C car-racing_amd/csrc -s clean all TRACE=1 2>&1 | grep -i error
( for w in cfg2 cfg3 cfg4; do python tools/gpu_solve_trace.py $w; done; python tools/gpu_lmpc_trace.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/final/r02_phase_cycles.txt
make -C car-racing_amd/csrc -s clean all 2>&1 | grep -i error
ls gpurun_out/final gpurun_out/prof2 | head -60
