# round 3, GPU call A: the whole -m gpu suite on the new synth / fixtures / slim <3,20> layout / opaque lane, the default bench
# line, and A/B of the kernel variants (tools/build_variant.sh): base = lane maps hoisted (round 2), noslim = full <3,20> layout,
# w2 = <2,12> pinned at two waves per SIMD.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3a
mkdir -p $O
make -C oracle -s
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
one() {  # variant workload steps
  L=""; [ $1 != default ] && L=$R/tools/ab/libcrx_$1.so
  CRX_LIB=$L python bench.py --workload $2 --no-cpu-baseline --steps $3 --warmup 3 > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err
  python -c "import json; d=json.loads(open('$O/bench_$2_$1.json').read().strip().splitlines()[-1]); print('$1 $2 value %.4g ms/step %.4f kernel_ms %.4f lds %d res %d conv %.4f it50 %g itmax %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['lds_bytes_per_problem'], d['roofline']['resident_problems_per_cu'], d['config']['converged_frac'], d['config']['iters_p50'], d['config']['iters_max']))" || tail -3 $O/bench_$2_$1.err
}
for v in default base; do
  one $v cfg2 100; one $v cfg3 100; one $v cfg4 15; one $v cfg5 20; one $v races 30; one $v overtake 40
done
one noslim cfg4 15
one w2 races 30
( CRX_LIB=$R/tools/ab/libcrx_noslim.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg4 or draws_cbf" 2>&1 | tail -5 ) > $O/pytest_noslim.log
tail -2 $O/pytest_noslim.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["value_converged"], d["ms_per_step"])
for k, v in d["summary"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_ms", "converged_frac", "iters_max", "roofline_frac", "resident_per_cu", "problems_launched")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread"]["value"])
PY
