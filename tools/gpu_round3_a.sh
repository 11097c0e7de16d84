# round 3, GPU call A: the whole -m gpu suite on the new synth / fixtures / slim <3,20> layout, the default bench line, and
# the slim-vs-full A/B on cfg4 (tools/ab/libcrx_noslim.so = make EXTRA=-DCRX_SLIM=0).
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3a
mkdir -p $O
make -C oracle -s
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
# the same suite on the full-layout build (is a failure above the slim layout's?)
( CRX_LIB=$R/tools/ab/libcrx_noslim.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg4 or draws_cbf or fuzz or synthetic_cbf" 2>&1 | tail -15 ) > $O/pytest_noslim.log
tail -3 $O/pytest_noslim.log
for v in slim noslim; do
  L=""; [ $v = noslim ] && L=$R/tools/ab/libcrx_noslim.so
  CRX_LIB=$L python bench.py --workload cfg4 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_cfg4_$v.json 2> $O/bench_cfg4_$v.err
  python -c "import json; d=json.loads(open('$O/bench_cfg4_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['lds_bytes_per_problem'], d['roofline']['resident_problems_per_cu'], d['config']['status_frac'], d['config']['iters_p50'], d['config']['iters_max'])"
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["value_converged"], d["ms_per_step"])
for k, v in d["summary"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_ms", "converged_frac", "iters_max", "roofline_frac", "resident_per_cu", "problems_launched")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread"]["value"])
PY
