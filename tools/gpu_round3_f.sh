# round 3, GPU call F: smoke(), the suite and the default bench line on the final build; slim vs full <3,20> layout, same session.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3f
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25 ) > $O/pytest_gpu.log
grep -h "passed\|failed" $O/pytest_gpu.log | tail -2
for v in default noslim default noslim; do
  L=""; [ $v != default ] && L=$R/tools/ab/libcrx_$v.so
  CRX_LIB=$L python bench.py --workload cfg4 --no-cpu-baseline --steps 15 --warmup 3 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $v', d['value'], d['roofline']['kernel_ms'], d['roofline']['lds_bytes_per_problem'], d['roofline']['resident_problems_per_cu'], d['config']['converged_frac'])"
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["value_converged"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60])
for k, v in d["summary"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_ms", "converged_frac", "iters_max", "traffic", "resident_per_cu")})
PY
