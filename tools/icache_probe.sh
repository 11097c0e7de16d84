R=$PWD; O=$R/gpurun_out/r6L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { tag=$1; b=$2; shift 2; rm -rf $O/pmc_$tag; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py --steps 20 --warmup 2 --workload cfg2 --batch $b --no-cpu-baseline > /dev/null 2> $O/err_$tag.log; find $O/pmc_$tag -type f ! -name "*counter_collection.csv" -delete; }
pmc ic256 256 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pmc w256 256 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("ic256", "w256"):
    f = glob.glob("gpurun_out/r6L/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for fn in f:
        for r in csv.DictReader(open(fn)):
            if "crx_solve_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 256 * 64:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(tag, k, "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
