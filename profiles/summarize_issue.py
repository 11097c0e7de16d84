"""What binds a resident solver wave -- issue / wait / instruction counters of the full-batch launches (rocprofv3 --pmc passes of
`tools/gpu_pass.sh TAG issue`, csv under gpurun_out/TAG/).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (4 clocks) summed
over all waves; SQ_INSTS_* count instructions.  (Mean iterations per solve: pass them as cfg4,cfg2,cfg3 -- the bench line of the same batch.)
Usage: python profiles/summarize_issue.py TAG [it_cfg4,it_cfg2,it_cfg3] > profiles/rNN_pmc_issue.txt
(the summaries of rounds 2-3 -- profiles/r03_pmc_issue*.txt -- were made by the round-3 version of this script, see git history)"""
import collections, csv, glob, os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r4g"
D = os.path.join(ROOT, "gpurun_out", TAG)
ITS = [float(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("18.0", "11.6", "6.0"))]
FINAL = True
RUNS = [("cfg4 <3,20>, 4 waves per CU, index order", "pmc_cfg4_default_", 16384, ITS[0]),
        ("cfg2 <1,12> at batch 16384, 8 waves per CU", "pmc_cfg2_", 16384, ITS[1]),
        ("cfg3 <0,12> at 65536 QPs, 12 waves per CU, reachability screen ON (41 % of the waves end before set-up)", "pmc_cfg3_", 65536, ITS[2])]


def load(prefix, grid):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(D, prefix + "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "crx_solve_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid * 64:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


print(__doc__.split("Usage")[0].strip() + "\n[" + TAG + "]")
print()
for title, prefix, n, it_mean in RUNS:
    c = load(prefix, n)
    if not c:
        print(title + ": no data"); continue
    print("%s   [%d problems per launch, ~%.1f interior-point iterations per solve]" % (title, n, it_mean))
    for k in sorted(c):
        print("  %-30s %.4g" % (k, c[k]))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        ins = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        print("  -- per resident wave-cycle: issuing an instruction %.0f %% (VALU %.0f %%, LDS %.0f %%, scalar %.0f %%), parked on s_waitcnt %.0f %%, issue-stalled %.0f %%" % (
            100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc, 100 * c["SQ_ACTIVE_INST_SCA"] / wc,
            100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc))
        print("  -- per solve: %.0f k clocks resident (%.1f k per iteration), of which %.1f k per iteration issuing (one instruction per 4 clocks and wave)" % (
            4 * wc / n / 1e3, 4 * wc / n / it_mean / 1e3, 4 * c["SQ_ACTIVE_INST_ANY"] / n / it_mean / 1e3))
        if ins:
            print("  -- per solve: %.1f k VALU + %.1f k LDS + %.1f k scalar instructions = %.1f k per iteration" % (
                c.get("SQ_INSTS_VALU", 0) / n / 1e3, c.get("SQ_INSTS_LDS", 0) / n / 1e3, c.get("SQ_INSTS_SALU", 0) / n / 1e3, ins / n / it_mean / 1e3))
    if "SQC_ICACHE_REQ" in c:
        print("  -- instruction cache: %.4g requests, %.4g misses (hit rate %.4f %%)" % (c["SQC_ICACHE_REQ"], c["SQC_ICACHE_MISSES"], 100 * c["SQC_ICACHE_HITS"] / c["SQC_ICACHE_REQ"]))
    print()
if not FINAL:
    print("""Reading.  A single-wave workgroup owns a SIMD's issue slot once every four clocks, so a wave retires at most one instruction per
4 clocks.  The obstacle instantiations keep 257..512 registers per lane = ONE wave per SIMD: their iteration is ~15 k (<3,20>) / ~10 k
(<1,12>) instructions = 60 k / 40 k clocks of pure issue, which is 70..90 % of the iteration time of a lone wave (89.7 k / 41 k clocks,
profiles/r02_phase_cycles.txt).  They are bound by the INSTRUCTION COUNT, not by latency, LDS or the instruction cache (hit rate
99.98 %).  With 3 -> 4 such waves per CU (slim layout) every wave still has its own SIMD, but the LDS pipe is shared: the time parked
on s_waitcnt grows from 26 % to 28 % of a longer residency and the launch gains 10 % (27.3 ms -> 24.75 ms, same draw and session:
tools/gpu_round3_f.sh) although 33 % more problems are in flight.  (The slim-layout pass above was taken on the build that also
recomputed the lane maps in <3,20> -- 18.2 k instructions per iteration; the shipped build keeps them hoisted there.)  The planner instantiation (3 waves per SIMD) interleaves waves on a SIMD and is bound by the VALU pipe.""")
