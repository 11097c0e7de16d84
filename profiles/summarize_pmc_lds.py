"""profiles/<round>_pmc_lds_valu.txt from the rocprofv3 --pmc passes of tools/gpu_pmc_lds.sh (SQ VALU / LDS counters of the solver
kernel, one counter group per pass, csv output under gpurun_out/pmc_lds/).  Usage: python profiles/summarize_pmc_lds.py [prefix]"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PRE = sys.argv[1] if len(sys.argv) > 1 else "r02"
FULL = {"cfg3": 4096 * 64, "cfg4": 16384 * 64, "cfg5": 65536 * 64}        # grid size of the full-batch launches


def solver_rows(path, grid):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "crx_solve" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def lds():
    lines = ["# rocprofv3 --pmc (separate passes, --kernel-trace only) on crx_solve_kernel; tools/gpu_pmc_lds.sh",
             "# values: mean over the full-batch launches, summed over the chip as rocprofv3 reports them.  SQ_*_CYCLES and",
             "# SQ_ACTIVE_INST_* count quad-cycles (4 clocks); GRBM_GUI_ACTIVE counts clocks summed over the 8 XCDs"]
    for wl, title, waves in (("cfg3", "planner QPs N=12, batch 4096 (12 resident per CU)", 4096), ("cfg5", "planner QPs N=12 from raw scenarios, 65536 per launch", 65536),
                             ("cfg4", "tracking NLP N=20, 3 obstacles, batch 16384, SURVEY 8d draw", 16384)):
        agg = {}
        for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_lds", wl + "_*"))):
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not f:
                continue
            for k, v in solver_rows(f[0], FULL[wl]).items():
                agg[k] = sum(v) / len(v)
        lines.append("")
        lines.append("%s (%s)" % (wl, title))
        for k in sorted(agg):
            lines.append("  %-24s %.4g" % (k, agg[k]))
        g = agg.get
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            lines.append("  -- per wave: %.0f VALU instructions, %.0f LDS instructions (VALU : LDS = %.1f : 1)"
                         % (g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_LDS") / waves, g("SQ_INSTS_VALU") / g("SQ_INSTS_LDS")))
            lines.append("  -- per resident wave-cycle: VALU active %.1f %%, LDS instruction active %.1f %%, waiting on LDS %.1f %%"
                         % (100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc))
            cu_cycles = g("GRBM_GUI_ACTIVE") / 8 * 256
            lines.append("  -- chip: VALU pipe busy %.1f %% (4 clocks per instruction over 1024 SIMDs), LDS unit busy %.1f %%, bank-conflict cycles / LDS-active cycles %.0f %%"
                         % (100 * 4 * g("SQ_INSTS_VALU") / (4 * cu_cycles), 100 * g("SQ_LDS_IDX_ACTIVE") / cu_cycles,
                            100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    open(os.path.join(ROOT, "profiles", "%s_pmc_lds_valu.txt" % PRE), "w").write("\n".join(lines) + "\n")



if __name__ == "__main__":
    lds()
    print(open(os.path.join(ROOT, "profiles", "%s_pmc_lds_valu.txt" % PRE)).read())
