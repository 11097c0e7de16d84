"""Regenerate profiles/r01_pmc_hbm_traffic.txt, profiles/pmc_hbm_traffic.json and profiles/r01_pmc_lds_valu.txt from
the rocprofv3 --pmc passes of tools/gpu_pmc.sh and tools/gpu_pmc_lds.sh (csv output under gpurun_out/).
Usage: python profiles/summarize_pmc.py [round-prefix, default r01]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PRE = sys.argv[1] if len(sys.argv) > 1 else "r01"
FULL = {"cfg2": 256 * 64, "cfg3": 4096 * 64, "cfg4": 16384 * 64}        # grid size of the full-batch launches
ALGO = {"cfg2": (1256, 256), "cfg3": (1200, 4096)}


def solver_rows(path, grid):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "crx_solve" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def hbm():
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace), bench.py --steps 10",
             "# counter unit: KiB (MI355X_MICROARCH.md section HBM); FETCH_SIZE on gfx950 reads 1/2 of a wide coalesced stream -> x2 correction shown beside the raw sum"]
    js = {"note": "HBM bytes per launch of crx_solve_kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                  "(FETCH_SIZE+WRITE_SIZE)*1024, uncorrected: the gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md is calibrated "
                  "for 16 B/lane streams, this kernel reads 8 B/lane; see profiles/%s_pmc_hbm_traffic.txt" % PRE}
    for wl in ("cfg2", "cfg3"):
        v = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "%s_%s" % (wl, ctr), "*counter_collection.csv"))
            v[ctr] = solver_rows(f[0], FULL[wl])[ctr]
        F, W = v["FETCH_SIZE"], v["WRITE_SIZE"]
        fm, wm = sum(F) / len(F), sum(W) / len(W)
        per, n = ALGO[wl]
        lines.append("%s: solver kernel, per launch (mean of %d launches): FETCH_SIZE=%.1f KiB (min %.1f max %.1f)  WRITE_SIZE=%.1f KiB (min %.1f max %.1f)"
                     "  -> raw (F+W)*1024 = %d B; with the gfx950 x2 read correction (2F+W)*1024 = %d B; algorithmic = %d B (%d B x %d solves)"
                     % (wl, len(F), fm, min(F), max(F), wm, min(W), max(W), round((fm + wm) * 1024), round((2 * fm + wm) * 1024), per * n, per, n))
        js[wl] = {"batch": n, "fetch_kib": round(fm, 1), "write_kib": round(wm, 1), "traffic_bytes": round((fm + wm) * 1024),
                  "traffic_bytes_x2_fetch": round((2 * fm + wm) * 1024)}
    open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % PRE), "w").write("\n".join(lines) + "\n")
    json.dump(js, open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json"), "w"), indent=1)


def lds():
    lines = ["# rocprofv3 --pmc (separate passes, --kernel-trace only) on crx_solve_kernel; tools/gpu_pmc_lds.sh",
             "# values: mean over the full-batch launches, summed over the chip as rocprofv3 reports them.  SQ_*_CYCLES and",
             "# SQ_ACTIVE_INST_* count quad-cycles (4 clocks); GRBM_GUI_ACTIVE counts clocks summed over the 8 XCDs"]
    for wl, title, waves in (("cfg3", "planner QPs N=12, batch 4096", 4096), ("cfg4", "tracking NLP N=20, 3 obstacles, batch 16384", 16384)):
        agg = {}
        for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_lds", wl + "_*"))):
            f = glob.glob(os.path.join(d, "*counter_collection.csv"))
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
    hbm()
    lds()
