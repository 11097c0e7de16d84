"""profiles/<round>_pmc_hbm_traffic.txt + profiles/pmc_hbm_traffic.json from the rocprofv3 --pmc passes of tools/gpu_pmc_r2.sh
(csv output under gpurun_out/pmc3/ -- round 3: tools/gpu_round3_d.sh), corrected with the calibration of tools/gpu_calib.sh (profiles/pmc_calibration.json):
FETCH_SIZE under-counts an 8 B/lane coalesced stream by the measured factor, WRITE_SIZE counts whole 64 B lines.
Usage: python profiles/summarize_pmc.py [round-prefix, default r04] [csv directory under gpurun_out/, default r4g/pmc]   (tools/gpu_pass.sh TAG pmc writes gpurun_out/TAG/pmc)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PRE = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = sys.argv[2] if len(sys.argv) > 2 else "r4g/pmc"   # directory under gpurun_out/ holding the csv output
# workload -> (kernel name fragment, problems per full launch, algorithmic bytes per problem in / out)
WL = {"cfg2": ("crx_solve_kernel", 256, 39 * 8, 118 * 8), "cfg3": ("crx_solve_kernel", 4096, 57 * 8, 93 * 8),
      "cfg4": ("crx_solve_kernel", 16384, 162 * 8, 232 * 8), "cfg5": ("crx_solve_kernel", 65536, 57 * 8, 93 * 8)}


def rows(path, frag, grid):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if frag in r["Kernel_Name"] and int(r["Grid_Size"]) == grid:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))
    f_read = cal["big_read/FETCH_SIZE"]["factor"]            # bytes per counted byte, 8 B/lane coalesced stream far beyond the Infinity Cache
    f_write = cal["big_write/WRITE_SIZE"]["factor"]
    fixed_kib = cal["big_write/FETCH_SIZE"]["counter_kib"]   # FETCH_SIZE of a launch that reads nothing: kernarg + code of a tiny kernel
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only), bench.py --workload <wl>",
             "# counter unit KiB.  Calibration on a known byte count in this kernel's access pattern (one wave per record, 8 B per lane;",
             "# profiles/r02_pmc_calibration.txt): FETCH_SIZE x 1024 x %.3f = bytes read, WRITE_SIZE x 1024 x %.3f = bytes written." % (f_read, f_write),
             "# The algorithmic bytes are the per-problem inputs / outputs of SURVEY.md section 8d; everything fetched beyond them is",
             "# instruction fetch (the solver kernel is ~150-300 KB of code, fetched once per XCD L2) and partial cache lines."]
    js = {"note": "HBM bytes per full-batch launch of the solver kernel, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on a known "
                  "byte count in the kernel's own access pattern (profiles/r02_pmc_calibration.txt): traffic_bytes = FETCH*1024*%.3f + WRITE*1024*%.3f" % (f_read, f_write)}
    for wl, (frag, n, b_in, b_out) in WL.items():
        v = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(ROOT, "gpurun_out", SRC, "%s_%s" % (wl, ctr), "**", "*counter_collection.csv"), recursive=True)
            v[ctr] = rows(f[0], frag, n * 64)[ctr] if f else []
        F, W = v["FETCH_SIZE"], v["WRITE_SIZE"]
        if not F or not W:
            lines.append("%s: no data" % wl)
            continue
        fm, wm = sum(F) / len(F), sum(W) / len(W)
        rd, wr = fm * 1024 * f_read, wm * 1024 * f_write
        lines.append("%s: per launch of %d problems (mean of %d / %d launches): FETCH_SIZE %.1f KiB (min %.1f max %.1f) -> %.0f B read (algorithmic %d B: x%.2f); "
                     "WRITE_SIZE %.1f KiB -> %.0f B written (algorithmic %d B: x%.2f); total %.0f B = x%.2f the algorithmic %d B" % (
                         wl, n, len(F), len(W), fm, min(F), max(F), rd, b_in * n, rd / (b_in * n), wm, wr, b_out * n, wr / (b_out * n),
                         rd + wr, (rd + wr) / ((b_in + b_out) * n), (b_in + b_out) * n))
        js[wl] = {"batch": n, "fetch_kib": round(fm, 1), "write_kib": round(wm, 1), "traffic_bytes": round(rd + wr),
                  "traffic_bytes_uncalibrated": round((fm + wm) * 1024), "algorithmic_bytes": (b_in + b_out) * n}
    # the kernel sources these numbers were measured on (written by the GPU script next to the csv files): bench.py reports the
    # traffic only while the sources are unchanged
    try:
        js["kernel_source_sha256"] = open(os.path.join(ROOT, "gpurun_out", SRC, "source_hash.txt")).read().strip()
    except Exception:
        js["kernel_source_sha256"] = None
    lines.append("# kernel sources: sha256[:16] = %s (bench.py kernel_source_hash())" % js["kernel_source_sha256"])
    lines.append("# FETCH_SIZE of a launch that reads no data at all (tools/ubench/fetch_calib write8): %.1f KiB -- the floor every launch pays." % fixed_kib)
    open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % PRE), "w").write("\n".join(lines) + "\n")
    json.dump(js, open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
