"""profiles/<round>_pmc_calibration.txt + profiles/pmc_calibration.json from the rocprofv3 passes of tools/gpu_calib.sh:
FETCH_SIZE / WRITE_SIZE (KiB) against the known byte counts of tools/ubench/fetch_calib.hip."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PRE = sys.argv[1] if len(sys.argv) > 1 else "r02"
O = os.path.join(ROOT, "gpurun_out", "calib")


def counter(tag, ctr, kernel):
    f = glob.glob(os.path.join(O, "%s_%s" % (tag, ctr), "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kernel in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    return v


def known(tag, ctr):
    m = re.search(r"bytes_per_launch=(\d+)", open(os.path.join(O, "%s_%s.log" % (tag, ctr))).read())
    rec = re.search(r"records=(\d+)", open(os.path.join(O, "%s_%s.log" % (tag, ctr))).read())
    return int(m.group(1)), int(rec.group(1))


def main():
    lines = ["# FETCH_SIZE / WRITE_SIZE calibration, tools/ubench/fetch_calib.hip (one wave per record, 8 B per lane, the access",
             "# pattern of crx_solve_kernel); rocprofv3 --pmc <counter> --kernel-trace, separate passes; counter unit KiB.",
             "# factor = known bytes / (counter * 1024): multiply a measured counter by it to get bytes in this pattern."]
    js = {}
    for tag, ctr, kernel in (("big_read", "FETCH_SIZE", "read8"), ("big_read", "WRITE_SIZE", "read8"), ("big_write", "WRITE_SIZE", "write8"),
                             ("big_write", "FETCH_SIZE", "write8"), ("cfg2_read", "FETCH_SIZE", "read8"), ("cfg2_write", "WRITE_SIZE", "write8"),
                             ("cfg3_read", "FETCH_SIZE", "read8"), ("cfg3_write", "WRITE_SIZE", "write8")):
        v = counter(tag, ctr, kernel)
        if not v:
            lines.append("%s %s: no data" % (tag, ctr))
            continue
        nbytes, rec = known(tag, ctr)
        if (kernel == "read8") != (ctr == "FETCH_SIZE"):        # the minor direction: read8 writes 8 B per record, write8 reads nothing
            nbytes = rec * 8 if kernel == "read8" else 0
        last = v[-1]                                                # later launches: code and kernarg already in cache
        fac = nbytes / (last * 1024) if last else float("nan")
        lines.append("%-10s %-10s launches=%d  counter(KiB) first=%.1f last=%.1f  known bytes=%d  factor(last)=%.3f" % (tag, ctr, len(v), v[0], last, nbytes, fac))
        js["%s/%s" % (tag, ctr)] = dict(counter_kib=last, counter_kib_first=v[0], known_bytes=nbytes, factor=fac)
    open(os.path.join(ROOT, "profiles", "%s_pmc_calibration.txt" % PRE), "w").write("\n".join(lines) + "\n")
    json.dump(js, open(os.path.join(ROOT, "profiles", "pmc_calibration.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
