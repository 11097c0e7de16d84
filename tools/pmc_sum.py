"""Mean per-dispatch PMC counter values of the solver kernels from rocprofv3 csv output: python tools/pmc_sum.py DIR [grid]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
grid = int(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "crx_solve_kernel" not in r["Kernel_Name"] and "crx_lmpc_kernel" not in r["Kernel_Name"]:
            continue
        if grid and int(r["Grid_Size"]) != grid:
            continue
        acc[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-42s grid %-8s %-28s mean %.4g (n=%d)" % (k[0], k[1], k[2], sum(v) / len(v), len(v)))
