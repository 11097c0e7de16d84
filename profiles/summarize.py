"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel-trace database into the text summary that
is committed under profiles/.  Usage: python profiles/summarize.py <results.db> [<bench.json>]"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
    print("%-48s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-48s %8d %14.1f %12.2f %8.3f" % (name[:48], calls, total, avg, pct))
    print()
    print("# per-kernel launch geometry / resources (first dispatch of each kernel)")
    print("%-48s %10s %6s %9s %8s %6s %6s" % ("kernel", "grid_x", "wg_x", "lds_B", "scratch", "vgpr", "sgpr"))
    seen = set()
    for r in db.execute("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels order by id"):
        if r[0] in seen:
            continue
        seen.add(r[0])
        print("%-48s %10d %6d %9d %8d %6d %6d" % (r[0][:48], r[1], r[2], r[3], r[4], r[5] + r[6], r[7]))
    print()
    print("# duration distribution of the solver kernel (us)")
    for (name,) in db.execute("select distinct name from kernels where name like '%crx_solve%'"):
        d = sorted(x[0] / 1e3 for x in db.execute("select duration from kernels where name=?", (name,)))
        n = len(d)
        print("%-48s n=%d min=%.1f p50=%.1f p90=%.1f max=%.1f" % (name[:48], n, d[0], d[n // 2], d[int(n * 0.9)], d[-1]))
    if len(sys.argv) > 2:
        print()
        print("# bench.py line of the same run")
        print(open(sys.argv[2]).read().strip())


if __name__ == "__main__":
    main()
