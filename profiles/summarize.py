"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel-trace database into the text summary that is committed under
profiles/.  Usage: python profiles/summarize.py <results.db> [<bench.json>]

Launches of one kernel are bucketed by grid size: a bench run issues the full-batch launches it times AND a few batch-1
launches (the "one control step with host arrays" latency probe); averaging the two would describe neither."""
import collections
import json
import sqlite3
import sys


def pct(d, q):
    return d[min(len(d) - 1, int(q * len(d)))]


def main():
    db = sqlite3.connect(sys.argv[1])
    print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
    print("%-48s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct_ in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-48s %8d %14.1f %12.2f %8.3f" % (name[:48], calls, total, avg, pct_))
    print()
    print("# per kernel and launch geometry: resources and duration distribution (us); grid_x = 64 x problems (one wave each)")
    print("%-44s %10s %6s %8s %6s %6s %6s | %9s %9s %9s %9s %9s" % ("kernel", "grid_x", "n", "lds_B", "scr", "vgpr", "sgpr", "min", "p50", "avg", "p90", "max"))
    rows = collections.defaultdict(list)
    meta = {}
    for name, gx, wg, lds, scr, vg, ag, sg, dur in db.execute(
            "select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count,duration from kernels order by id"):
        rows[(name, gx)].append(dur / 1e3)
        meta[(name, gx)] = (lds, scr, vg + ag, sg)
    for (name, gx), d in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        d.sort()
        lds, scr, vg, sg = meta[(name, gx)]
        print("%-44s %10d %6d %8d %6d %6d %6d | %9.1f %9.1f %9.1f %9.1f %9.1f" % (
            name[:44], gx, len(d), lds, scr, vg, sg, d[0], pct(d, 0.5), sum(d) / len(d), pct(d, 0.9), d[-1]))
    if len(sys.argv) > 2:
        print()
        print("# bench.py line of the same run (roofline.kernel_ms = HIP-event average over the full-batch launches)")
        line = open(sys.argv[2]).read().strip().splitlines()[-1]
        try:
            j = json.loads(line)
            keep = {k: j[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step") if k in j}
            keep["config"] = j.get("config")
            keep["roofline"] = j.get("roofline")
            print(json.dumps(keep))
        except Exception:
            print(line[:4000])


if __name__ == "__main__":
    main()
