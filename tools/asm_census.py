"""Static instruction census of one solver instantiation, phase by phase (no GPU needed): the translation unit is compiled to assembly with the
phase clocks of `make TRACE=1` in, and the instructions between successive clock reads are counted by class.  A lone solver wave issues one
instruction per ~4.4 clocks whatever its class, so these counts ARE the cost model of the iteration.
    python tools/asm_census.py NOBS NMAX DEG NFIX [extra hipcc flags]     e.g.  python tools/asm_census.py 1 12 6 12 -DCRX_STATIC_LDS=0
Loop bodies are counted once (the Riccati stage loop = the three `ric:` lines, executed N times per factorisation)."""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
src = os.path.join(ROOT, "car-racing_amd", "csrc", "crx_kernels.hip")
tpl = ",".join(sys.argv[1:5]); extra = sys.argv[5:]
sched = (["-DCRX_TU_OBSTACLES", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if sys.argv[1] != "0" else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])
def build(flags):
    out = tempfile.mktemp(suffix=".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-disable-machine-licm"] + sched +
                   ["-DCRX_PROBE_ONE=" + tpl] + flags + extra + ["--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    t = open(out).read().split("\n"); os.unlink(out)
    return t
def isins(x):
    s = x.strip()
    return x.startswith("\t") and s and not s.startswith(".") and not s.startswith(";")
def classes(seg):
    c = Counter(l.split()[0] for l in seg if isins(l))
    g = lambda f: sum(v for k, v in c.items() if f(k))
    return (sum(c.values()), g(lambda k: "f64" in k and k.startswith("v_")), g(lambda k: k.startswith("ds_read")), g(lambda k: k.startswith("ds_write")), c.get("v_readlane_b32", 0),
            g(lambda k: "cndmask" in k), g(lambda k: k.startswith("v_mov")), g(lambda k: k.startswith("v_") and ("_u32" in k or "_i32" in k or "_b32" in k) and "cndmask" not in k and "mov" not in k and "readlane" not in k),
            c.get("s_waitcnt", 0), g(lambda k: k.startswith("s_") and k != "s_waitcnt"), g(lambda k: k.startswith("v_accvgpr")))
prod = build([])
tot = classes(prod)
print("crx_solve_kernel<%s>  production build: %d instructions (static), %s" % (tpl, tot[0], " ".join(l.strip() for l in prod if "vgpr_count" in l or "group_segment_fixed_size:" in l)))
tr = build(["-DCRX_PHASE_CLOCKS"])
marks = [i for i, l in enumerate(tr) if "s_memtime" in l]
names = ["(loop top)", "adjoint (KKT error)", "barrier update", "assemble", "ric: terminal + lane maps", "ric: stage-invariant operands", "ric: set-up tail", "ric: T = P M   [per stage]", "ric: H = M'T + ..   [per stage]",
         "ric: factor + update  [per stage]", "ric: sigma_0 / retry logic", "forward sweep (loop body once)", "row steps", "line search (one trial)", "accept + first order"]
print("%-36s %6s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s" % ("phase (between clock reads)", "instr", "f64", "ds_r", "ds_w", "rdln", "cndm", "vmov", "vint", "wait", "salu", "agpr"))
def loops(a, b):   # bodies of the loops that lie inside [a, b): (label, instructions), innermost first
    lab = {m.group(1): i for i, l in enumerate(tr[a:b], a) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    out = {}
    for i, l in enumerate(tr[a:b], a):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            out[m.group(1)] = max(out.get(m.group(1), 0), sum(isins(x) for x in tr[lab[m.group(1)]:i + 1]))
    return sorted(out.items(), key=lambda kv: kv[1])
for n, (a, b) in enumerate(zip(marks[:-1], marks[1:])):
    lp = loops(a, b)
    print("%-36s %6d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d" % (((names[n] if n < len(names) else "?"),) + classes(tr[a + 1:b])), ("  loops: " + ", ".join("%d" % v for _, v in lp)) if lp else "")
