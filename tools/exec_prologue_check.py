"""LLVM places the EXEC restore of a divergent region / loop exit (s_or_b64 exec, exec, sN) at the top of the join block; a VGPR spill or copy that the
register allocator inserts BEFORE it runs with the EXEC mask the region ended with -- zero at the exit of a loop whose lanes drop out one by one -- and
stores nothing (round 5: `scratch_store ... ; Folded Spill` in front of the restore in crx_solve_kernel<3,24,6,0>; the reloads returned whatever the
scratch slot held from other kernels: DESIGN.md section 8).  This script compiles a translation unit to assembly and lists every vector instruction
(scratch / buffer spill STORE, copy into an AGPR) that sits at the very top of a block, in front of that block's EXEC restore.
    python tools/exec_prologue_check.py [plan|obs|gen|lmpc|prep|lmpcprep] [extra hipcc flags]          exit status 1 if anything is found"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S = os.path.join(ROOT, "car-racing_amd", "csrc")
which = sys.argv[1] if len(sys.argv) > 1 else "gen"
src, fl = {"plan": ("crx_kernels.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
           "obs": ("crx_kernels_obs.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]),
           "gen": ("crx_kernels_gen.hip", ["-mllvm", "-disable-machine-licm"]),
           "lmpc": ("crx_lmpc.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]), "prep": ("crx_prep.hip", []), "lmpcprep": ("crx_lmpcprep.hip", [])}[which]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + fl + sys.argv[2:] +
               ["--cuda-device-only", "-S", os.path.join(S, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
L = open(out).read().split("\n"); os.unlink(out)
VEC = re.compile(r"^(v_|ds_|scratch_|global_|buffer_|flat_)")
kernel, block, pending, found = "?", None, [], 0
for i, l in enumerate(L):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel = m.group(1); block = None; pending = []; continue
    if re.match(r"^\.LBB\d+_\d+:", l):
        block = l.split(":")[0]; pending = []; continue
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        continue
    if pending is None:
        continue
    if re.match(r"s_or_b64 exec, exec, ", t) or re.match(r"s_mov_b64 exec, ", t):
        for (ln, ins) in pending:
            found += 1
            print("%s %s line %d: `%s` in front of `%s`" % (kernel[:60], block, ln, ins, t))
        pending = None          # only the block's prologue matters
        continue
    if re.match(r"s_(c?branch|endpgm|and_saveexec|andn2_b64 exec|setpc)", t):
        pending = None
        continue
    if re.match(r"(scratch_store|buffer_store|v_accvgpr_write)", t):
        pending.append((i + 1, l.strip()[:90]))      # a spill store / a copy into an AGPR at the very top of the block
    elif re.match(r"(scratch_load|buffer_load|v_accvgpr_read)", t):
        pass                                          # (a reload there is what a then-block that needs the value starts with: not the pattern)
    elif VEC.match(t) and not re.match(r"v_(readlane|writelane)_b32", t):
        if not pending:
            pending = None      # ordinary code of the region comes first: the restore further down closes a region this block belongs to
        # (ordinary code AFTER a spill that is already in front: keep looking for the restore)
print("%s: %d vector instruction(s) in front of an EXEC restore" % (src, found))
sys.exit(1 if found else 0)
