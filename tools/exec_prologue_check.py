"""Guard against an LLVM (ROCm 7.2) code-generation defect, run on the code objects that SHIP -- not on a recompilation.

LLVM places the EXEC restore of a divergent region / loop exit (`s_or_b64 exec, exec, sN`) at the top of the join block.  A VGPR spill, reload or
copy that the register allocator inserts IN FRONT of it runs with the EXEC mask the region ended with.  Round 5: a
`scratch_store ... ; Folded Spill` in front of the restore at the exit of a loop whose lanes drop out one by one -- EXEC = 0, nothing stored, the
reloads returned what the scratch slot held from other kernels: wrong trajectories for 78 of 256 problems in crx_solve_kernel<3,24,6,0>.

What is checked: every gfx950 code object embedded in the given files (default: the in-tree libcrx.so, i.e. whatever flags, EXTRA= options or ROCm
install built it) is extracted (`llvm-objdump --offloading`), disassembled (`-d --symbolize-operands`) and scanned block by block.  A FINDING is a
register-allocator artefact -- spill store / reload (scratch_*, buffer_*), AGPR copy (v_accvgpr_write / _read), plain register-to-register v_mov -- between a block's label and that block's EXEC restore.  Each finding is classified by the mask it can run under:
  FATAL      some edge reaches the block with EXEC = 0: the block is the target of an `s_cbranch_execz`, or falls through from an
             `s_cbranch_execnz` (the exit of a divergent loop).  Nothing is stored / loaded / copied on that edge.  Exit status 1.
  narrowing  the block is only reached with a non-empty, narrowed mask (an if-region without a skip branch): the lanes outside keep stale data, which
             is harmless exactly when the value is dead outside the region.  Listed, exit status 0 (`--strict`: 1).
    python tools/exec_prologue_check.py [--strict] [--list] [FILE.so | FILE.o ...]
`make` runs it on libcrx.so right after linking (car-racing_amd/csrc/Makefile: a FATAL finding fails the build); tests/test_abi_cpu.py runs it on the
shipped library (skipped when the ROCm binutils are absent).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")

SPILL = re.compile(r"^(scratch_store|buffer_store|scratch_load|buffer_load|v_accvgpr_write|v_accvgpr_read|v_accvgpr_mov)")
COPY = re.compile(r"^v_mov_b(32|64)_e32 v(\d+|\[\d+:\d+\]), v(\d+|\[\d+:\d+\])$")     # a plain register-to-register copy (no DPP, no constant)
# the join of a divergent region / the exit of a divergent loop.  (`s_mov_b64 exec, sN` is NOT one: it is how a region is ENTERED when the mask was
# formed by s_and_b64 -- what precedes it runs under the block's own mask -- and how a whole-wave section ends, where every lane was on.  Round 5's
# version of this tool counted it and reported "milder forms" in three builds that were correct.)
RESTORE = re.compile(r"^s_or_b64 exec, exec, ")
ENDS_PROLOGUE = re.compile(r"^s_(c?branch|endpgm|and_saveexec|or_saveexec|andn2_saveexec|andn2_b64 exec|and_b64 exec|setpc|barrier)")
VECTOR = re.compile(r"^(v_|ds_|scratch_|global_|buffer_|flat_)")
HARMLESS = re.compile(r"^v_(readlane|writelane|readfirstlane)_b32")   # SGPR spills into VGPR lanes: EXEC-independent


def code_objects(path, tmp):
    """Extract the gfx950 code objects embedded in an x86 object / shared library (or take a bare code object as it is)."""
    base = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, base)
    out = subprocess.run([OBJDUMP, "--offloading", base], capture_output=True, text=True, cwd=tmp)
    objs = sorted(f for f in os.listdir(tmp) if f.startswith(os.path.basename(path) + ".") and "amdgcn" in f)
    return [os.path.join(tmp, f) for f in objs] or ([base] if "amdgpu" in out.stdout.lower() else [])


def scan(code_object):
    """-> list of (kernel, label, address, instruction, restore, fatal: bool)"""
    return scan_text(subprocess.run([OBJDUMP, "-d", "--symbolize-operands", code_object], capture_output=True, text=True, check=True).stdout)


def scan_text(txt):
    """The scan itself, on `llvm-objdump -d --symbolize-operands` text (tests/test_abi_cpu.py feeds it hand-made blocks)."""
    lines = txt.split("\n")
    # pass 1: per function, the labels that an `s_cbranch_execz` targets, and the labels that follow an `s_cbranch_execnz` (loop exits)
    func, prev, zero_edge = "?", "", set()
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            name = m.group(1)
            if re.match(r"L\d+$", name):
                if prev.startswith("s_cbranch_execnz"):
                    zero_edge.add((func, name))
            else:
                func = name
            continue
        t = l.split("//")[0].strip()
        if not t:
            continue
        mz = re.match(r"s_cbranch_execz (L\d+)", t)
        if mz:
            zero_edge.add((func, mz.group(1)))
        prev = t
    # pass 2: artefacts in front of a block's EXEC restore
    found, func, label, pending = [], "?", None, None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            name = m.group(1)
            if re.match(r"L\d+$", name):
                label, pending = name, []
            else:
                func, label, pending = name, None, None
            continue
        t = l.split("//")[0].strip()
        if not t or pending is None:
            continue
        addr = (l.split("//")[1].split(":")[0].strip() if "//" in l else "?")
        if RESTORE.match(t):
            for (a, ins) in pending:
                found.append((func, label, a, ins, t, (func, label) in zero_edge))
            pending = None
        elif ENDS_PROLOGUE.match(t):
            pending = None
        elif SPILL.match(t) or COPY.match(t):
            pending.append((addr, t))
        elif VECTOR.match(t) and not HARMLESS.match(t):
            if not pending:
                pending = None        # ordinary code of the region comes first: a restore further down closes a region this block belongs to
            # (ordinary code AFTER an artefact that is already in front: keep looking for the restore)
    return found


def demangle(n):
    m = re.match(r"_Z\d+([a-z_0-9]+)I((?:Li\d+E|Lb[01]E)+)E", n)
    return "%s<%s>" % (m.group(1), ",".join(re.findall(r"L[ib](\d+)E", m.group(2)))) if m else n[:60]


def main(argv):
    strict = "--strict" in argv
    files = [a for a in argv if not a.startswith("--")] or [os.path.join(ROOT, "car-racing_amd", "crx", "libcrx.so")]
    if not os.path.exists(OBJDUMP):
        print("exec_prologue_check: %s not found -- nothing checked" % OBJDUMP)
        return 2
    fatal = narrowing = n_obj = 0
    for f in files:
        tmp = tempfile.mkdtemp(prefix="crx_epc_")
        try:
            for co in code_objects(f, tmp):
                n_obj += 1
                for (func, label, addr, ins, restore, is_fatal) in scan(co):
                    fatal += is_fatal
                    narrowing += not is_fatal
                    print("%-9s %s %s @%s: `%s` in front of `%s`" % ("FATAL" if is_fatal else "narrowing", demangle(func), label, addr, ins, restore))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print("exec_prologue_check: %d code object(s) of %s: %d FATAL (EXEC = 0 on an incoming edge), %d narrowing" % (
        n_obj, ", ".join(os.path.basename(f) for f in files), fatal, narrowing))
    if n_obj == 0:
        print("exec_prologue_check: no gfx code object found in the input")
        return 2
    return 1 if (fatal or (strict and narrowing)) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
