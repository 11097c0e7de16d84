"""Registers, scratch and occupancy of the solver kernels, from the compiler's own remarks (no GPU needed):
    python tools/kernel_resources.py [obs|plan|gen|lmpc]     (recompiles the translation unit with -Rpass-analysis=kernel-resource-usage)"""
import os, re, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S = os.path.join(ROOT, "car-racing_amd", "csrc")
which = sys.argv[1] if len(sys.argv) > 1 else "obs"
src, sched = {"one": (None, None), "obs": ("crx_kernels_obs.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]),
              "plan": ("crx_kernels.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
              "gen": ("crx_kernels_gen.hip", ["-mllvm", "-disable-machine-licm"]),
              "lmpc": ("crx_lmpc.hip", ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"])}[which]
extra = sys.argv[2:]
if which == "one":   # python tools/kernel_resources.py one NOBS NMAX DEG NFIX [flags]
    tpl = ",".join(sys.argv[2:6]); extra = sys.argv[6:]
    src = "crx_kernels.hip"
    sched = (["-DCRX_TU_OBSTACLES", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if sys.argv[2] != "0" else
             ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]) + ["-DCRX_PROBE_ONE=" + tpl]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + sched + extra + [
    "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(S, src), "-o", "/dev/null"]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
def demangle(n):   # crx_solve_kernel<NOBS, NMAX, DEG, NFIX>: _Z16crx_solve_kernelILi1ELi12ELi6ELi12EEv10crx_kparams
    m = re.match(r"_Z\d+([a-z_]+)I((?:Li\d+E|Lb[01]E)+)E", n)
    return "%s<%s>" % (m.group(1), ",".join(re.findall(r"L[ib](\d+)E", m.group(2)))) if m else n
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print("%-44s vgpr %3s agpr %3s scratch %4s B/lane  waves/SIMD %s" % (demangle(b.split("\n")[0].split(" [")[0]), g("VGPRs"), g("AGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
