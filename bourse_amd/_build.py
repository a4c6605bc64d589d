"""Build the HIP extension in-tree: bourse_amd/csrc/libbourse_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so
travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libbourse_amd.so")
SOURCES = ["bourse_amd.hip", "book_device.hpp", "event_asm.hpp", "wave_agents.hpp", "mixed_agents.hpp", "wave_mixed.hpp", "pm_math.hpp",
           "host_pool.hpp", "host_math.hpp",
           os.path.join("..", "..", "include", "bourse_amd.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: bourse_amd needs the ROCm toolchain to build its HIP extension")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = False, out: str = None, defines=()) -> str:
    """Build libbourse_amd.so in-tree; `out` + `defines` build a VARIANT somewhere else (e.g. -DBOURSE_AMD_ASM_EVENTS=0:
    the compiled C++ event loop instead of the hand-written one) that BOURSE_AMD_LIBRARY=<path> makes _lib load."""
    if out is None and not force and not is_stale():
        return LIB
    # The step kernels are dominated by wave-UNIFORM control flow (scalar branches).  By default LLVM's StructurizeCFG
    # pass also rewrites uniform regions, which costs ~9 % extra scalar instructions (flag registers + s_andn2/vccnz
    # branches) on the SALU-bound k_step_batch: skip it for uniform regions (+7 % book-steps/s, parity tests green).
    extra = os.environ.get("BOURSE_AMD_HIPCC_FLAGS", "-mllvm -structurizecfg-skip-uniform-regions=1").split()
    target = out or LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra + [
        "-D" + d for d in defines] + ["-o", target, os.path.join(CSRC, "bourse_amd.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
