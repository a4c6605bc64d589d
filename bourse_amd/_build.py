"""Build the HIP extension in-tree: bourse_amd/csrc/libbourse_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so
travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libbourse_amd.so")
SOURCES = ["bourse_amd.hip", "fsm_unit.hip", "book_device.hpp", "event_asm.hpp", "event_asm_gen.hpp", "wave_agents.hpp", "mixed_agents.hpp", "wave_mixed.hpp", "step_events.hpp", "pm_math.hpp",
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


FSM_SCHED = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]  # fsm_unit.hip only (see its header)


def build(force: bool = False, verbose: bool = False, out: str = None, defines=(), remarks: list = None) -> str:
    """Build libbourse_amd.so in-tree; `out` + `defines` build a VARIANT somewhere else (e.g. -DBOURSE_AMD_ASM_EVENTS=0:
    the compiled C++ event loop instead of the hand-written one) that BOURSE_AMD_LIBRARY=<path> makes _lib load.
    Two translation units: bourse_amd.hip (everything) and fsm_unit.hip (k_agents_fsm under the max-ilp scheduler)."""
    if out is None and not force and not is_stale():
        return LIB
    # The step kernels are dominated by wave-UNIFORM control flow (scalar branches).  By default LLVM's StructurizeCFG
    # pass also rewrites uniform regions, which costs ~9 % extra scalar instructions (flag registers + s_andn2/vccnz
    # branches) on the SALU-bound k_step_batch: skip it for uniform regions (+7 % book-steps/s, parity tests green).
    extra = os.environ.get("BOURSE_AMD_HIPCC_FLAGS", "-mllvm -structurizecfg-skip-uniform-regions=1").split()
    fsm_extra = os.environ.get("BOURSE_AMD_FSM_HIPCC_FLAGS", " ".join(FSM_SCHED)).split()
    target = out or LIB
    base = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra + ["-D" + d for d in defines]
    if verbose or remarks is not None:  # (`remarks`: a list that receives the compiler's kernel-resource-usage report)
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    objs = []
    log = ""
    import concurrent.futures
    import tempfile

    with tempfile.TemporaryDirectory(prefix="bourse_amd_build_") as tmp:
        jobs = [(base + ["-c", "-o", os.path.join(tmp, "bourse_amd.o"), os.path.join(CSRC, "bourse_amd.hip")]),
                (base + fsm_extra + ["-c", "-o", os.path.join(tmp, "fsm_unit.o"), os.path.join(CSRC, "fsm_unit.hip")])]
        with concurrent.futures.ThreadPoolExecutor(2) as ex:
            results = list(ex.map(lambda c: subprocess.run(c, cwd=CSRC, capture_output=True, text=True), jobs))
        for cmd, res in zip(jobs, results):
            if res.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
            log += res.stderr
            objs.append(cmd[cmd.index("-o") + 1])
        res = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs,
                             cwd=CSRC, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    if remarks is not None:
        remarks.append(log)
    if verbose:
        print(log)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
