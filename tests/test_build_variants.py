"""The library also builds with the compiled C++ event loop (-DBOURSE_AMD_ASM_EVENTS=0) instead of the hand-written
gfx950 assembly of bourse_amd/csrc/event_asm.hpp, and exports the same C ABI.  (scripts/asm_ab.sh runs the GPU parity
suite and the bench on both builds; profiles/r02/asm_ab.txt.)  Same for the test build of the keyed event loop with an
8-bit arrival field (-DBOURSE_AMD_KEY_SEQ_BITS=8: ordinary runs leave the key's window and alternate between the keyed
loop and its fallback; scripts/keyed_variants.sh runs the GPU parity tests + a fuzz campaign on it)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compiled_event_loop_variant_builds_and_exports_the_abi(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    from bourse_amd import _build, _lib

    out = str(tmp_path / "libbourse_amd_cxx.so")
    assert _build.build(out=out, defines=["BOURSE_AMD_ASM_EVENTS=0"]) == out and os.path.getsize(out) > 100_000
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", "--defined-only", out], capture_output=True, text=True, check=True).stdout
    for name in _lib.SIGNATURES:
        assert f" {name}" in syms, name
    # the shipped build carries the assembly loop: its device code contains the statement's labels' instructions
    # (checked indirectly: both libraries exist and differ)
    assert open(out, "rb").read() != open(_build.LIB, "rb").read()


def test_keyed_loop_test_variant_builds(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    from bourse_amd import _build

    out = str(tmp_path / "libbourse_amd_sb8.so")
    remarks = []
    assert _build.build(out=out, defines=["BOURSE_AMD_KEY_SEQ_BITS=8"], remarks=remarks) == out and os.path.getsize(out) > 100_000
    assert open(out, "rb").read() != open(_build.LIB, "rb").read()
    # The register budget the split pipeline's throughput rests on (docs/EXPERIMENTS.md): the lane-per-book
    # agents kernel claims 232 VGPRs so that SEVEN event waves of <= 40 VGPRs fit beside one of its waves on a SIMD
    # (232 + 7 x 40 = 512).  An event kernel of 46 VGPRs (five waves) once cost C3 a quarter of its rate.
    def vgprs(mangled):
        import re

        m = re.search(r"Function Name: " + re.escape(mangled) + r"\b.*?VGPRs: (\d+)", remarks[0], re.S)
        assert m, mangled
        return int(m.group(1))

    # (an inequality, not the compiler's exact allocation: VGPRs are handed out in blocks of 8)
    def alloc(n):
        return (n + 7) // 8 * 8

    fsm = vgprs("_ZN3bkd12k_agents_fsmILi2EEEvNS_7DevArgsE")
    assert 200 <= fsm <= 264  # the measured plateau of its deliberate footprint (profiles/r02/fsm_vgpr_sweep.txt)
    for r in (1, 2):
        ev = vgprs(f"_ZN3bkd12k_step_batchILi{r}ELb0ELb0EEEvNS_7DevArgsEmj")
        assert alloc(fsm) + 7 * alloc(ev) <= 512, (r, fsm, ev)
