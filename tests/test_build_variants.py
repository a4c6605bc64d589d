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
    assert _build.build(out=out, defines=["BOURSE_AMD_KEY_SEQ_BITS=8"]) == out and os.path.getsize(out) > 100_000
    assert open(out, "rb").read() != open(_build.LIB, "rb").read()
