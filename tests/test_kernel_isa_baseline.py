"""A rebuild that changes what the compiler emits for a hand-tuned kernel is a RED TEST, not a slower BENCH (VERDICT r4 Weak #10 /
item 7).  The step kernels are issue-bound and tuned at the ISA level (assembly statements bound to fixed registers, scheduler
flags, a register claim, branch-free searches); tools/kernel_isa_counts.py disassembles the gfx950 code objects of the library
that the suite has just built and compares, per kernel, the instruction counts by issue class and the size of every loop (the
span of each backward branch: the keyed event loops' match blocks, the decode's window / walk / chase loops, k_agents_fsm's
66-instruction draw loop ...) with profiles/kernel_isa_baseline.json, within 3 %.
After an INTENDED kernel change: `python tools/kernel_isa_counts.py --update` and commit the baseline with it."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _tools_present():
    import kernel_isa_counts as K

    return all(K._tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def test_shipped_kernels_match_the_instruction_count_baseline():
    import kernel_isa_counts as K
    from bourse_amd import _build

    if not _tools_present():
        pytest.skip("no llvm-objdump / clang-offload-bundler")
    lib = _build.build()  # (rebuilds when a source is newer than the library: the counts are of THIS tree's code)
    base = json.load(open(K.BASELINE))
    now = K.measure(lib)
    bad = K.compare(base["kernels"], now, base["_tolerance"])
    assert not bad, ("kernel instruction counts moved by more than 3 % against profiles/kernel_isa_baseline.json "
                     f"(baseline toolchain: {base['_toolchain']}; now: {K.toolchain()}).  If the change is intended: "
                     "python tools/kernel_isa_counts.py --update.\n" + "\n".join(bad[:40]))
    # the hot loops this baseline exists for are really in it
    k = now["k_agents_fsm<2>"]
    assert 60 <= min(l for l in k["loops"] if l >= 60) <= 70, k["loops"]  # the draw loop: 66 instructions per draw
    assert now["k_step_batch<2, false, false>"]["counts"]["total"] > 3000  # carries the R = 2 assembly statements
    assert now["k_step_batch<8, false, false>"]["counts"]["total"] > 9000  # ... and the generated R = 8 ones


def test_compare_flags_a_grown_loop_and_a_grown_class():
    import kernel_isa_counts as K

    base = {"k": {"counts": {"total": 1000, "salu": 400, "branch": 100, "valu": 400, "lds": 50, "vmem": 50, "smem": 0, "other": 0},
                  "loops": [200, 66, 10]}}
    same = json.loads(json.dumps(base))
    assert K.compare(base, same) == []
    same["k"]["loops"][2] = 12  # +-2 instructions always pass
    same["k"]["counts"]["valu"] = 410  # 2.5 %
    assert K.compare(base, same) == []
    worse = json.loads(json.dumps(base))
    worse["k"]["loops"][1] = 73
    worse["k"]["counts"]["salu"] = 420
    assert K.compare(base, worse) == ["k: salu instructions 400 -> 420", "k: loop #1 (by size) 66 -> 73 instructions"]
    assert K.compare(base, {}) == ["k: kernel gone"]
