"""CPU test of the host-side integer thresholds and seeding (bourse_amd/csrc/host_math.hpp)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_math_matches_the_f32_definitions_and_the_oracle_seeding(tmp_path, oracle):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_math_test")
    res = subprocess.run([gxx, "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "host_math_test.cpp"), "-o", exe],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe, "print-seed"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = run.stdout.strip().splitlines()
    assert lines[-1].startswith("host_math ok")
    s0, s1 = (int(x) for x in lines[0].split())
    # the oracle's seed_from_u64 (itself pinned by tests/test_oracle_rng.py): first outputs of the same state
    r = oracle.Rng(101)
    st = r.state() if hasattr(r, "state") else None
    if st is not None:
        assert (int(st[0]), int(st[1])) == (s0, s1)
    else:  # compare through the first draw: result = rotl(s0 * 5, 7) * 9
        M = (1 << 64) - 1
        x = (s0 * 5) & M
        x = ((x << 7) | (x >> 57)) & M
        assert r.next_u64() == (x * 9) & M
