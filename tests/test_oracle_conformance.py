"""Build-container-only pin: the reference's OWN Python test-suite passes against the oracle.

Skipped wherever /root/reference is absent (e.g. the GPU box): nothing here reads it otherwise."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BOURSE_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference not present on this machine")
def test_reference_python_tests_pass_on_oracle():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_pytests.py")], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "18 passed" in r.stdout
