"""CPU test of the host thread pool behind the host-driven path (bourse_amd/csrc/host_pool.hpp), under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitize", ["", "-fsanitize=thread"])
def test_host_pool(tmp_path, sanitize):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_pool_test")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-pthread"] + ([sanitize] if sanitize else []) + \
          [os.path.join(ROOT, "tests", "cpp", "host_pool_test.cpp"), "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and sanitize:
        pytest.skip("ThreadSanitizer runtime not available: " + res.stderr[-200:])
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host_pool ok" in run.stdout
