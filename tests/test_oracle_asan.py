"""The CPU oracle's main workloads under AddressSanitizer + UBSan (GPU ASan is unavailable on this pool; the oracle is
the checker of every parity claim, so its own memory safety is checked here)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_workloads_under_asan_ubsan(tmp_path):
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    if not gxx or not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    so = str(tmp_path / "libbourse_oracle_asan.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("bourse_oracle.cpp", "bourse_oracle_agents.cpp", "bourse_oracle_capi.cpp")]
    res = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fsanitize=address,undefined",
                          "-fno-sanitize-recover=undefined", "-pthread", "-shared", "-o", so] + srcs, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", BOURSE_ORACLE_ASAN_LIB=so)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_oracle.py")], capture_output=True, text=True,
                         timeout=900, env=env)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "json ok" in run.stdout and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_oracle_threaded_runners_under_tsan(tmp_path):
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    if not gxx or not gcc:
        pytest.skip("no gcc")
    libtsan = subprocess.run([gcc, "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libtsan) or not os.path.exists(libtsan):
        pytest.skip("libtsan not installed")
    so = str(tmp_path / "libbourse_oracle_tsan.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("bourse_oracle.cpp", "bourse_oracle_agents.cpp", "bourse_oracle_capi.cpp")]
    res = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fsanitize=thread", "-pthread",
                          "-shared", "-o", so] + srcs, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=libtsan, TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0", BOURSE_ORACLE_TSAN_LIB=so)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tsan_oracle.py")], capture_output=True, text=True,
                         timeout=900, env=env)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this environment")
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "markets ok" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr
