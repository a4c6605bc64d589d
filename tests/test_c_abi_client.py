"""A plain-C11 client of include/bourse_amd.h (tests/c/abi_smoke.c): the boundary really is a C ABI — no C++, no
Python, no torch types.  CPU: it compiles, links and reports BK_NO_DEVICE (exit 77); GPU: it runs the reference's
three-step KAT, the error paths and an on-device run on both pipelines (exit 0)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import bourse_amd._build as b

    b.build()
    lib_dir = os.path.join(ROOT, "bourse_amd", "csrc")
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", lib_dir, "-lbourse_amd",
                    f"-Wl,-rpath,{lib_dir}", "-o", exe], check=True, capture_output=True)
    return exe


def test_c_client_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 77 and "no CPU execution path" in r.stdout


@pytest.mark.gpu
def test_c_client_on_gpu(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke: ok" in r.stdout


def _build_cpp(tmp_path):
    """tests/cpp/env_mirror.cpp: include/bourse_amd.hpp (C++ mirror of Env / Agent / sim_runner) + the oracle's C++ classes."""
    import bourse_amd._build as b

    b.build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libbourse_oracle.so"], check=True, capture_output=True)
    lib_dir, orc_dir = os.path.join(ROOT, "bourse_amd", "csrc"), os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "env_mirror")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", orc_dir,
                    os.path.join(ROOT, "tests", "cpp", "env_mirror.cpp"), "-L", lib_dir, "-lbourse_amd", "-L", orc_dir,
                    "-lbourse_oracle", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{orc_dir}", "-o", exe],
                   check=True, capture_output=True)
    return exe


def test_cpp_env_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([_build_cpp(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 77 and "no CPU execution path" in r.stdout


@pytest.mark.gpu
def test_cpp_env_mirror_user_agents_match_oracle_on_gpu(tmp_path):
    r = subprocess.run([_build_cpp(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "env_mirror: ok" in r.stdout
