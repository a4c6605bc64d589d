"""The sharded run's exchange on real hardware with one rank: zero-copy torch views of the library's device records
(bk_stats_device_ptr / bk_level2_device_ptr) all-gathered over RCCL equal the host-side readers.

Runs in a child process started BEFORE this pytest process has touched the GPU (this file sorts ahead of
test_gpu_parity.py): torch must be imported before the library so that the process uses a single HIP runtime, and a
process that has initialised the GPU must not spawn programs on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_stats_and_l1_gather_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_gather_child.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
