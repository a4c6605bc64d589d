"""The N > 1 path with REAL stepping on the one GPU the test box has: two rank processes (gloo between them, both on
GPU 0), each stepping its contiguous shard of one job through `ManyBookEnv(book_offset=...)` - concurrently, each with
its own part-stream probe - and checking it against the oracle seeded by global book index; the 64-byte records are
all-gathered and must describe the whole job.  (The 8-GPU RCCL run itself is the driver's; this is the closest thing one
GPU allows: VERDICT r2 "N > 1 has never executed on hardware".)

Children are started BEFORE this pytest process touches the GPU (the file sorts ahead of test_gpu_parity.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_and_shard_one_job():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BOURSE_AMD_VERBOSE="1")
    # 16 384 books in total: 8 192 per rank = the C4 shard size (wave_split, three parts: the parts' streams are probed)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_ranks_child.py"), str(r), "2", "16384", "8", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}/2 ok" in out, out[-3000:]
