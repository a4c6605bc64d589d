"""`python bench.py --gpus N` launches itself (CPU test of the spawn logic, gloo).

The driver runs `python bench.py --gpus N ...` directly for its 1/2/4/8-GPU scaling table: the parent must start the N
ranks as a CHILD `torch.distributed.run` job before it touches torch or a GPU, and by default shard BASELINE's 65 536
books over the ranks (strong scaling).  `--selftest-gloo` drives exactly that path on CPU: rendezvous, shard
arithmetic and the 64-byte stats all-gather, no stepping (there is no CPU execution path).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, timeout=240, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_line(out):
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in: " + out[-500:])


@pytest.mark.parametrize("world", [2, 3])
def test_self_spawn_strong_scaling_shards_the_workload(world):
    res = _run("--gpus", str(world), "--selftest-gloo")
    assert res.returncode == 0, res.stdout[-800:] + res.stderr[-800:]
    j = _json_line(res.stdout)
    assert j["n_gpus"] == world and j["ranks_seen"] == world and j["scaling"] == "strong"
    assert j["books_total"] == 65536  # BASELINE configs[3]: 65 536 books in total, not per GPU
    firsts = [65536 * r // world for r in range(world)]
    assert j["first_books_sum"] == sum(firsts)  # contiguous shards starting at total * r / world


def test_self_spawn_weak_scaling_is_opt_in():
    res = _run("--gpus", "2", "--selftest-gloo", "--scaling", "weak", "--books", "1000")
    assert res.returncode == 0, res.stdout[-800:] + res.stderr[-800:]
    j = _json_line(res.stdout)
    assert j["scaling"] == "weak" and j["books_total"] == 2000 and j["first_books_sum"] == 1000


def test_child_failure_is_propagated():
    """Without GPUs every rank exits with 'no GPU visible': the parent must return non-zero, not hang or mask it."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would really run")
    res = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert res.returncode != 0
    assert "no GPU visible" in res.stdout + res.stderr


def test_single_rank_does_not_spawn():
    """--gpus 1 runs in-process (no child): without a GPU it exits with the plain message."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    res = _run("--gpus", "1", "--steps", "1", "--warmup", "0")
    assert res.returncode != 0 and "no GPU visible" in res.stdout + res.stderr
    assert "torch.distributed.run" not in res.stderr


def test_a_rank_that_fails_before_the_rendezvous_ends_the_job_non_zero():
    """a rank whose initialisation fails (RCCL's on the real node; injected here) must not leave its peers
    waiting in init_process_group for ever: torch.distributed.run tears the job down, the parent returns its code."""
    import time

    t = time.time()
    res = _run("--gpus", "2", "--selftest-gloo", "--dist-timeout", "20", extra_env={"BOURSE_AMD_BENCH_TEST_FAULT": "exit:1"}, timeout=180)
    assert res.returncode != 0, res.stdout[-500:]
    assert "injected failure" in res.stdout + res.stderr
    assert time.time() - t < 150


def test_a_rank_that_hangs_is_ended_by_its_watchdog():
    """... and a rank that HANGS (a collective a peer never joins, a stream probe that never returns) dumps its stacks and exits
    non-zero after --rank-timeout; its peer's rendezvous times out after --dist-timeout.  Either ends the job."""
    import time

    t = time.time()
    res = _run("--gpus", "2", "--selftest-gloo", "--dist-timeout", "15", "--rank-timeout", "25",
               extra_env={"BOURSE_AMD_BENCH_TEST_FAULT": "hang:0"}, timeout=240)
    assert res.returncode != 0, res.stdout[-500:]
    assert time.time() - t < 200
