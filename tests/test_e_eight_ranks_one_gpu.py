"""First contact with an 8-GPU box, rehearsed on the one GPU there is: `bench.py --gpus 8 --dry-ranks`
starts EIGHT rank processes through torch.distributed.run exactly as the driver's command line does, every rank on GPU 0,
the collectives over gloo (RCCL refuses several ranks on one device).  Everything except RCCL-over-xGMI executes: the
rendezvous, eight concurrent part-stream probes, eight envs of 8 192 books with their book_offset, bk_warm, the region /
barrier logic, the per-launch stats all-gather and the books_total / ranks_seen consistency checks of the bench line.

The children are started BEFORE this pytest process touches the GPU (the file sorts ahead of test_gpu_parity.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_eight_ranks_dry_run_on_one_gpu():
    env = dict(os.environ, BOURSE_AMD_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-ranks", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--repeats", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_ranks"] is True and out["n_gpus"] == 8 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["books_total"] == 65536 and cfg["books_per_gpu"] == 8192 and cfg["ranks"] == 8
    assert cfg["stats_allgather"]["ranks_seen"] == 8 and cfg["stats_allgather"]["n_books"] == 65536
    assert cfg["stats_allgather"]["sum_trades"] > 0
    assert cfg["pipeline"].startswith("wave_split")  # the auto rule at the C4 shard size
    rr = out["rank_region_ms"]
    assert len(rr["all"]) == 8 and 0 < rr["min"] <= rr["median"] <= rr["max"]
    assert out["value"] > 0 and out["steps"] == 20
    # every rank probed its part streams (eight probes at the same time on one device)
    assert r.stderr.count("candidate streams on hardware queues of their own") + r.stderr.count("part streams taken unprobed") >= 8, r.stderr[-3000:]
