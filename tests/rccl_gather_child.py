"""Child process of tests/test_d_rccl_gather.py: torch FIRST (so that the process uses one HIP runtime, as bench.py
does), then the library; one-rank RCCL group; the sharded run's exchange checked against the host-side readers."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd as bk  # noqa: E402
from bourse_amd import parallel  # noqa: E402

C2_GROUPS = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
try:
    env = bk.ManyBookEnv(300, 11, 0, 2, 100_000, levels=16, max_live_orders=64, trade_capacity=4096, history_capacity=8,
                         stream=torch.cuda.current_stream().cuda_stream)
    env.set_random_agents(C2_GROUPS)
    gather, l1 = parallel.StatsGather(env, dist), parallel.L1Gather(env, dist)
    assert gather.zero_copy
    for _ in range(3):
        env.run(5, sync=False)
        gather.all_gather()
        l1.all_gather()
    torch.cuda.synchronize()
    got, want = gather.result(), env.stats()
    for k in parallel.STATS_FIELDS + ("min_bid", "max_bid", "min_ask", "max_ask"):
        assert got[k] == want[k], k
    assert got["sum_trades"] == int(env.trade_counts().sum()) > 0
    assert np.array_equal(l1.result(), env.level2()[:, :9])
    env.close()
finally:
    dist.destroy_process_group()
print("rccl gather ok")
