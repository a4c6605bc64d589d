"""Child of tests/test_e_two_ranks_one_gpu.py: ONE RANK of a two-rank sharded run in which both ranks share GPU 0.

There is no second GPU on the test box and RCCL refuses two ranks on one device, so the ranks talk over gloo (the
64-byte records travel as host tensors) - everything else is the real thing: two processes, each with its own HIP
context, part-stream probe (both probe at the same time, as eight ranks would at start-up) and shard
`ManyBookEnv(book_offset=...)`, stepping concurrently on the hardware.  torch is imported first (one HIP runtime)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bourse_amd as bk  # noqa: E402
import pyoracle  # noqa: E402
from bourse_amd import parallel  # noqa: E402

C3_GROUPS = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
rank, world, total, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[5]
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    first, B = parallel.shard_books(total, rank, world)
    env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=32, max_live_orders=128, trade_capacity=64 * T, history_capacity=T,
                         book_offset=first, device=0)
    env.set_random_agents(C3_GROUPS)
    dist.barrier()  # both ranks reach their first launch (and with it the stream probe) together
    for c in (T // 2, T - T // 2):
        env.run(c, sync=False)
    env.sync()
    ref = pyoracle.ManyBooks(B, 101 + first, 0, 2, 100_000, True, 32, C3_GROUPS)  # seeded by GLOBAL book index
    ref.run(T, 8)
    assert not env.flags().any()
    assert np.array_equal(env.history(), ref.history()), f"rank {rank}: shard history"
    assert np.array_equal(env.trade_counts(), ref.trade_counts())
    rec = torch.from_numpy(parallel.pack_stats(env.stats()).copy())
    allr = parallel.all_gather_records(rec, dist)
    got = parallel.combine_stats(allr.numpy())
    assert got["n_books"] == total and allr.shape[0] == world
    mine = int(env.trade_counts().sum())
    sums = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([mine], dtype=torch.int64))
    assert got["sum_trades"] == int(sum(int(x) for x in sums)) > 0
    print(f"rank {rank}/{world} ok: books [{first}, {first + B}) pipeline {env.pipeline()} trades {mine}", flush=True)
    env.close()
finally:
    dist.destroy_process_group()
