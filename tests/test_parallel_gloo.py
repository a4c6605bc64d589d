"""Multi-GPU path on CPU: world_size-2 gloo run of the book sharding + 64-byte stats all-gather.

Each rank derives its shard with shard_books(), "steps" it with the oracle (book b seeded from its GLOBAL
index, exactly the rule ManyBookEnv(book_offset=...) uses), packs the per-shard market statistics into the
wire record and all-gathers it; the combined result must equal the statistics of the unsharded run.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [(16, (40, 56), (10, 20), 2, 0.8), (16, (40, 56), (50, 70), 2, 0.2)]
TOTAL, STEPS, LEVELS = 10, 15, 16


def _shard_stats(first, n):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    m = pyoracle.ManyBooks(n, 101 + first, 0, 2, 100_000, True, LEVELS, GROUPS)
    m.run(STEPS, 1)
    last = m.history()[-1]
    nb = last[:, 6] > 0
    na = last[:, 8] > 0
    return {
        "n_books": n, "sum_trade_vol": int(last[:, 0].sum()), "sum_trades": int(m.trade_counts().sum()),
        "sum_events": 0, "sum_bid_vol": int(last[:, 4].sum()), "sum_ask_vol": int(last[:, 3].sum()),
        "min_bid": int(last[nb, 1].min()) if nb.any() else 0xFFFFFFFF, "max_bid": int(last[nb, 1].max()) if nb.any() else 0,
        "min_ask": int(last[na, 2].min()) if na.any() else 0xFFFFFFFF, "max_ask": int(last[na, 2].max()) if na.any() else 0,
    }, m.history()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from bourse_amd import parallel

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = parallel.shard_books(TOTAL, rank, world)
    stats, hist = _shard_stats(first, n)
    rec = torch.from_numpy(parallel.pack_stats(stats).copy())
    allr = parallel.all_gather_records(rec, dist)
    dist.barrier()
    combined = parallel.combine_stats(allr.numpy())
    q.put((rank, first, n, combined, hist))
    dist.destroy_process_group()


def test_shard_books_partition():
    from bourse_amd.parallel import shard_books

    for total, world in ((65536, 8), (10, 3), (7, 8), (1, 1)):
        parts = [shard_books(total, r, world) for r in range(world)]
        assert parts[0][0] == 0 and sum(n for _, n in parts) == total
        for (a, n), (b, _) in zip(parts, parts[1:]):
            assert a + n == b
    with pytest.raises(ValueError):
        shard_books(4, 4, 4)


def test_stats_record_roundtrip():
    from bourse_amd.parallel import pack_stats, unpack_stats

    d = {"n_books": 3, "sum_trade_vol": 2**40 + 5, "sum_trades": 7, "sum_events": 9, "sum_bid_vol": 11,
         "sum_ask_vol": 13, "min_bid": 0xFFFFFFFF, "max_bid": 0, "min_ask": 5, "max_ask": 0xFFFFFFFE}
    w = pack_stats(d)
    assert w.nbytes == 64 and unpack_stats(w) == d


def test_two_rank_gloo_gather_matches_unsharded():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full_stats, full_hist = _shard_stats(0, TOTAL)
    assert res[0][3] == res[1][3] == full_stats
    # sharded histories concatenate to the unsharded one: seeds depend only on the global book index
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]], axis=1), full_hist)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 5, 5, 5)
