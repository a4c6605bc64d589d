"""Multi-GPU sharding of independent books (one process per GPU, torch.distributed over RCCL).

Books never interact (each is its own ``Env`` + agents + RNG: ref crates/step_sim/src/runner.rs:46-69
touches nothing global), so stepping needs NO collective.  GPU ``g`` owns the contiguous block
``[g*B/G, (g+1)*B/G)`` and seeds book ``b`` from its GLOBAL index, so results are identical for any G.
The only exchange is an all-gather of one 64-byte market-statistics record per GPU, modelled on
``Market``'s array-valued queries (ref crates/order_book/src/market.rs:137-216).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

STATS_FIELDS = ("n_books", "sum_trade_vol", "sum_trades", "sum_events", "sum_bid_vol", "sum_ask_vol")
STATS_WORDS = 8  # 6 x u64 sums + (min_bid, max_bid) + (min_ask, max_ask) packed as 2 x u64 = 64 bytes


def shard_books(total_books: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: returns (first_global_book, n_books) of ``rank``."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = total_books * rank // world
    hi = total_books * (rank + 1) // world
    return lo, hi - lo


def pack_stats(d: Dict[str, int]) -> np.ndarray:
    """dict (as returned by ``ManyBookEnv.stats``) -> the 64-byte wire record as int64[8]."""
    w = np.zeros(STATS_WORDS, dtype=np.uint64)
    for i, k in enumerate(STATS_FIELDS):
        w[i] = d[k]
    w[6] = (int(d["max_bid"]) << 32) | int(d["min_bid"])
    w[7] = (int(d["max_ask"]) << 32) | int(d["min_ask"])
    return w.view(np.int64)


def unpack_stats(w: np.ndarray) -> Dict[str, int]:
    w = np.asarray(w).view(np.uint64)
    d = {k: int(w[i]) for i, k in enumerate(STATS_FIELDS)}
    d["min_bid"], d["max_bid"] = int(w[6]) & 0xFFFFFFFF, int(w[6]) >> 32
    d["min_ask"], d["max_ask"] = int(w[7]) & 0xFFFFFFFF, int(w[7]) >> 32
    return d


def combine_stats(records: np.ndarray) -> Dict[str, int]:
    """Reduce the gathered per-GPU records [G, 8] into whole-node statistics."""
    parts = [unpack_stats(r) for r in np.asarray(records).reshape(-1, STATS_WORDS)]
    out = {k: sum(p[k] for p in parts) for k in STATS_FIELDS}
    out["min_bid"] = min(p["min_bid"] for p in parts)
    out["max_bid"] = max(p["max_bid"] for p in parts)
    out["min_ask"] = min(p["min_ask"] for p in parts)
    out["max_ask"] = max(p["max_ask"] for p in parts)
    return out


def all_gather_records(record, dist):
    """All-gather one int64[8] torch tensor per rank -> int64[world, 8] (RCCL on GPU, gloo on CPU)."""
    import torch

    world = dist.get_world_size()
    out = torch.empty((world, STATS_WORDS), dtype=torch.int64, device=record.device)
    dist.all_gather_into_tensor(out.view(-1), record.contiguous().view(-1))
    return out


class _DevicePtr:
    """Expose a raw device allocation to torch without copying (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (ptr, False), "version": 2}


class StatsGather:
    """Per-launch market-stats all-gather for a sharded run.

    ``all_gather()`` reduces this GPU's books into the 64-byte record ON DEVICE (k_stats, on the env's
    stream), then all-gathers the records with RCCL on the same stream; nothing is copied to the host
    until ``result()`` is called."""

    def __init__(self, env, dist):
        import torch

        self.env, self.dist, self.torch = env, dist, torch
        self.out = None
        # gloo (CPU collectives: several ranks sharing one GPU in a dry run, or tests): the record travels as a host tensor
        self.host = dist.get_backend() == "gloo"
        if self.host:
            self.record, self.zero_copy = None, False
            return
        try:
            self.record = torch.as_tensor(_DevicePtr(env.stats_device_ptr(), 64), device="cuda")
            self.zero_copy = True
        except Exception:  # pragma: no cover - depends on the torch build
            self.record = torch.zeros(STATS_WORDS, dtype=torch.int64, device="cuda")
            self.zero_copy = False

    def all_gather(self):
        if self.host:
            self.out = all_gather_records(self.torch.from_numpy(pack_stats(self.env.stats()).copy()), self.dist)
            return self.out
        if self.zero_copy:
            self.env.stats_compute_async()
        else:
            host = pack_stats(self.env.stats())
            self.record.copy_(self.torch.from_numpy(host.copy()))
        self.out = all_gather_records(self.record, self.dist)
        return self.out

    def result(self) -> Dict[str, int]:
        return combine_stats(self.out.cpu().numpy())


L1_WORDS = 9  # [trade_vol, bid, ask, ask_vol, bid_vol, bid_touch_vol, bid_touch_n, ask_touch_vol, ask_touch_n]


class _DevicePtr32:
    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i4", "data": (ptr, False), "version": 2}


class L1Gather:
    """Optional second tier of the stats exchange: every book's level-1 record — the 9 leading words of
    its level-2 record, ``StepEnvNumpy.level_1_data`` layout — all-gathered over RCCL: 36 B per book, 2.4 MB per GPU at
    65 536 books, on the env's stream and overlappable with the next launch.  ``all_gather()`` returns an
    int32[world, n_books, 9] device tensor (a view of the words; reinterpret as uint32 on the host)."""

    def __init__(self, env, dist):
        import torch

        self.env, self.dist, self.torch = env, dist, torch
        self.l2 = torch.as_tensor(_DevicePtr32(env.level2_device_ptr(), (env.n_books, env.width)), device="cuda")
        self.out = torch.empty((dist.get_world_size(), env.n_books, L1_WORDS), dtype=torch.int32, device="cuda")

    def all_gather(self):
        mine = self.l2[:, :L1_WORDS].contiguous()  # zero-copy view of the library's records, packed on the device
        self.dist.all_gather_into_tensor(self.out.view(-1), mine.view(-1))
        return self.out

    def result(self) -> np.ndarray:
        """uint32[world * n_books, 9] on the host, books in global order (rank-major = contiguous shards)."""
        return self.out.cpu().numpy().view(np.uint32).reshape(-1, L1_WORDS)
