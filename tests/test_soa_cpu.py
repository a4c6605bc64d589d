"""The batched SoA CPU implementation (oracle/bourse_soa.cpp: price-level ladder + per-level FIFO + level bitmaps) equals
the literal map-based oracle on every shape the bench uses: level-2 history of every step and book, trade records
(order, times, prices, volumes, both ids), event counts, RNG states.  It is the second CPU figure of bench.py's
cpu_baseline (kind "soa") and, like the oracle, test / measurement infrastructure only."""
import numpy as np
import pytest

C2 = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]
C3 = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
C5 = [(256, (100, 164), (10, 20), 2, 0.8), (256, (100, 164), (50, 70), 2, 0.2)]
REF_BENCH = [(100, (40, 60), (10, 20), 2, 0.8), (100, (10, 90), (50, 70), 2, 0.2)]  # benches/benchmarks.rs:14-23 scaled
EDGE = [(5, (10, 14), (1, 3), 3, 0.0), (37, (10, 14), (0, 3), 3, 1.0), (23, (9, 13), (1, 2), 6, 0.5), (0, (1, 2), (1, 2), 3, 0.5)]


@pytest.mark.parametrize("name,groups,levels,tick,books,steps,threads", [
    ("C2", C2, 16, 2, 24, 40, 1), ("C3", C3, 32, 2, 40, 40, 3), ("C5", C5, 64, 2, 4, 12, 2),
    ("ref-bench", REF_BENCH, 10, 1, 5, 60, 1), ("edge", EDGE, 4, 3, 7, 50, 2),
])
def test_soa_equals_oracle(oracle, name, groups, levels, tick, books, steps, threads):
    soa = oracle.SoaBooks(books, 101, 0, tick, 100_000, levels, groups, history_capacity=steps, threads=threads)
    ref = oracle.ManyBooks(books, 101, 0, tick, 100_000, True, levels, groups, build_threads=threads)
    for chunk in (1, steps // 2, steps - 1 - steps // 2):
        soa.run(chunk)
        ref.run(chunk, threads)
    assert np.array_equal(soa.history(), ref.history()), name
    assert np.array_equal(soa.trade_counts(), ref.trade_counts())
    assert np.array_equal(soa.rng_states(), ref.rng_states())
    for b in range(books):
        got, exp = soa.trades(b), ref.book(b).trades_array()
        assert len(got) == len(exp)
        for f in ("t", "side", "price", "vol", "active_id", "passive_id"):
            assert np.array_equal(got[f], exp[f]), (name, b, f)
    assert int(soa.trade_counts().sum()) > 0 or name == "edge"


def test_soa_history_ring_and_unsupported_shapes(oracle):
    soa = oracle.SoaBooks(3, 5, 0, 2, 1000, 8, C2, history_capacity=4)
    ref = oracle.ManyBooks(3, 5, 0, 2, 1000, True, 8, C2)
    soa.run(11)
    ref.run(11, 1)
    assert np.array_equal(soa.history(), ref.history()[-4:])  # ring of the last 4 steps
    with pytest.raises(ValueError):
        oracle.SoaBooks(1, 1, 0, 2, 1000, 8, [(4, (10, 20), (1, 2), 3, 0.5)], history_capacity=1)  # agent tick off the grid
    with pytest.raises(ValueError):
        oracle.SoaBooks(1, 1, 0, 1, 1000, 8, [(4, (10, 100000), (1, 2), 1, 0.5)], history_capacity=1)  # window too wide
