"""Long-run parity soak: C3 agent mix, thousands of steps, every book's final L2 record / trade count / RNG state and the
full trade stream of a sample of books against the CPU oracle (GPU box; ~1 min)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, bourse_amd as bk, pyoracle as oracle
B, T, chunk, L = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 250, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * chunk, history_capacity=chunk)
env.set_random_agents(groups)
ref = oracle.ManyBooks(B, 101, 0, 2, 100_000, True, L, groups)
sample = [0, 1, B // 3, B - 1]
got_tr = {b: [] for b in sample}
t0 = time.time()
for i in range(T // chunk):
    env.set_pipeline(("split", "fused")[i % 2])
    env.run(chunk)
    for b in sample:
        got_tr[b].append(env.trades(b).copy())
    env.clear_trades()
t1 = time.time()
ref.run(T, os.cpu_count())
t2 = time.time()
assert not env.flags().any(), np.unique(env.flags())
assert np.array_equal(env.level2(), ref.history(T - 1, 1)[0]), "final L2"
assert np.array_equal(env.history(), ref.history(T - chunk, chunk)), "last chunk of history"
assert np.array_equal(env.trade_counts(), ref.trade_counts()), "trade counts"
want = ref.rng_states()
assert all(env.rng_state(b) == (int(want[b, 0]), int(want[b, 1])) for b in range(B)), "rng"
for b in sample:
    g, e = np.concatenate(got_tr[b]), ref.book(b).trades_array()
    assert len(g) == len(e) and all(np.array_equal(g[f], e[f]) for f in g.dtype.names), f"trades of book {b}"
print(f"soak ok: {B} books x {T} steps, {int(env.trade_counts().sum())} trades; gpu {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s")
