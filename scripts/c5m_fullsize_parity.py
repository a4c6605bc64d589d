"""C5 as written at its full size (8 192 books x 512 agents x 64 levels): the split pipeline (and the fused one on a
slice) against the oracle, every book's L2 history, RNG state and trade counts.  GPU box, ~20 s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, bourse_amd as bk, pyoracle as oracle
MOM_P = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=20.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0, price_dist_sigma=10.0)
NOISE_P = dict(tick_size=2, p_limit=0.3, p_market=0.2, p_cancel=0.2, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
members = [("momentum", 0, 256, MOM_P), ("noise", 256, 256, NOISE_P)]
B, T = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 24
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=64, max_live_orders=512, trade_capacity=96 * T, history_capacity=T, strict=False)
env.set_agents(members)
# mostly the auto pipeline (round 3: the wave-parallel members' decode in two parts), with the other pipelines in between
left = T
for c, pipe in ((T // 3, "auto"), (4, "split"), (2, "split_wave"), (T, "auto")):
    c = min(c, left); left -= c
    if c:
        env.set_pipeline(pipe); env.run(c)
ref = oracle.ManyBooks(B, 101, 0, 2, 100_000, True, 64, members=members)
t0 = time.time(); ref.run(T, os.cpu_count()); t1 = time.time()
f = env.flags()
assert not (f & ~np.uint32(64)).any(), np.unique(f)
assert np.array_equal(env.history(), ref.history()), "L2 history"
assert np.array_equal(env.trade_counts(), ref.trade_counts()), "trade counts"
want = ref.rng_states()
assert all(env.rng_state(b) == (int(want[b, 0]), int(want[b, 1])) for b in range(0, B, 7)), "rng"
print(f"C5 as written, full size: {B} books x {T} steps bit-exact vs the oracle ({int(env.trade_counts().sum())} trades; "
      f"books flagged off-tick clamp: {int((f & 64).astype(bool).sum())}); oracle {t1 - t0:.1f} s on {os.cpu_count()} threads")
