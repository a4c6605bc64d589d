"""Long-run parity soaks for the AgentSet and market pipelines vs the oracle (GPU box, ~1 min)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, bourse_amd as bk, pyoracle as oracle
MOM = dict(tick_size=1, p_cancel=0.1, trade_vol=100, decay=1.0, demand=20.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0, price_dist_sigma=10.0)
NOI = dict(tick_size=1, p_limit=0.3, p_market=0.2, p_cancel=0.2, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
NT = os.cpu_count() or 8

def check(env, ref, T, chunk, n_units, stride):
    want_hist = ref.history(T - chunk, chunk)
    assert not (env.flags() & ~np.uint32(64)).any(), np.unique(env.flags())
    assert np.array_equal(env.history(), want_hist), "history"
    want = ref.rng_states()
    assert all(env.rng_state(u * stride) == (int(want[u, 0]), int(want[u, 1])) for u in range(n_units)), "rng"

# 0. Noise / Momentum sets on the wave-parallel members' decode (k_agents_mixed_wave): long runs exercise the lists'
# compaction, slot re-use, the arrival-stamp window and thousands of ziggurat slow paths; every fourth chunk on another
# pipeline (lists rebuilt from the owner tags on the way back)
for B, T, chunk, pool, n_m, n_n in ((1024, 2000, 100, 512, 64, 64), (96, 600, 50, 512, 256, 256)):
    mem = [("momentum", 0, n_m, dict(MOM, demand=20.0 if n_m > 64 else 8.0)), ("noise", n_m, n_n, NOI)]
    env = bk.ManyBookEnv(B, 5, 0, 1, 1_000_000, levels=16, max_live_orders=pool, trade_capacity=256 * chunk, history_capacity=chunk, strict=False)
    env.set_agents(mem)
    ref = oracle.ManyBooks(B, 5, 0, 1, 1_000_000, True, 16, members=mem)
    t0 = time.time()
    for i in range(T // chunk):
        env.set_pipeline(("wave_split", "wave_split", "wave_split", "split", "wave_split", "fused", "wave_split", "split_wave")[i % 8]); env.run(chunk); env.clear_trades()
    t1 = time.time(); ref.run(T, NT); t2 = time.time()
    check(env, ref, T, chunk, B, 1)
    assert np.array_equal(env.trade_counts(), ref.trade_counts())
    print(f"wave-members soak ok: {B} books x {T} steps x {n_m + n_n} agents, {int(ref.trade_counts().sum())} trades (gpu {t1-t0:.1f} s, oracle {t2-t1:.1f} s)")
    del env, ref
# 1. AgentSet (momentum + noise + random), 1024 books x 1500 steps, pipelines cycling every chunk
members = [("momentum", 0, 64, MOM), ("noise", 64, 64, NOI), ("random", 32, (1_000_000_000, 1_000_000_032), (10, 20), 1, 0.5)]
B, T, chunk = 1024, 1500, 100
env = bk.ManyBookEnv(B, 7, 0, 1, 1_000_000, levels=16, max_live_orders=512, trade_capacity=256 * chunk, history_capacity=chunk, strict=False)
env.set_agents(members)
ref = oracle.ManyBooks(B, 7, 0, 1, 1_000_000, True, 16, members=members)
t0 = time.time()
for i in range(T // chunk):
    env.set_pipeline(("split", "fused", "wave_split", "split_wave")[i % 4]); env.run(chunk); env.clear_trades()
t1 = time.time(); ref.run(T, NT); t2 = time.time()
check(env, ref, T, chunk, B, 1)
assert np.array_equal(env.trade_counts(), ref.trade_counts())
print(f"agent-set soak ok: {B} books x {T} steps, {int(ref.trade_counts().sum())} trades (gpu {t1-t0:.1f} s, oracle {t2-t1:.1f} s)")
del env, ref
# 2. markets of 3 assets with every member kind, 512 markets x 1200 steps
mm = [(0, ("noise", 0, 40, NOI)), (2, ("momentum", 100, 40, MOM)), (1, ("random", 48, (1000, 1032), (10, 20), 1, 0.7)), (2, ("noise", 200, 24, NOI))]
NM, T, chunk = 512, 1200, 100
env = bk.ManyMarketEnv(NM, 9, 0, [1, 1, 1], 1_000_000, True, levels=16, max_live_orders=512, trade_capacity=256 * chunk, history_capacity=chunk, strict=False)
env.set_market_agents(mm)
ref = oracle.ManyMarkets(NM, 9, 0, [1, 1, 1], 1_000_000, True, 16, members=mm)
t0 = time.time()
for i in range(T // chunk):
    env.run(chunk); env.clear_trades()
t1 = time.time(); ref.run(T, NT); t2 = time.time()
check(env, ref, T, chunk, NM, 3)
print(f"market soak ok: {NM} markets x 3 assets x {T} steps (gpu {t1-t0:.1f} s, oracle {t2-t1:.1f} s)")
