"""Throughput of MarketEnv mode: markets of 4 books, 32 RandomMarketAgents per asset, 65 536 books in all (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
for A, per in ((4, 32), (2, 64), (8, 16)):
    NM = 65536 // A
    groups = []
    for a in range(A):
        groups += [(a, per // 2, (32, 64), (10, 20), 2, 0.8), (a, per // 2, (32, 64), (50, 70), 2, 0.2)][: 8 // A if A > 4 else 2]
    n_agents = sum(g[1] for g in groups)
    env = bourse_amd.ManyMarketEnv(NM, 101, 0, [2] * A, 100_000, levels=L, max_live_orders=n_agents,
                                   trade_capacity=64 * T, history_capacity=T)
    env.set_random_market_agents(groups)
    env.run(T); env.clear_trades()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
        env.clear_trades()
        best = max(best, NM * A * T / dt / 1e6)
    st = env.stats()
    print(f"{NM} markets x {A} assets, {n_agents} agents/market: {best:.1f} M book-steps/s ({best / A:.1f} M market-steps/s)  flags={int(env.flags().max())}", flush=True)
    del env
