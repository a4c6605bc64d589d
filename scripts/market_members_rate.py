"""Throughput of MarketEnv mode with a MarketAgentSet of Noise / Momentum / Random members (the lane-per-market members' update +
k_step_batch<R, MKT, POOLPEND>); BOURSE_AMD_LIBRARY=<another build> for an A/B.  GPU box:  python scripts/market_members_rate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bourse_amd as bk
from test_gpu_parity import MOM_P, NOISE_P
T = 40
for A, NM, per in ((2, 8192, 64), (4, 4096, 32)):  # (a set with Noise / Momentum members holds at most four members)
    members = []
    t = 0
    for a in range(A):
        if A == 2:
            members += [(a, ("momentum", t, per // 2, MOM_P)), (a, ("noise", t + per // 2, per // 2, NOISE_P))]
        else:
            members += [(a, ("noise", t, per, NOISE_P))]
        t += per
    env = bk.ManyMarketEnv(NM, 101, 0, [1] * A, 1_000_000, True, levels=16, max_live_orders=256, trade_capacity=64 * T * 4, history_capacity=T, strict=False)
    env.set_market_agents(members)
    env.run(T); env.clear_trades(); env.clear_history()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
        env.clear_trades(); env.clear_history()
        best = max(best, NM * A * T / dt / 1e6)
    print(f"{NM} markets x {A} assets, {per * A} members/market: {best:.1f} M book-steps/s, pipeline {env.pipeline()}, flags {int(env.flags().max())}", flush=True)
    env.close()
