"""Throughput of the C3 agent mix vs. batch size for both bk_run pipelines (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
for B in (32768, 65536, 131072):
    row = []
    for pipe in ("fused", "split"):
        env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
        env.set_random_agents(groups)
        env.set_pipeline(pipe)
        env.run(T); env.clear_trades()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
            env.clear_trades()
            best = max(best, B * T / dt / 1e6)
        row.append(f"{pipe} {best:7.1f} M")
        del env
    print(f"B={B:7d}  " + "  ".join(row), flush=True)
