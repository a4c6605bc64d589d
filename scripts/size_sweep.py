"""Throughput of the C3 agent mix vs. batch size for the bk_run pipelines (GPU box).
usage: python scripts/size_sweep.py [pipelines comma-separated] [sizes comma-separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
pipes = (sys.argv[1] if len(sys.argv) > 1 else "fused,split").split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4096,8192,16384,65536").split(",")]
for B in sizes:
    row = []
    for pipe in pipes:
        env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
        env.set_random_agents(groups)
        env.set_pipeline(pipe)
        env.run(T); env.clear_trades()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
            env.clear_trades()
            best = max(best, B * T / dt / 1e6)
        row.append(f"{pipe} {best:7.1f} M ({B / best:6.1f} us/step)")
        del env
    print(f"B={B:7d}  " + "  ".join(row), flush=True)
