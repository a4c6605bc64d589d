"""wave pipeline: throughput vs. number of parts at a given batch size (GPU box).  usage: wave_parts_sweep.py [books] [parts,..]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
parts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,6,8").split(",")]
pipe = sys.argv[3] if len(sys.argv) > 3 else "wave"
for P in parts:
    env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
    env.set_random_agents(groups)
    env.set_pipeline(pipe)
    if pipe == "wave":
        env.set_wave_options(64, P)
    else:
        env.set_split_parts(P, 64)
    env.run(T); env.clear_trades()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
        env.clear_trades()
        best = max(best, B * T / dt / 1e6)
    print(f"B={B} {pipe} parts={P}: {best:7.1f} M book-steps/s ({B / best:6.1f} us/step)", flush=True)
    del env
