"""RandomAgents throughput vs. batch size, number of parts and minimum part size (GPU box).
usage: parts_size_sweep.py [agents_per_group] [levels]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import bourse_amd
B, NA, L, pipe = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
T = 50
groups = [(NA, (32, 64), (10, 20), 2, 0.8), (NA, (32, 64), (50, 70), 2, 0.2)]
env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=2 * NA, trade_capacity=NA * T, history_capacity=T)
env.set_random_agents(groups); env.set_pipeline(pipe)
env.run(T); env.clear_trades()
best = 0.0
for rep in range(3):
    env.sync(); t0 = time.perf_counter(); env.run(T); env.sync(); dt = time.perf_counter() - t0
    env.clear_trades(); best = max(best, B * T / dt / 1e6)
print("%%.1f" %% best)
''' % ROOT
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for B in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
    row = []
    for pipe, parts, mp in (("fused", 1, 4096), ("split", 1, 4096), ("split", 2, 1024), ("split", 3, 1024), ("split", 3, 2048)):
        env = dict(os.environ, BOURSE_AMD_SPLIT_PARTS=str(parts), BOURSE_AMD_MIN_PART=str(mp))
        out = subprocess.run([sys.executable, "-c", CHILD, str(B), str(NA), str(L), pipe], env=env, capture_output=True, text=True, timeout=300)
        row.append(f"{pipe}/{parts}/{mp}: {out.stdout.strip() or out.stderr.strip()[-80:]:>6}")
    print(f"B={B:6d} agents={2*NA}  " + "  ".join(row), flush=True)
