"""Per-kernel HIP-event durations of the wave_split pipeline (k_agents_wave + k_step_batch) vs. batch size / parts (GPU box).
usage: wave_kernel_times.py [books[:parts] ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
for spec in (sys.argv[1:] or ["8192:1", "8192:3", "2731:1", "16384:3"]):
    B, P = (int(x) for x in (spec.split(":") + ["0"])[:2])
    env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
    env.set_random_agents(groups)
    env.set_pipeline("wave_split")
    env.set_wave_options(64, P)
    env.run(T); env.clear_trades()
    t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
    env.clear_trades()
    env.profile(1)
    env.run(T)
    ka, na = env.profile_read_kind(1)
    kb, nb = env.profile_read_kind(2)
    env.profile_read(True)
    env.profile(0)
    print(f"B={B:7d} parts={P}: {B*T/dt/1e6:6.1f} M book-steps/s, {dt/T*1e6:6.1f} us/step | k_agents_wave {ka/max(na,1)*1e3:6.1f} us x{na}"
          f" | k_step_batch {kb/max(nb,1)*1e3:6.1f} us x{nb}", flush=True)
    del env
