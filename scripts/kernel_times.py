"""Per-kernel HIP-event durations of the split pipeline vs. batch size and part count (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
parts = os.environ.get("BOURSE_AMD_SPLIT_PARTS", "3")
for B in [int(x) for x in (sys.argv[1:] or ["8192", "32768", "65536"])]:
    env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
    env.set_random_agents(groups)
    env.set_pipeline("split")
    env.run(T); env.clear_trades()
    t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
    env.clear_trades()
    env.profile(1)
    env.run(T)
    ka, na = env.profile_read_kind(1)
    kb, nb = env.profile_read_kind(2)
    env.profile_read(True)
    env.profile(0)
    print(f"parts={parts} B={B:7d}: {B*T/dt/1e6:6.1f} M book-steps/s, {dt/T*1e6:6.1f} us/step | k_agents_fsm {ka/max(na,1)*1e3:6.1f} us x{na}"
          f" | k_step_batch {kb/max(nb,1)*1e3:6.1f} us x{nb}", flush=True)
    del env
