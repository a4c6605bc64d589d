"""k_agents_fsm / k_step_batch launch time vs. agents per book (fixed overhead vs per-draw cost).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for n in (2, 8, 32, 64, 128, 256, 512):
    groups = [(n // 2, (32, 64), (10, 20), 2, 0.8), (n // 2, (32, 64), (50, 70), 2, 0.2)]
    env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=32, max_live_orders=max(64, n), trade_capacity=64 * 60, history_capacity=0)
    env.set_random_agents(groups); env.set_pipeline("split")
    env.run(30); env.clear_trades()
    env.profile(1); env.run(30)
    ka, na = env.profile_read_kind(1); kb, nb = env.profile_read_kind(2); env.profile_read(True)
    ev = env.stats()["sum_events"] / (B * 60)
    print(f"agents {n:4d}: k_agents_fsm {ka / na * 1e3:7.1f} us  k_step_batch {kb / nb * 1e3:7.1f} us  events/book-step {ev:.1f}", flush=True)
    del env
