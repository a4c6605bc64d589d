import os, sys, time
sys.path.insert(0, os.getcwd())
import bourse_amd
T, L = 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
B = 65536
for P in (2, 3, 4, 5, 6):
    env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
    env.set_random_agents(groups); env.set_pipeline("split"); env.set_split_parts(P, 2048)
    env.run(T); env.clear_trades()
    env.profile(1)
    t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
    env.profile(0)
    a = env.profile_read_kind(1); s = env.profile_read_kind(2); env.profile_read()
    print(f"parts={P}: {B*T/dt/1e6:6.1f} M, {dt/T*1e6:6.1f} us/step | fsm {a[0]/a[1]*1e3:6.1f} us x{a[1]} | step {s[0]/s[1]*1e3:6.1f} us x{s[1]}", flush=True)
    del env
