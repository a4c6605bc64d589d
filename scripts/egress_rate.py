"""Measure the PCIe-inclusive rate of the C3 workload when every L2 record is copied to the host (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bourse_amd
B, T, L = 65536, 50, 32
groups = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * T, history_capacity=T)
env.set_random_agents(groups)
env.run(T); env.clear_history(); env.clear_trades()
out = np.zeros((T, B, env.width), dtype=np.uint32)
for rep in range(3):
    t0 = time.perf_counter()
    env.run(T)
    t1 = time.perf_counter()
    f, n = env.history_len()
    bourse_amd._lib.check(env._L.bk_history(env._h, f, n, 0, B, bourse_amd._lib.p32(out)))
    t2 = time.perf_counter()
    env.clear_history(); env.clear_trades()
    print(f"run {B*T/(t1-t0)/1e6:.1f} M book-steps/s | D2H of {out.nbytes/1e9:.2f} GB history {out.nbytes/(t2-t1)/1e9:.1f} GB/s | "
          f"run+copy {B*T/(t2-t0)/1e6:.1f} M book-steps/s")

# streaming egress: ring of 2 chunks, the copy of chunk k overlaps the stepping of chunk k+1
for chunk in (10, 25):
    env2 = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * 200,
                                  history_capacity=2 * chunk)
    env2.set_random_agents(groups)
    env2.run(chunk)
    for rep in range(2):
        env2.clear_trades()
        st = env2.stream_history(100, chunk)
        print(f"streamed chunk={chunk}: {st['book_steps_per_s']/1e6:.1f} M book-steps/s, D2H {st['d2h_gb_per_s']:.1f} GB/s sustained")
    del env2

# L2 records AND trade records (compacted on the device) streamed together
for chunk in (10,):
    env3 = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=L, max_live_orders=128, trade_capacity=64 * chunk,
                                  history_capacity=2 * chunk)
    env3.set_random_agents(groups)
    env3.run(chunk); env3.clear_trades()
    for rep in range(2):
        st = env3.stream_history(100, chunk, trades=True, trade_records_per_chunk=48 * chunk * B)
        print(f"streamed L2 + trades chunk={chunk}: {st['book_steps_per_s']/1e6:.1f} M book-steps/s, D2H {st['d2h_gb_per_s']:.1f} GB/s sustained")
    del env3
