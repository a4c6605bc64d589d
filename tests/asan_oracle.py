"""Run the CPU oracle's main workloads under AddressSanitizer + UBSan (CPU only; GPU ASan is unavailable on this pool).
Usage: make -C oracle asan && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_oracle.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import pyoracle
pyoracle._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'libbourse_oracle_asan.so')
pyoracle.build = lambda force=False: pyoracle._LIB_PATH
import numpy as np
C3=[(64,(32,64),(10,20),2,0.8),(64,(32,64),(50,70),2,0.2)]
m = pyoracle.ManyBooks(8, 101, 0, 2, 100000, True, 32, C3); m.run(40, 2); print('random ok', m.trade_counts().sum())
MOM = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=5.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0, price_dist_sigma=10.0)
NOI = dict(tick_size=2, p_limit=0.2, p_market=0.2, p_cancel=0.1, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
m = pyoracle.ManyBooks(4, 101, 0, 1, 1000000, True, 10, members=[("momentum",0,10,MOM),("noise",10,20,NOI)]); m.run(60, 1); print('mixed ok', m.trade_counts().sum())
env = pyoracle.StepEnv(1, 0, 2, 100000)
rng = np.random.default_rng(1); ids=[]
for s in range(40):
    for k in range(10):
        u = rng.random()
        if u < 0.6 or not ids: ids.append(env.place_order(bool(rng.integers(0,2)), int(rng.integers(0,40)), 7, None if rng.random()<0.1 else int(rng.integers(40,60))*2))
        elif u < 0.8: env.cancel_order(int(rng.choice(ids)))
        else: env.modify_order(int(rng.choice(ids)), None if rng.random()<0.4 else int(rng.integers(40,60))*2, None if rng.random()<0.3 else int(rng.integers(0,50)))
    env.step()
print('host ok', len(env.get_trades()))
