"""Run the CPU oracle's main workloads under AddressSanitizer + UBSan (CPU only; GPU ASan is unavailable on this pool).
Usage: make -C oracle asan && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_oracle.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import pyoracle
pyoracle._LIB_PATH = os.environ.get('BOURSE_ORACLE_ASAN_LIB') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'libbourse_oracle_asan.so')
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
mk = pyoracle.ManyMarkets(6, 3, 0, [1, 2, 1], 1000000, True, 10, members=[(2, ("momentum", 0, 10, MOM)), (0, ("noise", 10, 20, dict(NOI, tick_size=1))),
                                                                           (1, ("random", 12, (40, 60), (1, 9), 2, 0.7)), (2, ("noise", 50, 10, NOI))])
mk.run(50, 2); print('market agents ok', sum(mk.book(m, a).n_orders() for m in range(6) for a in range(3)))
mk2 = pyoracle.ManyMarkets(2, 3, 0, [1, 2], 1000, True, 10)
for s in range(20):
    for k in range(8):
        a = int(rng.integers(0, 2)); n = mk2.book(0, a).n_orders()
        if n and rng.random() < 0.3: mk2.cancel_order(0, a, int(rng.integers(0, n)))
        elif n and rng.random() < 0.2: mk2.modify_order(0, a, int(rng.integers(0, n)), None, int(rng.integers(1, 9)))
        else: mk2.place_order(0, a, bool(rng.integers(0, 2)), int(rng.integers(1, 20)), 1, int(rng.integers(45, 55)) * 2)
    mk2.step()
print('market host ok', mk2.n_steps())
ob = pyoracle.OrderBook(0, 1)
for i in range(30):
    ob.set_time(i + 1); ob.place_order(bool(i % 2), 5 + i % 7, 1, 100 + (i * 7) % 11)
ob.save_json_snapshot('/tmp/_asan_ob.json'); lb = pyoracle.order_book_from_json('/tmp/_asan_ob.json')
assert lb.state() == ob.state(); print('json ok', ob.n_orders())
