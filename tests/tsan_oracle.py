"""The oracle's multi-threaded runners (books / markets statically partitioned over host threads) under ThreadSanitizer.
Run by tests/test_oracle_asan.py with LD_PRELOAD=libtsan.so and BOURSE_ORACLE_TSAN_LIB pointing at a -fsanitize=thread build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyoracle

pyoracle._LIB_PATH = os.environ["BOURSE_ORACLE_TSAN_LIB"]
pyoracle.build = lambda force=False: pyoracle._LIB_PATH
C3 = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
m = pyoracle.ManyBooks(64, 101, 0, 2, 100000, True, 32, C3)
m.run(20, 8)
print("random ok", m.trade_counts().sum())
MOM = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=5.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0, price_dist_sigma=10.0)
NOI = dict(tick_size=2, p_limit=0.2, p_market=0.2, p_cancel=0.1, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
m = pyoracle.ManyBooks(32, 101, 0, 1, 1000000, True, 10, members=[("momentum", 0, 10, MOM), ("noise", 10, 20, NOI)])
m.run(30, 8)
print("mixed ok", m.trade_counts().sum())
mk = pyoracle.ManyMarkets(24, 3, 0, [1, 2, 1], 1000000, True, 10,
                          members=[(2, ("momentum", 0, 10, MOM)), (0, ("noise", 10, 20, dict(NOI, tick_size=1))),
                                   (1, ("random", 12, (40, 60), (1, 9), 2, 0.7))])
mk.run(30, 8)
print("markets ok")
