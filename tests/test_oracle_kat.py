"""Pin the CPU oracle against the reference's OWN known-answer tests.

Each test re-expresses one reference test (cited file:line, paths relative to the reference
repo) against ``oracle/pyoracle.py``.  All vectors are permutation-invariant, i.e. independent of
RNG output, so they pin the matching engine / Env::step / layouts for real.
"""
import numpy as np
import pytest

MAX = 2**32 - 1


# ----------------------------------------------------------------- side.rs unit tests
def test_side_init(oracle):  # crates/order_book/src/side.rs:320-339
    a = oracle.BookSide("ask")
    assert (a.vol(), a.best_vol(), a.best_price(), a.best_order_idx()) == (0, 0, MAX, None)
    b = oracle.BookSide("bid")
    assert (b.vol(), b.best_vol(), b.best_vol_and_orders(), b.best_price(), b.best_order_idx()) == (0, 0, (0, 0), 0, None)


def test_side_insert_order(oracle):  # side.rs:342-384
    s = oracle.BookSide("raw")
    s.insert_order(10, 100, 1, 10)
    assert (s.vol(), s.best_vol(), s.best_vol_and_orders(), s.best_price(), s.best_order_idx()) == (10, 10, (10, 1), 100, 1)
    s.insert_order(11, 100, 2, 11)
    assert (s.vol(), s.best_vol(), s.best_vol_and_orders(), s.best_price(), s.best_order_idx()) == (21, 21, (21, 2), 100, 1)
    s.insert_order(12, 101, 3, 12)
    assert (s.vol(), s.best_vol(), s.best_vol_and_orders(), s.best_price(), s.best_order_idx()) == (33, 21, (21, 2), 100, 1)
    s.insert_order(13, 99, 4, 2)
    assert (s.vol(), s.best_vol(), s.best_vol_and_orders(), s.best_price(), s.best_order_idx()) == (35, 2, (2, 1), 99, 4)


def test_side_best_prices(oracle):  # side.rs:387-402
    b = oracle.BookSide("bid")
    b.insert_order(0, 100, 1, 10)
    assert b.best_price() == 100
    a = oracle.BookSide("ask")
    a.insert_order(0, 100, 1, 10)
    assert a.best_price() == 100


def test_side_remove_order(oracle):  # side.rs:405-443
    s = oracle.BookSide("ask")
    s.insert_order(0, 100, 1, 10)
    s.insert_order(1, 99, 2, 10)
    assert (s.best_price(), s.vol(), s.best_vol_and_orders(), s.best_order_idx()) == (99, 20, (10, 1), 2)
    s.remove_order(1, 99, 10)
    assert (s.best_price(), s.vol(), s.best_vol_and_orders(), s.best_order_idx()) == (100, 10, (10, 1), 1)
    s.insert_order(3, 100, 3, 15)
    assert (s.best_price(), s.vol(), s.best_vol_and_orders(), s.best_order_idx()) == (100, 25, (25, 2), 1)
    s.remove_order(3, 100, 15)
    assert (s.best_price(), s.vol(), s.best_vol_and_orders(), s.best_order_idx()) == (100, 10, (10, 1), 1)
    s.remove_order(0, 100, 10)
    assert (s.best_price(), s.vol(), s.best_vol_and_orders(), s.best_order_idx()) == (MAX, 0, (0, 0), None)


def test_side_remove_vol_and_level_lookup(oracle):  # side.rs:446-469
    s = oracle.BookSide("ask")
    s.insert_order(0, 100, 1, 10)
    s.remove_vol(100, 5)
    assert (s.best_vol(), s.best_vol_and_orders(), s.vol()) == (5, (5, 1), 5)
    s = oracle.BookSide("ask")
    s.insert_order(0, 100, 1, 10)
    s.insert_order(1, 100, 2, 20)
    s.insert_order(1, 101, 3, 40)
    assert s.vol_and_orders_at_price(100) == (30, 2)
    assert s.vol_and_orders_at_price(101) == (40, 1)
    assert s.vol_and_orders_at_price(102) == (0, 0)


# ------------------------------------------------------------ orderbook.rs unit tests
def test_book_init(oracle):  # C.1 orderbook.rs:926-936
    b = oracle.OrderBook(0, 1, True)
    assert b.bid_vol() == 0 and b.ask_vol() == 0
    assert b.best_bid_vol() == 0 and b.best_bid_vol_and_orders() == (0, 0)
    assert b.best_ask_vol_and_orders() == (0, 0)
    assert b.bid_ask() == (0, MAX)


def test_book_insert_order(oracle):  # C.2 orderbook.rs:939-980
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 0, 100)
    b.place_order(True, 10, 0, 50)
    assert b.bid_ask() == (50, 100)
    assert (b.ask_vol(), b.bid_vol()) == (10, 10)
    assert b.best_bid_vol_and_orders() == (10, 1) and b.best_ask_vol_and_orders() == (10, 1)
    b.place_order(False, 10, 0, 90)
    b.place_order(True, 10, 0, 60)
    assert b.bid_ask() == (60, 90)
    assert (b.ask_vol(), b.bid_vol()) == (20, 20)
    assert b.best_bid_vol_and_orders() == (10, 1) and b.best_ask_vol_and_orders() == (10, 1)
    b.place_order(False, 10, 0, 110)
    b.place_order(True, 10, 0, 40)
    assert b.bid_ask() == (60, 90)
    assert (b.ask_vol(), b.bid_vol()) == (30, 30)
    assert b.best_bid_vol_and_orders() == (10, 1) and b.best_ask_vol_and_orders() == (10, 1)


def test_book_level_data_with_gaps(oracle):  # C.3 orderbook.rs:983-1049 (tick 2, LEVELS 4)
    b = oracle.OrderBook(0, 2, True, levels=4)
    _, _, _, _, bl, al = b.level2()
    assert bl.tolist() == [[0, 0]] * 4 and al.tolist() == [[0, 0]] * 4
    for vol, price in ((10, 100), (10, 100), (12, 98), (14, 94)):
        b.place_order(True, vol, 0, price)
    for vol, price in ((11, 102), (11, 102), (13, 104), (15, 108)):
        b.place_order(False, vol, 0, price)
    bid, ask, bv, av, bl, al = b.level2()
    assert bl.tolist() == [[20, 2], [12, 1], [0, 0], [14, 1]]
    assert al.tolist() == [[22, 2], [13, 1], [0, 0], [15, 1]]
    assert (bid, ask, bv, av) == (100, 102, 46, 50)
    assert b.best_bid_vol_and_orders() == (20, 2) and b.best_ask_vol_and_orders() == (22, 2)


def test_book_cancel_order(oracle):  # C.4 orderbook.rs:1052-1098
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 0, 100)
    b.place_order(True, 10, 0, 50)
    b.place_order(False, 10, 0, 90)
    b.place_order(True, 10, 0, 60)
    assert b.bid_ask() == (60, 90) and (b.ask_vol(), b.bid_vol()) == (20, 20)
    b.cancel_order(0)
    b.cancel_order(3)
    assert b.bid_ask() == (50, 90) and (b.ask_vol(), b.bid_vol()) == (10, 10)
    assert b.best_bid_vol_and_orders() == (10, 1) and b.best_ask_vol_and_orders() == (10, 1)
    b.cancel_order(1)
    b.cancel_order(2)
    assert b.bid_ask() == (0, MAX) and (b.ask_vol(), b.bid_vol()) == (0, 0)
    assert b.best_bid_vol_and_orders() == (0, 0) and b.best_ask_vol_and_orders() == (0, 0)
    assert [b.order_status(i) for i in range(4)] == [3, 3, 3, 3]


def test_book_mod_order_vol(oracle):  # C.5 orderbook.rs:1101-1121
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 0, 100)
    b.place_order(True, 10, 0, 50)
    b.modify_order(0, None, 8)
    b.modify_order(1, None, 5)
    assert (b.ask_vol(), b.best_ask_vol_and_orders()) == (8, (8, 1))
    assert (b.bid_vol(), b.best_bid_vol_and_orders()) == (5, (5, 1))
    o = b.orders_array()
    assert (int(o["vol"][0]), int(o["vol"][1])) == (8, 5)


def test_book_modify_order(oracle):  # C.6 orderbook.rs:1124-1142
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 0, 100)
    b.place_order(True, 10, 0, 50)
    assert b.bid_ask() == (50, 100)
    b.modify_order(0, 110, 15)
    b.modify_order(1, 60, 20)
    assert (b.ask_vol(), b.best_ask_vol()) == (15, 15)
    assert (b.bid_vol(), b.best_bid_vol()) == (20, 20)
    assert b.bid_ask() == (60, 110)


def test_book_modify_order_crossing(oracle):  # C.7 orderbook.rs:1145-1168
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 0, 100)
    b.place_order(True, 10, 0, 50)
    b.modify_order(1, 100, 20)
    assert (b.ask_vol(), b.best_ask_vol_and_orders()) == (0, (0, 0))
    assert (b.bid_vol(), b.best_bid_vol_and_orders()) == (10, (10, 1))
    assert b.bid_ask() == (100, MAX)
    t = b.trades_array()
    assert len(t) == 1 and (int(t["price"][0]), int(t["vol"][0])) == (100, 10)


def test_book_trades(oracle):  # C.8 orderbook.rs:1171-1214
    b = oracle.OrderBook(0, 1, True)
    assert b.create_order(False, 101, 101, 20) == 0
    assert b.create_order(False, 101, 101, 18) == 1
    assert b.create_order(True, 202, 101, 12) == 2
    assert b.create_order(True, 202, 101, 14) == 3
    for t, i in enumerate(range(4)):
        b.place_order_id(i)
        b.set_time(t + 1)
    assert b.create_order(True, 102, 101, None) == 4
    b.place_order_id(4)
    assert b.ask_vol() == 100 and b.bid_ask() == (14, 20)
    tr = b.trades_array()
    assert len(tr) == 2
    assert (int(tr["price"][0]), int(tr["vol"][0])) == (18, 101)
    assert (int(tr["price"][1]), int(tr["vol"][1])) == (20, 1)
    assert b.trade_vol() == 102
    assert b.create_order(False, 204, 101, 14) == 5
    b.place_order_id(5)
    assert (b.bid_vol(), b.ask_vol()) == (202, 102)
    assert b.best_bid_vol_and_orders() == (202, 1) and b.best_ask_vol_and_orders() == (2, 1)
    assert b.bid_ask() == (12, 14)
    tr = b.trades_array()
    assert len(tr) == 3 and (int(tr["price"][2]), int(tr["vol"][2])) == (14, 202)
    assert b.trade_vol() == 304
    # record fields of match_orders (orderbook.rs:849-859): passive side/price, (active, passive) ids
    assert tr["active_id"].tolist() == [4, 4, 5] and tr["passive_id"].tolist() == [1, 0, 3]
    assert tr["side"].tolist() == [0, 0, 1] and tr["t"].tolist() == [4, 4, 4]


def test_book_market_order_no_trading(oracle):  # C.9 orderbook.rs:1217-1227
    b = oracle.OrderBook(0, 1, False)
    b.place_order(True, 101, 101, None)
    assert b.bid_ask() == (0, MAX) and (b.bid_vol(), b.ask_vol()) == (0, 0)
    assert b.order_status(0) == 4


def test_book_unfilled_market_order(oracle):  # C.9 orderbook.rs:1230-1242
    b = oracle.OrderBook(0, 1, True)
    b.place_order(False, 10, 101, 50)
    b.place_order(True, 20, 101, None)
    assert b.bid_ask() == (0, MAX) and (b.bid_vol(), b.ask_vol()) == (0, 0)
    assert b.order_status(1) == 3


def test_book_incorrect_price_err(oracle):  # C.10 orderbook.rs:1245-1257
    b = oracle.OrderBook(0, 2, True)
    with pytest.raises(ValueError, match="Price 51 was not a multiple of tick-size 2"):
        b.create_order(False, 100, 101, 51)
    assert b.n_orders() == 0  # the rejected order consumed no id (App. A.2)


def test_book_no_double_place(oracle):  # C.10 orderbook.rs:1260-1274
    b = oracle.OrderBook(0, 2, True)
    i = b.create_order(False, 100, 101, 50)
    b.place_order_id(i)
    assert b.bid_ask() == (0, 50) and b.best_ask_vol_and_orders() == (100, 1)
    b.place_order_id(i)
    assert b.bid_ask() == (0, 50) and b.best_ask_vol_and_orders() == (100, 1)


def test_python_trades_with_ids_and_times(oracle):  # C.14 tests/test_order_book.py:91-135
    ob = oracle.OrderBook(0, 1)
    ob.place_order(True, 10, 11, 50)
    id1 = ob.place_order(False, 20, 12, 60)
    id2 = ob.place_order(True, 10, 11, 55)
    id3 = ob.place_order(False, 20, 12, 65)
    ob.set_time(10)
    id4 = ob.place_order(True, 30, 11)
    assert ob.order_status(id4) == 2 and ob.order_status(id1) == 2
    assert ob.bid_ask() == (55, 65) and (ob.bid_vol(), ob.ask_vol()) == (20, 10)
    ob.set_time(20)
    id5 = ob.place_order(False, 20, 12, 55)
    assert ob.order_status(id5) == 1 and ob.order_status(id2) == 2
    assert ob.bid_ask() == (50, 55) and (ob.bid_vol(), ob.ask_vol()) == (10, 20)
    tr = ob.get_trades()
    assert [t[0] for t in tr] == [10, 10, 20]
    assert [t[2] for t in tr] == [60, 65, 55]
    assert [t[3] for t in tr] == [20, 10, 10]
    assert [t[4] for t in tr] == [id4, id4, id5]
    assert [t[5] for t in tr] == [id1, id3, id2]


# ------------------------------------------------------------------- env.rs unit test
def test_env_three_steps(oracle):  # C.11 crates/step_sim/src/env.rs:312-368 (LEVELS 10, seed 101)
    env = oracle.StepEnv(101, 0, 1, 1000)
    env.place_order(True, 10, 101, 10)
    env.place_order(False, 20, 101, 20)
    env.step()
    assert env.n_transactions() == 0
    assert env.book.bid_ask() == (10, 20) and env.book.n_orders() == 2
    assert [env.order_status(i) for i in range(2)] == [1, 1]
    assert env.time == 1000
    env.place_order(True, 10, 101, 11)
    env.place_order(False, 20, 101, 21)
    env.step()
    assert env.book.bid_ask() == (11, 20) and env.book.n_orders() == 4 and env.time == 2000
    env.place_order(True, 30, 101, None)
    env.step()
    assert env.book.bid_ask() == (11, 21) and env.book.ask_vol() == 10 and env.book.n_orders() == 5
    assert env.order_status(1) == 2 and env.order_status(4) == 2
    assert env.book.n_trades() == 2 and env.time == 3000
    bids, asks = env.get_prices()
    assert bids.tolist() == [10, 11, 11] and asks.tolist() == [20, 20, 21]
    bv, av = env.get_volumes()
    assert bv.tolist() == [10, 20, 20] and av.tolist() == [20, 40, 10]
    tb, ta = env.get_touch_volumes()
    assert tb.tolist() == [10, 10, 10] and ta.tolist() == [20, 20, 10]
    cb, ca = env.get_touch_order_counts()
    assert cb.tolist() == [1, 1, 1] and ca.tolist() == [1, 1, 1]
    assert env.get_trade_volumes().tolist() == [0, 0, 30]


def test_agent_set_declaration_order(oracle):  # C.16 crates/step_sim/tests/test_macros.rs:28-56
    env = oracle.StepEnv(101, 0, 1, 1000)
    for k in (1, 2):
        env.place_order(True, 10, 101, 20)   # agent a
        env.place_order(False, 10, 101, 40)  # agent b
        env.step()
        assert (env.book.ask_vol(), env.book.bid_vol()) == (10 * k, 10 * k)
        assert env.book.bid_ask() == (20, 40)


# --------------------------------------------------------- RandomAgents structure tests
def test_random_agents_activity_rate(oracle):  # C.15 random_agent.rs:255-267
    env = oracle.StepEnv(101, 0, 1, 1000)
    a0 = oracle.RandomAgentSet([(2, (10, 20), (20, 30), 1, 0.0)])
    a0.update(env)
    assert env.n_transactions() == 0
    a1 = oracle.RandomAgentSet([(2, (10, 20), (20, 30), 1, 1.0)])
    a1.update(env)
    assert env.n_transactions() == 2


def test_random_agents_place_then_cancel(oracle):  # C.15 random_agent.rs:270-296
    env = oracle.StepEnv(101, 0, 1, 1000)
    ag = oracle.RandomAgentSet([(1, (10, 20), (20, 30), 1, 1.0)])
    ag.update(env)
    assert env.transaction_kinds().tolist() == [0] and ag.held_ids(0).tolist() == [0]
    env.step()
    ag.update(env)
    assert env.transaction_kinds().tolist() == [1]
    env.step()
    ag.update(env)
    assert env.transaction_kinds().tolist() == [0] and ag.held_ids(0).tolist() == [1]
    o = env.book.orders_array()
    assert 10 <= int(o["price"][0]) < 20 and 20 <= int(o["start_vol"][0]) < 30  # ranges only (RNG unpinned)


# ------------------------------------------------------------- Python-surface vectors
def test_python_step_env(oracle):  # C.12 tests/test_step_sim/test_env.py:7-106
    env = oracle.StepEnv(101, 0, 1, 100_000)
    env.place_order(True, 100, 101, price=50)
    env.place_order(False, 100, 101, price=60)
    env.step()
    assert env.bid_ask == (50, 60) and (env.ask_vol, env.bid_vol, env.time) == (100, 100, 100_000)
    env.place_order(True, 100, 101, price=55)
    env.place_order(False, 100, 101, price=65)
    env.step()
    assert env.bid_ask == (55, 60) and (env.ask_vol, env.bid_vol, env.time) == (200, 200, 200_000)
    env.place_order(True, 150, 101)
    env.step()
    assert env.bid_ask == (55, 65) and (env.ask_vol, env.bid_vol, env.time) == (50, 200, 300_000)
    env.step()
    d = env.get_market_data()
    keys = {"bid_price", "ask_price", "bid_vol", "ask_vol", "trade_vol"}
    for i in range(10):
        keys |= {f"bid_vol_{i}", f"ask_vol_{i}", f"n_bid_{i}", f"n_ask_{i}"}
    assert set(d) == keys
    assert d["bid_price"].tolist() == [50, 55, 55, 55] and d["ask_price"].tolist() == [60, 60, 65, 65]
    assert d["bid_vol"].tolist() == [100, 200, 200, 200] and d["ask_vol"].tolist() == [100, 200, 50, 50]
    assert d["bid_vol_0"].tolist() == [100] * 4 and d["ask_vol_0"].tolist() == [100, 100, 50, 50]
    assert d["n_bid_0"].tolist() == [1] * 4 and d["n_ask_0"].tolist() == [1] * 4
    assert d["trade_vol"].tolist() == [0, 0, 150, 0]


def test_python_numpy_api(oracle):  # C.13 tests/test_step_sim/test_numpy_api.py:7-71
    env = oracle.StepEnvNumpy(101, 0, 1, 100_000)
    sides = np.array([True, True, True, False, False, False])
    vols = np.array([10, 11, 12, 10, 11, 12], dtype=np.uint32)
    tr = np.array([1, 1, 1, 2, 2, 2], dtype=np.uint32)
    prices = np.array([20, 20, 19, 22, 22, 23], dtype=np.uint32)
    ids = env.submit_limit_orders((sides, vols, tr, prices))
    env.step()
    assert ids.tolist() == list(range(6))
    assert env.level_1_data().tolist() == [0, 20, 22, 33, 33, 21, 2, 21, 2]
    l2 = env.level_2_data()
    assert l2.shape == (45,)
    assert l2[:13].tolist() == [0, 20, 22, 33, 33, 21, 2, 21, 2, 12, 1, 12, 1] and not l2[13:].any()
    env.submit_cancellations(np.array([0, 1, 3, 4], dtype=np.uint64))
    env.step()
    l1 = env.level_1_data()
    assert (l1[1], l1[2]) == (19, 23) and (l1[5], l1[6]) == (12, 1) and (l1[7], l1[8]) == (12, 1)
    bad = oracle.StepEnvNumpy(101, 0, 2, 100_000)
    with pytest.raises(ValueError):
        bad.submit_limit_orders((sides[:2], vols[:2], tr[:2], np.array([20, 21], dtype=np.uint32)))
    assert bad.n_transactions() == 1  # first element stayed created + queued (step_sim_numpy.rs:167-177)


def test_market_env_three_steps(oracle):  # crates/step_sim/src/market_env.rs:342-407 (MarketEnv<2>, one queue for both assets)
    MAXP = 2**32 - 1
    m = oracle.ManyMarkets(1, 101, 0, [1, 1], 1000, True, 10)
    m.place_order(0, 0, True, 10, 101, 10)
    m.place_order(0, 0, False, 20, 101, 20)
    m.step()
    assert m.book(0, 0).bid_ask() == (10, 20) and m.book(0, 1).bid_ask() == (0, MAXP)
    o = m.book(0, 0).orders_array()
    assert len(o) == 2 and o["status"].tolist() == [1, 1]
    m.place_order(0, 0, True, 10, 101, 11)
    m.place_order(0, 0, False, 20, 101, 21)
    m.step()
    assert m.book(0, 0).bid_ask() == (11, 20) and m.book(0, 0).n_orders() == 4
    m.place_order(0, 0, True, 30, 101, None)
    m.step()
    assert m.book(0, 0).bid_ask() == (11, 21) and (m.book(0, 0).ask_vol(), m.book(0, 1).ask_vol()) == (10, 0)
    o = m.book(0, 0).orders_array()
    assert len(o) == 5 and o["status"][1] == 2 and o["status"][4] == 2
    assert len(m.book(0, 0).trades_array()) == 2
    h = m.history()[:, 0]
    assert h[:, 1].tolist() == [10, 11, 11] and h[:, 2].tolist() == [20, 20, 21]
    assert h[:, 4].tolist() == [10, 20, 20] and h[:, 3].tolist() == [20, 40, 10]
    assert h[:, 5].tolist() == [10, 10, 10] and h[:, 7].tolist() == [20, 20, 10]
    assert h[:, 6].tolist() == [1, 1, 1] and h[:, 8].tolist() == [1, 1, 1]
    assert h[:, 0].tolist() == [0, 0, 30]
    assert m.history()[:, 1, :5].tolist() == [[0, 0, MAXP, 0, 0]] * 3


def test_market_shares_one_queue_and_clock(oracle):
    """market_env.rs:110-121: one shuffle over all assets' events, event i stamped t0 + i on EVERY book's clock."""
    m = oracle.ManyMarkets(1, 3, 0, [1, 2], 1000, True, 4)
    for k in range(6):
        m.place_order(0, k % 2, True, 5, 0, 10 + 2 * k)
    m.step()
    t = sorted(m.book(0, 0).orders_array()["arr_time"].tolist() + m.book(0, 1).orders_array()["arr_time"].tolist())
    assert t == [0, 1, 2, 3, 4, 5]  # a permutation of the global positions, split between the two books
    with pytest.raises(ValueError):
        m.place_order(0, 1, True, 5, 0, 11)  # asset 1 has tick 2 (Market::new(_, [1, 2], _), market.rs:74-81)


def test_random_market_agents_equal_single_book_agents(oracle):
    """RandomMarketAgents on a one-asset market == RandomAgents on an Env (random_agent.rs:84-120 vs :204-247)."""
    g = [(12, (40, 60), (10, 20), 2, 0.8), (7, (45, 55), (1, 5), 2, 0.4)]
    a = oracle.ManyBooks(3, 11, 0, 2, 100_000, True, 8, g)
    b = oracle.ManyMarkets(3, 11, 0, [2], 100_000, True, 8, [(0,) + x for x in g])
    a.run(30)
    b.run(30)
    assert np.array_equal(a.history(), b.history()) and np.array_equal(a.rng_states(), b.rng_states())


def test_json_snapshot_serde_layout(oracle, tmp_path):
    """serde_json layout of OrderBook (orderbook.rs:93-112, OrderEntry :33-39, Order types.rs:78-99, Trade :104-118,
    unit enums as strings, OrderKey = (Side, u32, u64) as an array; ask_side/bid_side skipped): written out by hand."""
    ob = oracle.OrderBook(5, 1)
    ob.place_order(True, 10, 7, 50)      # id 0 rests: key (Bid, u32::MAX - 50, 5)
    ob.set_time(6)
    ob.place_order(False, 4, 8, None)    # id 1 market sell, fills 4 @ 50: key stays provisional (Ask, 0, 0)
    p = tmp_path / "s.json"
    ob.save_json_snapshot(str(p))
    assert p.read_text() == (
        '{"t":6,"tick_size":1,"trade_vol":4,"orders":['
        '{"order":{"side":"Bid","status":"Active","arr_time":5,"end_time":18446744073709551615,"vol":6,"start_vol":10,'
        '"price":50,"trader_id":7,"order_id":0},"key":["Bid",4294967245,5]},'
        '{"order":{"side":"Ask","status":"Filled","arr_time":6,"end_time":6,"vol":0,"start_vol":4,'
        '"price":0,"trader_id":8,"order_id":1},"key":["Ask",0,0]}],'
        '"trades":[{"t":6,"side":"Bid","price":50,"vol":4,"active_order_id":1,"passive_order_id":0}],"trading":true}')
    lb = oracle.order_book_from_json(str(p))  # TryFrom<OrderBookState>, orderbook.rs:891-918
    assert lb.state() == ob.state() and lb.bid_ask() == ob.bid_ask() == (50, 2**32 - 1)
    assert lb.best_bid_vol_and_orders() == (6, 1)
    lb.set_time(7)
    ob.set_time(7)
    assert lb.place_order(False, 6, 9, 50) == ob.place_order(False, 6, 9, 50) == 2
    assert lb.state() == ob.state()


def test_rng_pin_file_is_what_the_oracle_produces(oracle):
    # tests/golden/rng_pin_expected.txt is the file integration/rust/pin_rng (the real rand / rand_xoshiro / rand_distr
    # crates) is diffed against; it must stay in sync with the oracle's restated sampling.
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rng_pin", os.path.join(root, "tools", "gen_rng_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = open(os.path.join(root, "tests", "golden", "rng_pin_expected.txt")).read().splitlines()
    assert mod.lines() == want
    # published known answers the file must carry: SplitMix64 seeding of xoroshiro128** (rand_core seed_from_u64)
    assert want[0].startswith("next_u64: ")
    assert len(want) == 12


def test_native_submit_instructions_equals_the_python_loop(oracle):
    """oracle.StepEnvNumpy.submit_instructions_native (the loop over the six arrays inside the library, as the reference's runs in
    Rust: rust/src/step_sim_numpy.rs:233-275) == submit_instructions (the same loop in Python): ids, the stop at a bad price
    with the earlier elements queued, every step's level 2, trades and orders.  bench.py's INGRESS CPU baseline times the native one."""
    import numpy as np

    rng = np.random.default_rng(11)
    a, b = oracle.StepEnvNumpy(3, 0, 2, 100_000), oracle.StepEnvNumpy(3, 0, 2, 100_000)
    made = 0
    for s in range(12):
        n = int(rng.integers(0, 40))
        action = rng.choice([0, 1, 2, 3], size=n, p=[0.05, 0.6, 0.3, 0.05]).astype(np.uint32)
        if made == 0:
            action[action == 2] = 1
        ins = (action, rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(1, 30, size=n).astype(np.uint32),
               rng.integers(0, 9, size=n).astype(np.uint32), (rng.integers(45, 56, size=n) * 2).astype(np.uint32),
               rng.integers(0, max(1, made), size=n).astype(np.uint64))
        if s == 5 and (action == 1).sum() > 2:
            ins[4][np.nonzero(action == 1)[0][2]] += 1  # an odd price at tick size 2: both stop there, the two before it stay queued
            with pytest.raises(ValueError):
                a.submit_instructions(ins)
            with pytest.raises(ValueError):
                b.submit_instructions_native(ins)
        else:
            ia, ib = a.submit_instructions(ins), b.submit_instructions_native(ins)
            assert np.array_equal(ia, ib)
        made = a.book.n_orders()
        assert made == b.book.n_orders() and a.n_transactions() == b.n_transactions()
        a.step(), b.step()
        assert np.array_equal(a.level_2_data(), b.level_2_data())
    assert np.array_equal(a.book.trades_array(), b.book.trades_array()) and np.array_equal(a.book.orders_array(), b.book.orders_array())
    assert a.book.n_trades() > 20
