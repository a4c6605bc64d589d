"""Oracle NoiseAgent / MomentumAgent / sampling: the reference's own structural tests
re-expressed, plus checks of the restated rand_distr sampling and of the portable math both the oracle and the HIP
path use.  RNG-dependent values are PARITY UNPINNED against Rust (see oracle/bourse_oracle_agents.hpp)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOISE = dict(tick_size=2, p_limit=0.2, p_market=0.2, p_cancel=0.1, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
MOM = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=5.0, scale=0.5, order_ratio=1.0,
           price_dist_mu=0.0, price_dist_sigma=10.0)  # doc example, ref crates/step_sim/src/lib.rs:53-73


def test_pm_math_copies_identical_and_close_to_libm(oracle):
    a = open(os.path.join(ROOT, "oracle", "pm_math.hpp")).read()
    b = open(os.path.join(ROOT, "bourse_amd", "csrc", "pm_math.hpp")).read()
    assert a == b
    ta = open(os.path.join(ROOT, "oracle", "zig_norm_tables.inc")).read()
    tb = open(os.path.join(ROOT, "bourse_amd", "csrc", "zig_norm_tables.inc")).read()
    assert ta == tb
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-40, 40, 4000), rng.uniform(-1, 1, 2000), [0.0, 1.0, -1.0, 700.0, -700.0, 1e-300]])
    for x in xs:
        got, ref = L.orc_pm_exp(float(x)), float(np.exp(x))
        assert abs(got - ref) <= 2 * np.spacing(ref), x
    for x in np.concatenate([rng.uniform(1e-12, 10, 3000), np.exp(rng.uniform(-700, 700, 2000)), [1.0, 5e-324]]):
        got, ref = L.orc_pm_log(float(x)), float(np.log(x))
        assert abs(got - ref) <= 2 * np.spacing(abs(ref)) + 1e-300, x
    for x in np.concatenate([rng.uniform(-25, 25, 3000), rng.uniform(-0.2, 0.2, 3000), [0.0, 0.125, -0.125]]):
        got, ref = L.orc_pm_tanh(float(x)), float(np.tanh(x))
        assert abs(got - ref) <= 4 * np.spacing(abs(ref)) + 1e-300, x


def test_rounding_helpers(oracle):  # ref crates/step_sim/src/agents/common.rs:268-305
    L = oracle.lib()
    up, down = L.orc_round_price_up, L.orc_round_price_down
    assert [up(5.0, 2.0), up(2.1, 2.0), up(3.9, 4.0), up(-2.2, 4.0), up(1.0 + 2.0**32, 4.0)] == [6, 4, 4, 0, 2**32 - 1]
    assert [down(5.0, 2.0), down(2.1, 2.0), down(3.9, 4.0), down(-2.2, 4.0), down(1.0 + 2.0**32, 4.0)] == [4, 2, 0, 0, 2**32 - 1]


def test_sampling_restatements(oracle):
    r = oracle.Rng(seed=3)
    q = oracle.Rng(seed=3)
    for _ in range(100):  # Standard f64: 53 bits * 2^-53
        assert r.gen_f64() == (q.next_u64() >> 11) * 2.0**-53
    z = np.array([oracle.Rng(seed=s).std_normal() for s in range(2000)] + [r.std_normal() for _ in range(20000)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03 and abs((z**3).mean()) < 0.1
    assert abs((np.abs(z) > 1.959964).mean() - 0.05) < 0.01 and (np.abs(z) > 3.6541528853610088).sum() >= 1  # tail branch hit
    ln = np.log([r.lognormal(0.5, 0.25) for _ in range(20000)])
    assert abs(ln.mean() - 0.5) < 0.01 and abs(ln.std() - 0.25) < 0.01


def test_noise_agent_init_place_and_cancel(oracle):  # ref noise_agent.rs:361-425
    env = oracle.StepEnv(101, 0, 1, 1_000_000)
    p = dict(NOISE, p_limit=1.0, p_market=0.0, p_cancel=1.0, price_dist_sigma=10.0)
    ag = oracle.AgentSet([("noise", 10, 10, p)])
    ag.update(env)
    assert len(ag.order_list(0)) == 10 and env.n_transactions() == 10 and set(env.transaction_kinds().tolist()) == {0}
    mid = env.book.mid_price()
    o = env.book.orders_array()
    assert o["trader_id"].tolist() == list(range(10, 20)) and set(o["vol"].tolist()) == {100}
    for r in o:
        assert r["price"] % 2 == 0
        assert (r["price"] <= mid) if r["side"] else (r["price"] >= mid)
    env.step()
    ag.set_noise_prob(0, "p_limit", 0.0)
    ag.update(env)
    assert len(ag.order_list(0)) == 0
    env.step()
    assert [env.order_status(i) for i in range(10)] == [3] * 10


def test_cancel_live_orders_extremes(oracle):  # ref common.rs:308-330
    for p_cancel, n_left in ((0.0, 10), (1.0, 0)):
        env = oracle.StepEnv(101, 0, 1, 1_000_000)
        ag = oracle.AgentSet([("noise", 0, 10, dict(NOISE, p_limit=1.0, p_market=0.0, p_cancel=p_cancel, tick_size=1))])
        ag.update(env)
        env.step()
        ag.set_noise_prob(0, "p_limit", 0.0)
        live_before = int((env.book.orders_array()["status"] == 1).sum())
        ag.update(env)
        assert len(ag.order_list(0)) == (live_before if n_left else 0)
        assert env.n_transactions() == (0 if n_left else live_before)


def test_momentum_agent_first_update_places_nothing(oracle):  # ref momentum_agent.rs:417-443
    env = oracle.StepEnv(101, 0, 1, 1_000_000)
    env.place_order(True, 100, 0, 1000)
    env.place_order(False, 100, 0, 1020)
    env.step()
    ag = oracle.AgentSet([("momentum", 10, 100, MOM)])
    ag.update(env)
    assert env.n_transactions() == 0


def test_momentum_agents_never_sell_quirk_and_doc_example_runs(oracle):
    # doc example (ref crates/step_sim/src/lib.rs:37-88): MomentumAgent(0,10) + NoiseAgent(10,20), 50 steps, seed 101
    m = oracle.ManyBooks(3, 101, 0, 1, 1_000_000, True, 10, members=[("momentum", 0, 10, MOM), ("noise", 10, 20, NOISE)])
    m.run(50, 1)
    h = m.history()
    assert h.shape == (50, 3, 45) and m.trade_counts().sum() > 0
    o = m.book(0).orders_array()
    mom = o[o["trader_id"] < 10]
    assert len(mom) > 0 and mom["side"].all()  # p_market < 0 when momentum < 0: momentum agents only ever buy
    assert np.all(o["price"][(o["price"] != 0) & (o["price"] != 2**32 - 1)] % 2 == 0)


def test_market_twins_on_a_one_asset_market_equal_the_single_asset_agents(oracle):
    """NoiseMarketAgent / MomentumMarketAgent (noise_agent.rs:281-339, momentum_agent.rs:328-396) run the same update as
    NoiseAgent / MomentumAgent against `env.get_market().get_order_book(asset)`: on MarketEnv<1> they are identical."""
    noise = dict(tick_size=2, p_limit=0.2, p_market=0.2, p_cancel=0.1, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
    mom = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=5.0, scale=0.5, order_ratio=1.0,
               price_dist_mu=0.0, price_dist_sigma=10.0)
    members = [("momentum", 0, 10, mom), ("noise", 10, 20, noise), ("random", 8, (1000, 1010), (5, 9), 1, 0.5)]
    a = oracle.ManyBooks(3, 11, 0, 1, 1_000_000, True, 10, members=members)
    b = oracle.ManyMarkets(3, 11, 0, [1], 1_000_000, True, 10, members=[(0, m) for m in members])
    a.run(40)
    b.run(40)
    assert np.array_equal(a.history(), b.history()) and np.array_equal(a.rng_states(), b.rng_states())
    assert a.trade_counts().sum() > 0


def test_market_twins_trade_their_own_asset_only(oracle):
    noise = dict(tick_size=1, p_limit=0.5, p_market=0.2, p_cancel=0.1, trade_vol=10, price_dist_mu=0.0, price_dist_sigma=1.0)
    m = oracle.ManyMarkets(2, 5, 0, [1, 1, 1], 1_000_000, True, 10, members=[(2, ("noise", 0, 10, noise))])
    m.run(20)
    for mk in range(2):
        assert m.book(mk, 0).n_orders() == 0 and m.book(mk, 1).n_orders() == 0 and m.book(mk, 2).n_orders() > 0
