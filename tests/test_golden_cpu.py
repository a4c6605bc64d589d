"""CPU tests against the committed golden fixtures (tests/golden/*.npz, made by make_golden.py).

* our own runner/agents (bourse_amd.step_sim) emit EXACTLY the instruction stream the reference's Python
  agents emitted (numpy PCG64 seed 101): pins draw order + loop order of the host-side mirror;
* the oracle reproduces the recorded outputs from the recorded instructions (regression pin).
"""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Rec:
    def __init__(self, env):
        self.env, self.log, self.step_no = env, [], 0

    def order_status(self, i):
        return self.env.order_status(i)

    def place_order(self, bid, vol, trader_id, price=None):
        oid = self.env.place_order(bid, vol, trader_id, price=price)
        self.log.append((self.step_no, 1, int(bool(bid)), int(vol), int(trader_id), int(price), oid))
        return oid

    def cancel_order(self, i):
        self.env.cancel_order(i)
        self.log.append((self.step_no, 2, 0, 0, 0, 0, int(i)))


def test_c1_our_random_agent_emits_reference_instruction_stream(oracle):
    from bourse_amd.step_sim.agents import RandomAgent

    fx = np.load(os.path.join(G, "c1_random_trades.npz"))
    env = oracle.StepEnv(101, 0, 2, 100_000)
    rec = _Rec(env)
    agents = [RandomAgent(i, 0.5, (10, 100), (20, 50), 2) for i in range(50)]
    rng = np.random.default_rng(101)
    for s in range(200):
        rec.step_no = s
        for a in agents:
            a.update(rng, rec)
        env.step()
    assert np.array_equal(np.array(rec.log, dtype=np.int64), fx["instructions"])
    md = env.get_market_data()
    assert len(md) == 45
    for k, v in md.items():
        assert np.array_equal(v, fx[f"md_{k}"]), k
    assert np.array_equal(np.array(env.get_trades(), dtype=np.uint64), fx["trades"])
    assert np.array_equal(np.array(env.get_orders(), dtype=np.uint64), fx["orders"])


def test_numpy_random_agents_emit_reference_instruction_stream(oracle):
    from bourse_amd.step_sim.agents import NumpyRandomAgents

    fx = np.load(os.path.join(G, "numpy_random_agents.npz"))
    ag = NumpyRandomAgents(30, (10, 100), (20, 50), 2)
    env = oracle.StepEnvNumpy(101, 0, 2, 100_000)
    rng = np.random.default_rng(101)
    for s in range(40):
        ins = ag.update(rng, env.level_2_data())
        assert ins[0].dtype == np.uint32 and ins[1].dtype == bool and ins[5].dtype == np.uint64
        got = np.stack([np.asarray(x).astype(np.uint64) for x in ins])
        assert np.array_equal(got, fx["instructions"][s]), f"step {s}"
        env.submit_instructions(ins)
        env.step()
    for k, v in env.get_market_data().items():
        assert np.array_equal(v, fx[f"md_{k}"]), k
    assert np.array_equal(np.array(env.get_trades(), dtype=np.uint64), fx["trades"])


def test_oracle_random_agents_regression(oracle):
    fx = np.load(os.path.join(G, "oracle_random_agents_c2x4.npz"))
    groups = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]
    m = oracle.ManyBooks(4, 101, 0, 2, 100_000, True, 16, groups)
    m.run(25, 2)
    assert np.array_equal(m.history(), fx["history"])
    assert np.array_equal(m.rng_states(), fx["rng"])
    assert np.array_equal(m.trade_counts(), fx["trade_counts"])
    t = m.book(0).trades_array()
    for f in t.dtype.names:
        assert np.array_equal(t[f], fx["trades0"][f])


def test_many_books_equals_single_env_sim_runner(oracle):
    # the many-book oracle runner is B copies of sim_runner(env, agents, seed + b, n)  (runner.rs:46-69)
    groups = [(5, (10, 20), (1, 9), 1, 0.7), (3, (12, 18), (2, 5), 2, 0.3)]
    m = oracle.ManyBooks(3, 7, 0, 1, 1000, True, 10, groups)
    m.run(30, 1)
    for b in range(3):
        env = oracle.StepEnv(0, 0, 1, 1000)
        ag = oracle.RandomAgentSet(groups)
        st = oracle.sim_runner(env, ag, 7 + b, 30)
        assert np.array_equal(env.history(), m.history()[:, b])
        assert tuple(st) == tuple(m.rng_states()[b])
        assert env.get_trades() == m.book(b).get_trades()


def test_golden_orderbook_snapshot_round_trips_through_the_oracle(oracle):
    import json
    path = os.path.join(G, "orderbook_snapshot.json")
    s = json.load(open(path))
    ob = oracle.order_book_from_json(path)
    assert ob.state() == s
    assert {e["order"]["status"] for e in s["orders"]} == {"Active", "Filled", "Cancelled", "Rejected"}
