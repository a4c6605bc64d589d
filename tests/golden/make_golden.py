"""Generate the golden fixtures under tests/golden/ (build container only; committed output).

Inputs come from the REFERENCE'S OWN Python callers, imported from /root/reference where they lie
(src/bourse/step_sim/runner.py, agents/random_agent.py, examples/random_trades.py), with the Rust
extension `bourse.core` supplied by the CPU oracle (oracle/run_reference_pytests.install_core_shim).
The fixtures are DATA: instruction streams the reference's agents emit for numpy's PCG64 seed 101, and
the outputs the oracle produced for them.  They pin (i) our own runner/agents against the reference's
Python behaviour (draw order, loop order), (ii) the oracle against regressions, (iii) the GPU path.
Caveat: RNG-dependent OUTPUTS (shuffle) are oracle outputs, not Rust outputs.

Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import run_reference_pytests as shim  # noqa: E402

core = shim.install_core_shim()
import bourse  # noqa: E402  (the reference's Python package, oracle-backed core)
from bourse.step_sim.agents import NumpyRandomAgents, RandomAgent  # noqa: E402


class Recorder:
    """Wraps an oracle StepEnv and records every instruction the reference's agents submit."""

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


def c1_random_trades():
    # examples/random_trades.py:7-16 with BASELINE's run(101, 200, 50)
    seed, n_steps, n_agents, tick = 101, 200, 50, 2
    agents = [RandomAgent(i, 0.5, (10, 100), (20, 50), tick) for i in range(n_agents)]
    env = core.StepEnv(seed, 0, tick, 100_000)
    data = bourse.step_sim.run(env, agents, n_steps, seed, show_progress=False)
    # same run again with a recording proxy to capture the instruction stream (runner.py:114-118 loop order)
    env2 = core.StepEnv(seed, 0, tick, 100_000)
    rec = Recorder(env2)
    agents = [RandomAgent(i, 0.5, (10, 100), (20, 50), tick) for i in range(n_agents)]
    rng = np.random.default_rng(seed)
    for s in range(n_steps):
        rec.step_no = s
        for a in agents:
            a.update(rng, rec)
        env2.step()
    assert all(np.array_equal(data[k], env2.get_market_data()[k]) for k in data)
    out = {f"md_{k}": v for k, v in data.items()}
    out["instructions"] = np.array(rec.log, dtype=np.int64)  # (step, action, bid, vol, trader, price, order_id)
    out["trades"] = np.array(env.get_trades(), dtype=np.uint64)
    out["orders"] = np.array(env.get_orders(), dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "c1_random_trades.npz"), **out)
    print("c1:", len(rec.log), "instructions,", len(env.get_trades()), "trades,", len(env.get_orders()), "orders")


def numpy_agents():
    # tests/test_step_sim/test_benchmarks.py:34-48 shape, smaller: NumpyRandomAgents over StepEnvNumpy
    seed, n_steps = 101, 40
    ag = NumpyRandomAgents(30, (10, 100), (20, 50), 2)
    env = core.StepEnvNumpy(seed, 0, 2, 100_000)
    rng = np.random.default_rng(seed)
    ins = []
    for _ in range(n_steps):
        i = ag.update(rng, env.level_2_data())
        ins.append(np.stack([np.asarray(x).astype(np.uint64) for x in i]))
        env.submit_instructions(i)
        env.step()
    out = {f"md_{k}": v for k, v in env.get_market_data().items()}
    out["instructions"] = np.stack(ins)  # [step, 6, n]
    out["trades"] = np.array(env.get_trades(), dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "numpy_random_agents.npz"), **out)
    print("numpy agents:", out["instructions"].shape, len(env.get_trades()), "trades")


def rust_random_agents():
    # crates/step_sim/examples/random_agents (sim_runner + RandomAgents), scaled; ORACLE outputs (RNG unpinned)
    import pyoracle

    groups = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]
    m = pyoracle.ManyBooks(4, 101, 0, 2, 100_000, True, 16, groups)
    m.run(25, 1)
    np.savez_compressed(os.path.join(HERE, "oracle_random_agents_c2x4.npz"), history=m.history(),
                        rng=m.rng_states(), trade_counts=m.trade_counts(), trades0=m.book(0).trades_array())
    print("oracle c2x4:", m.history().shape, m.trade_counts())


def orderbook_snapshot():
    # a serde-layout OrderBook snapshot (orderbook.rs:93-112) with every order status, re-keyed and reduced orders
    import pyoracle

    rng = np.random.default_rng(7)
    ob = pyoracle.OrderBook(100, 2)
    t = 100
    for _ in range(60):
        t += int(rng.integers(1, 4))
        ob.set_time(t)
        n, kind = ob.n_orders(), rng.random()
        if kind < 0.6 or n == 0:
            ob.place_order(bool(rng.integers(0, 2)), int(rng.integers(1, 40)), int(rng.integers(0, 9)),
                           None if rng.random() < 0.1 else int(rng.integers(45, 56)) * 2)
        elif kind < 0.8:
            ob.cancel_order(int(rng.integers(0, n)))
        else:
            ob.modify_order(int(rng.integers(0, n)), None if rng.random() < 0.4 else int(rng.integers(45, 56)) * 2,
                            None if rng.random() < 0.4 else int(rng.integers(1, 40)))
    ob.disable_trading()
    ob.set_time(t + 1)
    ob.place_order(True, 5, 1, None)  # Rejected
    ob.enable_trading()
    ob.save_json_snapshot(os.path.join(HERE, "orderbook_snapshot.json"))
    print("orderbook snapshot:", ob.n_orders(), "orders", len(ob.trades_array()), "trades", ob.bid_ask())


if __name__ == "__main__":
    orderbook_snapshot()
    c1_random_trades()
    numpy_agents()
    rust_random_agents()
