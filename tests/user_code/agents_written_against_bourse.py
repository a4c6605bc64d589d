"""USER CODE as it would be written for the reference's Python package - it imports `bourse`, never `bourse_amd`.
tests/test_install_as_bourse.py runs it unmodified after bourse_amd.install_as_bourse().

The workload is BASELINE configs[0] (the reference's examples/random_trades.py shape: run(101, 200, 50)) plus a
user-defined agent class deriving from the reference's BaseAgent."""
import bourse
from bourse.step_sim.agents import BaseAgent, RandomAgent

TICK = 2


def random_trades(seed=101, n_steps=200, n_agents=50):
    traders = [RandomAgent(i, 0.5, (10, 100), (20, 50), TICK) for i in range(n_agents)]
    env = bourse.core.StepEnv(seed, 0, TICK, 100_000)
    data = bourse.step_sim.run(env, traders, n_steps, seed, show_progress=False)
    return env, data


class Pinger(BaseAgent):
    """Places one bid every step and cancels the previous one - enough to show that a user's BaseAgent subclass passes
    the runner's interface check."""

    def __init__(self, trader_id, price):
        self.trader_id, self.price, self.last = trader_id, price, None

    def update(self, rng, env):
        if self.last is not None:
            env.cancel_order(self.last)
        self.last = env.place_order(True, 1 + int(rng.integers(0, 5)), self.trader_id, price=self.price)


def pingers(n_steps=12):
    env = bourse.core.StepEnv(7, 0, TICK, 1000)
    data = bourse.step_sim.run(env, [Pinger(0, 40), Pinger(1, 44)], n_steps, 3, show_progress=False)
    return env, data, bourse.data_processing.orders_to_dataframe(env.get_orders())
