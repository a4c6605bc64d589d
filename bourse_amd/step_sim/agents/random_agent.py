"""Uniform random order flow (ref src/bourse/step_sim/agents/random_agent.py:12-166).

The numpy draw ORDER is part of the behaviour (it fixes the instruction stream for a seed):
``RandomAgent``: random() -> integers(tick) -> integers(vol) -> choice(side)   (:77-91)
``NumpyRandomAgents``: choice(sides) -> integers(*tick_range) for VOLS (sic, :152) -> integers(tick)*tick_size
"""
import typing

import numpy as np

from .base_agent import BaseAgent, BaseNumpyAgent, InstructionArrays


class RandomAgent(BaseAgent):
    def __init__(self, i: int, activity_rate: float, tick_range: typing.Tuple[int, int],
                 vol_range: typing.Tuple[int, int], tick_size: int):
        self.i = i
        self.activity_rate = activity_rate
        self.tick_range = tick_range
        self.vol_range = vol_range
        self.tick_size = tick_size
        self.order_id = None

    def update(self, rng: np.random.Generator, env):
        if rng.random() >= self.activity_rate:
            return
        if self.order_id is not None and env.order_status(self.order_id) == 1:
            env.cancel_order(self.order_id)  # a live order: cancel it and forget the id
            self.order_id = None
            return
        tick = rng.integers(*self.tick_range)
        vol = rng.integers(*self.vol_range)
        side = bool(rng.choice([True, False]))
        self.order_id = env.place_order(side, vol, self.i, price=tick * self.tick_size)


class NumpyRandomAgents(BaseNumpyAgent):
    def __init__(self, n_agents: int, tick_range: typing.Tuple[int, int], vol_range: typing.Tuple[int, int],
                 tick_size: int):
        self.n_agents = n_agents
        self.tick_range = tick_range
        self.vol_range = vol_range
        self.tick_size = tick_size

    def update(self, rng: np.random.Generator, level_2_data: np.ndarray) -> InstructionArrays:
        n = self.n_agents
        sides = rng.choice([True, False], size=n).astype(bool)
        vols = rng.integers(*self.tick_range, size=n, dtype=np.uint32)  # the reference samples vols from tick_range
        prices = rng.integers(*self.tick_range, size=n, dtype=np.uint32) * self.tick_size
        return (np.ones(n, dtype=np.uint32), sides, vols, np.arange(n, dtype=np.uint32), prices,
                np.zeros(n, dtype=np.uint64))


class ManyBookNumpyRandomAgents:
    """``NumpyRandomAgents`` vectorised over books for ``run_many``: ``update_many(rng, level_2_data[B, W])`` draws the orders
    of ``n_agents`` agents for EVERY book in one go (same distributions as the one-book agent - including the reference's
    quirk that volumes come from ``tick_range``, random_agent.py:151-157 - but one generator stream over all books, so the
    draws are not those of B one-book agents called in turn)."""

    def __init__(self, n_agents: int, tick_range: typing.Tuple[int, int], vol_range: typing.Tuple[int, int], tick_size: int):
        self.n_agents, self.tick_range, self.vol_range, self.tick_size = n_agents, tick_range, vol_range, tick_size

    def update_many(self, rng: np.random.Generator, level_2_data: np.ndarray):
        n_books = level_2_data.shape[0]
        n = n_books * self.n_agents
        sides = rng.integers(0, 2, size=n, dtype=np.uint8)
        vols = rng.integers(*self.tick_range, size=n, dtype=np.uint32)
        prices = rng.integers(*self.tick_range, size=n, dtype=np.uint32) * np.uint32(self.tick_size)
        traders = np.tile(np.arange(self.n_agents, dtype=np.uint32), n_books)
        return (np.full(n_books, self.n_agents, dtype=np.uint64),
                (np.ones(n, dtype=np.uint32), sides, vols, traders, prices, np.zeros(n, dtype=np.uint64)))
