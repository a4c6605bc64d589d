"""Agent interfaces (ref src/bourse/step_sim/agents/base_agent.py:11-116)."""
import typing

import numpy as np

InstructionArrays = typing.Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]


class BaseAgent:
    """Object-API agent: ``update(rng, env)`` submits instructions to a ``StepEnv``."""

    def update(self, rng: np.random.Generator, env):
        raise NotImplementedError


class BaseNumpyAgent:
    """Array-API agent: ``update(rng, level_2_data)`` returns six parallel arrays
    ``(action u32 {0 none, 1 new, 2 cancel}, side bool (True = bid), vol u32, trader_id u32,
    price u32, order_id u64)``; ``level_2_data`` is the u32[45] array
    ``[trade_vol, bid, ask, ask_vol, bid_vol, (bid_vol_i, n_bid_i, ask_vol_i, n_ask_i) x 10]``."""

    def update(self, rng: np.random.Generator, level_2_data: np.ndarray) -> InstructionArrays:
        raise NotImplementedError
