"""``run``: fixed-number-of-steps simulation loop (ref src/bourse/step_sim/runner.py:12-120)."""
from __future__ import annotations

import typing

import numpy as np

from .. import core
from . import agents as _agents


def run(env, agents: typing.Iterable, n_steps: int, seed: int, show_progress: bool = True,
        use_numpy: bool = False) -> typing.Dict[str, np.ndarray]:
    """Each step: every agent updates (object API: ``agent.update(rng, env)``; numpy API:
    ``env.submit_instructions(agent.update(rng, level_2_data))``), then ``env.step()``.
    The agents' generator is ``numpy.random.default_rng(seed)``; the env's own xoroshiro
    stream only drives the shuffle.  Returns ``env.get_market_data()``."""
    agents = list(agents)
    if use_numpy:
        assert isinstance(env, core.StepEnvNumpy)
        assert all(isinstance(a, _agents.BaseNumpyAgent) for a in agents), \
            "Agents should implement bourse_amd.step_sim.agents.BaseNumpyAgent"
    else:
        assert isinstance(env, core.StepEnv)
        assert all(isinstance(a, _agents.BaseAgent) for a in agents), \
            "Agents should implement bourse_amd.step_sim.agents.BaseAgent"
    rng = np.random.default_rng(seed)
    steps = range(n_steps)
    if show_progress:
        try:
            import tqdm

            steps = tqdm.trange(n_steps)
        except ImportError:
            pass
    for _ in steps:
        if use_numpy:
            level_2_data = env.level_2_data()
            for agent in agents:
                env.submit_instructions(agent.update(rng, level_2_data))
        else:
            for agent in agents:
                agent.update(rng, env)
        env.step()
    return env.get_market_data()
