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


def _csr_for_all_books(agent, rng, level_2_data: np.ndarray):
    """One agent -> one CSR batch for ALL books: ``agent.update_many(rng, level_2_data[B, W])`` if the agent is vectorised
    over books (returns ``(counts[B], six arrays)``), else the reference's one-book ``update(rng, level_2_data[b])``
    (base_agent.py:67-116) called book by book, in book order, and concatenated."""
    n_books = level_2_data.shape[0]
    if hasattr(agent, "update_many"):
        counts, ins = agent.update_many(rng, level_2_data)
        counts = np.asarray(counts, dtype=np.uint64)
    else:
        per_book = [agent.update(rng, level_2_data[b]) for b in range(n_books)]
        counts = np.array([len(p[0]) for p in per_book], dtype=np.uint64)
        ins = tuple(np.concatenate([np.asarray(p[k]) for p in per_book]) if per_book else np.zeros(0) for k in range(6))
    off = np.zeros(n_books + 1, dtype=np.uint64)
    np.cumsum(counts, out=off[1:])
    return off, ins


def run_many(env, agents: typing.Iterable, n_steps: int, seed: int, show_progress: bool = False) -> np.ndarray:
    """The numpy-API loop of ``run(..., use_numpy=True)`` (ref src/bourse/step_sim/runner.py:103-112) over a
    ``ManyBookEnv``: each step every agent sees the level-2 records of ALL books (u32[n_books, 5 + 4 levels]; 45 wide at
    the reference's 10 levels) and its instructions for all books go out as ONE CSR batch of host arrays
    (``submit_instructions_all``: on a device-ingress env pinned staging -> async upload -> ``k_ingest``; the ids are not
    fetched - the reference's loop discards them too), then every book steps.  One submit per agent and step, as the
    reference.  Returns the level-2 history u32[n_steps, n_books, width] when the env retains it
    (``history_capacity >= n_steps``), else the last level-2 records."""
    agents = list(agents)
    rng = np.random.default_rng(seed)
    steps = range(n_steps)
    if show_progress:
        try:
            import tqdm

            steps = tqdm.trange(n_steps)
        except ImportError:
            pass
    ingress = getattr(env, "_device_ingress", False)
    pending = []
    for _ in steps:
        level_2_data = env.level2()
        for agent in agents:
            off, ins = _csr_for_all_books(agent, rng, level_2_data)
            if ingress:
                pending.append(env.submit_instructions_all_async(off, ins))
                if len(pending) > 1:  # two tickets in flight at most: the older one's status is looked at, its ids are not
                    _raise_for_status(env, pending.pop(0))
            else:
                env.submit_instructions_all(off, ins)
        if ingress:
            env.step(sync=False)
        else:
            env.step()
    for t in pending:
        _raise_for_status(env, t)
    if ingress:
        env.sync()
        if env.strict:
            env.raise_on_flags()
    return env.history() if env.history_capacity >= n_steps else env.level2()


def _raise_for_status(env, ticket):
    _, status, bad = env.submit_result(ticket, ids=False)
    if bad is not None:
        if int(status[bad, 0]) == 1:
            raise ValueError(f"book {bad}: a price of its batch was not a multiple of the tick size "
                             f"(element {int(status[bad, 1])} of the book's batch; earlier elements are queued)")
        raise RuntimeError(f"book {bad}: event queue / id space exhausted after {int(status[bad, 1])} elements")
