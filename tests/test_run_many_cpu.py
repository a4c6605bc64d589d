"""Host logic of bourse_amd.step_sim.run_many (the numpy-agent loop of ref src/bourse/step_sim/runner.py:103-112 over a many-book
env) against a recording stand-in for the env: call order, the CSR batch built from one-book agents (book order, one submit per
agent and step), the books-vectorised agent, and the ticket bookkeeping of the device-ingress flow.  No GPU, no oracle."""
import numpy as np
import pytest

from bourse_amd.step_sim import agents as A
from bourse_amd.step_sim import runner


class _Env:
    def __init__(self, n_books, ingress, bad_at=None):
        self.n_books, self._device_ingress, self.strict, self.history_capacity = n_books, ingress, True, 0
        self.calls, self.t, self.bad_at = [], 0, bad_at
        self.l2 = np.zeros((n_books, 45), dtype=np.uint32)

    def level2(self):
        self.calls.append(("level2",))
        return self.l2 + np.uint32(self.t)

    def submit_instructions_all(self, off, ins):
        self.calls.append(("submit", off.copy(), tuple(np.asarray(a).copy() for a in ins)))

    def submit_instructions_all_async(self, off, ins):
        self.calls.append(("submit_async", off.copy(), tuple(np.asarray(a).copy() for a in ins)))
        self.t += 0
        return len([c for c in self.calls if c[0] == "submit_async"]) - 1

    def submit_result(self, ticket, ids=True):
        self.calls.append(("result", ticket, ids))
        st = np.zeros((self.n_books, 2), dtype=np.uint32)
        if self.bad_at == ticket:
            st[3] = (1, 2)
            return None, st, 3
        return None, st, None

    def step(self, sync=True):
        self.calls.append(("step", sync))
        self.t += 1

    def sync(self):
        self.calls.append(("sync",))

    def raise_on_flags(self):
        self.calls.append(("flags",))


class _Counter(A.BaseNumpyAgent):
    """book b at step t gets (b % 3) instructions whose vol encodes (t, b): reads its own level-2 record"""

    def __init__(self):
        self.seen = []

    def update(self, rng, l2):
        t = int(l2[0])
        b = len(self.seen) % 5
        self.seen.append(t)
        n = b % 3
        return (np.ones(n, np.uint32), np.zeros(n, bool), np.full(n, 100 * t + b, np.uint32), np.arange(n, dtype=np.uint32),
                np.full(n, 50, np.uint32), np.zeros(n, np.uint64))


@pytest.mark.parametrize("ingress", [False, True])
def test_run_many_builds_one_csr_batch_per_agent_and_step(ingress):
    env = _Env(5, ingress)
    a1, a2 = _Counter(), A.ManyBookNumpyRandomAgents(2, (40, 60), (1, 9), 2)
    out = runner.run_many(env, [a1, a2], 3, seed=9)
    assert out.shape == (5, 45)
    kinds = [c[0] for c in env.calls]
    sub = "submit_async" if ingress else "submit"
    per_step = ["level2", sub, sub, "step"]
    body = [k for k in kinds if k not in ("result", "sync", "flags")]
    assert body[:12] == per_step * 3 and body[12:] == ["level2"]
    subs = [c for c in env.calls if c[0] == sub]
    for t in range(3):
        off, ins = subs[2 * t][1], subs[2 * t][2]
        assert off.tolist() == [0, 0, 1, 3, 3, 4]  # books 0..4 hand over 0, 1, 2, 0, 1 instructions, in book order
        assert ins[2].tolist() == [100 * t + 1, 100 * t + 2, 100 * t + 2, 100 * t + 4]
        assert all(len(x) == 4 for x in ins)
        off2, ins2 = subs[2 * t + 1][1], subs[2 * t + 1][2]
        assert off2.tolist() == [0, 2, 4, 6, 8, 10] and ins2[3].tolist() == [0, 1] * 5 and (ins2[4] % 2 == 0).all()
    assert a1.seen == [t for t in range(3) for _ in range(5)]  # every book's agent saw the step's own level-2 record
    assert [c[1] for c in env.calls if c[0] == "step"] == [not ingress] * 3
    if ingress:  # every ticket's status is looked at exactly once, none of the ids fetched; at most two in flight
        res = [c for c in env.calls if c[0] == "result"]
        assert sorted(r[1] for r in res) == list(range(6)) and not any(r[2] for r in res)
        inflight = 0
        for c in env.calls:
            inflight += c[0] == "submit_async"
            inflight -= c[0] == "result"
            assert inflight <= 2
        assert kinds[-3:] == ["sync", "flags", "level2"]


def test_run_many_raises_the_reference_error_for_a_book_that_stopped_at_a_bad_price():
    env = _Env(5, True, bad_at=1)
    with pytest.raises(ValueError, match="book 3: a price of its batch was not a multiple of the tick size \\(element 2"):
        runner.run_many(env, [_Counter()], 4, seed=1)
