"""CPU test of the wave-parallel RNG decode ALGORITHM (tools/wave_decode_proto.py, the plain-Python model of
bourse_amd/csrc/wave_agents.hpp): jump-ahead lane states via nibble tables, per-window ballot masks + per-lane
placement look-ahead + scalar walk over the activity hits, fixed-point shuffle acceptance - against a straightforward
serial restatement of RandomAgents::update + shuffle (ref random_agent.rs:85-119, env.rs:121).
The HIP kernels themselves are checked against the oracle on the GPU (tests/test_gpu_parity.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_wave_decode_model_equals_serial_semantics():
    import wave_decode_proto as w

    st = w.selftest(n_cases=25, seed=3)
    assert st["steps"] == 150 and st.get("slow", 0) == 0  # 64-draw look-ahead: the slow path is (practically) never taken


def test_wave_decode_model_slow_path():
    import wave_decode_proto as w

    st = w.selftest(n_cases=15, seed=11, lookahead=2)
    assert st["slow"] > 20  # a 2-draw look-ahead forces most placements through the draw-by-draw path; results unchanged


def test_jump_tables_model():
    import wave_decode_proto as w

    tab = w.build_jump_tables(37)
    s0, s1 = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    assert w.jump(tab, s0, s1) == w.advance(s0, s1, 37)
    # published xoroshiro128** vector: state (1, 2) -> 5760, 97769243520, ...
    a, b, x0 = w.step(1, 2)
    a, b, x1 = w.step(a, b)
    assert (x0, x1) == (5760, 97769243520 & 0xFFFFFFFF)


def test_parallel_fisher_yates_resolution_equals_sequential_swaps():
    import random

    import wave_decode_proto as w

    rnd = random.Random(5)
    for _ in range(2000):
        n = rnd.randrange(1, 129)
        j = [0] + [rnd.randrange(0, i + 1) for i in range(1, n)]
        assert w.fisher_yates_parallel(n, j) == w.fisher_yates_serial(n, j)
    for n in (1, 2, 128):  # all self-swaps / all to position 0
        assert w.fisher_yates_parallel(n, list(range(n))) == list(range(n))
        assert w.fisher_yates_parallel(n, [0] * n) == w.fisher_yates_serial(n, [0] * n)


def test_two_round_resolution_of_up_to_256_positions_equals_sequential_swaps():
    import random

    import wave_decode_proto as w

    rnd = random.Random(6)
    for _ in range(1500):
        n = rnd.choice([1, 2, 127, 128, 129, 130, 160, 255, 256, rnd.randrange(1, 257)])
        j = [0] + [rnd.randrange(0, i + 1) for i in range(1, n)]
        assert w.two_round_resolution(n, j) == w.fisher_yates_serial(n, j), n
    for n in (129, 200, 256):
        assert w.two_round_resolution(n, list(range(n))) == list(range(n))
        assert w.two_round_resolution(n, [0] * n) == w.fisher_yates_serial(n, [0] * n)
