#!/usr/bin/env python3
"""Write tests/golden/rng_pin_expected.txt: what the CPU oracle's restated rand 0.8.5 / rand_xoshiro 0.6.0 /
rand_distr 0.4.3 produce for the call patterns the reference's hot path uses (seed 101).

integration/rust/pin_rng prints the same lines from the real crates; an empty diff pins the RNG-dependent half
of the parity claim.  tests/test_oracle_kat.py checks this file against the oracle.
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def lines():
    import pyoracle as o

    def line(name, v):
        return f"{name}: " + " ".join(str(x) for x in v)

    def bits(x):
        return struct.unpack("<Q", struct.pack("<d", x))[0]

    R = lambda s=101: o.Rng(s)  # noqa: E731
    out = []
    r = R(); out.append(line("next_u64", [r.next_u64() for _ in range(8)]))
    r = R(); out.append(line("next_u32", [r.next_u32() for _ in range(16)]))
    r = R(); out.append(line("f32_bits", [int(np.float32(r.gen_f32()).view(np.uint32)) for _ in range(16)]))
    r = R(); out.append(line("choose_of_2", [r.gen_range(0, 2) for _ in range(16)]))
    r = R(); out.append(line("gen_range_10_100", [r.gen_range(10, 100) for _ in range(16)]))
    r = R(); out.append(line("gen_range_32_64", [r.gen_range(32, 64) for _ in range(16)]))
    r = R(); out.append(line("shuffle_16", r.shuffle(np.arange(16)).tolist()))
    r = R(); out.append(line("f64_bits", [bits(r.gen_f64()) for _ in range(8)]))
    r = R(); out.append(line("gen_bool_half", [int(r.next_u64() < 2**63) for _ in range(16)]))
    r = R(); out.append(line("std_normal_bits", [bits(r.std_normal()) for _ in range(16)]))
    r = R(); out.append(line("lognormal_0_10", ["%.17e" % r.lognormal(0.0, 10.0) for _ in range(8)]))
    r = R(108)
    acc, M = 0, 2**64
    for _ in range(1000):
        if r.gen_f32() < np.float32(0.8):
            acc = (acc * 31 + r.gen_range(0, 2)) % M
            acc = (acc * 31 + r.gen_range(32, 64)) % M
            acc = (acc * 31 + r.gen_range(10, 20)) % M
    out.append(line("random_agents_pattern_acc_then_next_u64", [acc, r.next_u64()]))
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "rng_pin_expected.txt")
    open(path, "w").write("\n".join(lines()) + "\n")
    print("wrote", path)
