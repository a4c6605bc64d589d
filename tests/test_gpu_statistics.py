"""Statistical parity of the Noise/Momentum agents' sampling chain ON THE DEVICE.

Bit parity with a Rust build is impossible here for this agent family: rand_distr 0.4.3's ziggurat tables and the
platform libm behind `exp` / `ln` / `tanh` are third-party code that is not in the reference tree, and the
oracle shares pm_math.hpp / the regenerated tables with the kernels - bit equality between the two proves control flow
and rounding order, not the DISTRIBUTIONS.  These tests check the distributions themselves, on the GPU:

* NoiseAgent limit prices: price = round_down(mid - d) / round_up(mid + d) with d = LogNormal(mu, sigma).sample()
  (ref crates/step_sim/src/agents/noise_agent.rs:134-149, common.rs:96-141) -> ln d ~ N(mu, sigma^2): Kolmogorov-
  Smirnov distance, first four moments, the ziggurat's tail region, the gen_bool(0.5) side split, over 1.6 M orders;
* pm_math.hpp's exp / log / tanh as compiled for gfx950: equal to the HOST build of the same header bit for bit (the
  oracle's), and within 2 ulp (log: 3, tanh: 5 measured, 6 allowed) of libm over dense sweeps.
"""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MID = 2147483647.5  # OrderBook::mid_price of an empty book: 0 + 0.5 * (u32::MAX - 0) (orderbook.rs:272-276)


@pytest.fixture(scope="module")
def bk():
    import bourse_amd

    return bourse_amd


def _resting_orders(env, R):
    """(price, is_bid) of every live pool slot of every book, parsed from the checkpoint image (64-byte header, then per
    book: 64 header dwords + per pool register {price, vol, id, seq, meta} x 64 lanes)."""
    img = env.checkpoint()
    stride = 64 + 320 * R
    st = img[64:64 + env.n_books * stride * 4].view(np.uint32).reshape(env.n_books, stride)
    pool = st[:, 64:].reshape(env.n_books, R, 5, 64)
    live = (pool[:, :, 4, :] & 1) != 0
    return pool[:, :, 0, :][live].astype(np.float64), ((pool[:, :, 4, :] >> 1) & 1)[live].astype(bool)


@pytest.mark.parametrize("mu,sigma", [(8.0, 1.0), (9.0, 1.5)])
def test_noise_agent_offsets_are_lognormal(bk, mu, sigma):
    from scipy import stats

    B, A = 4096, 400
    env = bk.ManyBookEnv(B, 20240917, 0, 1, 1_000_000, True, levels=4, max_live_orders=512, trade_capacity=64)
    env.set_agents([("noise", 0, A, dict(tick_size=1, p_limit=1.0, p_market=0.0, p_cancel=0.0, trade_vol=1,
                                          price_dist_mu=mu, price_dist_sigma=sigma))])
    env.run(1)  # empty books: every agent places one limit order around MID; bids < MID < asks, so nothing crosses
    price, is_bid = _resting_orders(env, 8)
    n = len(price)
    assert n == B * A and int(env.trade_counts().sum()) == 0
    # undo the tick rounding (floor for bids, ceil for asks; tick 1): the offset lies within half a tick of this
    d = np.where(is_bid, MID - price, price - MID) - 0.5
    keep = d > 50.0  # below that the half-tick uncertainty would show in ln d; < 4e-4 of the mass at these parameters
    assert keep.mean() > 0.999
    z = (np.log(d[keep]) - mu) / sigma
    # the discarded lower tail, accounted for exactly: compare with the normal law truncated at the cut
    cut = (math.log(50.0) - mu) / sigma
    p_cut = stats.norm.cdf(cut)
    u = (stats.norm.cdf(z) - p_cut) / (1.0 - p_cut)          # ~ U(0, 1) under the hypothesis
    ks = stats.kstest(u, "uniform").statistic
    assert ks < 2.2 / math.sqrt(len(z)), ks                   # alpha ~ 1e-4 for n = 1.6e6 (critical value 1.95 / sqrt n at 1e-3)
    if p_cut < 1e-6:                                          # moments of the (practically) untruncated sample
        m1, m2 = z.mean(), z.var()
        m3, m4 = ((z - m1) ** 3).mean() / m2 ** 1.5, ((z - m1) ** 4).mean() / m2 ** 2
        se = 1.0 / math.sqrt(len(z))
        assert abs(m1) < 5 * se and abs(m2 - 1.0) < 5 * math.sqrt(2) * se
        assert abs(m3) < 5 * math.sqrt(6) * se and abs(m4 - 3.0) < 5 * math.sqrt(24) * se
        # the ziggurat's base layer / tail algorithm (|z| > R = 3.654...): expected mass 2 (1 - Phi(R))
        R = 3.654152885361008796
        tail, p_tail = int((np.abs(z) > R).sum()), 2.0 * stats.norm.sf(R)
        assert abs(tail - len(z) * p_tail) < 5 * math.sqrt(len(z) * p_tail), (tail, len(z) * p_tail)
    # gen_bool(0.5): next_u64 < 2^63
    assert abs(is_bid.mean() - 0.5) < 5 * 0.5 / math.sqrt(n)


def test_pm_math_on_device_matches_host_build_and_libm(bk, oracle):
    L = bk._lib.load()
    H = oracle.lib()
    rng = np.random.default_rng(1)

    def dev(op, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        bk._lib.check(L.bk_selftest_math(op, x.ctypes.data_as(C.POINTER(C.c_double)), len(x),
                                         out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def ulps(got, ref):
        return np.abs(got - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)

    sweeps = {
        0: (np.concatenate([np.linspace(-745.0, 709.0, 400_001), rng.uniform(-40, 40, 200_000), rng.uniform(-1e-3, 1e-3, 50_000)]),
            np.exp, H.orc_pm_exp, 2.0),
        1: (np.concatenate([np.exp(np.linspace(-700.0, 700.0, 400_001)), rng.uniform(0.5, 2.0, 200_000),
                            1.0 + rng.uniform(-1e-6, 1e-6, 50_000)]), np.log, H.orc_pm_log, 3.0),
        2: (np.concatenate([np.linspace(-30.0, 30.0, 400_001), rng.uniform(-0.3, 0.3, 200_000), rng.uniform(-1e-8, 1e-8, 50_000)]),
            np.tanh, H.orc_pm_tanh, 6.0),
    }
    for op, (x, libm, host, tol) in sweeps.items():
        got = dev(op, x)
        ref = libm(x)
        ok = np.isfinite(ref) & (np.abs(ref) > 1e-300)
        worst = float(ulps(got[ok], ref[ok]).max())
        assert worst <= tol, (op, worst)
        # the gfx950 build equals the host (x86-64) build of the same header bit for bit: no FMA contraction, same
        # operation order (a sample; the oracle entry point is one ctypes call per value)
        idx = rng.choice(len(x), 20_000, replace=False)
        hv = np.array([host(float(v)) for v in x[idx]])
        assert np.array_equal(got[idx].view(np.uint64), hv.view(np.uint64)), op
