"""Prototype (plain Python) of the wave-parallel RNG decode used by k_agents_wave: validates the ALGORITHM on CPU against
a straightforward serial restatement of RandomAgents::update + shuffle (random_agent.rs:85-119, env.rs:121).

  * xoroshiro128** is F2-linear: lane j holds the state 4*j draws ahead; a block = 64 lanes x K=4 draws; the lane states
    advance by T^(256) with 32 nibble-table lookups (tables built here by stepping basis vectors).
  * agents: per 64-draw window, ballot masks (activity hit per group, range accepts) + per-lane f(p) = stream position
    after a placement that starts at p; a scalar walk over the hit mask consumes one hit per iteration.
  * shuffle: acceptance of draw p depends on how many earlier draws were accepted (range i+1 shrinks): fixed-point
    iteration over the window's accept mask, exact because lane 0's count is always right.
"""
import random

M64 = (1 << 64) - 1
K = 4
BLOCK = 64 * K


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


def step(s0, s1):
    r = (rotl((s0 * 5) & M64, 7) * 9) & M64
    s1 ^= s0
    s0 = rotl(s0, 24) ^ s1 ^ ((s1 << 16) & M64)
    s1 = rotl(s1, 37)
    return s0, s1, r & 0xFFFFFFFF


def advance(s0, s1, n):
    for _ in range(n):
        s0, s1, _ = step(s0, s1)
    return s0, s1


def build_jump_tables(n_steps):
    """tab[k][v] = T^n_steps applied to the state whose only non-zero nibble is nibble k = v (k = 0..31, 128-bit LE)."""
    # columns of T^n: image of each basis bit
    cols = []
    for b in range(128):
        s = 1 << b
        s0, s1 = advance(s & M64, s >> 64, n_steps)
        cols.append(s0 | (s1 << 64))
    tab = []
    for k in range(32):
        row = [0] * 16
        for v in range(16):
            acc = 0
            for bit in range(4):
                if v >> bit & 1:
                    acc ^= cols[4 * k + bit]
            row[v] = acc
        tab.append(row)
    return tab


def jump(tab, s0, s1):
    s = s0 | (s1 << 64)
    acc = 0
    for k in range(32):
        acc ^= tab[k][(s >> (4 * k)) & 15]
    return acc & M64, acc >> 64


def zone_of(r):
    return ((r << (32 - r.bit_length())) - 1) & 0xFFFFFFFF


# ------------------------------------------------------------------ serial reference
def serial_step(s0, s1, groups, live):
    """groups: (n, thr, tick_lo, tick_rng, vol_lo, vol_rng, tick_size); live: set of agent slots holding an Active order.
    Returns (s0, s1, events [(slot, is_new, side)], orders {slot: (price, vol)}, shuffled order)."""
    draws = 0

    def nxt():
        nonlocal s0, s1, draws
        s0, s1, x = step(s0, s1)
        draws += 1
        return x

    def below(r):
        z = zone_of(r)
        while True:
            m = nxt() * r
            if (m & 0xFFFFFFFF) <= z:
                return m >> 32

    ev, orders = [], {}
    a = 0
    for (n, thr, tlo, trng, vlo, vrng, ts) in groups:
        for _ in range(n):
            x = nxt()
            if (x >> 8) < thr:
                if a in live:
                    ev.append((a, 0, 0))
                else:
                    side = below(2)
                    tick = tlo + below(trng)
                    vol = vlo + below(vrng)
                    orders[a] = (tick * ts, vol)
                    ev.append((a, 1, side))
            a += 1
    lst = list(ev)
    for i in range(len(lst) - 1, 0, -1):
        j = below(i + 1)
        lst[i], lst[j] = lst[j], lst[i]
    return s0, s1, ev, orders, lst, draws


# ------------------------------------------------------------------ wave-parallel version
class WaveRng:
    """Lane j holds the chunk-start state of draws [4j, 4j+4) of the current block; ring of generated draws."""

    def __init__(self, s0, s1, tab_block, lane_tabs):
        # lane states by log-doubling from the canonical state (lane j = T^(4j))
        self.cs = []
        for j in range(64):
            a, b = s0, s1
            for bit in range(6):
                if j >> bit & 1:
                    a, b = jump(lane_tabs[bit], a, b)
            self.cs.append((a, b))
        self.tab = tab_block
        self.draws = []   # absolute stream of generated draws (the device keeps a 512-entry ring)
        self.block0_states = None

    def gen_block(self):
        out = [0] * BLOCK
        for j in range(64):
            a, b = self.cs[j]
            for i in range(K):
                a, b, x = step(a, b)
                out[j * K + i] = x
        self.draws.extend(out)
        self.prev_cs = list(self.cs)
        self.cs = [jump(self.tab, a, b) for (a, b) in self.cs]

    def ensure(self, upto):
        while len(self.draws) < upto:
            self.gen_block()

    def canonical_at(self, pos):
        """state after `pos` draws, recovered from the lane states of the block containing pos"""
        self.ensure(pos + 1)
        nb = len(self.draws) // BLOCK
        blk = pos // BLOCK
        assert blk == nb - 1, "device keeps only the last block's chunk-start states"
        c = pos - blk * BLOCK
        a, b = self.prev_cs[c // K]
        return advance(a, b, c % K)


def first_above(mask128, i):
    """index of the first set bit > i among the next 64 positions (and below 128), or 128 - the device's
    first_above_near (wave_agents.hpp): a set bit further away is reported as none, the placement then counts as not
    resolvable inside the look-ahead and takes the exact draw-by-draw path"""
    n = i + 1
    if n >= 128:
        return 128
    m = (mask128 >> n) & ((1 << min(64, 128 - n)) - 1)
    if m == 0:
        return 128
    return n + ((m & -m).bit_length() - 1)


def wave_step(s0, s1, groups, live, tab_block, lane_tabs, lookahead=64, stats=None):
    W = WaveRng(s0, s1, tab_block, lane_tabs)
    pos = 0
    ev, orders = [], {}
    a = 0
    gi = 0
    gends = []
    t = 0
    for g in groups:
        t += g[0]
        gends.append(t)
    # skip empty groups
    while gi < len(groups) and groups[gi][0] == 0:
        gi += 1
    n_windows = 0
    while gi < len(groups):
        w0 = pos & ~63
        W.ensure(w0 + 128)
        x = W.draws[w0:w0 + 128]
        n, thr, tlo, trng, vlo, vrng, ts = groups[gi]
        gend = gends[gi]
        H = sum(1 << l for l in range(64) if (x[l] >> 8) < thr)
        A2 = sum(1 << l for l in range(128) if ((x[l] * 2) & 0xFFFFFFFF) <= 0x7FFFFFFF)
        zt, zv = zone_of(trng), zone_of(vrng)
        AT = sum(1 << l for l in range(128) if ((x[l] * trng) & 0xFFFFFFFF) <= zt)
        AV = sum(1 << l for l in range(128) if ((x[l] * vrng) & 0xFFFFFFFF) <= zv)
        lim = 64 + lookahead
        f = [None] * 64
        for p_ in range(64):
            q1 = first_above(A2, p_)
            q2 = first_above(AT, q1) if q1 < lim else 128
            q3 = first_above(AV, q2) if q2 < lim else 128
            if q3 < lim:
                f[p_] = (q3 + 1, x[q1] >> 31, (tlo + ((x[q2] * trng) >> 32)) * ts, vlo + ((x[q3] * vrng) >> 32))
        n_windows += 1
        p = pos - w0
        # scalar walk over this window (same group)
        while True:
            m = H >> p
            if m == 0:
                adv = min(64 - p, gend - a)
                a += adv
                p += adv
            else:
                d = (m & -m).bit_length() - 1
                if a + d >= gend:
                    p += gend - a
                    a = gend
                else:
                    a += d
                    p += d
                    if a in live:
                        ev.append((a, 0, 0))
                        a += 1
                        p += 1
                    elif f[p] is not None:
                        fp, side, price, vol = f[p]
                        orders[a] = (price, vol)
                        ev.append((a, 1, side))
                        a += 1
                        p = fp
                    else:
                        # slow path: serial resolution of this one placement (a stage ran past the look-ahead)
                        if stats is not None:
                            stats["slow"] = stats.get("slow", 0) + 1
                        q = w0 + p + 1
                        vals = []
                        for r_, z_ in ((2, 0x7FFFFFFF), (trng, zt), (vrng, zv)):
                            while True:
                                W.ensure(q + 1)
                                mm = W.draws[q] * r_
                                q += 1
                                if (mm & 0xFFFFFFFF) <= z_:
                                    vals.append(mm >> 32)
                                    break
                        orders[a] = ((tlo + vals[1]) * ts, vlo + vals[2])
                        ev.append((a, 1, vals[0]))
                        a += 1
                        p = q - w0
            if a >= gend or p >= 64:
                break
        pos = w0 + p
        while gi < len(groups) and a >= gends[gi]:
            gi += 1
    # ---- shuffle: fixed-point iteration per 64-draw window
    lst = list(ev)
    i = len(lst) - 1
    iters = 0
    while i >= 1:
        w0 = pos & ~63
        W.ensure(w0 + 64)
        x = W.draws[w0:w0 + 64]
        p0 = pos - w0
        valid = [(l >= p0) for l in range(64)]
        acc = [valid[l] for l in range(64)]  # initial guess: every draw accepted
        while True:
            iters += 1
            k = 0
            new = [False] * 64
            jv = [0] * 64
            iv = [0] * 64
            for l in range(64):
                if not valid[l]:
                    continue
                ii = i - k   # index this draw would serve, given the accepts counted so far under the OLD mask
                iv[l] = ii
                if ii >= 1:
                    r_ = ii + 1
                    mm = x[l] * r_
                    new[l] = (mm & 0xFFFFFFFF) <= zone_of(r_)
                    jv[l] = mm >> 32
                k += 1 if acc[l] else 0
            if new == acc:
                break
            acc = new
        # apply in order
        used = p0
        for l in range(p0, 64):
            if iv[l] < 1:
                break
            used = l + 1
            if acc[l]:
                ii, jj = iv[l], jv[l]
                lst[ii], lst[jj] = lst[jj], lst[ii]
                i = ii - 1
                if i < 1:
                    break
        pos = w0 + used
    if stats is not None:
        stats["windows"] = stats.get("windows", 0) + n_windows
        stats["shuffle_iters"] = stats.get("shuffle_iters", 0) + iters
        stats["steps"] = stats.get("steps", 0) + 1
    # canonical state at pos: the device recovers it from the last block's lane states
    if pos == 0:
        c0, c1 = s0, s1
    else:
        W.ensure(pos)  # (pos may sit exactly at a block end)
        if pos % BLOCK == 0 and len(W.draws) == pos:
            W.gen_block()
        c0, c1 = W.canonical_at(pos) if pos // BLOCK == len(W.draws) // BLOCK - 1 else advance(s0, s1, pos)
    return c0, c1, ev, orders, lst, pos


def fisher_yates_serial(n, j):
    """positions after `for i in (1..n).rev(): swap(i, j[i])` applied to the identity"""
    a = list(range(n))
    for i in range(n - 1, 0, -1):
        a[i], a[j[i]] = a[j[i]], a[i]
    return a


def fisher_yates_parallel(n, j):
    """The same permutation, every position resolved independently (k_agents_wave's shuffle, R <= 2): W[y] = mask of
    the steps that target position y; the value that ends at x sat at j[x] just before step x; position y holds, before
    step t, what the most recent earlier step s* = min{s > t : j[s] == y} moved there - the value position s* held
    before step s* - or its original entry when there is none."""
    W = [0] * n
    for s in range(1, n):
        if j[s] != s:
            W[j[s]] |= 1 << s
    out = [0] * n
    for x in range(n):
        y, t = (j[x], x) if x >= 1 else (0, 0)
        while True:
            m = W[y] >> (t + 1)
            if m == 0:
                break
            y = t = t + 1 + ((m & -m).bit_length() - 1)
        out[x] = y
    return out


def two_round_resolution(n, j):
    """fisher_yates_parallel for up to 256 positions with 128-bit masks only (round 5: the members' decode of the larger pools).
    A step s < 128 only touches positions <= s, so the shuffle is (steps 127 .. 1) after (steps n-1 .. 128):
    round A - masks of the steps >= 128 (bit s - 128) per target position; a position x >= 128 is final after its own step:
    chase from (j[x], x); a position x < 128 holds what the most recent of those steps moved there: chase from (x, "before
    step 127"); round B - the rule above on positions 0 .. 127 of round A's array."""
    assert n <= 256

    def chase(W, y, t, base):
        while True:
            m = W[y] >> (t + 1)
            if m == 0:
                return y
            t = t + 1 + ((m & -m).bit_length() - 1)
            y = t + base

    jx = [j[x] if 1 <= x < n else x for x in range(n)]
    WA = [0] * n
    for x in range(128, n):
        if jx[x] != x:
            WA[jx[x]] |= 1 << (x - 128)
    a = [chase(WA, jx[x], x - 128, 128) if x >= 128 else chase(WA, x, -1, 128) for x in range(n)]
    nb = min(n, 128)
    WB = [0] * nb
    for x in range(nb):
        if jx[x] != x:
            WB[jx[x]] |= 1 << x
    return [a[chase(WB, jx[x], x, 0)] if x < nb else a[x] for x in range(n)]


def selftest(n_cases=60, seed=1, lookahead=64, verbose=False):
    rnd = random.Random(seed)
    tab_block = build_jump_tables(BLOCK)
    lane_tabs = [build_jump_tables(K << b) for b in range(6)]
    stats = {}
    for case in range(n_cases):
        ng = rnd.choice([1, 2, 2, 3])
        groups = []
        for _ in range(ng):
            n = rnd.choice([0, 1, 5, 32, 64, 100])
            rate = rnd.choice([0.0, 0.2, 0.5, 0.8, 1.0])
            thr = int(rate * (1 << 24))
            trng = rnd.choice([1, 2, 16, 32, 64, 90])
            vrng = rnd.choice([1, 10, 20, 30])
            groups.append((n, thr, rnd.randrange(1, 50), trng, rnd.randrange(1, 60), vrng, rnd.choice([1, 2, 3])))
        total = sum(g[0] for g in groups)
        s0, s1 = rnd.getrandbits(64), rnd.getrandbits(64)
        live = set()
        for stepno in range(6):
            r = serial_step(s0, s1, groups, live)
            w = wave_step(s0, s1, groups, live, tab_block, lane_tabs, lookahead, stats)
            assert r[0:2] == w[0:2], ("rng", case, stepno)
            assert r[2] == w[2] and r[3] == w[3], ("events", case, stepno)
            assert r[4] == w[4], ("shuffle", case, stepno)
            assert r[5] == w[5], ("draws", case, stepno)
            s0, s1 = r[0], r[1]
            # next live set: random subset evolves (stand-in for the event phase)
            live = set(a for a in range(total) if rnd.random() < 0.5)
    if verbose:
        print(stats)
    return stats


if __name__ == "__main__":
    import sys
    print(selftest(int(sys.argv[1]) if len(sys.argv) > 1 else 60, verbose=True))
    print(selftest(30, seed=7, lookahead=3, verbose=True))  # tiny look-ahead: forces the slow path
