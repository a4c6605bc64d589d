"""Generate bourse_amd/csrc/event_asm_gen.hpp: the KEYED event loop of Env::step for pools of 256 / 512 slots (R = 4, 8),
hand-written gfx950 assembly like event_asm.hpp's R <= 2 loops ("emit the R = 4, 8 loop from the same
macros, pool registers addressed by fixed VGPR numbers").

What limits an asm statement is its 30 OPERANDS, not its size: the pool rows travel as register TUPLES bound to fixed
physical registers (`"+{v[40:47]}"`), so the text below names v40.. / s36.. directly, and one statement takes 14 operands.
Semantics are event_asm.hpp's keyed loop (orderbook.rs:429-487, 583-611, 622-644, 843-870 through the 32-bit keys of
book_device.hpp keys_begin): cancellation, new order with the bound test, match against the best key of the other side,
compact trade records {k word, volume, passive slot}, rest with a fresh arrival field (a market order - the members' lists
of an AgentSet carry them - never rests).  Differences, all forced by R:
  * live asks / live bids are kept as TWO mask sets (askm, bidm: no `live & ~bid` algebra per match step);
  * the slot's own pool row (its key / volume / mask words) is addressed DYNAMICALLY - s_set_gpr_idx_on for the vector rows,
    s_movrels / s_movreld for the mask pairs - instead of one copy of the code per pool register (8 x 2 x 8 copies);
  * the passive order's row is found by comparing the best key against all R rows at once (R SGPR pairs), then one
    specialised PICK block per row;
  * the step's event list spans R registers: the current one is copied into a scratch register whenever k crosses a
    multiple of 64.
usage: gen_event_asm.py [--check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bourse_amd", "csrc", "event_asm_gen.hpp")


def gen(N):
    KB, VB, EB = 40, 40 + N, 40 + 2 * N                     # VGPR rows: key, vol, event list
    AB = 36                                                   # SGPR: askm pairs, bidm pairs, E pairs, scratch
    BB, EQB = AB + 2 * N, AB + 4 * N
    SC = AB + 6 * N
    names = ["EW", "KP", "V", "KK", "BEST", "PV", "LS", "X", "SLOT", "ALO", "BHI", "RG", "T0", "T1"]
    S = {n: f"s{SC + i}" for i, n in enumerate(names)}
    T = f"s[{SC + 12}:{SC + 13}]"
    last_s = SC + 13
    bits = {4: 2, 8: 3}[N]

    def key(r): return f"v{KB + r}"
    def vol(r): return f"v{VB + r}"
    def askm(r): return f"s[{AB + 2 * r}:{AB + 2 * r + 1}]"
    def bidm(r): return f"s[{BB + 2 * r}:{BB + 2 * r + 1}]"
    def eq(r): return f"s[{EQB + 2 * r}:{EQB + 2 * r + 1}]"

    L = []
    def e(x): L.append(x)

    DPP = ["quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf",
           "row_half_mirror row_mask:0xf bank_mask:0xf", "row_mirror row_mask:0xf bank_mask:0xf",
           "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]

    def loop():  # next event (or the end of the list); reload the list register when k crosses a multiple of 64
        e("s_branch L_loop_%=")

    def row_read(dst, base):   # dst = row RG of a vector array (RG in S[RG])
        e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(SRC0)")
        e(f"v_mov_b32 {dst}, v{base}")
        e("s_set_gpr_idx_off")

    def row_write(base, src):
        e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(DST)")
        e(f"v_mov_b32 v{base}, {src}")
        e("s_set_gpr_idx_off")

    def mask_bit(base, op):    # op = s_bitset0_b64 / s_bitset1_b64 on pair RG of a mask array, bit EW[5:0]
        e(f"s_lshl_b32 m0, {S['RG']}, 1")
        e("s_nop 0")
        e(f"s_movrels_b64 {T}, s[{base}:{base + 1}]")
        e(f"{op} {T}, {S['EW']}")
        e(f"s_movreld_b64 s[{base}:{base + 1}], {T}")

    # ---------------------------------------------------------------- entry
    e(f"s_mov_b32 {S['ALO']}, 0")
    e(f"s_mov_b32 {S['BHI']}, -1")
    e("s_cmp_lt_u32 %[k], %[nev]")
    e("s_cbranch_scc0 L_done_%=")
    e("L_reload_%=:")                                       # evc = ev[k >> 6]
    e(f"s_lshr_b32 {S['RG']}, %[k], 6")
    e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(SRC0)")
    e(f"v_mov_b32 %[evc], v{EB}")
    e("s_set_gpr_idx_off")
    e("s_nop 0")
    # ---------------------------------------------------------------- one event
    e("L_top_%=:")
    e(f"v_readlane_b32 {S['EW']}, %[evc], %[k]")
    e(f"s_bfe_u32 {S['RG']}, {S['EW']}, {hex((bits << 16) | 6)}")   # the slot's pool row
    e(f"s_bitcmp1_b32 {S['EW']}, 15")
    e("s_cbranch_scc1 L_new_%=")
    # Cancellation: whichever side the order rests on
    mask_bit(AB, "s_bitset0_b64")
    e(f"s_movrels_b64 {T}, s[{BB}:{BB + 1}]")
    e(f"s_bitset0_b64 {T}, {S['EW']}")
    e(f"s_movreld_b64 s[{BB}:{BB + 1}], {T}")
    e("L_loop_%=:")
    e("s_add_u32 %[k], %[k], 1")
    e("s_cmp_lt_u32 %[k], %[nev]")
    e("s_cbranch_scc0 L_done_%=")
    e(f"s_and_b32 {S['X']}, %[k], 63")
    e("s_cbranch_scc1 L_top_%=")
    e("s_branch L_reload_%=")
    # New order: its key prefix and volume (ew as a lane select: 4 wait states behind its v_readlane - bfe, bitcmp, branch, idx_on)
    e("L_new_%=:")
    e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(SRC0)")
    e(f"v_mov_b32 %[vt], v{KB}")
    e(f"v_mov_b32 %[vm], v{VB}")
    e("s_set_gpr_idx_off")
    e(f"v_readlane_b32 {S['KP']}, %[vt], {S['EW']}")
    e(f"v_readlane_b32 {S['V']}, %[vm], {S['EW']}")
    e(f"s_bitcmp1_b32 {S['EW']}, 14")
    e("s_cbranch_scc1 L_bid_%=")

    def side(tag, agg_bid):
        # agg_bid: searches the asks (min key), rests among the bids
        opp = askm if agg_bid else bidm
        own_base = BB if agg_bid else AB
        sent = "-1" if agg_bid else "0"
        vop = "v_min_u32" if agg_bid else "v_max_u32"
        oppb, ownb = (S["ALO"], S["BHI"]) if agg_bid else (S["BHI"], S["ALO"])
        e(f"s_and_b32 {S['X']}, {S['V']}, %[tmask]")            # no volume or trading disabled: no match
        e(f"s_cbranch_scc0 L_restq_{tag}_%=")
        e(f"{'s_cmp_lt_u32' if agg_bid else 's_cmp_gt_u32'} {S['KP']}, {oppb}")   # beyond the bound: cannot cross
        e(f"s_cbranch_scc1 L_restq_{tag}_%=")
        e(f"s_mov_b32 {S['KK']}, %[k]" if agg_bid else f"s_or_b32 {S['KK']}, %[k], 0x80000000")
        e(f"L_match_{tag}_%=:")
        e(f"v_cndmask_b32_e64 %[vm], {sent}, {key(0)}, {opp(0)}")
        for r in range(1, N):
            e(f"v_cndmask_b32_e64 %[vt], {sent}, {key(r)}, {opp(r)}")
            e(f"{vop} %[vm], %[vm], %[vt]")
        for ctl in DPP:
            e("s_nop 1")
            e(f"{vop}_dpp %[vm], %[vm], %[vm] {ctl}")
        e("s_nop 1")
        e(f"v_readlane_b32 {S['BEST']}, %[vm], 63")
        e("s_nop 1")                                            # BEST as a VALU operand: 2 wait states
        for r in range(N):
            e(f"v_cmp_eq_u32_e64 {eq(r)}, {S['BEST']}, {key(r)}")
        e(f"s_mov_b32 {oppb}, {S['BEST']}")                     # the bound is exact now
        e(f"{'s_cmp_gt_u32' if agg_bid else 's_cmp_lt_u32'} {S['BEST']}, {S['KP']}")   # no cross
        e(f"s_cbranch_scc1 L_rest_{tag}_%=")
        for r in range(N - 1):
            e(f"s_cmp_lg_u64 {eq(r)}, 0")
            e(f"s_cbranch_scc1 L_pick{r}_{tag}_%=")
        for r in reversed(range(N)):                            # (the last row falls through into its block)
            e(f"L_pick{r}_{tag}_%=:")
            e(f"s_ff1_i32_b64 {S['LS']}, {eq(r)}")
            if r:
                e(f"s_or_b32 {S['SLOT']}, {S['LS']}, {64 * r}")
            slot = S["SLOT"] if r else S["LS"]
            e("s_mov_b32 m0, %[trn]")
            e(f"v_readlane_b32 {S['PV']}, {vol(r)}, {S['LS']}")
            e(f"v_writelane_b32 %[trk], {S['KK']}, m0")
            e(f"v_writelane_b32 %[trs], {slot}, m0")
            e(f"s_sub_u32 {S['X']}, {S['PV']}, {S['V']}")       # SCC = borrow: the passive order is the smaller one
            e(f"s_cbranch_scc1 L_A{r}_{tag}_%=")
            e(f"v_writelane_b32 %[trv], {S['V']}, m0")          # aggressor exhausted; X = passive remainder
            e(f"s_mov_b32 m0, {S['LS']}")
            e(f"v_writelane_b32 {vol(r)}, {S['X']}, m0")
            e(f"s_cmp_eq_u32 {S['X']}, 0")
            e(f"s_cbranch_scc0 L_B{r}_{tag}_%=")
            e(f"s_andn2_b64 {opp(r)}, {opp(r)}, {eq(r)}")
            e(f"L_B{r}_{tag}_%=:")
            e("s_add_u32 %[trn], %[trn], 1")                    # SCC = carry = buffer full
            e("s_cbranch_scc1 L_fullnext_%=")
            loop()
            e(f"L_A{r}_{tag}_%=:")                              # passive order exhausted, aggressor goes on
            e(f"v_writelane_b32 %[trv], {S['PV']}, m0")
            e(f"s_mov_b32 m0, {S['LS']}")
            e(f"v_writelane_b32 {vol(r)}, 0, m0")
            e(f"s_andn2_b64 {opp(r)}, {opp(r)}, {eq(r)}")
            e(f"s_sub_u32 {S['V']}, {S['V']}, {S['PV']}")
            e("s_add_u32 %[trn], %[trn], 1")
            e(f"s_cbranch_scc1 L_fullA_{tag}_%=")
            e(f"s_branch L_match_{tag}_%=")
        # buffer full, volume left: the event restarts with what remains
        e(f"L_fullA_{tag}_%=:")
        row_read("%[vm]", VB)
        e(f"s_mov_b32 m0, {S['EW']}")
        e(f"v_writelane_b32 %[vm], {S['V']}, m0")
        row_write(VB, "%[vm]")
        e("s_branch L_flush_%=")
        # rests with what the trades left ... (a market order's remainder is dropped: orderbook.rs:521-524; its prefix is
        # the sentinel of book_device.hpp keys_begin<MARKETS> - 0xFFFFFFFE for a bid, 1 for an ask - which no limit order has)
        e(f"L_rest_{tag}_%=:")
        e(f"s_cmp_eq_u32 {S['KP']}, {'0xfffffffe' if agg_bid else '1'}")
        e("s_cbranch_scc1 L_loop_%=")
        row_read("%[vm]", VB)
        e(f"s_mov_b32 m0, {S['EW']}")
        e(f"v_writelane_b32 %[vm], {S['V']}, m0")
        row_write(VB, "%[vm]")
        e(f"s_branch L_restk_{tag}_%=")
        # ... or untouched (no volume / trading disabled / beyond the bound; a market order can only get here by the first two)
        e(f"L_restq_{tag}_%=:")
        e(f"s_cmp_eq_u32 {S['KP']}, {'0xfffffffe' if agg_bid else '1'}")
        e("s_cbranch_scc1 L_loop_%=")
        e(f"L_restk_{tag}_%=:")
        e(f"s_xor_b32 {S['X']}, {S['KP']}, %[sq]")
        e(f"{'s_max_u32' if agg_bid else 's_min_u32'} {ownb}, {ownb}, {S['X']}")   # this side's bound covers the new order
        row_read("%[vt]", KB)
        e(f"s_mov_b32 m0, {S['EW']}")
        e(f"v_writelane_b32 %[vt], {S['X']}, m0")
        row_write(KB, "%[vt]")
        mask_bit(own_base, "s_bitset1_b64")
        e("s_add_u32 %[sq], %[sq], 2")
        loop()

    side("a", False)
    e("L_bid_%=:")
    side("b", True)
    e("L_done_%=:")
    e("s_mov_b32 %[st], 0")
    e("s_branch L_out_%=")
    e("L_fullnext_%=:")
    e("s_add_u32 %[k], %[k], 1")
    e("L_flush_%=:")
    e("s_mov_b32 %[st], 1")
    e("L_out_%=:")
    e("s_nop 1")

    text = "".join(f'      "{x}\\n\\t"\n' for x in L)
    clob = ", ".join(f'"s{i}"' for i in range(EQB, last_s + 1))
    vt = {4: ("u32x4", "u32x8"), 8: ("u32x8", "u32x16")}[N]
    return f'''
// ---- R = {N}: {len(L)} instructions; rows key v[{KB}:{KB + N - 1}], vol v[{VB}:{VB + N - 1}], list v[{EB}:{EB + N - 1}]; masks askm s[{AB}:{AB + 2 * N - 1}],
// bidm s[{BB}:{BB + 2 * N - 1}]; row-compare results s[{EQB}:{EQB + 2 * N - 1}]; scratch s{SC}..s{last_s}
__device__ __forceinline__ uint32_t events_key_r{N}(uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                  uint64_t (&askm)[{N}], uint64_t (&bidm)[{N}], uint32_t (&vol)[{N}], uint32_t (&key)[{N}],
                                                  const uint32_t (&ev)[{N}], uint32_t& trk, uint32_t& trv, uint32_t& trs) {{
  uint32_t st, vm, vt, evc;
  auto u32 = [](uint32_t x) {{ return (uint32_t)__builtin_amdgcn_readfirstlane(x); }};
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;  // biased: the increment carries out exactly when the buffer is full
  sq = u32(sq);
  {vt[0]} kv, vv, evv;
  {vt[1]} am, bm;
#pragma unroll
  for (int r = 0; r < {N}; ++r) {{
    kv[r] = key[r];
    vv[r] = vol[r];
    evv[r] = ev[r];
    am[2 * r] = u32((uint32_t)askm[r]);
    am[2 * r + 1] = u32((uint32_t)(askm[r] >> 32));
    bm[2 * r] = u32((uint32_t)bidm[r]);
    bm[2 * r + 1] = u32((uint32_t)(bidm[r] >> 32));
  }}
  asm volatile(
{text}      : [st] "=&s"(st), [vm] "=&v"(vm), [vt] "=&v"(vt), [evc] "=&v"(evc), [k] "+s"(k), [trn] "+s"(trn), [sq] "+s"(sq),
        [key] "+{{v[{KB}:{KB + N - 1}]}}"(kv), [vol] "+{{v[{VB}:{VB + N - 1}]}}"(vv), [askm] "+{{s[{AB}:{AB + 2 * N - 1}]}}"(am),
        [bidm] "+{{s[{BB}:{BB + 2 * N - 1}]}}"(bm), [trk] "+v"(trk), [trv] "+v"(trv), [trs] "+v"(trs)
      : [ev] "{{v[{EB}:{EB + N - 1}]}}"(evv), [nev] "s"(n_ev), [tmask] "s"(tmask)
      : {clob}, "vcc", "scc", "memory");
#pragma unroll
  for (int r = 0; r < {N}; ++r) {{
    key[r] = kv[r];
    vol[r] = vv[r];
    askm[r] = ((uint64_t)am[2 * r + 1] << 32) | am[2 * r];
    bidm[r] = ((uint64_t)bm[2 * r + 1] << 32) | bm[2 * r];
  }}
  tr_n = trn + 64u;
  return st;
}}
'''


HEADER = '''// event_asm_gen.hpp - GENERATED by tools/gen_event_asm.py (do not edit; tests/test_capi_symbols.py checks it is current).
// The keyed event loop of Env::step for pools of 256 / 512 slots (R = 4, 8) in gfx950 assembly: see the generator for the
// design, event_asm.hpp for the semantics and the wait-state rules it shares with the R <= 2 loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bkd {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
'''

if __name__ == "__main__":
    src = HEADER + gen(4) + gen(8) + "\n}  // namespace bkd\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == src else 1)
    open(OUT, "w").write(src)
    print("wrote", OUT, len(src.splitlines()), "lines")
