"""Generate bourse_amd/csrc/event_asm_gen.hpp: the KEYED event loop of Env::step for pools of 256 / 512 slots (R = 4, 8),
hand-written gfx950 assembly like event_asm.hpp's R <= 2 loops ("emit the R = 4, 8 loop from the same
macros, pool registers addressed by fixed VGPR numbers").

What limits an asm statement is its 30 OPERANDS, not its size: the pool rows travel as register TUPLES bound to fixed
physical registers (`"+{v[40:47]}"`), so the text below names v40.. directly, and one statement takes 15 operands.
Semantics are event_asm.hpp's keyed loop (orderbook.rs:429-487, 583-611, 622-644, 843-870 through the SIGNED 32-bit keys of
book_device.hpp keys_begin: asks negative, bids positive, every other pool lane 0): cancellation, new order with the bound
test, match against the best key of the other side, compact trade records {k word, volume, passive slot}, rest with a
fresh arrival field (a market order - the members' lists of an AgentSet carry them - never rests).  What R forces:
  * the best key of a side is the signed minimum / maximum over ALL pool lanes: v_min3 / v_max3 over the rows, no masks;
  * the slot's own pool row is addressed DYNAMICALLY (s_set_gpr_idx_on) and its lane through EXEC = 1 << lane: a cancellation
    is five instructions, an order comes to rest with one masked v_mov per field - instead of one copy of the code per pool
    register (8 x 2 x 8 copies);
  * the passive order's row is found by comparing the best key against all R rows at once (R SGPR pairs), then one
    specialised PICK block per row;
  * the step's event list spans R registers: the current one is copied into a scratch register whenever k crosses a
    multiple of 64.
The event words carry the new order's compare value in their upper half (book_device.hpp key_event_words).
usage: gen_event_asm.py [--check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bourse_amd", "csrc", "event_asm_gen.hpp")


VOL_LIST = True  # the new orders' volumes travel in a second list (by event position) instead of being read from the slot


def gen(N, markets, ext_bounds=False):
    KB, VB, EB = 40, 40 + N, 40 + 2 * N                     # VGPR rows: key, vol, event list
    QB = 40 + 3 * N                                           # ... and the new orders' volumes by event position
    EQB = 36                                                  # SGPR: row-compare results, scratch
    SC = EQB + 2 * N
    names = ["EW", "KP", "V", "KK", "BEST", "PV", "LS", "X", "SLOT", "ALO", "BHI", "RG", "KEND"]
    S = {n: f"s{SC + i}" for i, n in enumerate(names)}
    if ext_bounds:  # the bounds are the caller's (operands): k_step_events cuts its list at the modifications and keeps them across the cuts
        S["ALO"], S["BHI"] = "%[alo]", "%[bhi]"
    last_s = SC + len(names) - 1
    bits = {4: 2, 8: 3}[N]

    def key(r): return f"v{KB + r}"
    def vol(r): return f"v{VB + r}"
    def eq(r): return f"s[{EQB + 2 * r}:{EQB + 2 * r + 1}]"

    L = []
    def e(x): L.append(x)

    DPP = ["quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf",
           "row_half_mirror row_mask:0xf bank_mask:0xf", "row_mirror row_mask:0xf bank_mask:0xf",
           "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]

    def slot_write(pairs):     # row RG, lane EW[5:0] of the given arrays := the given scalars (EXEC = that one lane)
        e(f"s_lshl_b64 exec, 1, {S['EW']}")
        e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(DST)")
        for base, src in pairs:
            e(f"v_mov_b32 v{base}, {src}")
        e("s_set_gpr_idx_off")
        e("s_mov_b64 exec, -1")

    # ---------------------------------------------------------------- entry
    if not ext_bounds:
        e(f"s_mov_b32 {S['ALO']}, 0x80000000")
        e(f"s_mov_b32 {S['BHI']}, 0x7fffffff")
    e("s_cmp_lt_u32 %[k], %[nev]")
    e("s_cbranch_scc0 L_done_%=")
    e("s_cmp_eq_u32 %[chk], 0")
    e("s_cbranch_scc1 L_reload_f_%=")

    def copy(c, vchk):
        # c: label suffix of this copy of the loop; vchk: test "no volume or trading disabled" on every new order (the
        # caller has established that no step of this statement needs it when %[chk] = 0)
        def loop():
            e(f"s_branch L_loop_{c}_%=")

        e(f"L_reload_{c}_%=:")                                  # evc = ev[k >> 6]; this list register ends at KEND
        e(f"s_lshr_b32 {S['RG']}, %[k], 6")
        e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(SRC0)")
        e(f"v_mov_b32 %[evc], v{EB}")
        if VOL_LIST:
            e(f"v_mov_b32 %[evq], v{QB}")
        e("s_set_gpr_idx_off")
        e(f"s_or_b32 {S['X']}, %[k], 63")
        e(f"s_add_u32 {S['X']}, {S['X']}, 1")
        e(f"s_min_u32 {S['KEND']}, {S['X']}, %[nev]")
        # ------------------------------------------------------------ one event
        e(f"L_top_{c}_%=:")
        e(f"v_readlane_b32 {S['EW']}, %[evc], %[k]")
        e(f"s_bfe_u32 {S['RG']}, {S['EW']}, {hex((bits << 16) | 6)}")   # the slot's pool row
        e(f"s_bitcmp1_b32 {S['EW']}, 15")
        e(f"s_cbranch_scc1 L_new_{c}_%=")
        slot_write([(KB, "0")])                                 # Cancellation: the key goes (whichever side it rests on)
        e(f"L_loop_{c}_%=:")
        e("s_add_u32 %[k], %[k], 1")
        e(f"s_cmp_lt_u32 %[k], {S['KEND']}")
        e(f"s_cbranch_scc1 L_top_{c}_%=")
        e("s_cmp_lt_u32 %[k], %[nev]")
        e(f"s_cbranch_scc1 L_reload_{c}_%=")
        e("s_branch L_done_%=")
        # New order: its volume from the slot (ew as a lane select: 4 wait states behind its v_readlane - bfe, bitcmp, branch, idx_on)
        e(f"L_new_{c}_%=:")

        def read_volume():
            if VOL_LIST:
                e(f"v_readlane_b32 {S['V']}, %[evq], %[k]")
            else:
                e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(SRC0)")
                e(f"v_mov_b32 %[vm], v{VB}")
                e("s_set_gpr_idx_off")
                e(f"v_readlane_b32 {S['V']}, %[vm], {S['EW']}")

        if vchk:                                                # (the copy without the test reads it once the order is
            read_volume()                                       # known to reach the reduction: below)
        e(f"s_bitcmp1_b32 {S['EW']}, 14")
        e(f"s_cbranch_scc1 L_bid_{c}_%=")

        def side(tag, agg_bid):
            # agg_bid: searches the asks (signed min), rests among the bids
            tag = tag + c
            op3 = "v_min3_i32" if agg_bid else "v_max3_i32"
            op2 = "v_min_i32" if agg_bid else "v_max_i32"
            oppb, ownb = (S["ALO"], S["BHI"]) if agg_bid else (S["BHI"], S["ALO"])
            mkt = "-1" if agg_bid else "0x10000"
            # the compare value from the event word's upper half (book_device.hpp "SIGNED KEYS")
            e(f"s_or_b32 {S['KP']}, {S['EW']}, 0xffff" if agg_bid else f"s_and_b32 {S['KP']}, {S['EW']}, 0xffff0000")
            if vchk:
                e(f"s_and_b32 {S['X']}, {S['V']}, %[tmask]")        # no volume or trading disabled: no match
                e(f"s_cbranch_scc0 L_restq_{tag}_%=")
            e(f"{'s_cmp_lt_i32' if agg_bid else 's_cmp_gt_i32'} {S['KP']}, {oppb}")   # beyond the bound: cannot cross
            e(f"s_cbranch_scc1 L_restq_{tag}_%=")
            if not vchk:
                read_volume()
            e(f"s_mov_b32 {S['KK']}, %[k]" if agg_bid else f"s_or_b32 {S['KK']}, %[k], 0x80000000")
            e(f"L_match_{tag}_%=:")
            if N == 4:
                e(f"{op3} %[vm], {key(0)}, {key(1)}, {key(2)}")
                e(f"{op2} %[vm], %[vm], {key(3)}")
            else:
                e(f"{op3} %[vm], {key(0)}, {key(1)}, {key(2)}")
                e(f"{op3} %[vt], {key(3)}, {key(4)}, {key(5)}")
                e(f"{op3} %[vm], %[vm], {key(6)}, {key(7)}")
                e(f"{op2} %[vm], %[vm], %[vt]")
            for ctl in DPP:
                e("s_nop 1")
                e(f"{op2}_dpp %[vm], %[vm], %[vm] {ctl}")
            e("s_nop 1")
            e(f"v_readlane_b32 {oppb}, %[vm], 63")                  # the best key, read INTO the bound: exact now
            e("s_nop 1")                                            # ... as a VALU operand: 2 wait states
            for r in range(N):
                e(f"v_cmp_eq_u32_e64 {eq(r)}, {oppb}, {key(r)}")
            e(f"{'s_cmp_gt_i32' if agg_bid else 's_cmp_lt_i32'} {oppb}, {S['KP']}")   # no cross (or nothing there)
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
                e(f"v_writelane_b32 {key(r)}, 0, m0")               # ... and the passive order with it
                e(f"L_B{r}_{tag}_%=:")
                e("s_add_u32 %[trn], %[trn], 1")                    # SCC = carry = buffer full
                e("s_cbranch_scc1 L_fullnext_%=")
                loop()
                e(f"L_A{r}_{tag}_%=:")                              # passive order exhausted, aggressor goes on
                e(f"v_writelane_b32 %[trv], {S['PV']}, m0")
                e(f"s_mov_b32 m0, {S['LS']}")
                e(f"v_writelane_b32 {vol(r)}, 0, m0")
                e(f"v_writelane_b32 {key(r)}, 0, m0")
                e(f"s_sub_u32 {S['V']}, {S['V']}, {S['PV']}")
                e("s_add_u32 %[trn], %[trn], 1")
                e(f"s_cbranch_scc1 L_fullA_{tag}_%=")
                e(f"s_branch L_match_{tag}_%=")
            # buffer full, volume left: the event restarts with what remains
            e(f"L_fullA_{tag}_%=:")
            slot_write([(VB, S["V"])])
            if VOL_LIST:                                            # ... also where the restarted event reads it
                e(f"s_lshl_b64 exec, 1, %[k]")
                e(f"s_lshr_b32 {S['RG']}, %[k], 6")
                e(f"s_set_gpr_idx_on {S['RG']}, gpr_idx(DST)")
                e(f"v_mov_b32 v{QB}, {S['V']}")
                e("s_set_gpr_idx_off")
                e("s_mov_b64 exec, -1")
            e("s_branch L_flush_%=")
            # rests with what the trades left ... (a market order's remainder is dropped: orderbook.rs:521-524)
            e(f"L_rest_{tag}_%=:")
            if markets:
                e(f"s_cmp_eq_u32 {S['KP']}, {mkt}")
                e(f"s_cbranch_scc1 L_loop_{c}_%=")
            e(f"s_xor_b32 {S['X']}, {S['KP']}, %[sq]")
            e(f"{'s_max_i32' if agg_bid else 's_min_i32'} {ownb}, {ownb}, {S['X']}")   # this side's bound covers the new order
            slot_write([(VB, S["V"]), (KB, S["X"])])
            e("s_add_u32 %[sq], %[sq], 1")
            loop()
            # ... or untouched (no volume / trading disabled / beyond the bound - which an EMPTY other side also is: a
            # market order that finds nobody gets here)
            e(f"L_restq_{tag}_%=:")
            if markets:
                e(f"s_cmp_eq_u32 {S['KP']}, {mkt}")
                e(f"s_cbranch_scc1 L_loop_{c}_%=")
            e(f"s_xor_b32 {S['X']}, {S['KP']}, %[sq]")
            e(f"{'s_max_i32' if agg_bid else 's_min_i32'} {ownb}, {ownb}, {S['X']}")
            slot_write([(KB, S["X"])])
            e("s_add_u32 %[sq], %[sq], 1")
            loop()

        side("a", False)
        e(f"L_bid_{c}_%=:")
        side("b", True)

    copy("c", True)
    copy("f", False)
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
    vt = {4: "u32x4", 8: "u32x8"}[N]
    return f'''
// ---- R = {N}{', lists that may carry market orders' if markets else ''}{', bounds kept by the caller across statements (k_step_events)' if ext_bounds else ''}: {len(L)} instructions; rows key v[{KB}:{KB + N - 1}], vol v[{VB}:{VB + N - 1}], lists v[{EB}:{EB + N - 1}] (event words), v[{QB}:{QB + N - 1}] (volumes);
// row-compare results s[{EQB}:{EQB + 2 * N - 1}]; scratch s{SC}..s{last_s}.  EXEC must be all ones on entry (it is restored to that).
__device__ __forceinline__ uint32_t events_key_r{N}{('x' if ext_bounds else 'm') if markets else ''}(uint32_t checked, uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                  uint32_t (&vol)[{N}], uint32_t (&key)[{N}], const uint32_t (&ev)[{N}], uint32_t (&evq)[{N}],
                                                  uint32_t& trk, uint32_t& trv, uint32_t& trs{', uint32_t& alo, uint32_t& bhi' if ext_bounds else ''}) {{
  uint32_t st, vm, vt, evc, evqc;
{'  alo = (uint32_t)__builtin_amdgcn_readfirstlane(alo);' + chr(10) + '  bhi = (uint32_t)__builtin_amdgcn_readfirstlane(bhi);' + chr(10) if ext_bounds else ''}
  auto u32 = [](uint32_t x) {{ return (uint32_t)__builtin_amdgcn_readfirstlane(x); }};
  checked = u32(checked);
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;  // biased: the increment carries out exactly when the buffer is full
  sq = u32(sq);
  {vt} kv, vv, evv, evqv;
#pragma unroll
  for (int r = 0; r < {N}; ++r) {{
    kv[r] = key[r];
    vv[r] = vol[r];
    evv[r] = ev[r];
    evqv[r] = evq[r];
  }}
  asm volatile(
{text}      : [st] "=&s"(st), [vm] "=&v"(vm), [vt] "=&v"(vt), [evc] "=&v"(evc), [evq] "=&v"(evqc), [k] "+s"(k), [trn] "+s"(trn), [sq] "+s"(sq),
        [key] "+{{v[{KB}:{KB + N - 1}]}}"(kv), [vol] "+{{v[{VB}:{VB + N - 1}]}}"(vv), [evqr] "+{{v[{QB}:{QB + N - 1}]}}"(evqv), [trk] "+v"(trk), [trv] "+v"(trv), [trs] "+v"(trs){', [alo] "+s"(alo), [bhi] "+s"(bhi)' if ext_bounds else ''}
      : [ev] "{{v[{EB}:{EB + N - 1}]}}"(evv), [nev] "s"(n_ev), [tmask] "s"(tmask), [chk] "s"(checked)
      : {clob}, "m0", "vcc", "scc", "memory");
#pragma unroll
  for (int r = 0; r < {N}; ++r) {{
    key[r] = kv[r];
    vol[r] = vv[r];
    evq[r] = evqv[r];
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
    src = (HEADER + gen(4, False) + gen(4, True) + gen(8, False) + gen(8, True) + gen(4, True, True) + gen(8, True, True) +
           "\n}  // namespace bkd\n")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == src else 1)
    open(OUT, "w").write(src)
    print("wrote", OUT, len(src.splitlines()), "lines")
