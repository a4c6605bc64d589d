// event_asm.hpp — the event loop of Env::step for a 128-slot pool (R = 2), hand-written for gfx950.
//
// Why assembly: every kernel on this path is bound by the CU's single scalar issue port, and the event
// loop is where the scalar instructions are: the compiled loop spends ≈47 scalar + branch instructions per fill and
// ≈46 per new order that only rests (SQ_INSTS_SALU / SQ_INSTS_BRANCH, profiles/r02/pmc_sq_*.csv) — flag registers for
// loop exits, 64-bit mask juggling, per-stage re-derivation of the slot's register.  Written by hand the same
// semantics take ≈28 and ≈23: the code is specialised by list register (events 0..63 / 64..127), by the slot's pool
// register and by the aggressor's side, so nothing is selected at run time; every exit is one compare + one branch;
// lane selects and shift counts use the event word directly (the hardware reads its low 6 bits).
// ONE asm statement processes a whole step's list, so the operand copies hipcc places around it are paid once per
// step, not once per fill (round 1's per-match asm block lost to exactly that).
//
// Semantics (identical to slot_event_at / match_side in book_device.hpp; the parity suite runs on both):
//   event word ew = slot | EV_NEW (bit 15) | EV_BID (bit 14); pool slot (r = ew bit 6, lane = ew & 63)
//   Cancellation (orderbook.rs:622-644): live[r] &= ~bit (a no-op if the order was filled meanwhile)
//   New (place_order, orderbook.rs:583-611): while volume remains and the book crosses (inclusive test :430/:463):
//     touch = DPP min/max over the opposite side's live prices; oldest order at the touch = the single candidate, or
//     the min `seq` among them (second DPP reduction); match_orders (:843-870): trade record into lane tr_n of the
//     trade buffer, passive volume / liveness updated; the remainder of a limit order rests with a fresh `seq`,
//     a market order's (price sentinel) is dropped.  `tmask` = all ones while trading is enabled.
//   When the 64-record trade buffer fills the statement returns 1 for a flush; an event with volume left has it
//   written back to its slot and is simply dispatched again (it resumes matching with what remains).
//
// Wait states (hipcc pads nothing inside an asm string): VALU-written SGPR -> v_readlane/v_writelane LANE SELECT needs
// 4 (ew: two s_bitcmp1 + two s_cbranch precede its first use as a select); VALU-written SGPR -> VALU operand needs 2
// (touch: s_cmp + s_cbranch; tie stamp: s_nop 1); VGPR write -> DPP read needs 2 (s_nop 1 inside the reduction);
// VGPR write -> v_readlane needs 1 (the reduction ends with s_nop 1).  SALU-written SGPRs / M0 need none.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bkd {

// fixed scratch SGPRs of the statement (declared as clobbers): 64-bit pairs first
#define EA_BIT "s[40:41]"
#define EA_C0 "s[42:43]"
#define EA_C1 "s[44:45]"
#define EA_E0 "s[46:47]"
#define EA_E1 "s[48:49]"
#define EA_EW "s50"
#define EA_P "s51"
#define EA_V "s52"
#define EA_ID "s53"
#define EA_KK "s54"
#define EA_BEST "s55"
#define EA_PV "s56"
#define EA_PID "s57"
#define EA_TV "s58"
#define EA_LS "s59"
#define EA_X "s60"
#define EA_X2 "s61"

// text that only exists for a pool of NR = 2 registers
#define EA_IF2_1(x) ""
#define EA_IF2_2(x) x
#define EA_IF1_1(x) x
#define EA_IF1_2(x) ""

// full-wave DPP reduction of %[vm] (same network as BK_DPP_REDUCE)
#define EA_DPP(OP)                                                                                \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"      \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"      \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"          \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] row_mirror row_mask:0xf bank_mask:0xf\n\t"               \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"             \
  "s_nop 1\n\t" OP " %[vm], %[vm], %[vm] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"             \
  "s_nop 1\n\t"

// next event of list phase PH (inlined at the end of every path instead of a branch to one shared copy: one taken
// branch - an instruction-buffer refill - fewer per event)
#define EA_LOOP(PH, KEND)                       \
  "s_add_u32 %[k], %[k], 1\n\t"                 \
  "s_cmp_lt_u32 %[k], " KEND "\n\t"             \
  "s_cbranch_scc1 L_top_" PH "_%=\n\t"          \
  "s_branch L_end_" PH "_%=\n\t"

// the passive order sits in pool register Q (lane = first set bit of its mask EQ): match_orders on it
#define EA_PICK(Q, EQ, L)                                              \
  "s_ff1_i32_b64 " EA_LS ", " EQ "\n\t"                                \
  "v_readlane_b32 " EA_PV ", %[vol" Q "], " EA_LS "\n\t"               \
  "v_readlane_b32 " EA_PID ", %[id" Q "], " EA_LS "\n\t"               \
  "s_mov_b32 m0, " EA_LS "\n\t"                                        \
  "s_min_u32 " EA_TV ", " EA_V ", " EA_PV "\n\t"                       \
  "s_sub_u32 " EA_PV ", " EA_PV ", " EA_TV "\n\t"                      \
  "v_writelane_b32 %[vol" Q "], " EA_PV ", m0\n\t"                     \
  "s_cmp_lg_u32 " EA_PV ", 0\n\t"                                      \
  "s_cbranch_scc1 L_trade_" L "\n\t"                                   \
  "s_andn2_b64 %[live" Q "], %[live" Q "], " EQ "\n\t"

// "exactly one order at the touch?" - then the second reduction is skipped.  BOURSE_AMD_ASM_ALWAYS_TIE=1 drops the test
// (4 scalar instructions + a branch per fill) and always runs the 13-instruction vector reduction instead.
#ifndef BOURSE_AMD_ASM_ALWAYS_TIE
#define BOURSE_AMD_ASM_ALWAYS_TIE 0
#endif
#if BOURSE_AMD_ASM_ALWAYS_TIE
#define EA_TIE_TEST(L, NR) ""
#else
#define EA_TIE_TEST(L, NR)                                                                 \
  "s_bcnt1_i32_b64 " EA_X ", " EA_E0 "\n\t"                                                \
  EA_IF2_##NR("s_bcnt1_i32_b64 " EA_X2 ", " EA_E1 "\n\t"                                   \
              "s_add_u32 " EA_X ", " EA_X ", " EA_X2 "\n\t")                               \
  "s_cmp_eq_u32 " EA_X ", 1\n\t"                                                           \
  "s_cbranch_scc1 L_pick_" L "\n\t"
#endif

// A New event whose slot is in pool register RG, aggressor side given by the five side-specific pieces:
//   CAND  "s_andn2_b64" (bid: opposite = asks = live & ~bid) / "s_and_b64" (ask: opposite = bids)
//   SENT  "-1" / "0"      neutral element of the touch reduction = price sentinel of a market order of this side
//   VOP, DOP  v_min_u32 / v_max_u32 and the _dpp form
//   NOX   "s_cmp_lt_u32" (bid: p < best ask) / "s_cmp_gt_u32" (ask: p > best bid): no cross
//   KKI   instruction that forms the trade's k word: passive side bit 31
#define EA_SIDE(L, PH, KEND, RG, NR, CAND, SENT, VOP, DOP, NOX, KKI)                                     \
  KKI "\n\t"                                                                                        \
  "s_and_b32 " EA_X ", " EA_V ", %[tmask]\n\t"          /* no volume or trading disabled: no match */ \
  "s_cbranch_scc0 L_rest_" L "\n\t"                                                                 \
  "L_match_" L ":\n\t"                                                                              \
  CAND " " EA_C0 ", %[live0], %[bid0]\n\t"               /* SCC = candidates exist (NR = 1) */       \
  EA_IF2_##NR(CAND " " EA_C1 ", %[live1], %[bid1]\n\t"                                              \
              "s_or_b64 " EA_E0 ", " EA_C0 ", " EA_C1 "\n\t")                                       \
  "s_cbranch_scc0 L_rest_" L "\n\t"                     /* best_order_idx() == None */              \
  "v_cndmask_b32_e64 %[vm], " SENT ", %[price0], " EA_C0 "\n\t"                                     \
  EA_IF2_##NR("v_cndmask_b32_e64 %[vm2], " SENT ", %[price1], " EA_C1 "\n\t"                        \
              VOP " %[vm], %[vm], %[vm2]\n\t")                                                      \
  EA_DPP(DOP)                                                                                       \
  "v_readlane_b32 " EA_BEST ", %[vm], 63\n\t"                                                       \
  NOX " " EA_P ", " EA_BEST "\n\t"                                                                  \
  "s_cbranch_scc1 L_rest_" L "\n\t"                                                                 \
  "v_cmp_eq_u32_e64 " EA_E0 ", " EA_BEST ", %[price0]\n\t"                                          \
  EA_IF2_##NR("v_cmp_eq_u32_e64 " EA_E1 ", " EA_BEST ", %[price1]\n\t")                             \
  "s_and_b64 " EA_E0 ", " EA_E0 ", " EA_C0 "\n\t"                                                   \
  EA_IF2_##NR("s_and_b64 " EA_E1 ", " EA_E1 ", " EA_C1 "\n\t")                                      \
  EA_TIE_TEST(L, NR)                                                                                \
  /* several orders rest at the touch: the oldest (min seq stamp, unique per book) is next in the queue */ \
  "v_cndmask_b32_e64 %[vm], -1, %[seq0], " EA_E0 "\n\t"                                             \
  EA_IF2_##NR("v_cndmask_b32_e64 %[vm2], -1, %[seq1], " EA_E1 "\n\t"                                \
              "v_min_u32 %[vm], %[vm], %[vm2]\n\t")                                                 \
  EA_DPP("v_min_u32_dpp")                                                                           \
  "v_readlane_b32 " EA_X ", %[vm], 63\n\t"                                                          \
  "s_nop 1\n\t"                                                                                     \
  "v_cmp_eq_u32_e64 " EA_C0 ", " EA_X ", %[seq0]\n\t"                                               \
  EA_IF2_##NR("v_cmp_eq_u32_e64 " EA_C1 ", " EA_X ", %[seq1]\n\t")                                  \
  "s_and_b64 " EA_E0 ", " EA_E0 ", " EA_C0 "\n\t"                                                   \
  EA_IF2_##NR("s_and_b64 " EA_E1 ", " EA_E1 ", " EA_C1 "\n\t")                                      \
  "L_pick_" L ":\n\t"                                                                               \
  EA_IF2_##NR("s_cmp_lg_u64 " EA_E0 ", 0\n\t"                                                       \
              "s_cbranch_scc0 L_pick1_" L "\n\t")                                                   \
  EA_PICK("0", EA_E0, L)                                                                            \
  EA_IF2_##NR("s_branch L_trade_" L "\n\t"                                                          \
              "L_pick1_" L ":\n\t"                                                                  \
              EA_PICK("1", EA_E1, L))                                                               \
  "L_trade_" L ":\n\t"                                  /* Trade record, lane tr_n of the buffer */ \
  "s_sub_u32 " EA_V ", " EA_V ", " EA_TV "\n\t"                                                     \
  "s_add_u32 %[tvol], %[tvol], " EA_TV "\n\t"                                                       \
  "s_mov_b32 m0, %[trn]\n\t"                                                                        \
  "v_writelane_b32 %[trk], " EA_KK ", m0\n\t"                                                       \
  "v_writelane_b32 %[trp], " EA_BEST ", m0\n\t"                                                     \
  "v_writelane_b32 %[trv], " EA_TV ", m0\n\t"                                                       \
  "v_writelane_b32 %[tra], " EA_ID ", m0\n\t"                                                       \
  "v_writelane_b32 %[trs], " EA_PID ", m0\n\t"                                                      \
  "s_add_u32 %[trn], %[trn], 1\n\t"                                                                 \
  "s_cmp_eq_u32 %[trn], 64\n\t"                                                                     \
  "s_cbranch_scc1 L_full_" L "\n\t"                                                                 \
  "s_cmp_lg_u32 " EA_V ", 0\n\t"                                                                    \
  "s_cbranch_scc1 L_match_" L "\n\t"                                                                \
  EA_LOOP(PH, KEND)                                     /* Filled: nothing rests */                 \
  "L_full_" L ":\n\t"                                   /* buffer full: flush outside */            \
  "s_cmp_eq_u32 " EA_V ", 0\n\t"                                                                    \
  "s_cbranch_scc1 L_fullnext_%=\n\t"                                                                \
  "s_mov_b32 m0, " EA_EW "\n\t"                                                                     \
  "v_writelane_b32 %[vol" RG "], " EA_V ", m0\n\t"      /* the event restarts with what remains */  \
  "s_branch L_flush_%=\n\t"                                                                         \
  "L_rest_" L ":\n\t"                                                                               \
  "s_cmp_eq_u32 " EA_P ", " SENT "\n\t"                 /* market remainder: dropped (:521-524) */  \
  "s_cbranch_scc1 L_next_" PH "_%=\n\t"                                                             \
  "s_mov_b32 m0, " EA_EW "\n\t"                                                                     \
  "v_writelane_b32 %[vol" RG "], " EA_V ", m0\n\t"                                                  \
  "v_writelane_b32 %[seq" RG "], %[seqc], m0\n\t"                                                   \
  "s_lshl_b64 " EA_BIT ", 1, " EA_EW "\n\t"                                                         \
  "s_or_b64 %[live" RG "], %[live" RG "], " EA_BIT "\n\t"                                           \
  "s_add_u32 %[seqc], %[seqc], 1\n\t"                                                               \
  EA_LOOP(PH, KEND)

// a New event in pool register RG of list phase PH: read the order, dispatch on its side
#define EA_NEW(PH, KEND, RG, NR)                                                                          \
  "v_readlane_b32 " EA_P ", %[price" RG "], " EA_EW "\n\t"                                          \
  "v_readlane_b32 " EA_V ", %[vol" RG "], " EA_EW "\n\t"                                            \
  "v_readlane_b32 " EA_ID ", %[id" RG "], " EA_EW "\n\t"                                            \
  "s_bitcmp1_b32 " EA_EW ", 14\n\t"                                                                 \
  "s_cbranch_scc1 L_bid_" PH RG "_%=\n\t"                                                           \
  EA_SIDE("a" PH RG "_%=", PH, KEND, RG, NR, "s_and_b64", "0", "v_max_u32", "v_max_u32_dpp", "s_cmp_gt_u32", \
          "s_or_b32 " EA_KK ", %[k], 0x80000000")                                                   \
  "L_bid_" PH RG "_%=:\n\t"                                                                         \
  EA_SIDE("b" PH RG "_%=", PH, KEND, RG, NR, "s_andn2_b64", "-1", "v_min_u32", "v_min_u32_dpp", "s_cmp_lt_u32", \
          "s_mov_b32 " EA_KK ", %[k]")

// the events of one list register (PH = "0": k < kend0 from ev0, "1": k < nev from ev1)
#define EA_PHASE(PH, KEND, NR)                                                                      \
  "L_top_" PH "_%=:\n\t"                                                                            \
  "v_readlane_b32 " EA_EW ", %[ev" PH "], %[k]\n\t"                                                 \
  "s_bitcmp1_b32 " EA_EW ", 15\n\t"                                                                 \
  "s_cbranch_scc1 L_new_" PH "_%=\n\t"                                                              \
  "s_lshl_b64 " EA_BIT ", 1, " EA_EW "\n\t"             /* Cancellation */                          \
  EA_IF2_##NR("s_bitcmp1_b32 " EA_EW ", 6\n\t"                                                      \
              "s_cbranch_scc1 L_can1_" PH "_%=\n\t")                                                \
  "s_andn2_b64 %[live0], %[live0], " EA_BIT "\n\t"                                                  \
  EA_IF2_##NR(EA_LOOP(PH, KEND)                                                                     \
              "L_can1_" PH "_%=:\n\t"                                                               \
              "s_andn2_b64 %[live1], %[live1], " EA_BIT "\n\t")                                     \
  "L_next_" PH "_%=:\n\t"                                                                           \
  EA_LOOP(PH, KEND)                                                                                 \
  "L_new_" PH "_%=:\n\t"                                                                            \
  EA_IF2_##NR("s_bitcmp1_b32 " EA_EW ", 6\n\t"                                                      \
              "s_cbranch_scc1 L_new1_" PH "_%=\n\t")                                                \
  EA_IF1_##NR("s_nop 1\n\t")                            /* ew as a lane select: 4 wait states */    \
  EA_NEW(PH, KEND, "0", NR)                                                                               \
  EA_IF2_##NR("L_new1_" PH "_%=:\n\t"                                                               \
              EA_NEW(PH, KEND, "1", NR))

// Processes events k .. n_ev-1 of the step's (shuffled) list.  Returns 0 when the list is done, 1 when the trade
// buffer is full (flush it, call again).  All scalars are wave-uniform.
__device__ __forceinline__ uint32_t events_asm_r2(uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n,
                                                  uint32_t& seq_ctr, uint32_t& trade_vol, uint64_t& live0,
                                                  uint64_t& live1, uint64_t bid0, uint64_t bid1, uint32_t price0,
                                                  uint32_t price1, uint32_t& vol0, uint32_t& vol1, uint32_t id0,
                                                  uint32_t id1, uint32_t& seq0, uint32_t& seq1, uint32_t ev0,
                                                  uint32_t ev1, uint32_t& trk, uint32_t& trp, uint32_t& trv,
                                                  uint32_t& tra, uint32_t& trs) {
  uint32_t st, vm, vm2;
  // "s" operands must be provably wave-uniform: the compiler parks some of these book scalars in VGPRs
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  auto u64 = [&](uint64_t x) { return ((uint64_t)u32((uint32_t)(x >> 32)) << 32) | u32((uint32_t)x); };
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  tr_n = u32(tr_n);
  seq_ctr = u32(seq_ctr);
  trade_vol = u32(trade_vol);
  live0 = u64(live0);
  live1 = u64(live1);
  bid0 = u64(bid0);
  bid1 = u64(bid1);
  const uint32_t kend0 = n_ev < 64u ? n_ev : 64u;
  asm volatile(
      "s_cmp_lt_u32 %[k], %[kend0]\n\t"
      "s_cbranch_scc1 L_top_0_%=\n\t"
      "s_branch L_end_0_%=\n\t"
      EA_PHASE("0", "%[kend0]", 2)
      "L_end_0_%=:\n\t"
      "s_cmp_lt_u32 %[k], %[nev]\n\t"
      "s_cbranch_scc0 L_done_%=\n\t"
      EA_PHASE("1", "%[nev]", 2)
      "L_end_1_%=:\n\t"
      "L_done_%=:\n\t"
      "s_mov_b32 %[st], 0\n\t"
      "s_branch L_out_%=\n\t"
      "L_fullnext_%=:\n\t"
      "s_add_u32 %[k], %[k], 1\n\t"
      "L_flush_%=:\n\t"
      "s_mov_b32 %[st], 1\n\t"
      "L_out_%=:\n\t"
      "s_nop 1"
      : [st] "=&s"(st), [vm] "=&v"(vm), [vm2] "=&v"(vm2), [k] "+s"(k), [trn] "+s"(tr_n), [seqc] "+s"(seq_ctr),
        [tvol] "+s"(trade_vol), [live0] "+s"(live0), [live1] "+s"(live1), [vol0] "+v"(vol0), [vol1] "+v"(vol1),
        [seq0] "+v"(seq0), [seq1] "+v"(seq1), [trk] "+v"(trk), [trp] "+v"(trp), [trv] "+v"(trv), [tra] "+v"(tra),
        [trs] "+v"(trs)
      : [price0] "v"(price0), [price1] "v"(price1), [id0] "v"(id0), [id1] "v"(id1), [ev0] "v"(ev0), [ev1] "v"(ev1),
        [bid0] "s"(bid0), [bid1] "s"(bid1), [nev] "s"(n_ev), [kend0] "s"(kend0), [tmask] "s"(tmask)
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
        "s56", "s57", "s58", "s59", "s60", "s61", "m0", "vcc", "scc", "memory");
  return st;
}

// The same for a 64-slot pool (R = 1): one list register, one pool register.
__device__ __forceinline__ uint32_t events_asm_r1(uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n,
                                                  uint32_t& seq_ctr, uint32_t& trade_vol, uint64_t& live0, uint64_t bid0,
                                                  uint32_t price0, uint32_t& vol0, uint32_t id0, uint32_t& seq0,
                                                  uint32_t ev0, uint32_t& trk, uint32_t& trp, uint32_t& trv,
                                                  uint32_t& tra, uint32_t& trs) {
  uint32_t st, vm;
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  auto u64 = [&](uint64_t x) { return ((uint64_t)u32((uint32_t)(x >> 32)) << 32) | u32((uint32_t)x); };
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  tr_n = u32(tr_n);
  seq_ctr = u32(seq_ctr);
  trade_vol = u32(trade_vol);
  live0 = u64(live0);
  bid0 = u64(bid0);
  asm volatile(
      "s_cmp_lt_u32 %[k], %[nev]\n\t"
      "s_cbranch_scc0 L_done_%=\n\t"
      EA_PHASE("0", "%[nev]", 1)
      "L_end_0_%=:\n\t"
      "L_done_%=:\n\t"
      "s_mov_b32 %[st], 0\n\t"
      "s_branch L_out_%=\n\t"
      "L_fullnext_%=:\n\t"
      "s_add_u32 %[k], %[k], 1\n\t"
      "L_flush_%=:\n\t"
      "s_mov_b32 %[st], 1\n\t"
      "L_out_%=:\n\t"
      "s_nop 1"
      : [st] "=&s"(st), [vm] "=&v"(vm), [k] "+s"(k), [trn] "+s"(tr_n), [seqc] "+s"(seq_ctr), [tvol] "+s"(trade_vol),
        [live0] "+s"(live0), [vol0] "+v"(vol0), [seq0] "+v"(seq0), [trk] "+v"(trk), [trp] "+v"(trp), [trv] "+v"(trv),
        [tra] "+v"(tra), [trs] "+v"(trs)
      : [price0] "v"(price0), [id0] "v"(id0), [ev0] "v"(ev0), [bid0] "s"(bid0), [nev] "s"(n_ev), [tmask] "s"(tmask)
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
        "s56", "s57", "s58", "s59", "s60", "s61", "m0", "vcc", "scc", "memory");
  return st;
}

// ==================================================================================
// KEYED event loop: price-time priority as ONE 32-bit sort key per resting order, so a match step is a single DPP
// reduction - no "how many orders at the touch" test, no second reduction over the arrival stamps, no candidate-mask
// algebra before or after it (the key is unique, so `key == best` selects exactly one lane).
//
// SIGNED keys (round 4; book_device.hpp "SIGNED KEYS" has the layout and the proof obligations):
//   ask:  1 | price - pbase (15 bits) | seq - sbase            negative as an i32    best ask = signed MIN over all lanes
//   bid:  0 | price - pbase           | 0xFFFF - (seq - sbase) positive              best bid = signed MAX over all lanes
//   any other pool lane (free, cancelled, filled, still pending): 0 - invisible to both searches, so the reduction
//   starts from the raw key registers: round 3's side-in-bit-0 keys needed `live & ~bid` / `live & bid` (scalar) and a
//   select of the neutral element (vector) per pool register in front of EVERY reduction, and live masks kept current
//   by every cancellation, fill and rest.  Here an order leaves by a zero written to its key lane and the live masks are
//   rebuilt once per step from `key != 0` (keys_end).
//   book_device.hpp (key_window) checks that the step's prices and stamps fit - if not, the step runs on the loop above -
//   and rebuilds the keys from {price, seq} at the start of every step, so nothing about them is persistent state.
//   A NEW order's compare value kp comes from the upper half of its event word (key_event_words), one scalar instruction:
//     bid: kp = ew | 0xFFFF (sign bit set by the set-up)  crosses iff best ask <= kp;   ask: kp = ew & 0xFFFF0000  crosses
//     iff best bid >= kp; with nothing on the other side the reduction returns a value of the wrong sign and the same
//     signed compare says "no cross".  It rests as kp ^ sq, sq = 0x80000000 | (seq_ctr - sbase), advancing by 1.
//   These lists (the event words of RandomAgents) carry no market orders.
//
// The fill itself is specialised three ways on one scalar subtract (SCC = borrow of passive - aggressor volume):
//   aggressor wants more  -> passive order gone, trade = its volume, match again (unconditional branch)
//   aggressor exhausted   -> trade = aggressor's volume, next event (the passive order dies too iff nothing is left)
// and the trade-buffer index is kept biased by -64: `s_add_u32 trn, trn, 1` carries out exactly when the buffer is full
// (lane selects read the low 6 bits).  The step's traded volume is summed from the buffer at every flush (book_device.hpp)
// instead of one scalar add per trade.
// (Round 3's LAZY CANCELLATIONS - cancellations as a death time per pool lane instead of loop iterations - were measured
// twice and never shipped: C3 253 -> 250 M, round 4 279.9 -> 262.9 M; docs/EXPERIMENTS.md.  The code left with the
// side-in-bit-0 keys it was written for.)
// ==================================================================================
#define EK_KP EA_P  // the aggressor's compare value lives where the loop above keeps its price
// Bounds that let a new order that cannot cross skip the reduction: EK_ALO <= best ask key, EK_BHI >= best bid key (i32).
// Exact right after a reduction of that side, still valid after any removal (the best only moves away), pulled in when
// an order rests beyond them; the loosest values at statement entry.
#define EK_ALO "s62"
#define EK_BHI "s63"

// COMPACT TRADE RECORDS (round 4).  The loop writes THREE words per trade - the k word (event position | passive side),
// the volume, and the passive order's POOL SLOT - instead of five, and reads ONE lane of the passive order (its volume)
// instead of three: the trade's price and both order ids are gathered from the pool registers when the buffer is flushed
// (book_device.hpp flush_trades_compact; ids and prices of pool slots never change inside a step: a slot gets its new order
// before the loop and is never re-used in the step).  The record words that do not depend on the passive order's volume
// are written right behind its v_readlane, in front of the scalar subtract that needs it: a scalar instruction issued
// right after a VALU write of an SGPR waits ~16 clocks for it (scripts/micro/mixed_issue_bench.hip), dependent or not.
//   EK_PICK: the passive order is the single lane of mask EQ in pool register Q; SLOTW = its slot word (Q = 0: the lane;
//   Q = 1: lane | 64, formed by SLOTI)
#define EK_PICK(Q, EQ, L, PH, KEND, SLOTI, SLOTW)                                                     \
  "s_ff1_i32_b64 " EA_LS ", " EQ "\n\t"                                                               \
  SLOTI                                                                                               \
  "s_mov_b32 m0, %[trn]\n\t"                                                                          \
  "v_readlane_b32 " EA_PV ", %[vol" Q "], " EA_LS "\n\t"                                              \
  "v_writelane_b32 %[trk], " EA_KK ", m0\n\t"                                                         \
  "v_writelane_b32 %[trs], " SLOTW ", m0\n\t"                                                         \
  "s_sub_u32 " EA_X ", " EA_PV ", " EA_V "\n\t"         /* SCC = borrow: the passive order is the smaller one */ \
  "s_cbranch_scc1 L_A" Q "_" L "\n\t"                                                                 \
  "v_writelane_b32 %[trv], " EA_V ", m0\n\t"            /* aggressor exhausted; X = passive remainder */ \
  "s_mov_b32 m0, " EA_LS "\n\t"                                                                       \
  "v_writelane_b32 %[vol" Q "], " EA_X ", m0\n\t"                                                     \
  "s_cmp_eq_u32 " EA_X ", 0\n\t"                                                                      \
  "s_cbranch_scc0 L_B" Q "_" L "\n\t"                                                                 \
  "v_writelane_b32 %[key" Q "], 0, m0\n\t"              /* ... and the passive order with it */       \
  "L_B" Q "_" L ":\n\t"                                                                               \
  "s_add_u32 %[trn], %[trn], 1\n\t"                     /* SCC = carry = buffer full */               \
  "s_cbranch_scc1 L_fullnext_%=\n\t"                                                                  \
  EA_LOOP(PH, KEND)                                                                                   \
  "L_A" Q "_" L ":\n\t"                                 /* passive order exhausted, aggressor goes on */ \
  "v_writelane_b32 %[trv], " EA_PV ", m0\n\t"                                                         \
  "s_mov_b32 m0, " EA_LS "\n\t"                                                                       \
  "v_writelane_b32 %[vol" Q "], 0, m0\n\t"                                                            \
  "v_writelane_b32 %[key" Q "], 0, m0\n\t"                                                            \
  "s_sub_u32 " EA_V ", " EA_V ", " EA_PV "\n\t"                                                       \
  "s_add_u32 %[trn], %[trn], 1\n\t"                                                                   \
  "s_cbranch_scc1 L_fullA_" L "\n\t"                                                                  \
  "s_branch L_match_" L "\n\t"

// VCHK: EK_VCHK(L) - "no volume or trading disabled: no match" - or "" when the caller has established that trading is
// enabled and no new order of this step has volume 0 (two scalar instructions per new order; book_device.hpp)
#define EK_VCHK(L)                                   \
  "s_and_b32 " EA_X ", " EA_V ", %[tmask]\n\t"       \
  "s_cbranch_scc0 L_restq_" L "\n\t"
//   KPI   the compare value from the event word;  SKIP / OPPB "beyond the other side's bound: cannot cross";
//   VOP / DOP the reduction (v_min_i32 / v_max_i32 and its DPP form);  NOX "no cross" on the reduction's result;
//   KKI   the trade records' k word;  OWNB / PULL this side's bound covers the order that rests
//   VLATE the order's volume, read only once it is known to reach the reduction (the copies without VCHK: an order beyond
//   the bound rests with the volume its slot already holds - one lane read and one hand-over wait less)
#define EK_VRD(RG) "v_readlane_b32 " EA_V ", %[vol" RG "], " EA_EW "\n\t"
//   MK    "" or EK_MKT(value): a MARKET order's compare value never rests (orderbook.rs:521-524, :564-567) - the lists of the
//   host-driven step (step_events.hpp) can carry them; two scalar instructions per order that would rest
#define EK_MKT(PH, KEND, L, MKTV)                                                                     \
  "s_cmp_eq_u32 " EK_KP ", " MKTV "\n\t"                                                              \
  "s_cbranch_scc0 L_dorest_" L "\n\t"                                                                \
  EA_LOOP(PH, KEND)                                                                                   \
  "L_dorest_" L ":\n\t"
#define EK_SIDE(L, PH, KEND, RG, NR, KPI, VOP, DOP, NOX, KKI, OPPB, SKIP, OWNB, PULL, VCHK, VLATE, MK) \
  KPI "\n\t"                                                                                          \
  VCHK                                                                                                \
  SKIP " " EK_KP ", " OPPB "\n\t"                      /* beyond the bound: cannot cross */           \
  "s_cbranch_scc1 L_restq_" L "\n\t"                                                                  \
  VLATE                                                                                               \
  KKI "\n\t"                                           /* (only a trade needs the k word) */          \
  "L_match_" L ":\n\t"                                                                                \
  EA_IF1_##NR("v_mov_b32 %[vm], %[key0]\n\t")                                                         \
  EA_IF2_##NR(VOP " %[vm], %[key0], %[key1]\n\t")                                                     \
  EA_DPP(DOP)                                                                                         \
  "v_readlane_b32 " OPPB ", %[vm], 63\n\t"             /* the best key, read INTO the bound: exact now */ \
  "s_nop 1\n\t"                                        /* ... as a VALU operand: 2 wait states */     \
  /* both "which lane is it" compares BEFORE any scalar instruction: one wait for the vector unit's SGPR writes   \
     instead of one behind the v_readlane and one behind the compares */                             \
  "v_cmp_eq_u32_e64 " EA_E0 ", " OPPB ", %[key0]\n\t"                                                 \
  EA_IF2_##NR("v_cmp_eq_u32_e64 " EA_E1 ", " OPPB ", %[key1]\n\t")                                    \
  NOX " " OPPB ", " EK_KP "\n\t"                                                                      \
  "s_cbranch_scc1 L_rest_" L "\n\t"                                                                   \
  EA_IF2_##NR("s_cmp_lg_u64 " EA_E0 ", 0\n\t"                                                         \
              "s_cbranch_scc0 L_pick1_" L "\n\t")                                                     \
  EK_PICK("0", EA_E0, L, PH, KEND, "", EA_LS)                                                         \
  EA_IF2_##NR("L_pick1_" L ":\n\t"                                                                    \
              EK_PICK("1", EA_E1, L, PH, KEND, "s_or_b32 " EA_PID ", " EA_LS ", 64\n\t", EA_PID))        \
  "L_fullA_" L ":\n\t"                                  /* buffer full, volume left: the event restarts */ \
  "s_mov_b32 m0, " EA_EW "\n\t"                                                                       \
  "v_writelane_b32 %[vol" RG "], " EA_V ", m0\n\t"                                                    \
  "s_branch L_flush_%=\n\t"                                                                           \
  "L_rest_" L ":\n\t"                                  /* rests with what the trades left ... */      \
  "s_mov_b32 m0, " EA_EW "\n\t"                                                                       \
  "v_writelane_b32 %[vol" RG "], " EA_V ", m0\n\t"                                                    \
  "L_restq_" L ":\n\t"                                 /* ... or untouched */                         \
  MK                                                                                                  \
  "s_mov_b32 m0, " EA_EW "\n\t"                                                                       \
  "s_xor_b32 " EA_X ", " EK_KP ", %[sq]\n\t"                                                          \
  PULL " " OWNB ", " OWNB ", " EA_X "\n\t"             /* this side's bound covers the new order */  \
  "v_writelane_b32 %[key" RG "], " EA_X ", m0\n\t"                                                    \
  "s_add_u32 %[sq], %[sq], 1\n\t"                                                                     \
  EA_LOOP(PH, KEND)

#define EK_NOMK(PH, KEND, L, MKTV) ""
#define EK_NEW(PH, KEND, RG, NR, CHK, VE, VL, MKM)                                                    \
  VE(EK_VRD(RG))                                                                                      \
  "s_bitcmp1_b32 " EA_EW ", 14\n\t"                                                                   \
  "s_cbranch_scc1 L_bid_" PH RG "_%=\n\t"                                                             \
  /* an ask: searches the bids (cannot cross if kp > bid bound), rests among the asks (ask bound = min) */ \
  EK_SIDE("a" PH RG "_%=", PH, KEND, RG, NR, "s_and_b32 " EK_KP ", " EA_EW ", 0xffff0000", "v_max_i32", "v_max_i32_dpp", \
          "s_cmp_lt_i32", "s_or_b32 " EA_KK ", %[k], 0x80000000", EK_BHI, "s_cmp_gt_i32", EK_ALO, "s_min_i32", \
          CHK("a" PH RG "_%="), VL(EK_VRD(RG)), MKM(PH, KEND, "a" PH RG "_%=", "0x10000"))             \
  "L_bid_" PH RG "_%=:\n\t"                                                                           \
  EK_SIDE("b" PH RG "_%=", PH, KEND, RG, NR, "s_or_b32 " EK_KP ", " EA_EW ", 0xffff", "v_min_i32", "v_min_i32_dpp", \
          "s_cmp_gt_i32", "s_mov_b32 " EA_KK ", %[k]", EK_ALO, "s_cmp_lt_i32", EK_BHI, "s_max_i32", CHK("b" PH RG "_%="), \
          VL(EK_VRD(RG)), MKM(PH, KEND, "b" PH RG "_%=", "-1"))

#define EK_NOCHK(L) ""
#define EK_ID(x) x
#define EK_NONE(x) ""
#define EK_PHASE(PH, EVN, KEND, NR, CHK, VE, VL) EK_PHASE_M(PH, EVN, KEND, NR, CHK, VE, VL, EK_NOMK)
#define EK_PHASE_M(PH, EVN, KEND, NR, CHK, VE, VL, MKM)                                               \
  "L_top_" PH "_%=:\n\t"                                                                              \
  "v_readlane_b32 " EA_EW ", %[ev" EVN "], %[k]\n\t"                                                  \
  "s_bitcmp1_b32 " EA_EW ", 15\n\t"                                                                   \
  "s_cbranch_scc1 L_new_" PH "_%=\n\t"                                                                \
  /* Cancellation: a zero into the slot's key lane, whichever side it rests on (ew as a lane select: 4 wait states   \
     behind its v_readlane) */                                                                        \
  EA_IF2_##NR("s_bitcmp1_b32 " EA_EW ", 6\n\t"                                                        \
              "s_cbranch_scc1 L_can1_" PH "_%=\n\t")                                                  \
  EA_IF1_##NR("s_nop 1\n\t")                                                                          \
  "v_writelane_b32 %[key0], 0, " EA_EW "\n\t"                                                         \
  EA_IF2_##NR(EA_LOOP(PH, KEND)                                                                       \
              "L_can1_" PH "_%=:\n\t"                                                                 \
              "v_writelane_b32 %[key1], 0, " EA_EW "\n\t")                                            \
  EA_LOOP(PH, KEND)                                                                                   \
  "L_new_" PH "_%=:\n\t"                                                                              \
  EA_IF2_##NR("s_bitcmp1_b32 " EA_EW ", 6\n\t"                                                        \
              "s_cbranch_scc1 L_new1_" PH "_%=\n\t")                                                  \
  EA_IF1_##NR("s_nop 1\n\t")                            /* ew as a lane select: 4 wait states */      \
  EK_NEW(PH, KEND, "0", NR, CHK, VE, VL, MKM)                                                         \
  EA_IF2_##NR("L_new1_" PH "_%=:\n\t"                                                                 \
              EK_NEW(PH, KEND, "1", NR, CHK, VE, VL, MKM))

#define EK_TAIL                     \
  "L_done_%=:\n\t"                  \
  "s_mov_b32 %[st], 0\n\t"          \
  "s_branch L_out_%=\n\t"           \
  "L_fullnext_%=:\n\t"              \
  "s_add_u32 %[k], %[k], 1\n\t"     \
  "L_flush_%=:\n\t"                 \
  "s_mov_b32 %[st], 1\n\t"          \
  "L_out_%=:\n\t"                   \
  "s_nop 1"

#define EK_CLOBBERS                                                                                                 \
  "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",  \
      "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "m0", "vcc", "scc", "memory"

// ONE statement holds the loop twice: with the "no volume or trading disabled" test on every new order (phases 0 / 1) and
// without it (phases 2 / 3; %[chk] = 0: the caller has established that trading is on and no new order of the step has
// volume 0).  Two statements - one per variant - cost the kernel 10 VGPRs (46 instead of 36), and with them two of the
// seven event waves that fit beside a k_agents_fsm wave on a SIMD: C3 277 -> 210 M.
#define EK_BINIT "s_mov_b32 " EK_ALO ", 0x80000000\n\t" "s_mov_b32 " EK_BHI ", 0x7fffffff\n\t"
#define EK_NOBOPS
#define EK_R2_STMT_X(MKM, BINIT, BOPS) \
  asm volatile( \
      BINIT \
      "s_cmp_eq_u32 %[chk], 0\n\t" \
      "s_cbranch_scc1 L_fast_%=\n\t" \
      "s_cmp_lt_u32 %[k], %[kend0]\n\t" \
      "s_cbranch_scc1 L_top_0_%=\n\t" \
      "s_branch L_end_0_%=\n\t" \
      EK_PHASE_M("0", "0", "%[kend0]", 2, EK_VCHK, EK_ID, EK_NONE, MKM) \
      "L_end_0_%=:\n\t" \
      "s_cmp_lt_u32 %[k], %[nev]\n\t" \
      "s_cbranch_scc0 L_done_%=\n\t" \
      EK_PHASE_M("1", "1", "%[nev]", 2, EK_VCHK, EK_ID, EK_NONE, MKM) \
      "L_end_1_%=:\n\t" \
      "s_branch L_done_%=\n\t" \
      "L_fast_%=:\n\t" \
      "s_cmp_lt_u32 %[k], %[kend0]\n\t" \
      "s_cbranch_scc1 L_top_2_%=\n\t" \
      "s_branch L_end_2_%=\n\t" \
      EK_PHASE_M("2", "0", "%[kend0]", 2, EK_NOCHK, EK_NONE, EK_ID, MKM) \
      "L_end_2_%=:\n\t" \
      "s_cmp_lt_u32 %[k], %[nev]\n\t" \
      "s_cbranch_scc0 L_done_%=\n\t" \
      EK_PHASE_M("3", "1", "%[nev]", 2, EK_NOCHK, EK_NONE, EK_ID, MKM) \
      "L_end_3_%=:\n\t" \
      EK_TAIL \
      : [st] "=&s"(st), [vm] "=&v"(vm), [k] "+s"(k), [trn] "+s"(trn), [sq] "+s"(sq), \
        [vol0] "+v"(vol0), [vol1] "+v"(vol1), [key0] "+v"(key0), \
        [key1] "+v"(key1), [trk] "+v"(trk), [trv] "+v"(trv), [trs] "+v"(trs) BOPS \
      : [ev0] "v"(ev0), [ev1] "v"(ev1), \
        [nev] "s"(n_ev), [kend0] "s"(kend0), [tmask] "s"(tmask), [chk] "s"(checked) \
      : EK_CLOBBERS);

#define EK_R1_STMT_X(MKM, BINIT, BOPS) \
  asm volatile( \
      BINIT \
      "s_cmp_lt_u32 %[k], %[nev]\n\t" \
      "s_cbranch_scc0 L_done_%=\n\t" \
      "s_cmp_eq_u32 %[chk], 0\n\t" \
      "s_cbranch_scc1 L_top_2_%=\n\t" \
      EK_PHASE_M("0", "0", "%[nev]", 1, EK_VCHK, EK_ID, EK_NONE, MKM) \
      "L_end_0_%=:\n\t" \
      "s_branch L_done_%=\n\t" \
      EK_PHASE_M("2", "0", "%[nev]", 1, EK_NOCHK, EK_NONE, EK_ID, MKM) \
      "L_end_2_%=:\n\t" \
      EK_TAIL \
      : [st] "=&s"(st), [vm] "=&v"(vm), [k] "+s"(k), [trn] "+s"(trn), [sq] "+s"(sq), \
        [vol0] "+v"(vol0), [key0] "+v"(key0), [trk] "+v"(trk), [trv] "+v"(trv), [trs] "+v"(trs) BOPS \
      : [ev0] "v"(ev0), [nev] "s"(n_ev), [tmask] "s"(tmask), [chk] "s"(checked) \
      : EK_CLOBBERS);

#define EK_R2_STMT EK_R2_STMT_X(EK_NOMK, EK_BINIT, EK_NOBOPS)
#define EK_R1_STMT EK_R1_STMT_X(EK_NOMK, EK_BINIT, EK_NOBOPS)
// ... and for lists that may carry MARKET orders (the host-driven step, step_events.hpp).  Their bounds are the CALLER's
// (operands [alo] / [bhi], the loosest values before a step's first statement): that step cuts its list at every modification,
// and a statement that started from the loosest bounds again made the next new bid and the next new ask pay a reduction each.
// The bounds stay valid across the cuts: between two statements an order only LEAVES the book (a replacement's key := 0).
#define EK_BOPS , [alo] "+s"(alo), [bhi] "+s"(bhi)
#define EK_R2M_STMT EK_R2_STMT_X(EK_MKT, "", EK_BOPS)
#define EK_R1M_STMT EK_R1_STMT_X(EK_MKT, "", EK_BOPS)

// Keyed form of events_asm_r2: `sq` = 0x80000000 | (seq_ctr - sbase) (the caller converts back), key0 / key1 and the event
// words (compare value in the upper half) as described above.
// checked = 0: the caller guarantees trading is enabled and every new order of the step has volume > 0
// (trk / trv / trs: the compact trade records - k word, volume, passive order's pool slot; book_device.hpp flush_trades_compact)
__device__ __forceinline__ uint32_t events_key_r2(uint32_t checked, uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                  uint32_t& vol0, uint32_t& vol1, uint32_t& key0, uint32_t& key1, uint32_t ev0,
                                                  uint32_t ev1, uint32_t& trk, uint32_t& trv, uint32_t& trs) {
  uint32_t st, vm;
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  checked = u32(checked);
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;  // biased: see above
  sq = u32(sq);
  const uint32_t kend0 = n_ev < 64u ? n_ev : 64u;
  EK_R2_STMT
  tr_n = trn + 64u;
  return st;
}

__device__ __forceinline__ uint32_t events_key_r1(uint32_t checked, uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                  uint32_t& vol0, uint32_t& key0, uint32_t ev0, uint32_t& trk, uint32_t& trv,
                                                  uint32_t& trs) {
  uint32_t st, vm;
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  checked = u32(checked);
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;
  sq = u32(sq);
  EK_R1_STMT
  tr_n = trn + 64u;
  return st;
}

// The same two loops for lists that may carry MARKET orders (compare value -1 for a market bid, 0x10000 for a market ask:
// book_device.hpp keys_begin<R, MARKETS>): they match like any order and never rest.  k_step_events only - the agent
// pipelines' lists of these pool sizes (RandomAgents) carry none and keep the loops above.
#undef EK_ALO
#undef EK_BHI
#define EK_ALO "%[alo]"
#define EK_BHI "%[bhi]"
__device__ __forceinline__ uint32_t events_key_r2m(uint32_t checked, uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                   uint32_t& vol0, uint32_t& vol1, uint32_t& key0, uint32_t& key1, uint32_t ev0,
                                                   uint32_t ev1, uint32_t& trk, uint32_t& trv, uint32_t& trs, uint32_t& alo,
                                                   uint32_t& bhi) {
  uint32_t st, vm;
  alo = (uint32_t)__builtin_amdgcn_readfirstlane(alo);
  bhi = (uint32_t)__builtin_amdgcn_readfirstlane(bhi);
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  checked = u32(checked);
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;
  sq = u32(sq);
  const uint32_t kend0 = n_ev < 64u ? n_ev : 64u;
  EK_R2M_STMT
  tr_n = trn + 64u;
  return st;
}

__device__ __forceinline__ uint32_t events_key_r1m(uint32_t checked, uint32_t& k, uint32_t n_ev, uint32_t tmask, uint32_t& tr_n, uint32_t& sq,
                                                   uint32_t& vol0, uint32_t& key0, uint32_t ev0, uint32_t& trk, uint32_t& trv,
                                                   uint32_t& trs, uint32_t& alo, uint32_t& bhi) {
  uint32_t st, vm;
  alo = (uint32_t)__builtin_amdgcn_readfirstlane(alo);
  bhi = (uint32_t)__builtin_amdgcn_readfirstlane(bhi);
  auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  checked = u32(checked);
  k = u32(k);
  n_ev = u32(n_ev);
  tmask = u32(tmask);
  uint32_t trn = u32(tr_n) - 64u;
  sq = u32(sq);
  EK_R1M_STMT
  tr_n = trn + 64u;
  return st;
}

}  // namespace bkd
