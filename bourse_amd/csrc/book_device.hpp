// book_device.hpp — gfx950 device code of the many-book LOB step simulator.
//
// Execution model: ONE WAVEFRONT (64 lanes) OWNS ONE BOOK.  The whole book lives in the
// wave's registers for the duration of a launch:
//   * live-order pool: 64*R slots, slot (r, lane) = {price, vol, id, seq} in VGPRs,
//     liveness / side / pending as three 64-bit wave-uniform masks per r (SGPRs);
//   * RNG state, clock, counters: wave-uniform scalars (SGPRs, SALU arithmetic);
//   * the step's event list and the step's trade records: one entry per lane (VGPRs).
// Strictly sequential semantics (RNG stream, shuffled event order) run as scalar control
// flow; wave parallelism is used INSIDE an event: DPP min/max reductions over the pool to
// find the touch, ballot + s_ff1 to pick the oldest order at the touch, and lane-parallel
// LDS binning for the level-2 ladder.  No MFMA: this is branchy integer work.
//
// Semantics restated from the reference (paths relative to the reference repo):
//   Env::step                     crates/step_sim/src/env.rs:116-135
//   OrderBook::place/cancel/modify crates/order_book/src/orderbook.rs:429-792,843-870
//   OrderBook::level_2_data       orderbook.rs:229-264,314-324
//   RandomAgents::update          crates/step_sim/src/agents/random_agent.rs:85-119
//   RNG (rand 0.8.5 / rand_xoshiro 0.6.0): SURVEY App. B
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>

#include "event_asm.hpp"
#include "event_asm_gen.hpp"

// 1: the event loop of a 128-slot pool runs the hand-written gfx950 code of event_asm.hpp; 0: the compiled C++ below
// (same semantics; the parity suite passes on both - build with -DBOURSE_AMD_ASM_EVENTS=0 to compare)
#ifndef BOURSE_AMD_ASM_EVENTS
#define BOURSE_AMD_ASM_EVENTS 1
#endif
// 1: pools of 256 / 512 slots (R = 4, 8) run the generated assembly loop of event_asm_gen.hpp (round 4); 0: the compiled
// C++ keyed loop (match_side_keyed / slot_event_keyed), which stays the path of the members' lists and of markets
#ifndef BOURSE_AMD_ASM_R48
#define BOURSE_AMD_ASM_R48 1
#endif

namespace bkd {

// -DBOURSE_AMD_STAMPS=1 (a diagnostic build, never the shipped library): every wave of k_step_batch / k_agents_wave adds
// the shader-clock length of its phases to g_stamps[kernel * 8 + phase] and its count to [.. + 7]
// (scripts/wave_phases.py reads them through bk_debug_stamps): where a wave's time goes UNDER the real pipeline's load.
#ifndef BOURSE_AMD_STAMPS
#define BOURSE_AMD_STAMPS 0
#endif
#if BOURSE_AMD_STAMPS && !defined(BOURSE_AMD_FSM_UNIT)
// per-book accumulators, [book][kernel * 8 + phase] (plain read-modify-write by lane 0: one wave per book and kernel at a
// time; contended atomics on a few shared words made the first version of this build 14x slower than the library)
__device__ unsigned int* g_stamp_ptr;
#define BK_STAMP_WORDS 24  // per book: k_step_batch's eight, k_agents_wave's eight, the decode's inner phases' eight
#define BK_STAMP_FIELD    \
  unsigned long long stamp_t; \
  unsigned int stamp_book, stamp_t0;
#define BK_STAMP_START(obj, book)                  \
  (obj).stamp_t = __builtin_amdgcn_s_memtime(); \
  (obj).stamp_t0 = (unsigned int)(obj).stamp_t;   \
  (obj).stamp_book = (book)
#define BK_STAMP(obj, kernel, phase, lane)                                                           \
  do {                                                                                               \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                      \
    if ((lane) == 0) g_stamp_ptr[(size_t)(obj).stamp_book * BK_STAMP_WORDS + (kernel) * 8 + (phase)] += (unsigned int)(n_ - (obj).stamp_t); \
    (obj).stamp_t = n_;                                                                              \
  } while (0)
// (+ the wave's absolute start / end of its LATEST run, low 32 bits of the clock: words 6 / 2 of k_step_batch's eight, 4 / 5
// of k_agents_wave's - scripts/wave_phases.py --skew)
#define BK_STAMP_COUNT(obj, kernel, lane)                                                                       \
  if ((lane) == 0) {                                                                                            \
    g_stamp_ptr[(size_t)(obj).stamp_book * BK_STAMP_WORDS + (kernel) * 8 + 7] += 1u;                                        \
    g_stamp_ptr[(size_t)(obj).stamp_book * BK_STAMP_WORDS + (kernel) * 8 + ((kernel) ? 4 : 6)] = (obj).stamp_t0;            \
    g_stamp_ptr[(size_t)(obj).stamp_book * BK_STAMP_WORDS + (kernel) * 8 + ((kernel) ? 5 : 2)] = (unsigned int)(obj).stamp_t; \
  }
// (diagnostic tallies in the words a run does not stamp: C5 as written leaves 8..23 - the k_agents_wave rows - free)
#define BK_STAMP_TALLY(obj, word, lane) \
  if ((lane) == 0) g_stamp_ptr[(size_t)(obj).stamp_book * BK_STAMP_WORDS + (word)] += 1u
#else
#define BK_STAMP_TALLY(obj, word, lane)
#define BK_STAMP_FIELD
#define BK_STAMP_START(obj, book)
#define BK_STAMP(obj, kernel, phase, lane)
#define BK_STAMP_COUNT(obj, kernel, lane)
#endif

constexpr int HDR_DW = 64;  // per-book header: 64 dwords, lane i holds dword i
enum Hdr : int {
  H_T_LO = 0, H_T_HI, H_S0_LO, H_S0_HI, H_S1_LO, H_S1_HI,
  H_NEXT_ID, H_SEQ, H_TRADES_LO, H_TRADES_HI, H_FLAGS, H_TRADING,
  H_STEPS_LO, H_STEPS_HI, H_EVENTS_LO, H_EVENTS_HI, H_TRADE_VOL,
  H_TRADE_BASE_LO, H_TRADE_BASE_HI, H_LAST_NTRADES, H_LAST_NEVENTS,
  H_EV_KEYED,    // host-driven steps of this book that ran on the keyed loop (step_events_keyed; a diagnostic, bk_event_steps_keyed)
  H_LIVE0 = 32,  // live masks of the pool: dwords 32 + 2r (lo), 33 + 2r (hi), r < 8 (read by k_agents_fsm)
};
constexpr int HDR_SCALARS = 20;  // header dwords 0 .. 19: every scalar field a step reads (Hdr, up to H_TRADE_BASE_HI)
constexpr int POOL_FIELDS = 5;  // price, vol, id, seq, meta(bit0 live, bit1 bid, bit2 pending New, bits 8..15 owner tag)
constexpr int MAX_GROUPS = 8;
constexpr int MAX_ASSETS = 8;  // books per market (MarketEnv<ASSETS>)

constexpr uint32_t FLAG_POOL_OVERFLOW = 1u, FLAG_TRADE_OVERFLOW = 2u, FLAG_STEP_SIZE = 4u,
                   FLAG_ORDER_LOG_FULL = 8u, FLAG_UNKNOWN_ORDER = 16u, FLAG_HIST_OVERFLOW = 32u, FLAG_PRICE_TICK = 64u,
                   FLAG_EVENT_OVERFLOW = 128u;

struct Group {  // RandomAgents::new, host-preprocessed
  uint32_t n;          // agents in the group
  uint32_t thr;        // activity: (u32 >> 8) < thr  <=>  f32 draw < activity_rate (exact, see host)
  uint32_t tick_lo, tick_rng, tick_zone;
  uint32_t vol_lo, vol_rng, vol_zone;
  uint32_t tick_size;  // the agents' tick size
  uint32_t asset;      // RandomMarketAgents: the asset (book of the market) the group trades
  uint32_t pad[2];
};

// n / d for a wave-uniform divisor without the 25-instruction u32 division sequence (Granlund-Montgomery round-up
// method, exact for every u32 n): t = umulhi(m, n); q = (t + ((n - t) >> sh1)) >> sh2.  Built on the host
// (host_math.hpp make_udiv, checked against `/` by tests/cpp/host_math_test.cpp).
struct UDiv {
  uint32_t m, sh1, sh2, d;
};
__device__ __forceinline__ uint32_t udiv(uint32_t n, const UDiv& dv) {
  const uint32_t t = __umulhi(dv.m, n);
  return (t + ((n - t) >> dv.sh1)) >> dv.sh2;
}

struct DevTrade {  // 32 B device trade record
  uint32_t t_lo, t_hi, price, vol, active, passive, side_is_bid, pad;
};

struct DevOrderLog {  // 48 B mutable part of an order (host keeps the immutable part)
  uint32_t status, vol, price;
  uint32_t key_price;  // OrderEntry.key (orderbook.rs:34-39): the price and time the priority key was last set with
  uint32_t arr_lo, arr_hi, end_lo, end_hi;
  uint32_t key_lo, key_hi, pad[2];
};

struct DevArgs {
  uint32_t n_books, levels, tick_size, n_groups;
  uint32_t step_lo, step_hi;    // step_size
  uint32_t state_stride;        // dwords per book
  uint32_t l2_width;            // 5 + 4*levels
  uint32_t trade_cap, hist_cap;
  uint32_t hist_slot0, step_prio;  // history is a ring of hist_cap steps: slot of the launch's first step; k_step_batch's wave priority (0 / 1)
  uint32_t n_agents_total, log_cap;
  uint32_t* state;
  uint32_t* l2_last;
  uint32_t* hist;
  DevTrade* trades;
  DevOrderLog* order_log;
  // host-driven event batch (k_step_events only); CSR per book
  const uint32_t* ev_off;    // [n_books + 1]
  const uint4* ev;           // {word = kind | bid<<8 | has_price<<9 | has_vol<<10 | asset<<16, id, price, vol}
  // device-resident ingress (k_ingest): the queues are fixed-capacity rows filled ON the device - market m's events are
  // ev[m * ev_stride ..] and their number is ev_len[m] (null: the CSR form above, uploaded by the host)
  const uint32_t* ev_len;
  uint32_t ev_stride;
  // split pipeline: per-book step batch written by k_agents_fsm, consumed by k_step_batch
  uint32_t* batch;
  uint32_t batch_stride;  // dwords per book: 64 + 160 * R
  uint32_t book_begin, book_end;  // split pipeline: this launch covers books (markets if assets > 1) [begin, end)
  // MarketEnv mode (market_env.rs): `assets` consecutive books form a market sharing one clock, one RNG stream and one
  // shuffled event queue; 1 = independent books.  asset_tick = the books' tick sizes (market.rs:74-81).
  uint32_t assets;
  uint32_t asset_tick[MAX_ASSETS];
  UDiv tick_div, asset_div[MAX_ASSETS];  // division by tick_size / asset_tick[i] (the level-2 snapshot's level index)
  Group groups[MAX_GROUPS];
};
// step batch layout (dwords): [0] n_ev; [64, 64+32R) shuffled event list (u16 event words: agent slot | EV_NEW | EV_BID);
// then uint2 {price, vol} per agent slot.  (Dwords 8..39 held placing / bid-side masks until k_step_batch rebuilt them
// from the event words itself.)
constexpr int BT_NEV = 0, BT_EV = 64;
// event words of k_agents_fsm: slot in bits 0..8, EV_BID / EV_NEW classify the event
constexpr uint32_t EV_NEW = 0x8000u, EV_BID = 0x4000u, EV_SLOT = 0x1FFu;
// k_step_events only (step_events.hpp): a MODIFICATION's word - never handed to the loops as an event: the list is cut there -
// carries its flags where the event record has them (has price 1 << 9, has volume 1 << 10) and its new price's field above
constexpr uint32_t EV_MOD = 0x800u, EV_MOD_P = 0x200u, EV_MOD_V = 0x400u;

// ----------------------------------------------------------------------------------
// wave primitives
// ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Explicit scalar loads of wave-uniform words (header fields, the lane-state record's tag) that the SAME kernel stores to
// later.  Rounds 3-4 read them through an address_space(4) ("constant") pointer so that the compiler would issue s_load -
// which also tells it the memory never changes, although store_book / finish() write it: correct only as long as every use
// precedes the store, and silently wrong in a persistent or several-books-per-wave kernel.  The statement is
// volatile with a memory clobber: it is not moved across the kernel's loads and stores, not rematerialised, and the
// scalar cache's coherence rests on what it always rested on (invalidated at kernel start; nothing in the kernel stores
// these words before it runs).  It waits for its own loads (the compiler does not count asm-issued ones): callers issue
// their vector loads FIRST, so the wait hides under those loads' longer round trip.
typedef uint32_t bk_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bk_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t bk_u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sload_x16_x4(const uint32_t* p, bk_u32x16& a, bk_u32x4& b) {  // words 0..15, 16..19
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
__device__ __forceinline__ bk_u32x8 sload_x8(const uint32_t* p) {
  bk_u32x8 a;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(p) : "memory");
  return a;
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
// v_writelane_b32: clang 22 has no builtin for it; bind the LLVM intrinsic by its mangled name
// (the compiler then legalises the operands, e.g. lane select through M0 on gfx9).
extern "C" __device__ uint32_t bk_llvm_writelane(uint32_t val, uint32_t lane, uint32_t old) __asm(
    "llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t wrl(uint32_t val, uint32_t lane, uint32_t old) {
  return bk_llvm_writelane(val, lane, old);
}

// dst[lane] = mask[lane] ? if_set : if_clear, mask wave-uniform in an SGPR pair (one VALU op)
__device__ __forceinline__ uint32_t sel(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
  return __builtin_amdgcn_inverse_ballot_w64(mask) ? if_set : if_clear;  // v_cndmask_b32 with the SGPR mask
}
__device__ __forceinline__ bool lane_bit(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// Full-wave (all 64 lanes active) reductions on the DPP network: butterfly inside each row
// of 16, then row_bcast:15 / row_bcast:31 carry the row results into lane 63.
#define BK_DPP_REDUCE(OP)                                                                      \
  asm("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"        \
      "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"        \
      "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"            \
      "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                 \
      "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"               \
      "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"               \
      "s_nop 1"                                                                                \
      : "+v"(x))
__device__ __forceinline__ uint32_t wave_umin(uint32_t x) {
  BK_DPP_REDUCE("v_min_u32_dpp");
  return rdl(x, 63);
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t x) {
  BK_DPP_REDUCE("v_max_u32_dpp");
  return rdl(x, 63);
}
__device__ __forceinline__ int32_t wave_imin(int32_t x) {
  BK_DPP_REDUCE("v_min_i32_dpp");
  return (int32_t)rdl((uint32_t)x, 63);
}
__device__ __forceinline__ int32_t wave_imax(int32_t x) {
  BK_DPP_REDUCE("v_max_i32_dpp");
  return (int32_t)rdl((uint32_t)x, 63);
}
__device__ __forceinline__ uint32_t wave_add(uint32_t x) {  // wrapping u32 sum
  BK_DPP_REDUCE("v_add_u32_dpp");
  return rdl(x, 63);
}

// The four reductions of the level-2 snapshot (max, min, add, add) interleaved: every DPP op is separated from the
// previous op on the same register by three others, so the 2 wait states a VGPR-write -> DPP-read needs are filled with
// useful work instead of s_nop (28 fewer instructions per step, and the four chains overlap).
#define BK_DPP4(CTRL)                                                     \
  "v_max_u32_dpp %0, %0, %0 " CTRL "\n\tv_min_u32_dpp %1, %1, %1 " CTRL "\n\t" \
  "v_add_u32_dpp %2, %2, %2 " CTRL "\n\tv_add_u32_dpp %3, %3, %3 " CTRL "\n\t"
__device__ __forceinline__ void wave_reduce4(uint32_t& mx, uint32_t& mn, uint32_t& s1, uint32_t& s2) {
  asm("s_nop 1\n\t" BK_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") BK_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
          BK_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf") BK_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
              BK_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf") BK_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
      : "+v"(mx), "+v"(mn), "+v"(s1), "+v"(s2));
  mx = rdl(mx, 63);
  mn = rdl(mn, 63);
  s1 = rdl(s1, 63);
  s2 = rdl(s2, 63);
}

// max / min / max interleaved the same way (key_window below)
#define BK_DPP3(CTRL) \
  "v_max_u32_dpp %0, %0, %0 " CTRL "\n\tv_min_u32_dpp %1, %1, %1 " CTRL "\n\tv_max_u32_dpp %2, %2, %2 " CTRL "\n\t"
__device__ __forceinline__ void wave_reduce3(uint32_t& mx, uint32_t& mn, uint32_t& m2) {
  asm("s_nop 1\n\t" BK_DPP3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") BK_DPP3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
          BK_DPP3("row_half_mirror row_mask:0xf bank_mask:0xf") BK_DPP3("row_mirror row_mask:0xf bank_mask:0xf")
              BK_DPP3("row_bcast:15 row_mask:0xa bank_mask:0xf") BK_DPP3("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
      : "+v"(mx), "+v"(mn), "+v"(m2));
  mx = rdl(mx, 63);
  mn = rdl(mn, 63);
  m2 = rdl(m2, 63);
}

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// ----------------------------------------------------------------------------------
// RNG: xoroshiro128** in wave-uniform scalars (the compiler keeps this on the SALU)
// ----------------------------------------------------------------------------------
struct Rng {
  uint64_t s0, s1;
  __device__ __forceinline__ uint32_t next_u32() {  // low 32 bits of next_u64
    uint64_t r = s0 * 5ull;
    r = (r << 7) | (r >> 57);
    r *= 9ull;
    uint64_t t1 = s1 ^ s0;
    s0 = ((s0 << 24) | (s0 >> 40)) ^ t1 ^ (t1 << 16);
    s1 = (t1 << 37) | (t1 >> 27);
    return (uint32_t)r;
  }
  // UniformInt<u32>::sample_single with precomputed zone (App. B.3): returns value in [0, range)
  __device__ __forceinline__ uint32_t below(uint32_t range, uint32_t zone) {
    for (;;) {
      uint32_t v = next_u32();
      uint64_t m = (uint64_t)v * range;
      if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
  }
  __device__ __forceinline__ uint32_t below(uint32_t range) {
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    return below(range, zone);
  }
};

// The same generator for the lane-per-book kernel, on 32-bit halves (s0 = a1:a0, s1 = b1:b0): full-rate VALU only
// (v_alignbit / v_lshl_add / v_xor3) instead of 64-bit shifts and quarter-rate 32x32 multiplies.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // gfx950 v_bitop3_b32: any 3-input boolean; 0x96 = a ^ b ^ c
}
struct RngLane {
  uint32_t a0, a1, b0, b1;
  // s0 * 5 as ONE v_mad_u64_u32 (low word + carry) + two adds for the high word: the compiler's shift-and-add form of
  // the constant multiply needs five instructions for the 64-bit product (the 5 is hidden in an SGPR to keep it from it)
  static __device__ __forceinline__ uint32_t opaque5() {
    uint32_t five = 5u;
    asm("" : "+s"(five));
    return five;
  }
  static __device__ __forceinline__ uint32_t mul5_rotl7_lo(uint32_t a0, uint32_t a1, uint32_t five) {
    const uint64_t p = (uint64_t)a0 * five;
    const uint32_t hi5 = (a1 << 2) + (uint32_t)(p >> 32) + a1;             // (s0 * 5) bits 32..63
    return __builtin_amdgcn_alignbit((uint32_t)p, hi5, 25);                // low word of rotl(s0 * 5, 7)
  }
  __device__ __forceinline__ uint32_t next_u32() {
    const uint32_t r = mul5_rotl7_lo(a0, a1, opaque5());
    const uint32_t t0 = b0 ^ a0, t1 = b1 ^ a1;
    const uint32_t n0 = xor3(__builtin_amdgcn_alignbit(a0, a1, 8), t0, t0 << 16);  // rotl(s0, 24) ^ t ^ (t << 16)
    const uint32_t n1 = xor3(__builtin_amdgcn_alignbit(a1, a0, 8), t1, __builtin_amdgcn_alignbit(t1, t0, 16));
    a0 = n0;
    a1 = n1;
    b0 = __builtin_amdgcn_alignbit(t1, t0, 27);  // rotl(t, 37)
    b1 = __builtin_amdgcn_alignbit(t0, t1, 27);
    return (r << 3) + r;
  }
  // the same in two halves: the output word of the current state, and the state update (which does not depend on it)
  // (five = opaque5(), taken once outside the caller's loop)
  __device__ __forceinline__ uint32_t output(uint32_t five) const {
    const uint32_t r = mul5_rotl7_lo(a0, a1, five);
    return (r << 3) + r;
  }
  __device__ __forceinline__ void advance() {
    uint32_t t0 = b0 ^ a0, t1 = b1 ^ a1;
    uint32_t r0 = __builtin_amdgcn_alignbit(a0, a1, 8), r1 = __builtin_amdgcn_alignbit(a1, a0, 8);  // rotl(s0, 24)
    // every read of the old s0 before the new one is formed: the new state then takes the old one's registers (the
    // scheduler had left one rotate behind the new low word, which cost a register copy per draw in k_agents_fsm)
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(r0), "+v"(r1));
    const uint32_t n0 = xor3(r0, t0, t0 << 16);
    const uint32_t n1 = xor3(r1, t1, __builtin_amdgcn_alignbit(t1, t0, 16));
    a0 = n0;
    a1 = n1;
    b0 = __builtin_amdgcn_alignbit(t1, t0, 27);
    b1 = __builtin_amdgcn_alignbit(t0, t1, 27);
  }
};

// ----------------------------------------------------------------------------------
// Book state in registers
// ----------------------------------------------------------------------------------
template <int R>
struct Book {
  // pool (VGPR): slot (r, lane)
  uint32_t price[R], vol[R], id[R], seq[R];
  // wave-uniform masks
  uint64_t live[R], bid[R], pend[R];
  // scalars
  uint64_t t;
  uint32_t next_id, seq_ctr, trade_vol, flags, trading;
  uint64_t n_trades, n_events, trade_base;
  // this step's trade records, one per lane (flushed when 64 are buffered)
  uint32_t tr_k, tr_price, tr_vol, tr_act, tr_pas;
  uint32_t tr_n;  // records buffered
  uint32_t hdr0;  // this lane's header dword as loaded (store_book<KEEP_HDR> rewrites it without reading it again)
  BK_STAMP_FIELD
};

template <int R>
__device__ __forceinline__ uint32_t slot_read(const uint32_t (&v)[R], uint32_t n) {
  uint32_t out = rdl(v[0], n & 63);
#pragma unroll
  for (int r = 1; r < R; ++r) {
    uint32_t x = rdl(v[r], n & 63);
    out = ((n >> 6) == (uint32_t)r) ? x : out;
  }
  return out;
}
template <int R>
__device__ __forceinline__ void slot_write(uint32_t (&v)[R], uint32_t n, uint32_t val) {
#pragma unroll
  for (int r = 0; r < R; ++r)
    if ((n >> 6) == (uint32_t)r) v[r] = wrl(val, n & 63, v[r]);
}
// Both written so that every element is touched unconditionally with a per-element select: an `if (index == r) m[r] = ..`
// chain is turned back into ONE dynamically indexed access by LLVM at R >= 4, which defeats scalar replacement and
// moves the whole Book into scratch memory.
template <int R>
__device__ __forceinline__ bool mask_test(const uint64_t (&m)[R], uint32_t n) {
  uint64_t w = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) w |= m[r] & (((n >> 6) == (uint32_t)r) ? ~0ull : 0ull);
  return (w >> (n & 63)) & 1ull;
}
template <int R>
__device__ __forceinline__ void mask_set(uint64_t (&m)[R], uint32_t n, bool on) {
  const uint64_t bit = 1ull << (n & 63);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t b = ((n >> 6) == (uint32_t)r) ? bit : 0ull;
    m[r] = on ? (m[r] | b) : (m[r] & ~b);
  }
}

// ----------------------------------------------------------------------------------
// trade records: buffered one per lane, written out as coalesced 32-byte records
// ----------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void flush_trades(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane) {
  if (B.tr_n == 0) return;
  const uint64_t first = B.n_trades;  // global index of lane 0's record (n_trades counts flushed records)
  B.n_trades += B.tr_n;
  const uint64_t pos0 = first - B.trade_base;
  if ((uint32_t)lane < B.tr_n) {
    const uint64_t pos = pos0 + (uint32_t)lane;
    if (pos < a.trade_cap) {
      const uint64_t t = t0 + (B.tr_k & 0x7FFFFFFFu);
      DevTrade rec;
      rec.t_lo = (uint32_t)t;
      rec.t_hi = (uint32_t)(t >> 32);
      rec.price = B.tr_price;
      rec.vol = B.tr_vol;
      rec.active = B.tr_act;
      rec.passive = B.tr_pas;
      rec.side_is_bid = B.tr_k >> 31;
      rec.pad = 0;
      uint4* dst = reinterpret_cast<uint4*>(a.trades + (size_t)book * a.trade_cap + pos);
      const uint4* src = reinterpret_cast<const uint4*>(&rec);
      dst[0] = src[0];
      dst[1] = src[1];
    }
  }
  if (pos0 + B.tr_n > a.trade_cap) B.flags |= FLAG_TRADE_OVERFLOW;
  B.tr_n = 0;
}

template <int R>
__device__ __forceinline__ void emit_trade(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                           uint32_t k, bool passive_is_bid, uint32_t price, uint32_t vol,
                                           uint32_t active, uint32_t passive) {
  if (B.tr_n == 64) flush_trades(B, a, book, t0, lane);
  const uint32_t l = B.tr_n;
  B.tr_k = wrl(k | (passive_is_bid ? 0x80000000u : 0u), l, B.tr_k);
  B.tr_price = wrl(price, l, B.tr_price);
  B.tr_vol = wrl(vol, l, B.tr_vol);
  B.tr_act = wrl(active, l, B.tr_act);
  B.tr_pas = wrl(passive, l, B.tr_pas);
  B.tr_n = l + 1;
}

// Compact records of the keyed assembly loop (event_asm.hpp EK_PICK): lane i < tr_n holds {tr_k = event position |
// passive side << 31, tr_vol, tr_pas = the passive order's pool slot}.  The trade's price is the passive order's
// (match_orders, orderbook.rs:843-870), its ids are the aggressor's - the order of the event at that position - and the
// passive order's: all three sit in pool registers that do not change inside a step, so they are gathered here, once per
// flush, instead of being read and written lane by lane for every trade.  ev: the step's (shuffled) event list.
template <int R>
__device__ __forceinline__ uint32_t pool_gather(const uint32_t (&v)[R], uint32_t idx) {
  uint32_t out = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)v[0]);
#pragma unroll
  for (int r = 1; r < R; ++r) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)v[r]);
    out = (idx >> 6) == (uint32_t)r ? x : out;
  }
  return out;
}
template <int R>
__device__ __forceinline__ void flush_trades_compact(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                                     const uint32_t (&ev)[R]) {
  if (B.tr_n == 0) return;
  const uint32_t k = B.tr_k & 0x7FFFFFFFu, ps = B.tr_pas & (64u * R - 1u);
  const uint32_t as = pool_gather<R>(ev, k < 64u * R ? k : 0u) & EV_SLOT & (64u * R - 1u);  // the aggressor's slot
  B.tr_price = pool_gather<R>(B.price, ps);
  B.tr_pas = pool_gather<R>(B.id, ps);
  B.tr_act = pool_gather<R>(B.id, as);
  flush_trades<R>(B, a, book, t0, lane);
}

// ----------------------------------------------------------------------------------
// order log (host-driven path only): single-lane scattered 32-byte updates
// ----------------------------------------------------------------------------------
struct LogCtx {
  DevOrderLog* base;  // this book's log or nullptr
  uint32_t cap;
};
__device__ __forceinline__ void log_write(const LogCtx& lg, uint32_t& flags, int lane, uint32_t id, uint32_t status,
                                          uint32_t vol, uint32_t price, uint64_t arr, uint64_t end,
                                          bool set_arr, bool set_key = false, uint64_t key_t = 0) {
  if (!lg.base) return;
  if (id >= lg.cap) {
    flags |= FLAG_ORDER_LOG_FULL;
    return;
  }
  if (lane == 0) {
    DevOrderLog* e = lg.base + id;
    e->status = status;
    e->vol = vol;
    e->price = price;
    if (set_arr) {
      e->arr_lo = (uint32_t)arr;
      e->arr_hi = (uint32_t)(arr >> 32);
    }
    e->end_lo = (uint32_t)end;
    e->end_hi = (uint32_t)(end >> 32);
    if (set_key) {  // key = (side, price_key(price), key_t): create (t = 0), rest on placement / replace (t = now)
      e->key_price = price;
      e->key_lo = (uint32_t)key_t;
      e->key_hi = (uint32_t)(key_t >> 32);
    }
  }
}
// passive order touched by a fill: vol / status / end_time only
__device__ __forceinline__ void log_fill(const LogCtx& lg, uint32_t& flags, int lane, uint32_t id, uint32_t vol,
                                         uint64_t t) {
  if (!lg.base) return;
  if (id >= lg.cap) {
    flags |= FLAG_ORDER_LOG_FULL;
    return;
  }
  if (lane == 0) {
    DevOrderLog* e = lg.base + id;
    e->vol = vol;
    if (vol == 0) {
      e->status = 2;  // Filled
      e->end_lo = (uint32_t)t;
      e->end_hi = (uint32_t)(t >> 32);
    }
  }
}

// ----------------------------------------------------------------------------------
// matching: match_bid / match_ask + match_orders (orderbook.rs:429-487, 843-870)
// Returns true iff the aggressor ended Filled (its volume hit zero in a match).
// ----------------------------------------------------------------------------------
// The aggressor's side is a template parameter: each instantiation has a fixed opposite side, so the loop carries no
// per-iteration side tests (they were ~15 % of the scalar instructions of the SALU-bound event kernel).
template <int R, bool agg_bid>
__device__ __forceinline__ bool match_side(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                           uint32_t k, uint32_t p, uint32_t& v, uint32_t agg_id, const LogCtx& lg) {
  const uint32_t v0 = v;
  while (v > 0) {
    // candidates: live orders on the opposite side
    uint64_t cand[R];
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cand[r] = B.live[r] & (agg_bid ? ~B.bid[r] : B.bid[r]);
      any |= cand[r];
    }
    if (!any) break;  // best_order_idx() == None (orderbook.rs:449-451)
    // touch of the opposite side: min ask / max bid
    uint32_t m = agg_bid ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t pr = sel(cand[r], B.price[r], agg_bid ? 0xFFFFFFFFu : 0u);
      m = agg_bid ? min(m, pr) : max(m, pr);
    }
    const uint32_t best = agg_bid ? wave_umin(m) : wave_umax(m);
    if (agg_bid ? (p < best) : (p > best)) break;  // inclusive crossing test (:430 / :463)
    // oldest order at the touch: min seq among equal-price candidates
    uint64_t eq[R];
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      eq[r] = cand[r] & __ballot(B.price[r] == best);
      cnt += __builtin_popcountll(eq[r]);
    }
    // match_orders (orderbook.rs:843-870) on the chosen passive order.  All pool accesses use a
    // compile-time register index r: the single-candidate case (the common one) is handled inside the
    // unrolled loop, the multi-candidate case first scans the few candidates' seq stamps.
    if (cnt != 1) {
      // several orders rest at the touch: the oldest (min seq stamp, unique per book) is next in the queue.
      // A second DPP reduction instead of a scalar scan of the candidates: the VALU has slack, the SALU does not.
      uint32_t sm = 0xFFFFFFFFu;
#pragma unroll
      for (int r = 0; r < R; ++r) sm = min(sm, sel(eq[r], B.seq[r], 0xFFFFFFFFu));
      const uint32_t bs = wave_umin(sm);
#pragma unroll
      for (int r = 0; r < R; ++r) eq[r] &= __ballot(B.seq[r] == bs);
    }
    uint32_t pv = 0, pid = 0, tv = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (eq[r]) {  // exactly one r has a (single-bit) mask now
        const uint32_t l = __builtin_ctzll(eq[r]);
        pv = rdl(B.vol[r], l);
        pid = rdl(B.id[r], l);
        tv = v < pv ? v : pv;
        pv -= tv;
        B.vol[r] = wrl(pv, l, B.vol[r]);
        if (pv == 0) B.live[r] &= ~eq[r];  // passive Filled -> remove_order
      }
    }
    v -= tv;
    emit_trade(B, a, book, t0, lane, k, !agg_bid, best, tv, agg_id, pid);
    B.trade_vol += tv;
    log_fill(lg, B.flags, lane, pid, pv, t0 + k);
  }
  return v0 != 0 && v == 0;  // Filled = its volume hit zero in a match (a zero-volume order never enters the loop, :430)
}
template <int R>
__device__ __forceinline__ bool match(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane, uint32_t k,
                                      bool agg_bid, uint32_t p, uint32_t& v, uint32_t agg_id, const LogCtx& lg) {
  return agg_bid ? match_side<R, true>(B, a, book, t0, lane, k, p, v, agg_id, lg)
                 : match_side<R, false>(B, a, book, t0, lane, k, p, v, agg_id, lg);
}

// One event of the fused/split RandomAgents paths (slot == agent): a New if the slot's pend bit is set
// (place_order, orderbook.rs:583-611), else the Cancellation of the slot's order (orderbook.rs:622-644; a
// no-op if the order was filled meanwhile).  The whole handler is instantiated per pool register RS and selected by
// ONE uniform branch on the slot index, so every pool access inside uses a compile-time register and no flag has to
// survive between stages (scalar instructions are the scarce resource of this kernel).
// CLS: the event word itself says New / Cancellation and the side (`ew`, k_agents_fsm's lists); otherwise the slot's
// pend and side masks do.
template <int R, int RS, bool CLS = false>
__device__ __forceinline__ void slot_event_at(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                              uint32_t k, uint32_t sl, uint32_t ew = 0) {
  const LogCtx nolog{nullptr, 0};
  const uint64_t bit = 1ull << sl;
  if (CLS ? !(ew & EV_NEW) : !(B.pend[RS] & bit)) {
    B.live[RS] &= ~bit;  // Cancellation
    return;
  }
  if (!CLS) B.pend[RS] &= ~bit;
  const uint32_t p = rdl(B.price[RS], sl);
  uint32_t v = rdl(B.vol[RS], sl);
  const uint32_t id = rdl(B.id[RS], sl);
  bool filled = false, market;
  if (CLS ? (ew & EV_BID) != 0 : (B.bid[RS] & bit) != 0) {
    market = p == 0xFFFFFFFFu;
    if (B.trading) filled = match_side<R, true>(B, a, book, t0, lane, k, p, v, id, nolog);
  } else {
    market = p == 0u;
    if (B.trading) filled = match_side<R, false>(B, a, book, t0, lane, k, p, v, id, nolog);
  }
  if (!market && !filled) {  // rest the remainder with a fresh priority stamp
    B.vol[RS] = wrl(v, sl, B.vol[RS]);
    B.seq[RS] = wrl(B.seq_ctr, sl, B.seq[RS]);
    B.live[RS] |= bit;
    B.seq_ctr += 1;
  }
}

template <int R, int RS = 0, bool CLS = false>
__device__ __forceinline__ void process_slot_event(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                                   uint32_t k, uint32_t n, uint32_t ew = 0) {
  if constexpr (RS + 1 < R) {
    if ((n >> 6) == (uint32_t)RS)
      slot_event_at<R, RS, CLS>(B, a, book, t0, lane, k, n & 63, ew);
    else
      process_slot_event<R, RS + 1, CLS>(B, a, book, t0, lane, k, n, ew);
  } else {
    slot_event_at<R, RS, CLS>(B, a, book, t0, lane, k, n & 63, ew);
  }
}

// ----------------------------------------------------------------------------------
// level-2 snapshot (orderbook.rs:229-264,314-324) -> [trade_vol, bid, ask, ask_vol, bid_vol, levels...]
// ----------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void snapshot(const Book<R>& B, const DevArgs& a, uint32_t book, int lane,
                                         uint32_t* __restrict__ bins /* LDS, >= 4*levels */, uint32_t hist_slot,
                                         uint32_t& flags, bool write_last, const UDiv& tick) {
  const uint32_t L = a.levels;
  uint32_t mb = 0u, ma = 0xFFFFFFFFu, sb = 0u, sa = 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t lb = B.live[r] & B.bid[r], la = B.live[r] & ~B.bid[r];
    mb = max(mb, sel(lb, B.price[r], 0u));
    ma = min(ma, sel(la, B.price[r], 0xFFFFFFFFu));
    sb += sel(lb, B.vol[r], 0u);
    sa += sel(la, B.vol[r], 0u);
  }
  wave_reduce4(mb, ma, sb, sa);
  const uint32_t bid_best = mb;  // empty side -> 0        (side.rs:194-196)
  const uint32_t ask_best = ma;  // empty side -> u32::MAX (side.rs:99-104)
  const uint32_t bid_vol = sb, ask_vol = sa;

  for (uint32_t j = lane; j < 4 * L; j += 64) bins[j] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool is_live = lane_bit(B.live[r]);
    const bool is_bid = lane_bit(B.bid[r]);
    if (is_live) {
      // level i of a side holds the orders priced touch -/+ i*tick (wrapping arithmetic never matches)
      const uint32_t d = is_bid ? (bid_best - B.price[r]) : (B.price[r] - ask_best);
      const uint32_t q = udiv(d, tick);
      if (q * tick.d == d && q < L) {
        atomicAdd(&bins[4 * q + (is_bid ? 0 : 2)], B.vol[r]);
        atomicAdd(&bins[4 * q + (is_bid ? 1 : 3)], 1u);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const uint32_t W = a.l2_width;
  uint32_t* last = a.l2_last + (size_t)book * W;
  uint32_t* hist = nullptr;
  if (a.hist_cap) hist = a.hist + ((size_t)hist_slot * a.n_books + book) * W;  // ring: the oldest record is overwritten
  (void)flags;
  uint32_t h = 0;
  h = lane == 0 ? B.trade_vol : h;
  h = lane == 1 ? bid_best : h;
  h = lane == 2 ? ask_best : h;
  h = lane == 3 ? ask_vol : h;
  h = lane == 4 ? bid_vol : h;
  if (lane < 5) {
    if (write_last) last[lane] = h;
    if (hist) hist[lane] = h;
  }
  for (uint32_t j = lane; j < 4 * L; j += 64) {
    const uint32_t x = bins[j];
    if (write_last) last[5 + j] = x;
    if (hist) hist[5 + j] = x;
  }
}

// ----------------------------------------------------------------------------------
// state load / store (coalesced: lane-contiguous dwords)
// ----------------------------------------------------------------------------------
// What one step of one book reads from memory, as loaded (nothing unpacked yet): issued together, so that the one exposed
// memory round trip of a wave covers all of it - and k_step_batch issues the NEXT book's while this one's events run.
template <int R>
struct StepRaw {
  uint32_t hdr, f[R][POOL_FIELDS];
  uint32_t sh[HDR_SCALARS];  // the header's scalar fields once more, through the scalar cache (wave-uniform address)
  uint32_t bh, ev[R];  // step batch: header words, event list
  uint2 pv[R];         // step batch: new orders {price, vol} by slot
};
template <int R>
__device__ __forceinline__ void load_state_raw(StepRaw<R>& w, const uint32_t* __restrict__ st, int lane) {
  w.hdr = st[lane];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t* p = st + HDR_DW + r * POOL_FIELDS * 64;
#pragma unroll
    for (int f = 0; f < POOL_FIELDS; ++f) w.f[r][f] = p[f * 64 + lane];
  }
}
// The header's scalar fields once more through the scalar cache (s_load_dwordx16 + x4: sixteen-odd v_readlane of the line
// above and their vector-to-scalar hand-over at the head of every wave's chain otherwise).  Call it AFTER the wave's vector
// loads have been issued (sload_x16_x4 waits for its own loads).
// Pools of <= 128 slots only: k_step_batch<8> sits at the scalar-register limit already, and twenty more live scalars
// cost the C5 stand-in 3 % (31.6 -> 30.6 M, same box) where the smaller pools gain ~1 %; the larger pools read the
// fields from the vector copy (unpack_book).
template <int R>
__device__ __forceinline__ void load_state_scalars(StepRaw<R>& w, const uint32_t* __restrict__ st) {
  if constexpr (R <= 2) {
    bk_u32x16 a;
    bk_u32x4 b;
    sload_x16_x4(st, a, b);
#pragma unroll
    for (int i = 0; i < 16; ++i) w.sh[i] = a[i];
#pragma unroll
    for (int i = 16; i < HDR_SCALARS; ++i) w.sh[i] = b[i - 16];
  }
}
template <int R>
__device__ __forceinline__ void unpack_book(Book<R>& B, Rng& rng, const StepRaw<R>& w) {
  const uint32_t hdr = w.hdr;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    B.price[r] = w.f[r][0];
    B.vol[r] = w.f[r][1];
    B.id[r] = w.f[r][2];
    B.seq[r] = w.f[r][3];
  }
  auto H = [&](int i) { return R <= 2 ? w.sh[i] : rdl(hdr, (uint32_t)i); };  // (load_state_raw: which pools take which path)
  B.t = mk64(H(H_T_LO), H(H_T_HI));
  rng.s0 = mk64(H(H_S0_LO), H(H_S0_HI));
  rng.s1 = mk64(H(H_S1_LO), H(H_S1_HI));
  B.next_id = H(H_NEXT_ID);
  B.seq_ctr = H(H_SEQ);
  B.n_trades = mk64(H(H_TRADES_LO), H(H_TRADES_HI));
  B.flags = H(H_FLAGS);
  B.trading = H(H_TRADING);
  B.n_events = mk64(H(H_EVENTS_LO), H(H_EVENTS_HI));
  B.trade_vol = H(H_TRADE_VOL);
  B.trade_base = mk64(H(H_TRADE_BASE_LO), H(H_TRADE_BASE_HI));
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t meta = w.f[r][4];
    B.live[r] = __ballot((meta & 1u) != 0);
    B.bid[r] = __ballot((meta & 2u) != 0);
    B.pend[r] = __ballot((meta & 4u) != 0);  // only the split mixed-agent pipeline stores books with pending orders
  }
  B.tr_k = B.tr_price = B.tr_vol = B.tr_act = B.tr_pas = 0;
  B.tr_n = 0;
  B.hdr0 = hdr;
}
template <int R>
__device__ __forceinline__ void load_book(Book<R>& B, Rng& rng, const uint32_t* __restrict__ st, int lane) {
  StepRaw<R> w;
  load_state_raw<R>(w, st, lane);
  load_state_scalars<R>(w, st);
  unpack_book<R>(B, rng, w);
}

// KEEP_HDR: the reserved / untouched header words come from the copy load_book took (k_step_batch: one register across the
// step instead of a second, exposed memory round trip in every wave's tail - at 8 waves per SIMD the event kernel is bound
// by the waves' chain latency).  The kernels that keep a book for a whole launch re-read the line instead.
template <int R, bool KEEP_HDR = false>
__device__ __forceinline__ void store_book(const Book<R>& B, const Rng& rng, uint32_t* __restrict__ st, int lane,
                                           uint64_t steps_done, uint32_t last_ntrades, uint32_t last_nevents,
                                           uint32_t ev_keyed = 0u) {
  uint32_t hdr = KEEP_HDR ? B.hdr0 : st[lane];  // keep reserved words
  if (!KEEP_HDR) hdr += lane == H_EV_KEYED ? ev_keyed : 0u;  // (k_step_events only; 0 elsewhere)
  // (wave-uniform values into their lanes with one v_writelane each - a compare + select per field before)
  auto put = [&](int idx, uint32_t v) { hdr = wrl(rfl(v), (uint32_t)idx, hdr); };
  put(H_T_LO, (uint32_t)B.t);
  put(H_T_HI, (uint32_t)(B.t >> 32));
  put(H_S0_LO, (uint32_t)rng.s0);
  put(H_S0_HI, (uint32_t)(rng.s0 >> 32));
  put(H_S1_LO, (uint32_t)rng.s1);
  put(H_S1_HI, (uint32_t)(rng.s1 >> 32));
  put(H_NEXT_ID, B.next_id);
  put(H_SEQ, B.seq_ctr);
  put(H_TRADES_LO, (uint32_t)B.n_trades);
  put(H_TRADES_HI, (uint32_t)(B.n_trades >> 32));
  put(H_FLAGS, B.flags);
  put(H_STEPS_LO, (uint32_t)steps_done);
  put(H_STEPS_HI, (uint32_t)(steps_done >> 32));
  put(H_EVENTS_LO, (uint32_t)B.n_events);
  put(H_EVENTS_HI, (uint32_t)(B.n_events >> 32));
  put(H_TRADE_VOL, B.trade_vol);
  put(H_LAST_NTRADES, last_ntrades);
  put(H_LAST_NEVENTS, last_nevents);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    put(H_LIVE0 + 2 * r, (uint32_t)B.live[r]);
    put(H_LIVE0 + 2 * r + 1, (uint32_t)(B.live[r] >> 32));
  }
  st[lane] = hdr;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t* p = st + HDR_DW + r * POOL_FIELDS * 64;
    p[0 * 64 + lane] = B.price[r];
    p[1 * 64 + lane] = B.vol[r];
    p[2 * 64 + lane] = B.id[r];
    p[3 * 64 + lane] = B.seq[r];
    p[4 * 64 + lane] = (lane_bit(B.live[r]) ? 1u : 0u) | (lane_bit(B.bid[r]) ? 2u : 0u) | (lane_bit(B.pend[r]) ? 4u : 0u);
  }
}

// ==================================================================================
// SIGNED KEYS (round 4).  One 32-bit sort key per resting order:
//     ask:  1 | price - pbase (15 bits) | seq - sbase (16 bits)          -> NEGATIVE as an i32
//     bid:  0 | price - pbase (15 bits) | 0xFFFF - (seq - sbase)         -> POSITIVE (price field >= 2)
//     every other pool lane (free, cancelled, filled, or holding an order that is still pending): 0
// so the best ask is the signed MINIMUM over all pool lanes and the best bid the signed MAXIMUM - with no mask, select or
// neutral element in front of the reduction: an empty side answers with a value of the wrong sign (>= 0 for the asks,
// <= 0 for the bids), which fails the crossing test by the same signed compare.  Price-time priority (orderbook.rs:429-487)
// is the order of the keys: lower price then earlier arrival among the asks, higher price then earlier arrival (the
// complemented field) among the bids.  An order leaves the book by zeroing its key; the live masks are rebuilt from
// `key != 0` after the loop (keys_end).  Round 3's keys carried the side in bit 0 and needed `live & ~bid` / `live & bid`
// as a select in front of every reduction (2 scalar + 2 vector instructions per pool register and match step).
//   A NEW order's compare value does not sit in the key array (it would be found): it travels in the upper half of its
//   event word (`pk` below, per pool slot, gathered into the list by key_event_words): pk = 0x8000 | price field for a bid,
//   the price field for an ask, and
//     bid:  kp = pk << 16 | 0xFFFF   crosses the best ask a iff a <= kp (signed: a >= 0 - no ask - never does)
//     ask:  kp = pk << 16            crosses the best bid b iff b >= kp (b <= 0 - no bid - never does)
//     rests as kp ^ sq, sq = 0x80000000 | (seq_ctr - sbase): the sign flips and the arrival field lands (complemented for a
//     bid by the 0xFFFF it XORs with).
//   Market orders (price u32::MAX for a bid, 0 for an ask: the members of an AgentSet place them): pk = 0xFFFF / 1, i.e.
//   kp = -1 (at or above every ask key) / 0x10000 (at or below every bid key: limit orders have a price field >= 2);
//   they never rest (orderbook.rs:521-524).
// Can this step run on the keyed loop?  Yes iff the prices of the live and the new LIMIT orders span <= 32 762 (and none is
// 0 or u32::MAX) and the live arrival stamps plus this step's (at most n_ev) new ones span < 65 534.
#ifndef BOURSE_AMD_KEYED_EVENTS
#define BOURSE_AMD_KEYED_EVENTS 1
#endif
// Test builds shrink the windows so that ordinary runs leave them (tests/test_build_variants.py): a value below 16 narrows
// the ARRIVAL window to that many bits, a value above 16 the PRICE window to 31 - value bits.  The key layout is fixed.
#ifndef BOURSE_AMD_KEY_SEQ_BITS
#define BOURSE_AMD_KEY_SEQ_BITS 16
#endif
// 0: a step whose prices do not fit the window always runs the two-reduction loop (rounds 2 - 4); 1: keys_begin_wide first
#ifndef BOURSE_AMD_KEYED_WIDE
#define BOURSE_AMD_KEYED_WIDE 1
#endif
constexpr uint32_t KEY_SB = BOURSE_AMD_KEY_SEQ_BITS;
constexpr uint32_t KEY_SMASK = (1u << (KEY_SB < 16u ? KEY_SB : 16u)) - 1u;       // arrival window
constexpr uint32_t KEY_PSPAN = (1u << (31u - (KEY_SB > 16u ? KEY_SB : 16u))) - 6u;  // price window
constexpr uint32_t KEY_ASK = 0x80000000u;
constexpr uint32_t KP_MKT_BID = 0xFFFFFFFFu, KP_MKT_ASK = 0x10000u;
template <int R>
__device__ __forceinline__ bool key_window(const Book<R>& B, const uint64_t (&newm)[R], uint32_t n_ev, uint32_t& pbase,
                                           uint32_t& sbase, uint32_t xmin = 0xFFFFFFFFu, uint32_t xmax = 0u) {
  // (xmin / xmax: per-lane extra prices that must fit the window too - the new prices of a host-driven step's modifications)
  uint32_t pmax = xmax, pmin = xmin, age = 0;  // age = seq_ctr - seq of the oldest live order (wrapping)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool lv = lane_bit(B.live[r]), in = lv | lane_bit(newm[r]);
    pmax = max(pmax, in ? B.price[r] : 0u);
    pmin = min(pmin, in ? B.price[r] : 0xFFFFFFFFu);
    age = max(age, lv ? B.seq_ctr - B.seq[r] : 0u);
  }
  wave_reduce3(pmax, pmin, age);
  pbase = pmin - 2u;  // price field >= 2: 1 is the market ask's
  sbase = B.seq_ctr - age - 1u;
  return pmin >= 2u && pmax != 0xFFFFFFFFu && pmax - pmin <= KEY_PSPAN && age + n_ev < KEY_SMASK - 1u;
}

// The keys of one step: key[r] per pool lane, pk[r] = the compare value (upper half) of the lane's PENDING order,
// sq = KEY_ASK | the running arrival field (advances by 1 per resting order).
template <int R>
struct KeyState {
  uint32_t key[R], pk[R];
  uint32_t sq, sbase, pbase;
  // alo <= best ask key, bhi >= best bid key (as i32): exact after a reduction of that side, still valid after any
  // removal, pulled in when an order rests beyond them - a new order on the far side of the bound cannot cross and skips
  // the reduction
  int32_t alo, bhi;
};
// MARKETS: the new orders may include market orders; they stay out of the window test
template <int R, bool MARKETS = false>
__device__ __forceinline__ bool keys_begin(const Book<R>& B, const uint64_t (&newm)[R], uint32_t n_ev, KeyState<R>& K,
                                           uint32_t xmin = 0xFFFFFFFFu, uint32_t xmax = 0u) {
  uint32_t pbase;
  uint64_t lim[R];  // the new LIMIT orders
#pragma unroll
  for (int r = 0; r < R; ++r)
    lim[r] = MARKETS ? newm[r] & ~__ballot(B.price[r] == (lane_bit(B.bid[r]) ? 0xFFFFFFFFu : 0u)) : newm[r];
  if (!key_window<R>(B, lim, n_ev, pbase, K.sbase, xmin, xmax)) return false;
  K.pbase = pbase;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool bidl = lane_bit(B.bid[r]);
    const uint32_t pf = B.price[r] - pbase, sf = B.seq[r] - K.sbase;
    K.key[r] = lane_bit(B.live[r]) ? ((pf << 16) | (bidl ? 0xFFFFu - sf : KEY_ASK | sf)) : 0u;
    // (0 for a lane without a pending order: a members' list, which holds bare slots, is classified by this word alone -
    // New iff it is not 0, a bid iff bit 15 is set)
    K.pk[r] = lane_bit(lim[r]) ? (bidl ? 0x8000u | pf : pf) : 0u;
    if (MARKETS) K.pk[r] = lane_bit(newm[r] & ~lim[r]) ? (bidl ? 0xFFFFu : 1u) : K.pk[r];
  }
  K.sq = KEY_ASK | (B.seq_ctr - K.sbase);
  K.alo = (int32_t)0x80000000u;
  K.bhi = 0x7FFFFFFF;
  return true;
}
// The keyed loop for a step whose prices do NOT fit the window (round 5).  C5 as written: MomentumAgent limit buys sit at
// mid - exp(N(0, 10^2)), a third of them clamped to price 0 - far below everything that trades - and 12 % of the book-steps
// failed key_window and ran the two-reduction loop at ~3 x the cost per event; one such book holds its whole launch.
// Here the window is anchored at the TOP price and the bids below it are SATURATED: price field 1, below every in-window
// field (>= 2) - so they lose every reduction to any in-window bid, no limit ask of the window crosses them (its compare
// value has field >= 2), a market ask does (KP_MKT_ASK is field 1, arrival 0) - exactly the reference's semantics as long as
// no aggressor REACHES one, because among themselves they are ordered by arrival only, not by price.  That is guaranteed up
// front: the volume that this step's new asks can take - the market asks', and the limit asks' priced at or below the
// highest bid that rests in this step - must not exceed the volume of the in-window bids that rest now and are not
// cancelled in this step (new bids only add to it): asks consume the best bids first, and those outlast them.  An ask below the window, a price at u32::MAX, a volume >= 2^22 (the sums stay in 32 bits) or a
// failed guard: the caller falls back to the two-reduction loop as before.  ev / n_ev: the step's event list (a slot per
// entry: the cancellations are the listed slots without a pending order); bins: >= 16 words of LDS not in use yet.
template <int R, bool MARKETS>
__device__ __forceinline__ bool keys_begin_wide(const Book<R>& B, const uint64_t (&newm)[R], uint32_t n_ev, KeyState<R>& K,
                                                const uint32_t (&ev)[R], uint32_t* bins, int lane) {
  uint64_t lim[R];
#pragma unroll
  for (int r = 0; r < R; ++r)
    lim[r] = MARKETS ? newm[r] & ~__ballot(B.price[r] == (lane_bit(B.bid[r]) ? 0xFFFFFFFFu : 0u)) : newm[r];
  BK_STAMP_TALLY(B, 9, lane);  // (diagnostic build: the narrow window failed)
  uint32_t pmax = 0, age = 0, vbig = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool lv = lane_bit(B.live[r]), in = lv | lane_bit(lim[r]);
    pmax = max(pmax, in ? B.price[r] : 0u);
    age = max(age, lv ? B.seq_ctr - B.seq[r] : 0u);
    vbig = max(vbig, (lv | lane_bit(newm[r])) ? B.vol[r] : 0u);
  }
  uint32_t dummy = 0xFFFFFFFFu;
  wave_reduce3(pmax, dummy, age);
  vbig = wave_umax(vbig);
  if (pmax == 0xFFFFFFFFu || pmax < 2u || age + n_ev >= KEY_SMASK - 1u || vbig >= (1u << 22)) {
    BK_STAMP_TALLY(B, pmax == 0xFFFFFFFFu ? 10 : (age + n_ev >= KEY_SMASK - 1u ? 11 : 12), lane);
    return false;
  }
  const uint32_t pbase = pmax > KEY_PSPAN ? pmax - KEY_PSPAN : 0u, lowp = pbase + 2u;  // in the window: lowp <= price <= pmax
  // this step's listed slots (one event per slot at most): a 512-bit map in LDS
  if (lane < 2 * R) bins[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int re = 0; re < R; ++re) {
    const uint32_t slot = ev[re] & EV_SLOT & (64u * R - 1u);
    if ((uint32_t)(re * 64 + lane) < n_ev) atomicOr(&bins[slot >> 5], 1u << (slot & 31u));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // Which asks can take bid volume at all?  A MARKET ask always; a LIMIT ask only from bids priced at or above it - in-window
  // bids by construction (its compare value has field >= 2) - and only if some bid that rests in this step is priced that
  // high: the highest price among the live and the new limit bids bounds it.  (Counting every new ask, the first version of
  // this guard turned away half of the steps it was built for: most of a NoiseAgent's limit asks sit above the book.)
  uint32_t pbid = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    pbid = max(pbid, ((lane_bit(B.live[r]) | lane_bit(lim[r])) && lane_bit(B.bid[r])) ? B.price[r] : 0u);
  pbid = wave_umax(pbid);
  uint32_t w_bid = 0, a_ask = 0;
  uint64_t bad = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool lv = lane_bit(B.live[r]), bidl = lane_bit(B.bid[r]), nw = lane_bit(newm[r]);
    const bool below = (lv | lane_bit(lim[r])) && B.price[r] < lowp;
    const bool listed = ((bins[(uint32_t)(r * 64 + lane) >> 5] >> ((uint32_t)lane & 31u)) & 1u) != 0u;
    bad |= __ballot(below && !bidl);  // an ask below the window: not this path
    w_bid += (lv && bidl && !below && !listed) ? B.vol[r] : 0u;
    // (a market ask's price is 0 <= pbid: counted by the same compare)
    a_ask += (nw && !bidl && B.price[r] <= pbid) ? B.vol[r] : 0u;
  }
  if (bad) {
    BK_STAMP_TALLY(B, 13, lane);
    return false;
  }
  w_bid = wave_add(w_bid);
  a_ask = wave_add(a_ask);
  if (a_ask > w_bid) {
    BK_STAMP_TALLY(B, 14, lane);
    return false;
  }
  BK_STAMP_TALLY(B, 15, lane);
  K.pbase = pbase;
  K.sbase = B.seq_ctr - age - 1u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool bidl = lane_bit(B.bid[r]);
    const uint32_t pf = B.price[r] < lowp ? 1u : B.price[r] - pbase, sf = B.seq[r] - K.sbase;  // (below the window: a bid, saturated)
    K.key[r] = lane_bit(B.live[r]) ? ((pf << 16) | (bidl ? 0xFFFFu - sf : KEY_ASK | sf)) : 0u;
    K.pk[r] = lane_bit(lim[r]) ? (bidl ? 0x8000u | pf : pf) : 0u;
    if (MARKETS) K.pk[r] = lane_bit(newm[r] & ~lim[r]) ? (bidl ? 0xFFFFu : 1u) : K.pk[r];
  }
  K.sq = KEY_ASK | (B.seq_ctr - K.sbase);
  K.alo = (int32_t)0x80000000u;
  K.bhi = 0x7FFFFFFF;
  return true;
}
// The MIRROR case (round 6, the host-driven step only - step_events.hpp): asks far ABOVE the book.  The window is anchored at the
// BOTTOM price and the asks above it are saturated at price field 0x7FFF - above every in-window field (<= KEY_PSPAN + 2), so they
// lose every reduction to an in-window ask, no in-window limit bid crosses them (its compare value has a smaller field), a market
// bid does (-1 is at or above every ask key) - exact as long as no aggressor reaches one: the volume of this step's new bids that
// can take ask volume at all (market bids; limit bids priced at or above the lowest ask that rests in this step) must not exceed
// the volume of the in-window asks that rest now and have no cancellation in this step's list.  A bid above the window, a price
// below 2, a volume >= 2^22 or a failed guard: false.
template <int R, bool MARKETS>
__device__ __forceinline__ bool keys_begin_wide_high(const Book<R>& B, const uint64_t (&newm)[R], uint32_t n_ev, KeyState<R>& K,
                                                     const uint32_t (&ev)[R], uint32_t* bins, int lane) {
  uint64_t lim[R];
#pragma unroll
  for (int r = 0; r < R; ++r)
    lim[r] = MARKETS ? newm[r] & ~__ballot(B.price[r] == (lane_bit(B.bid[r]) ? 0xFFFFFFFFu : 0u)) : newm[r];
  uint32_t pmin = 0xFFFFFFFFu, age = 0, vbig = 0, pdummy = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool lv = lane_bit(B.live[r]), in = lv | lane_bit(lim[r]);
    pmin = min(pmin, in ? B.price[r] : 0xFFFFFFFFu);
    age = max(age, lv ? B.seq_ctr - B.seq[r] : 0u);
    vbig = max(vbig, (lv | lane_bit(newm[r])) ? B.vol[r] : 0u);
  }
  wave_reduce3(pdummy, pmin, age);
  vbig = wave_umax(vbig);
  if (pmin < 2u || pmin > 0xFFFFFFFFu - KEY_PSPAN - 4u || age + n_ev >= KEY_SMASK - 1u || vbig >= (1u << 22)) return false;
  const uint32_t pbase = pmin - 2u, highp = pmin + KEY_PSPAN;  // in the window: pmin <= price <= highp
  if (lane < 2 * R) bins[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int re = 0; re < R; ++re) {
    const uint32_t slot = ev[re] & EV_SLOT & (64u * R - 1u);
    if ((uint32_t)(re * 64 + lane) < n_ev) atomicOr(&bins[slot >> 5], 1u << (slot & 31u));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t pask = 0xFFFFFFFFu;  // the lowest price among the asks that rest in this step
#pragma unroll
  for (int r = 0; r < R; ++r)
    pask = min(pask, ((lane_bit(B.live[r]) | lane_bit(lim[r])) && !lane_bit(B.bid[r])) ? B.price[r] : 0xFFFFFFFFu);
  pask = wave_umin(pask);
  uint32_t w_ask = 0, b_bid = 0;
  uint64_t bad = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool lv = lane_bit(B.live[r]), bidl = lane_bit(B.bid[r]), nw = lane_bit(newm[r]);
    const bool above = (lv | lane_bit(lim[r])) && B.price[r] > highp;
    const bool listed = ((bins[(uint32_t)(r * 64 + lane) >> 5] >> ((uint32_t)lane & 31u)) & 1u) != 0u;
    bad |= __ballot(above && bidl);  // a bid above the window: not this path
    w_ask += (lv && !bidl && !above && !listed) ? B.vol[r] : 0u;
    // (a market bid's price is u32::MAX >= pask: counted by the same compare)
    b_bid += (nw && bidl && B.price[r] >= pask) ? B.vol[r] : 0u;
  }
  if (bad) return false;
  w_ask = wave_add(w_ask);
  b_bid = wave_add(b_bid);
  if (b_bid > w_ask) return false;
  K.pbase = pbase;
  K.sbase = B.seq_ctr - age - 1u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool bidl = lane_bit(B.bid[r]);
    const uint32_t pf = B.price[r] > highp ? 0x7FFFu : B.price[r] - pbase, sf = B.seq[r] - K.sbase;  // (above the window: an ask, saturated)
    K.key[r] = lane_bit(B.live[r]) ? ((pf << 16) | (bidl ? 0xFFFFu - sf : KEY_ASK | sf)) : 0u;
    K.pk[r] = lane_bit(lim[r]) ? (bidl ? 0x8000u | pf : pf) : 0u;
    if (MARKETS) K.pk[r] = lane_bit(newm[r] & ~lim[r]) ? (bidl ? 0xFFFFu : 1u) : K.pk[r];
  }
  K.sq = KEY_ASK | (B.seq_ctr - K.sbase);
  K.alo = (int32_t)0x80000000u;
  K.bhi = 0x7FFFFFFF;
  return true;
}
// after the loop: the live masks and the arrival stamps of the orders resting now (the others' are never read again)
template <int R>
__device__ __forceinline__ void keys_end(Book<R>& B, const KeyState<R>& K) {
  B.seq_ctr = K.sbase + (K.sq & 0x7FFFFFFFu);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    B.live[r] = __ballot(K.key[r] != 0u);
    const uint32_t f = K.key[r] & 0xFFFFu;
    B.seq[r] = K.key[r] != 0u ? K.sbase + ((int32_t)K.key[r] < 0 ? f : 0xFFFFu - f) : B.seq[r];
  }
}
// The step's event words for the assembly loops: slot | EV_NEW | EV_BID in the lower half (as the decode kernels write
// them), the event's compare value pk[slot] in the upper half (one ds_bpermute per pool register and list register in use).
template <int R>
__device__ __forceinline__ void key_event_words(const KeyState<R>& K, const uint32_t (&ev)[R], uint32_t n_ev, uint32_t (&evw)[R]) {
#pragma unroll
  for (int re = 0; re < R; ++re) {
    evw[re] = ev[re] & 0xFFFFu;
    if (n_ev > (uint32_t)re * 64u) evw[re] |= pool_gather<R>(K.pk, ev[re] & EV_SLOT & (64u * R - 1u)) << 16;
  }
}

// (a pack expansion, not a loop: `#pragma unroll` was not honoured here in the R = 8 instantiation, and a rolled loop indexes
// the pool dynamically, which sends the whole Book to scratch memory - C5 15 -> 3 M book-steps/s)
template <int R, bool agg_bid, int... I>
__device__ __forceinline__ int32_t key_touch(const KeyState<R>& K, std::integer_sequence<int, I...>) {
  int32_t m = (int32_t)K.key[0];
  ((m = agg_bid ? min(m, (int32_t)K.key[I]) : max(m, (int32_t)K.key[I])), ...);
  return m;
}

// The keyed loop in C++ (pool sizes the assembly does not cover, market books' own lists, the -DBOURSE_AMD_ASM_EVENTS=0
// build): same semantics as match_side / slot_event_at with ONE reduction per match step and no tie handling.
template <int R>
__device__ __forceinline__ void flush_trades_compact(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                                     const uint32_t (&ev)[R]);
template <int R, bool agg_bid>
__device__ __forceinline__ bool match_side_keyed(Book<R>& B, KeyState<R>& K, const DevArgs& a, uint32_t book, uint64_t t0,
                                                 int lane, uint32_t k, int32_t kp, uint32_t& v, const uint32_t (&ev)[R]) {
  const uint32_t v0 = v;
  while (v > 0) {
    // an empty side answers with a value of the wrong sign: "no cross" by the same compare
    const int32_t m = key_touch<R, agg_bid>(K, std::make_integer_sequence<int, R>());
    const int32_t best = agg_bid ? wave_imin(m) : wave_imax(m);
    (agg_bid ? K.alo : K.bhi) = best;
    if (agg_bid ? (best > kp) : (best < kp)) break;  // inclusive crossing test (:430 / :463) in key space
    uint32_t pv = 0, tv = 0, slot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t eq = __ballot(K.key[r] == (uint32_t)best);  // the key is unique: exactly one lane of one register
      if (eq) {
        const uint32_t l = __builtin_ctzll(eq);
        pv = rdl(B.vol[r], l);
        tv = v < pv ? v : pv;
        pv -= tv;
        B.vol[r] = wrl(pv, l, B.vol[r]);
        if (pv == 0) K.key[r] = wrl(0u, l, K.key[r]);  // passive Filled -> remove_order
        slot = (uint32_t)r * 64u + l;
      }
    }
    v -= tv;
    // compact record (as the assembly loop's): position | passive side, volume, passive slot - price and ids are gathered
    // from the pool at the flush (flush_trades_compact)
    if (B.tr_n == 64) flush_trades_compact<R>(B, a, book, t0, lane, ev);
    const uint32_t tl = B.tr_n;
    B.tr_k = wrl(k | (agg_bid ? 0u : 0x80000000u), tl, B.tr_k);
    B.tr_vol = wrl(tv, tl, B.tr_vol);
    B.tr_pas = wrl(slot, tl, B.tr_pas);
    B.tr_n = tl + 1;
    B.trade_vol += tv;
  }
  return v0 != 0 && v == 0;
}
template <int R, int RS, bool CLS = true>
__device__ __forceinline__ void slot_event_keyed_at(Book<R>& B, KeyState<R>& K, const DevArgs& a, uint32_t book, uint64_t t0,
                                                    int lane, uint32_t k, uint32_t sl, uint32_t ew, const uint32_t (&ev)[R]) {
  const uint64_t bit = 1ull << sl;
  if (CLS ? !(ew & EV_NEW) : !(B.pend[RS] & bit)) {
    K.key[RS] = wrl(0u, sl, K.key[RS]);  // Cancellation
    return;
  }
  if (!CLS) B.pend[RS] &= ~bit;
  const bool is_bid = CLS ? (ew & EV_BID) != 0 : (B.bid[RS] & bit) != 0;
  const uint32_t kpu = (rdl(K.pk[RS], sl) << 16) | (is_bid ? 0xFFFFu : 0u);
  const int32_t kp = (int32_t)kpu;
  uint32_t v = rdl(B.vol[RS], sl);
  // a market order's remainder is dropped (orderbook.rs:521-524); the event words' lists (CLS) carry none
  const bool market = !CLS && kpu == (is_bid ? KP_MKT_BID : KP_MKT_ASK);
  bool filled = false;
  if (B.trading && !(is_bid ? kp < K.alo : kp > K.bhi))  // (beyond the bound: cannot cross)
    filled = is_bid ? match_side_keyed<R, true>(B, K, a, book, t0, lane, k, kp, v, ev)
                    : match_side_keyed<R, false>(B, K, a, book, t0, lane, k, kp, v, ev);
  if (!market && !filled) {  // rest the remainder with a fresh arrival field
    B.vol[RS] = wrl(v, sl, B.vol[RS]);
    const uint32_t x = kpu ^ K.sq;
    if (is_bid)
      K.bhi = max(K.bhi, (int32_t)x);
    else
      K.alo = min(K.alo, (int32_t)x);
    K.key[RS] = wrl(x, sl, K.key[RS]);
    K.sq += 1;
  }
}
template <int R, int RS = 0, bool CLS = true>
__device__ __forceinline__ void slot_event_keyed(Book<R>& B, KeyState<R>& K, const DevArgs& a, uint32_t book, uint64_t t0,
                                                 int lane, uint32_t k, uint32_t n, uint32_t ew, const uint32_t (&ev)[R]) {
  if constexpr (RS + 1 < R) {  // ONE uniform branch per pool register, every pool access below with a compile-time index
    if ((n >> 6) == (uint32_t)RS)
      slot_event_keyed_at<R, RS, CLS>(B, K, a, book, t0, lane, k, n & 63, ew, ev);
    else
      slot_event_keyed<R, RS + 1, CLS>(B, K, a, book, t0, lane, k, n, ew, ev);
  } else {
    slot_event_keyed_at<R, RS, CLS>(B, K, a, book, t0, lane, k, n & 63, ew, ev);
  }
}

// ----------------------------------------------------------------------------------
// Env::step body after the shuffle (env.rs:117-134): process the (already shuffled) event list of
// agent/slot indices, advance the clock, snapshot, flush trades.  Returns this step's trade count.
// An event is a New if the slot's pend bit is set, else a Cancellation of the slot's order.
// ----------------------------------------------------------------------------------
// MKT: the list is the MARKET's queue (market_env.rs:110-121); this book processes the events of its own agents'
// slots (`mine`) at their global positions t0 + k and skips the rest.  Returns trades; `n_own` = events processed.
// TAGGED (MKT lists written by k_agents_mixed_lanes): entry = slot | asset << 12, ownership by the tag.
template <int R, bool MKT = false, bool TAGGED = false, bool CLS = false, bool PENDKEY = false>
__device__ __forceinline__ uint32_t step_from_list(Book<R>& B, const DevArgs& a, uint32_t book, int lane,
                                                   const uint32_t (&ev)[R], uint32_t n_ev, uint32_t* bins,
                                                   uint32_t hist_slot, bool write_last, const UDiv& tick,
                                                   const uint64_t (&mine)[R], uint32_t& n_own, uint32_t asset = 0) {
  const uint64_t step_size = mk64(a.step_lo, a.step_hi);
  const uint64_t t0 = B.t;
  B.trade_vol = 0;  // reset_trade_vol (env.rs:118)
  const uint64_t trades_before = B.n_trades;
  // step_size 0 = immediate mode (the clock is the caller's, OrderBook::set_time): no step window to overflow
  if (step_size != 0 && (uint64_t)n_ev >= step_size) B.flags |= FLAG_STEP_SIZE;
  // CLS callers hand over this step's new-order lanes in B.pend (the event words classify themselves, so the mask is
  // not carried through the event loop - only the keyed loop's set-up wants it)
  // (PENDKEY - the members' lists of k_step_batch<POOLPEND> - keeps the pend bits: there they classify the events)
  uint64_t newm[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    newm[r] = (CLS || PENDKEY) ? B.pend[r] : 0ull;
    if (CLS) B.pend[r] = 0;
  }
  // MKTK (round 5): a MARKET's self-classifying list on the keyed / assembly loops too.  The other assets' events become events
  // that do nothing - a Cancellation of a slot that belongs to another asset's agent, which this book never fills - at their
  // positions in the market's queue (the time stamps are the market's, market_env.rs:110-121); n_own = this book's events.
  constexpr bool MKTK = MKT && CLS && !TAGGED;
  uint32_t evm[R], own_cnt = 0;
  bool listed = false;
#pragma unroll
  for (int re = 0; re < R; ++re) {
    evm[re] = ev[re];
    if constexpr (MKTK) {
      const uint32_t slot = ev[re] & EV_SLOT & (64u * R - 1u);
      bool own = false;
#pragma unroll
      for (int r = 0; r < R; ++r) own = (slot >> 6) == (uint32_t)r ? ((mine[r] >> (slot & 63u)) & 1ull) != 0ull : own;
      evm[re] = own ? ev[re] : (ev[re] & ~(EV_NEW | EV_BID));
      own_cnt += (uint32_t)__builtin_popcountll(__ballot(own && (uint32_t)(re * 64 + lane) < n_ev));
    }
  }
  if constexpr ((R == 2 || R == 1) && (!MKT || MKTK) && CLS && BOURSE_AMD_ASM_EVENTS) {
    listed = true;
    // hand-written event loops (event_asm.hpp); they return whenever the 64-record trade buffer is full
    uint32_t k = 0;
    const uint32_t nev = rfl(n_ev), tmask = B.trading ? 0xFFFFFFFFu : 0u;
    KeyState<R> K;
    if (BOURSE_AMD_KEYED_EVENTS && keys_begin<R>(B, newm, nev, K)) {
      // keyed loop: one signed sort key per order (side | price field | arrival field), rebuilt from {price, seq} every
      // step; the event words get their new orders' compare values here
      uint32_t evw[R];
      key_event_words<R>(K, evm, nev, evw);
      // the loop tests "no volume or trading disabled" on every new order (two scalar instructions) unless this step is
      // known not to need it: trading enabled and no new order with volume 0 (one ballot per pool register here)
      uint64_t zero_vol = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) zero_vol |= __ballot(lane_bit(newm[r]) && B.vol[r] == 0u);
      const uint32_t checked = (!B.trading || zero_vol != 0) ? 1u : 0u;
      for (;;) {
        uint32_t full;
        if constexpr (R == 2)
          full = events_key_r2(checked, k, nev, tmask, B.tr_n, K.sq, B.vol[0], B.vol[1], K.key[0], K.key[1], evw[0], evw[1],
                               B.tr_k, B.tr_vol, B.tr_pas);
        else
          full = events_key_r1(checked, k, nev, tmask, B.tr_n, K.sq, B.vol[0], K.key[0], evw[0], B.tr_k, B.tr_vol, B.tr_pas);
        // Env::get_trade_vol: the loop leaves the sum to the vector unit (one reduction per flush, not an add per trade)
        if (B.tr_n) B.trade_vol += wave_add((uint32_t)lane < B.tr_n ? B.tr_vol : 0u);
        // (the records are compact - k word, volume, passive slot: price and ids are gathered from the pool at the flush,
        // which therefore happens HERE, before the step's snapshot / store, also for the last buffer of the step)
        flush_trades_compact<R>(B, a, book, t0, lane, evw);
        if (!full) break;
      }
      keys_end<R>(B, K);
    } else if constexpr (R == 2) {
      while (events_asm_r2(k, nev, tmask, B.tr_n, B.seq_ctr, B.trade_vol, B.live[0], B.live[1], B.bid[0], B.bid[1],
                           B.price[0], B.price[1], B.vol[0], B.vol[1], B.id[0], B.id[1], B.seq[0], B.seq[1], evm[0], evm[1],
                           B.tr_k, B.tr_price, B.tr_vol, B.tr_act, B.tr_pas))
        flush_trades<R>(B, a, book, t0, lane);
    } else {
      while (events_asm_r1(k, nev, tmask, B.tr_n, B.seq_ctr, B.trade_vol, B.live[0], B.bid[0], B.price[0], B.vol[0], B.id[0],
                           B.seq[0], evm[0], B.tr_k, B.tr_price, B.tr_vol, B.tr_act, B.tr_pas))
        flush_trades<R>(B, a, book, t0, lane);
    }
  } else if (KeyState<R> K; (CLS || PENDKEY) && (!MKT || MKTK) && BOURSE_AMD_KEYED_EVENTS &&
             (keys_begin<R, !CLS>(B, newm, rfl(n_ev), K) ||
              // (members' lists only: RandomAgents draw their prices from a bounded tick window, and the extra path costs
              // k_step_batch<8>'s RandomAgents instantiation 32 B of scratch at its 96-register claim)
              (BOURSE_AMD_KEYED_WIDE && !CLS && keys_begin_wide<R, true>(B, newm, rfl(n_ev), K, ev, bins, lane)))) {
   if constexpr ((R == 4 || R == 8) && BOURSE_AMD_ASM_EVENTS && BOURSE_AMD_ASM_R48) {
    // the generated assembly loop (event_asm_gen.hpp): signed keys, compact trade records.
    // It reads SELF-CLASSIFYING event words (slot | EV_NEW | EV_BID) with the new order's compare value in the upper half.
    // The members' lists of an AgentSet (PENDKEY) hold bare slots - New iff the slot's pend bit is set, the side in its bid
    // bit: both are in the compare value the gather fetches anyway (0 = nothing pending, bit 15 = bid), so the lists are
    // classified by it (round 4's first version selected the slot's mask words from 4 R scalar pairs per list register:
    // ~100 vector instructions each).  A slot holds at most one event per step (a pending order sits in a slot that was
    // free when the step began), so classifying up front is the same as classifying at the event.
    listed = true;
    uint32_t evw[R];
    key_event_words<R>(K, evm, rfl(n_ev), evw);
    if (!CLS) {
#pragma unroll
      for (int re = 0; re < R; ++re) {
        const uint32_t pk = evw[re] >> 16;  // (the gather above: the slot's compare value, 0 = no pending order there)
        evw[re] = (evw[re] & (0xFFFF0000u | EV_SLOT)) | (pk != 0u ? EV_NEW | ((pk & 0x8000u) ? EV_BID : 0u) : 0u);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) B.pend[r] = 0;  // every pending order is an event of this step
    }
    // ... and the new orders' volumes by event position (the loop reads an event's volume by the lane it reads the event
    // word from instead of addressing the slot's pool row)
    uint32_t evq[R];
#pragma unroll
    for (int re = 0; re < R; ++re)
      evq[re] = n_ev > (uint32_t)re * 64u ? pool_gather<R>(B.vol, evw[re] & EV_SLOT & (64u * R - 1u)) : 0u;
    uint32_t k = 0;
    const uint32_t nev = rfl(n_ev), tmask = B.trading ? 0xFFFFFFFFu : 0u;
    // (as for the smaller pools: the "no volume or trading disabled" test only in steps that need it)
    uint64_t zero_vol = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) zero_vol |= __ballot(lane_bit(newm[r]) && B.vol[r] == 0u);
    const uint32_t checked = (!B.trading || zero_vol != 0) ? 1u : 0u;
    for (;;) {
      uint32_t full;
      if constexpr (R == 4 && CLS)
        full = events_key_r4(checked, k, nev, tmask, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas);
      else if constexpr (R == 4)
        full = events_key_r4m(checked, k, nev, tmask, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas);
      else if constexpr (CLS)
        full = events_key_r8(checked, k, nev, tmask, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas);
      else
        full = events_key_r8m(checked, k, nev, tmask, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas);
      if (B.tr_n) B.trade_vol += wave_add((uint32_t)lane < B.tr_n ? B.tr_vol : 0u);
      flush_trades_compact<R>(B, a, book, t0, lane, evw);
      if (!full) break;
    }
   } else {
#pragma unroll
    for (int re = 0; re < R; ++re) {
      const uint32_t kb = re * 64;
      if (n_ev > kb) {
        const uint32_t cnt = rfl((n_ev - kb) < 64u ? (n_ev - kb) : 64u);
        for (uint32_t l = 0; l < cnt; ++l) {
          const uint32_t ew = rdl(evm[re], l);
          slot_event_keyed<R, 0, CLS>(B, K, a, book, t0, lane, kb + l, CLS ? (ew & EV_SLOT) : ew, ew, evm);
        }
      }
    }
    listed = true;
    flush_trades_compact<R>(B, a, book, t0, lane, evm);  // (the loop's records are compact: filled in before the snapshot)
   }
    keys_end<R>(B, K);
  } else
#pragma unroll
  for (int re = 0; re < R; ++re) {  // events at t0 + k (env.rs:123-127); entry k lives in lane k & 63 of ev[k >> 6]
    const uint32_t kb = re * 64;
    if (n_ev > kb) {
      const uint32_t cnt = rfl((n_ev - kb) < 64u ? (n_ev - kb) : 64u);
      for (uint32_t l = 0; l < cnt; ++l) {
        uint32_t slot = rdl(ev[re], l);
        const uint32_t ew = slot;
        if (CLS) slot &= EV_SLOT;
        if (MKT && TAGGED) {
          if ((slot >> 12) != asset) continue;
          slot &= 0xFFFu;
          ++n_own;
        } else if (MKT) {
          if (!mask_test<R>(mine, slot)) continue;
          ++n_own;
        }
        process_slot_event<R, 0, CLS>(B, a, book, t0, lane, kb + l, slot, ew);
      }
    }
  }
  if (!MKT) n_own = n_ev;
  if (MKTK && listed) n_own = own_cnt;
  BK_STAMP(B, 0, 3, lane);  // key set-up + event loop
  B.n_events += n_own;
  B.t = t0 + step_size;  // env.rs:129
  // env.rs:132-134.  Env::level_2_data (the "latest" record) only needs a launch's final snapshot;
  // with no history buffer every step's record is written there.
  snapshot<R>(B, a, book, lane, bins, hist_slot, B.flags, write_last, tick);
  flush_trades<R>(B, a, book, t0, lane);
  BK_STAMP(B, 0, 4, lane);  // level-2 snapshot + trade flush
  return (uint32_t)(B.n_trades - trades_before);
}

constexpr int LDS_DW_PER_WAVE = 4 * 64;  // level bins: 4 * levels dwords, levels <= 64

// ==================================================================================
// Kernel 1: fused on-device order flow.  n_steps x { RandomAgents::update for every
// group (random_agent.rs:85-119); Env::step (env.rs:116-135) } per book, state kept in
// registers across the steps of the launch (sim_runner, runner.rs:53-68).
// ==================================================================================
template <int R>
__global__ __launch_bounds__(256) void k_run_random(DevArgs a, uint64_t first_step, uint32_t n_steps) {
  __shared__ uint32_t lds[4][LDS_DW_PER_WAVE];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t book = rfl(blockIdx.x * 4 + wv);
  if (book >= a.n_books) return;
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t* bins = lds[wv];

  Book<R> B;
  Rng rng;
  load_book<R>(B, rng, st, lane);
  uint32_t last_ntr = 0, last_nev = 0;
  uint32_t ev[R];  // this step's event list: entry k (lane k & 63 of ev[k >> 6]) = agent/slot index
#pragma unroll
  for (int r = 0; r < R; ++r) ev[r] = 0;

  for (uint32_t s = 0; s < n_steps; ++s) {
    // ---------------- agents.update(env, rng): groups in declaration order -------------
    uint32_t n_ev = 0;
    {
      uint32_t g = 0;
      Group G = a.groups[0];
      uint32_t gend = G.n;
      uint32_t n = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        for (uint32_t l = 0; l < 64; ++l, ++n) {
          if (n >= a.n_agents_total) break;
          while (n >= gend) {
            ++g;
            G = a.groups[g];
            gend += G.n;
          }
          const uint32_t x = rng.next_u32();  // p = gen::<f32>()  (random_agent.rs:91)
          if ((x >> 8) < G.thr) {             // p < activity_rate
            slot_write<R>(ev, n_ev, n);
            n_ev += 1;
            const uint64_t bit = 1ull << l;
            if (!(B.live[r] & bit)) {
              // no Active order held: place a new one.  Draw order: side, tick, vol (:99-101)
              // [Ask, Bid].choose -> gen_range(0..2): zone = (2 << 30) - 1, i.e. half the draws are rejected
              const uint32_t side = rng.below(2u, 0x7FFFFFFFu);  // 0 = Ask, 1 = Bid
              const uint32_t tick = G.tick_lo + rng.below(G.tick_rng, G.tick_zone);
              const uint32_t vol = G.vol_lo + rng.below(G.vol_rng, G.vol_zone);
              B.price[r] = wrl(tick * G.tick_size, l, B.price[r]);
              B.vol[r] = wrl(vol, l, B.vol[r]);
              B.id[r] = wrl(B.next_id, l, B.id[r]);  // create_order: id = orders.len() (orderbook.rs:363)
              B.next_id += 1;
              B.bid[r] = side ? (B.bid[r] | bit) : (B.bid[r] & ~bit);
              B.pend[r] |= bit;
            }
            // else: holds an Active order -> queue its cancellation (:95-97); nothing to store,
            // the event is told apart from a New by the slot's pend bit.
          }
        }
      }
    }
    // ---------------- Env::step -----------------------------------------------------
    // transactions.shuffle(rng) (env.rs:121; App. B.4)
    for (uint32_t i = n_ev; i-- > 1;) {
      const uint32_t j = rng.below(i + 1);
      const uint32_t ai = slot_read<R>(ev, i), aj = slot_read<R>(ev, j);
      slot_write<R>(ev, i, aj);
      slot_write<R>(ev, j, ai);
    }
    last_ntr = step_from_list<R>(B, a, book, lane, ev, n_ev, bins, a.hist_cap ? (a.hist_slot0 + s) % a.hist_cap : 0u,
                                 s + 1 == n_steps || a.hist_cap == 0, a.tick_div, B.pend, last_nev);
  }
  store_book<R>(B, rng, st, lane, first_step + n_steps, last_ntr, last_nev);
}

// ==================================================================================
// Split pipeline for large batches (the RNG-serial phases dominate the fused kernel's SALU issue):
//
//   k_agents_fsm<R>  ONE LANE PER BOOK (64 books per wave).  RandomAgents::update for every group
//                    (random_agent.rs:85-119) + the Fisher-Yates shuffle of Env::step (env.rs:121), i.e.
//                    everything that consumes the book's RNG stream, as a per-lane state machine in which
//                    every iteration performs exactly one next_u32() draw - lanes never wait for each
//                    other's rejection loops.  Emits a per-book "step batch" (event list + new orders).
//   k_step_batch<R>  ONE WAVE PER BOOK.  Applies the batch to the register-resident pool and runs the
//                    event loop / snapshot of Env::step (env.rs:117-134) exactly as the fused kernel.
// ==================================================================================
enum Phase : uint32_t { PH_ACT = 0, PH_SIDE = 1, PH_TICK = 2, PH_VOL = 3, PH_SHUF = 4, PH_DONE = 5 };

#ifndef BOURSE_AMD_FSM_FREERUN
#define BOURSE_AMD_FSM_FREERUN 0
#endif
#ifndef BOURSE_AMD_FSM_TOP_VGPR
#define BOURSE_AMD_FSM_TOP_VGPR 231  // highest VGPR k_agents_fsm claims (0 = only what it uses); see the kernel's prologue
#endif
#define BK_STR2(x) #x
#define BK_STR(x) BK_STR2(x)

constexpr uint32_t fsm_lds_bytes(int R) { return 64u * R * 32u * 4u; }

template <int R>
__global__ __launch_bounds__(64) void k_agents_fsm(DevArgs a) {
  // LDS is DYNAMIC (fsm_lds_bytes(R) at launch) so that the kernel's register footprint is the one chosen below: with a
  // static size the compiler derives "at most 2 waves per SIMD" from it and pads the kernel descriptor's register request
  // up to that occupancy's budget (R = 2: 169 VGPRs + 102 SGPRs for a kernel using 34 + 46; 257 VGPRs for R = 4, 8).
  extern __shared__ uint32_t fsm_lds[];
  uint16_t* list = reinterpret_cast<uint16_t*>(fsm_lds);  // event list of lane l: list[k * 64 + l], 64 * R * 64 entries
  // (which agents place, and on which side, is not tracked here: the event words say it - bit 15 New, bit 14 bid - and
  // k_step_batch rebuilds the masks from them with four LDS atomics per book instead of two per new order in this loop)
  const int lane = threadIdx.x;
  // This kernel is a dependent chain of ~600 iterations on ONE wave per SIMD, co-resident with up to 7 waves of the
  // issue-bound event kernel of another part: top issue priority lets the chain run at its lone-wave pace (the part's
  // next k_step_batch cannot start before it ends) at no cost to the event kernel's throughput
  __builtin_amdgcn_s_setprio(3);
  // ... and it CLAIMS far more VGPRs than it uses (34): the chain is VALU-latency bound, and every k_step_batch wave
  // sharing its SIMD's VALU stretches it (151 us alone, 185-194 us under 8 event waves).  A 232-VGPR footprint leaves
  // room for 7 event waves beside one of these waves and 1 beside two of them, instead of 8 and 8; measured on C3
  // (profiles/r02/fsm_vgpr_sweep.txt): no pad 184 M, 104: 198, 168: 214, 200-264: 215-220 (plateau), 296: 169 M
  // book-steps/s (from 296 up the dispatcher cannot place these waves until a whole SIMD drains).
#if BOURSE_AMD_FSM_TOP_VGPR > 0
  asm volatile("" ::: "v" BK_STR(BOURSE_AMD_FSM_TOP_VGPR));
#endif
  // MarketEnv mode (assets = M > 1): the lane owns a MARKET = books [b*M, b*M + M) with one RNG stream and one event
  // queue (market_env.rs:110-121, runner.rs:108-131); RandomMarketAgents::update is RandomAgents::update addressed to
  // the group's asset (random_agent.rs:204-247), so the state machine below is unchanged.
  const uint32_t M = a.assets;
  const uint32_t b = a.book_begin + blockIdx.x * 64 + lane;
  if (b >= a.book_end) return;
  uint32_t* st = a.state + (size_t)b * M * a.state_stride;
  uint32_t* bt = a.batch + (size_t)b * M * a.batch_stride;

  RngLane rng;
  {
    const uint2 x0 = *reinterpret_cast<const uint2*>(st + H_S0_LO);
    const uint2 x1 = *reinterpret_cast<const uint2*>(st + H_S1_LO);
    rng.a0 = x0.x, rng.a1 = x0.y, rng.b0 = x1.x, rng.b1 = x1.y;
  }
  uint64_t live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint2 x = *reinterpret_cast<const uint2*>(st + H_LIVE0 + 2 * r);
    for (uint32_t as = 1; as < M; ++as) {  // slot = agent index in every book of the market: the masks are disjoint
      const uint2 y = *reinterpret_cast<const uint2*>(st + (size_t)as * a.state_stride + H_LIVE0 + 2 * r);
      x.x |= y.x;
      x.y |= y.y;
    }
    live[r] = mk64(x.x, x.y);
  }
  uint2* pv = reinterpret_cast<uint2*>(bt + BT_EV + 32 * R);

  // ---- loop 1: agents.update, group by group (declaration order).  Inside a group every lane runs a state
  // machine that performs exactly ONE next_u32() draw per iteration (a lane never waits for another lane's rejection
  // loop); the group's parameters are wave-uniform (SGPRs).  Lanes re-converge at each group boundary.
  // Select-style body (v_cndmask) with three short predicated blocks: list append, new-order store, next agent.
  const uint32_t five = RngLane::opaque5();
  uint32_t n = 0, n_ev = 0, gbase = 0;
  for (uint32_t g = 0; g < a.n_groups; ++g) {
    const Group G = a.groups[g];
    const uint32_t gend = gbase + G.n;
    gbase = gend;
    if (G.n == 0) continue;
    // complete the scalar loads of the group's parameters HERE: a wait parked inside the loop would be
    // s_waitcnt lgkmcnt(0), which also waits for the iteration's own LDS writes (list append, ds_or) to drain
    asm volatile("" ::"s"(G.thr), "s"(G.tick_lo), "s"(G.tick_rng), "s"(G.tick_zone));
    asm volatile("" ::"s"(G.vol_lo), "s"(G.vol_rng), "s"(G.vol_zone), "s"(G.tick_size));
    // Every predicate of a draw is taken as a WAVE MASK first (v_cmp into an SGPR pair), then the generator's state
    // update runs (11 vector instructions that depend on none of them), and only then does the scalar unit combine the
    // masks: an SALU instruction that reads an SGPR a vector compare has JUST written stalls the wave ~16 clocks
    // (scripts/micro/lone_wave_latency.hip), and the straightforward form - `bool` predicates combined where they are
    // used - had five of those per draw (k_agents_fsm 161 -> 141 us per launch under load).  Same instructions, same
    // counts: the two fences only fix their order.
    // the group's ranges and zones as VECTOR registers for the loop: a select under an SGPR mask cannot also read an
    // SGPR source (one scalar operand per VOP3), so the compiler copied each of the four into a VGPR on every draw
    uint32_t v_trng = G.tick_rng, v_vrng = G.vol_rng, v_tzone = G.tick_zone, v_vzone = G.vol_zone;
    asm volatile("" : "+v"(v_trng), "+v"(v_vrng), "+v"(v_tzone), "+v"(v_vzone));
    const uint64_t thr8 = (uint64_t)G.thr << 8;
    // price = (tick_lo + val) * tick_size as one multiply-add: val * tick_size + tick_lo * tick_size (mod 2^32)
    uint64_t price0 = (uint64_t)(G.tick_lo * G.tick_size);
    asm volatile("" : "+v"(price0));  // (kept in a VGPR pair: the addend of the multiply-add below)
    // The group is walked in SEGMENTS that stay inside one 64-slot pool register, so that the live word of the agent
    // at hand is one register pair per segment, not a per-draw select over the pool's registers (lanes re-converge at
    // a segment's end as they do at a group's; the benchmark groups are 64-aligned: no extra boundary there).
    for (uint32_t sbeg = gend - G.n; sbeg < gend;) {
#if BOURSE_AMD_FSM_FREERUN
      // MEASUREMENT BUILD (docs/EXPERIMENTS.md): no re-convergence at the 64-slot segment boundary inside a
      // group - the live word is selected per lane from the lane's own agent index (2 (R - 1) more vector instructions per draw)
      const uint32_t send = gend;
#else
      const uint32_t send = gend < (sbeg | 63u) + 1u ? gend : (sbeg | 63u) + 1u;
#endif
      uint64_t w = live[0];
#pragma unroll
      for (int r = 1; r < R; ++r) w = ((sbeg >> 6) == (uint32_t)r) ? live[r] : w;
      uint32_t cur_side = 0, cur_price = 0;
      // The phase of every lane lives in four WAVE MASKS carried across the iterations in scalar registers (one-hot per
      // lane) and is advanced by scalar mask algebra at the end of the iteration - round 3 kept it in a vector register:
      // four compares to get the masks and four selects to write the next phase, every draw.  Bits of lanes that have
      // left the loop go stale, harmlessly: every predicate they are combined with is a ballot of the lanes still in it.
      uint64_t P_ACT = ~0ull, P_SIDE = 0, P_TICK = 0, P_VOL = 0;
      while (n < send) {
        const uint32_t x = rng.output(five);
        // range and zone of the phase at hand, from its masks (two selects each; no loop-carried copies)
        const uint32_t range = sel(P_SIDE, 2u, sel(P_TICK, v_trng, v_vrng));
        const uint32_t zone = sel(P_SIDE, 0x7FFFFFFFu, sel(P_TICK, v_tzone, v_vzone));
        const uint64_t m = (uint64_t)x * range;  // sample_single step of the current phase: accept iff lo <= zone
        const uint32_t val = (uint32_t)(m >> 32);
        // gen::<f32>() < activity_rate (:91-93): (x >> 8) < thr as ONE 64-bit compare x < thr << 8 (thr <= 2^24)
        uint64_t C_HIT = __builtin_amdgcn_ballot_w64((uint64_t)x < thr8);
        uint64_t C_ACC = __builtin_amdgcn_ballot_w64((uint32_t)m <= zone);
#if BOURSE_AMD_FSM_FREERUN
        uint64_t wl = live[0];
#pragma unroll
        for (int r = 1; r < R; ++r) wl = ((n >> 6) == (uint32_t)r) ? live[r] : wl;
        uint64_t C_LIVE = __builtin_amdgcn_ballot_w64(((wl >> (n & 63)) & 1ull) != 0);
#else
        uint64_t C_LIVE = __builtin_amdgcn_ballot_w64(((w >> (n & 63)) & 1ull) != 0);  // Active order held (:95-97)
#endif
        asm volatile("" : "+v"(rng.a0), "+v"(rng.a1), "+v"(rng.b0), "+v"(rng.b1)
                     : "s"(P_ACT), "s"(P_SIDE), "s"(P_TICK), "s"(P_VOL), "s"(C_HIT), "s"(C_ACC), "s"(C_LIVE));
        rng.advance();
        asm volatile("" : "+v"(rng.a0), "+v"(rng.a1), "+v"(rng.b0), "+v"(rng.b1), "+s"(P_ACT), "+s"(P_SIDE), "+s"(P_TICK),
                       "+s"(P_VOL), "+s"(C_HIT), "+s"(C_ACC), "+s"(C_LIVE));
        const uint64_t HIT = P_ACT & C_HIT, CANCEL = HIT & C_LIVE, TO_SIDE = HIT & ~C_LIVE;
        const uint64_t A_SIDE = C_ACC & P_SIDE, A_TICK = C_ACC & P_TICK, A_VOL = C_ACC & P_VOL;
        const uint64_t QUEUE = CANCEL | A_VOL, ADV = (P_ACT & ~C_HIT) | QUEUE;
        cur_side = sel(A_SIDE, val, cur_side);                                    // 0 = Ask, 1 = Bid ([Ask, Bid].choose, :99)
        {  // tick * tick_size (:100,:107)
          uint64_t pr, cy;
          asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(pr), "=s"(cy) : "v"(val), "s"(G.tick_size), "v"(price0));
          cur_price = sel(A_TICK, (uint32_t)pr, cur_price);
        }
        // the agent's event, queued once its kind is known (agent order): bit 15 = New, bit 14 = bid
        if (lane_bit(QUEUE)) list[n_ev * 64 + lane] = (uint16_t)sel(A_VOL, n | EV_NEW | (cur_side << 14), n);
        asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(n_ev) : "s"(QUEUE) : "vcc");  // n_ev += lane_bit(QUEUE)
        if (lane_bit(A_VOL)) pv[n] = make_uint2(cur_price, G.vol_lo + val);       // vol drawn last (:101): the order is complete
        // next phase; an agent that is done (inactive, cancelled or placed) hands over to the next one
        P_SIDE = TO_SIDE | (P_SIDE & ~C_ACC);
        P_TICK = A_SIDE | (P_TICK & ~C_ACC);
        P_VOL = A_TICK | (P_VOL & ~C_ACC);
        P_ACT = ADV;
        asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(n) : "s"(ADV) : "vcc");  // n += lane_bit(ADV)
      }
      sbeg = send;
    }
  }

  // ---- loop 2: transactions.shuffle(rng) (env.rs:121): for i in (1..n_ev).rev() { swap(i, gen_index(i + 1)) }
  // (rand SliceRandom::shuffle), again one draw per iteration per lane.
  {
    uint32_t i = n_ev > 1 ? n_ev - 1 : 0;
    uint32_t rg = i + 1;
    uint32_t zn = (rg << __builtin_clz(rg)) - 1u;
    while (i != 0) {
      const uint32_t x = rng.output(five);
      const uint64_t m = (uint64_t)x * rg;
      uint64_t ACC = __builtin_amdgcn_ballot_w64((uint32_t)m <= zn);  // (same ordering as in loop 1)
      asm volatile("" : "+v"(rng.a0), "+v"(rng.a1), "+v"(rng.b0), "+v"(rng.b1) : "s"(ACC));
      rng.advance();
      asm volatile("" : "+v"(rng.a0), "+v"(rng.a1), "+v"(rng.b0), "+v"(rng.b1), "+s"(ACC));
      if (lane_bit(ACC)) {
        const uint32_t j = (uint32_t)(m >> 32);
        const uint16_t ai = list[i * 64 + lane], aj = list[j * 64 + lane];
        list[i * 64 + lane] = aj;
        list[j * 64 + lane] = ai;
        --i;
        rg = i + 1;
        zn = (rg << __builtin_clz(rg)) - 1u;
      }
    }
  }

  // publish: RNG state back to the book header, the step batch for k_step_batch
  for (uint32_t as = 0; as < M; ++as) {  // every book of a market carries a copy of the market's RNG state
    *reinterpret_cast<uint2*>(st + (size_t)as * a.state_stride + H_S0_LO) = make_uint2(rng.a0, rng.a1);
    *reinterpret_cast<uint2*>(st + (size_t)as * a.state_stride + H_S1_LO) = make_uint2(rng.b0, rng.b1);
  }
  bt[BT_NEV] = n_ev;
  for (uint32_t k = 0; k < n_ev; k += 2) {
    const uint32_t lo = list[k * 64 + lane];
    const uint32_t hi = (k + 1 < n_ev) ? list[(k + 1) * 64 + lane] : 0u;
    bt[BT_EV + (k >> 1)] = lo | (hi << 16);
  }
}

template <int R, bool MKT, bool POOLPEND>
__device__ __forceinline__ void step_batch_book(const DevArgs& a, uint32_t book, int lane, uint32_t* lds, uint64_t step_index,
                                                uint32_t write_last, Book<R>& B, Rng& rng, uint32_t hist_slot);
template <int R, bool MKT, bool POOLPEND>
__device__ __forceinline__ void step_batch_raw(const DevArgs& a, uint32_t book, int lane, uint32_t* lds, uint64_t step_index,
                                               uint32_t write_last, const StepRaw<R>& w, Book<R>& B, Rng& rng, uint32_t hist_slot);
// POOLPEND (split pipeline of AgentSets with Noise/Momentum members, k_agents_mixed): the new orders already sit in the
// pool with their pend bit and id (created by the members' update); the batch only carries the shuffled event list.
template <int R, bool MKT, bool POOLPEND = false>
// (512-slot pools: asked to fit 5 waves per SIMD - 96 VGPRs, 44-52 B of scratch - instead of the 119 VGPRs / 4 waves the
// compiler takes by itself: C5 stand-in 272 -> 186 us per launch, 19.0 -> 21.1 M book-steps/s; 6 waves: 201 us.  Since
// round 4's load restructuring the compiler reaches 80 VGPRs / 6 waves without scratch by itself, which is that slower
// 201 us form (20.2 instead of 21.3 M): the kernel CLAIMS 96 registers - a clobber of v95, as k_agents_fsm claims its
// footprint - so that the hardware places five waves per SIMD again; -DBOURSE_AMD_SB8_TOP_VGPR=0 drops the claim)
#ifndef BOURSE_AMD_SB8_TOP_VGPR
#define BOURSE_AMD_SB8_TOP_VGPR 95
#endif
__global__ __launch_bounds__(64, R >= 8 ? 5 : 1) void k_step_batch(DevArgs a, uint64_t step_index, uint32_t write_last) {
  // one-wave workgroups: the dispatcher places every wave independently, so the wave slots left beside the
  // co-running k_agents_fsm waves are all usable (4-wave workgroups needed a free slot on every SIMD)
  __shared__ uint32_t lds[LDS_DW_PER_WAVE];
#if BOURSE_AMD_SB8_TOP_VGPR > 0
  if constexpr (R >= 8) asm volatile("" ::: "v" BK_STR(BOURSE_AMD_SB8_TOP_VGPR));
#endif
  // behind the wave-parallel decode the event waves go first from a batch size that depends on the pool (bourse_amd.hip
  // wave_step_prio_books: +5-7 % at 16 384 - 24 576 books, +2 % at 8 192 for pools of <= 128 slots since round 4's kernels,
  // -2 % there for the 512-slot pools; nothing beside k_agents_fsm, which runs at priority 3 anyway)
  if (a.step_prio) __builtin_amdgcn_s_setprio(1);
  const int lane = threadIdx.x;
  // MKT: book = market * assets + asset; the step batch is the market's (stored at the market's first book)
  const uint32_t book = MKT ? a.book_begin * a.assets + blockIdx.x : a.book_begin + blockIdx.x;
  if (book >= (MKT ? a.book_end * a.assets : a.book_end)) return;
  Book<R> B;
  Rng rng;
  // (Round 4 tried TWO books per wave, the second one's loads in flight while the first one's events run - the memory
  // round trip at a wave's start is the longest single link of its chain: the 18 registers of the second book take the
  // kernel from 40 to 64 VGPRs, only four event waves then fit beside a k_agents_fsm wave, and C3 fell from 281 to
  // 221 - 236 M whatever the lane kernel's own footprint was set to: docs/EXPERIMENTS.md.)
  step_batch_book<R, MKT, POOLPEND>(a, book, lane, lds, step_index, write_last, B, rng, a.hist_slot0);
}

// One book's Env::step from its step batch (the body of k_step_batch; round 4's two experimental kernels ran it in front of the next step's
// decode).  lds: LDS_DW_PER_WAVE dwords of this wave's.  Leaves the stored book in B / rng for a caller that goes on.
template <int R, bool MKT, bool POOLPEND>
__device__ __forceinline__ void step_batch_book(const DevArgs& a, uint32_t book, int lane, uint32_t* lds, uint64_t step_index,
                                                uint32_t write_last, Book<R>& B, Rng& rng, uint32_t hist_slot) {
  BK_STAMP_START(B, book);
  StepRaw<R> w;
  const uint32_t mkt_book0 = MKT ? (book / a.assets) * a.assets : book;
  const uint32_t* st = a.state + (size_t)book * a.state_stride;
  const uint32_t* bt = a.batch + (size_t)mkt_book0 * a.batch_stride;
  load_state_raw<R>(w, st, lane);
  w.bh = bt[lane];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    w.ev[r] = reinterpret_cast<const uint16_t*>(bt + BT_EV)[r * 64 + lane];
    w.pv[r] = POOLPEND ? make_uint2(0u, 0u) : reinterpret_cast<const uint2*>(bt + BT_EV + 32 * R)[r * 64 + lane];
  }
  load_state_scalars<R>(w, st);  // (behind the vector loads: its wait hides under their round trip)
#if BOURSE_AMD_STAMPS && !defined(BOURSE_AMD_FSM_UNIT)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase 0 = the loads' round trip
#endif
  BK_STAMP(B, 0, 0, lane);
  step_batch_raw<R, MKT, POOLPEND>(a, book, lane, lds, step_index, write_last, w, B, rng, hist_slot);
  BK_STAMP(B, 0, 5, lane);  // store
  BK_STAMP_COUNT(B, 0, lane);
}
// ... from what load_step_raw / the wrapper above loaded
template <int R, bool MKT, bool POOLPEND>
__device__ __forceinline__ void step_batch_raw(const DevArgs& a, uint32_t book, int lane, uint32_t* lds, uint64_t step_index,
                                               uint32_t write_last, const StepRaw<R>& w, Book<R>& B, Rng& rng, uint32_t hist_slot) {
  const uint32_t mkt_book0 = MKT ? (book / a.assets) * a.assets : book;
  const uint32_t asset = book - mkt_book0;
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint64_t mine[R];  // MKT: the pool slots (= agent indices) of the groups trading this asset
#pragma unroll
  for (int r = 0; r < R; ++r) mine[r] = MKT ? 0ull : ~0ull;
  if (MKT) {
    uint32_t gb = 0;
    for (uint32_t g = 0; g < a.n_groups; ++g) {
      const uint32_t ge = gb + a.groups[g].n;
      if (a.groups[g].asset == asset) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t lo = gb > 64u * r ? gb : 64u * r, hi = ge < 64u * r + 64u ? ge : 64u * r + 64u;
          if (lo < hi) mine[r] |= ((hi - lo) == 64u ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << (lo - 64u * r);
        }
      }
      gb = ge;
    }
  }

  unpack_book<R>(B, rng, w);
  // the step batch: header words, event list (u16), new-order {price, vol} per agent slot
  const uint32_t n_ev = rdl(w.bh, BT_NEV);
  uint32_t ev[R];
  uint32_t owner[R];  // POOLPEND: the members' owner tags ride in meta bits 8..15 and must survive the store
  uint32_t base = B.next_id;
#pragma unroll
  for (int r = 0; r < R; ++r) ev[r] = w.ev[r];
  // Which agent slots place an order in this step, and on which side: rebuilt from the event words (bit 15 New, bit 14
  // bid, slot below) - two LDS atomics per list register scatter the bits, lanes 0 .. 4R-1 read the words back.
  uint32_t mw = 0;
  if (!POOLPEND) {
    uint32_t* pm = lds;  // 2R words placing, 2R words bid side (the level bins are not in use yet)
    if (lane < 4 * R) pm[lane] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t ew = ev[r], slot = ew & EV_SLOT, bit = 1u << (slot & 31);
      const bool is_new = (uint32_t)(r * 64 + lane) < n_ev && (ew & EV_NEW);
      atomicOr(&pm[slot >> 5], is_new ? bit : 0u);
      atomicOr(&pm[2 * R + (slot >> 5)], (is_new && (ew & EV_BID)) ? bit : 0u);
    }
    mw = pm[lane & (4 * R - 1)];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (POOLPEND) {
      owner[r] = w.f[r][4] & 0xFF00u;
      continue;
    }
    const uint64_t pend = mk64(rdl(mw, 2 * r), rdl(mw, 2 * r + 1)) & mine[r];
    const uint64_t side = mk64(rdl(mw, 2 * R + 2 * r), rdl(mw, 2 * R + 2 * r + 1));
    const uint2 pv = w.pv[r];
    B.price[r] = sel(pend, pv.x, B.price[r]);
    B.vol[r] = sel(pend, pv.y, B.vol[r]);
    // create_order ids: dense, in agent order (orderbook.rs:363): base + #placing agents below this slot
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
    B.id[r] = sel(pend, base + rank, B.id[r]);
    base += __builtin_popcountll(pend);
    B.bid[r] = (B.bid[r] & ~pend) | (side & pend);
    B.pend[r] = pend;  // handed to step_from_list, which clears it (the event words classify themselves: EV_NEW)
  }
  B.next_id = base;
  uint32_t n_own = 0;
  BK_STAMP(B, 0, 1, lane);  // unpack + placing masks + new orders into the pool
  const uint32_t ntr = step_from_list<R, MKT, MKT && POOLPEND, !POOLPEND, POOLPEND && !MKT>(B, a, book, lane, ev, n_ev, lds, hist_slot,
                                                               write_last != 0, MKT ? a.asset_div[asset] : a.tick_div,
                                                               mine, n_own, asset);
  store_book<R, !POOLPEND>(B, rng, st, lane, step_index + 1, ntr, n_own);
  if (POOLPEND) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t* p = st + HDR_DW + r * POOL_FIELDS * 64 + 4 * 64;
      p[lane] = (lane_bit(B.live[r]) ? 1u : 0u) | (lane_bit(B.bid[r]) ? 2u : 0u) | owner[r];
    }
  }
}

// (Kernel 2, the host-driven order flow - k_step_events -, lives in step_events.hpp: it borrows the wave-parallel shuffle of
// wave_agents.hpp)

// ==================================================================================
// small service kernels (not templates: defined in ONE translation unit - fsm_unit.hip leaves them out)
// ==================================================================================
#ifndef BOURSE_AMD_FSM_UNIT
// set trade_base = n_trades for every book (bk_clear_trades) / set trading flag
__global__ void k_book_service(uint32_t* state, uint32_t stride, uint32_t n_books, int op, uint32_t value) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_books) return;
  uint32_t* h = state + (size_t)b * stride;
  if (op == 0) {
    h[H_TRADE_BASE_LO] = h[H_TRADES_LO];
    h[H_TRADE_BASE_HI] = h[H_TRADES_HI];
  } else if (op == 1) {
    h[H_TRADING] = value;
  } else if (op == 2) {
    h[H_FLAGS] &= ~value;  // bk_clear_flags
  }
}

// ==================================================================================
// Device-resident instruction ingress (rust/src/step_sim_numpy.rs:233-275 `submit_instructions` + the host half of
// Env::place_order / cancel_order / modify_order, env.rs:166-219, for EVERY book in one launch, no host in the loop).
// The six SoA arrays live in device memory (an agent layer running on the GPU wrote them); the instructions of book b are
// elements [off[b], off[b + 1]).  One wave per MARKET (per book when assets == 1) walks its books' batches 64 elements at
// a time:
//   * create_order's tick check (orderbook.rs:367-382): the first new order whose price is not a multiple of the book's
//     tick size stops THAT BOOK's batch - earlier elements stay created and queued, later ones are not looked at
//     (`.collect::<Result<Vec<_>, _>>()` short-circuits, step_sim_numpy.rs:255-268); status = {code, elements applied};
//   * ids: dense per book, in element order - base + exclusive prefix count of the new orders (a ballot + mbcnt);
//   * the events are appended to the market's queue in element (then asset) order, the position = queue length +
//     exclusive prefix count of the event-producing elements; action 0 / unknown actions produce nothing (:266);
//   * the immutable half of every new order and its initial order-log entry (status New, arr_time = now) are written
//     for the readers (bk_get_orders ...).
// BK_ACTION_MODIFY (0x80000003, outside the numpy API's 0 / 1 / 2; the host entry takes the same code) = Env::modify_order
// (env.rs:208-219): side bit 1 = has price, bit 2 = has volume.  Every other action is a no-op (:266).
// ==================================================================================
struct IngestArgs {
  const unsigned long long* off;  // [n_books + 1]
  const uint32_t* action;
  const uint8_t* side;
  const uint32_t* vol;
  const uint32_t* trader;
  const uint32_t* price;
  const unsigned long long* order_id;
  unsigned long long* out_ids;    // nullable: id of the order an element created, else u64::MAX
  uint32_t* status;               // nullable: [2 * n_books] {code (bk_status), elements of the book's batch applied}
  uint4* q;                       // [n_markets][qcap] event records (HostEvent layout)
  uint32_t* qlen;                 // [n_markets]
  uint32_t qcap;
  uint4* dorders;                 // [n_books][log_cap][2]: {start_vol, trader, price, bid} {create_lo, create_hi, 0, 0}
  uint32_t* mods_flag;            // nullable: a word in mapped HOST memory, set to 1 when a BK_ACTION_MODIFY element is seen (a hint
                                  // for the host's choice of k_step_events instantiation: sticky, may lag by a step)
};
constexpr uint32_t ING_ACTION_MODIFY = 0x80000003u;  // == BK_ACTION_MODIFY (include/bourse_amd.h)
constexpr uint32_t ING_OK = 0u, ING_PRICE = 1u, ING_CAPACITY = 3u;  // == BK_OK / BK_PRICE_NOT_TICK_MULTIPLE / BK_CAPACITY

__global__ __launch_bounds__(64) void k_ingest(DevArgs a, IngestArgs g) {
  const int lane = threadIdx.x;
  const uint32_t mkt = blockIdx.x, M = a.assets;
  uint32_t qn = g.qlen[mkt];
  uint4* q = g.q + (size_t)mkt * g.qcap;
  for (uint32_t asset = 0; asset < M; ++asset) {
    const uint32_t book = mkt * M + asset;
    uint32_t* hdr = a.state + (size_t)book * a.state_stride;
    const uint32_t tick = a.asset_tick[asset];
    uint32_t next_id = rfl(hdr[H_NEXT_ID]);
    const uint32_t t_lo = rfl(hdr[H_T_LO]), t_hi = rfl(hdr[H_T_HI]);  // now: the book's clock (Env::place_order stamps it)
    const unsigned long long lo = g.off[book], hi = g.off[book + 1];
    uint32_t code = ING_OK;
    unsigned long long applied = 0;
    for (unsigned long long base = lo; base < hi && code == ING_OK; base += 64) {
      const unsigned long long i = base + (uint32_t)lane;
      const bool in = i < hi;
      const uint32_t act = in ? g.action[i] : 0u, sd = in ? g.side[i] : 0u, vol = in ? g.vol[i] : 0u;
      const uint32_t price = in ? g.price[i] : 0u, trader = in ? g.trader[i] : 0u;
      const unsigned long long oid = in ? g.order_id[i] : 0ull;
      const bool is_new = act == 1u, is_ev = act == 1u || act == 2u || act == ING_ACTION_MODIFY;
      if (g.mods_flag && __ballot(in && act == ING_ACTION_MODIFY) && lane == 0) *g.mods_flag = 1u;
      // the first bad price of the chunk stops the book's batch (earlier elements are applied)
      const uint64_t badm = __ballot(is_new && price % tick != 0u);
      uint32_t cut = badm ? (uint32_t)__builtin_ctzll(badm) : 64u;
      if (badm) code = ING_PRICE;
      // ... so does a full queue / an exhausted id space (cannot occur in the reference: BK_CAPACITY)
      uint64_t evm = __ballot(in && is_ev && (uint32_t)lane < cut);
      const uint32_t room = g.qcap - qn;
      if ((uint32_t)__builtin_popcountll(evm) > room) {
        uint64_t m = evm;  // the events that do not fit: all but the first `room`
        for (uint32_t r = 0; r < room; ++r) m &= m - 1ull;
        cut = (uint32_t)__builtin_ctzll(m);  // (before the bad price, if there is one: the earlier failure is reported)
        code = ING_CAPACITY;
        evm = __ballot(in && is_ev && (uint32_t)lane < cut);
      }
      if ((uint64_t)next_id + 64u >= 0xFFFFFFFFull) {  // ids are u32 on the device: refuse the chunk
        cut = 0;
        code = ING_CAPACITY;
        evm = 0;
      }
      const bool valid = in && (uint32_t)lane < cut;
      const uint64_t newm = __ballot(valid && is_new);
      const uint32_t rank_ev = __builtin_amdgcn_mbcnt_hi((uint32_t)(evm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)evm, 0u));
      const uint32_t rank_new = __builtin_amdgcn_mbcnt_hi((uint32_t)(newm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)newm, 0u));
      const uint32_t id = next_id + rank_new;
      if (valid && is_ev) {
        uint4 e;
        const uint32_t idc = oid > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)oid;
        if (is_new) {
          e = make_uint4(0u | ((sd & 1u) << 8) | (asset << 16), id, price, vol);
        } else if (act == 2u) {
          e = make_uint4(1u | (asset << 16), idc, 0u, 0u);
        } else {
          e = make_uint4(2u | ((sd & 2u) ? 1u << 9 : 0u) | ((sd & 4u) ? 1u << 10 : 0u) | (asset << 16), idc, price, vol);
        }
        q[qn + rank_ev] = e;
      }
      if (valid && is_new && id < a.log_cap) {
        uint4* d = g.dorders + ((size_t)book * a.log_cap + id) * 2;
        d[0] = make_uint4(vol, trader, price, sd & 1u);
        d[1] = make_uint4(t_lo, t_hi, 0u, 0u);
        // initial order-log entry: status New, nothing traded, provisional key (price, 0) (orderbook.rs:388-391)
        uint4* l = reinterpret_cast<uint4*>(a.order_log + (size_t)book * a.log_cap + id);
        l[0] = make_uint4(0u, vol, price, price);
        l[1] = make_uint4(t_lo, t_hi, 0xFFFFFFFFu, 0xFFFFFFFFu);
        l[2] = make_uint4(0u, 0u, 0u, 0u);
      }
      if (valid && g.out_ids) g.out_ids[i] = is_new ? (unsigned long long)id : ~0ull;
      next_id += (uint32_t)__builtin_popcountll(newm);
      qn += (uint32_t)__builtin_popcountll(evm);
      applied += cut < 64u ? cut : (hi - base < 64ull ? hi - base : 64ull);
    }
    if (lane == 0) {
      hdr[H_NEXT_ID] = next_id;
      if (g.status) {
        g.status[2 * book] = code;
        g.status[2 * book + 1] = applied > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)applied;
      }
    }
  }
  if (lane == 0) g.qlen[mkt] = qn;
}

// OR of every book's sticky flags and the largest number of retained trade records: what a strict caller polls after a
// step (two words instead of n_books flag words; the per-book array is only fetched when a new bit shows up)
__global__ void k_flags_summary(const uint32_t* state, uint32_t stride, uint32_t n_books, uint32_t* out /* [2] */) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t f = 0, r = 0;
  if (b < n_books) {
    const uint32_t* h = state + (size_t)b * stride;
    f = h[H_FLAGS];
    const uint64_t n = mk64(h[H_TRADES_LO], h[H_TRADES_HI]) - mk64(h[H_TRADE_BASE_LO], h[H_TRADE_BASE_HI]);
    r = n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
  }
  for (int o = 32; o > 0; o >>= 1) {
    f |= (uint32_t)__shfl_xor((int)f, o);
    r = max(r, (uint32_t)__shfl_xor((int)r, o));
  }
  if ((threadIdx.x & 63) == 0 && (f | r)) {
    if (f) atomicOr(&out[0], f);
    atomicMax(&out[1], r);
  }
}

// holds a stream for `ticks` x 10 ns (s_memrealtime runs at 100 MHz): the timed stagger of the split pipelines' parts
// header word(s) `word` (.. word + n_words - 1, n_words = 1 or 2) of every book, gathered into a contiguous array of u64:
// a strided 2-D copy of 65 536 eight-byte rows takes milliseconds, this kernel + one contiguous copy ~0.1 ms
__global__ void k_gather_header(const uint32_t* state, uint32_t stride, uint32_t word, uint32_t n_words, uint32_t n_books,
                                unsigned long long* out) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_books) return;
  const uint32_t* h = state + (size_t)b * stride + word;
  out[b] = n_words == 2 ? mk64(h[0], h[1]) : (unsigned long long)h[0];
}

__global__ void k_delay(uint32_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

struct DevStats {  // == bk_stats
  unsigned long long n_books, sum_trade_vol, sum_trades, sum_events, sum_bid_vol, sum_ask_vol;
  uint32_t min_bid, max_bid, min_ask, max_ask;
};

// per-shard market statistics: one 64-byte record (the unit all-gathered across GPUs)
__global__ void k_stats(const uint32_t* state, uint32_t stride, const uint32_t* l2_last, uint32_t W, uint32_t n_books,
                        DevStats* out) {
  unsigned long long tv = 0, tr = 0, ev = 0, bv = 0, av = 0;
  uint32_t mnb = 0xFFFFFFFFu, mxb = 0, mna = 0xFFFFFFFFu, mxa = 0;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_books; b += gridDim.x * blockDim.x) {
    const uint32_t* h = state + (size_t)b * stride;
    const uint32_t* l = l2_last + (size_t)b * W;
    tv += l[0];
    tr += mk64(h[H_TRADES_LO], h[H_TRADES_HI]);
    ev += mk64(h[H_EVENTS_LO], h[H_EVENTS_HI]);
    bv += l[4];
    av += l[3];
    if (l[4] != 0 || l[5 + 1] != 0) {  // bid side non-empty (touch level holds >= 1 order)
      mnb = min(mnb, l[1]);
      mxb = max(mxb, l[1]);
    }
    if (l[3] != 0 || l[5 + 3] != 0) {
      mna = min(mna, l[2]);
      mxa = max(mxa, l[2]);
    }
  }
  // one set of atomics per WAVE (nine contended atomics per thread made this kernel 114 us for 65 536 books, and it sits
  // between two launches whenever the stats are gathered)
  for (int o = 32; o > 0; o >>= 1) {
    tv += __shfl_xor(tv, o);
    tr += __shfl_xor(tr, o);
    ev += __shfl_xor(ev, o);
    bv += __shfl_xor(bv, o);
    av += __shfl_xor(av, o);
    mnb = min(mnb, (uint32_t)__shfl_xor((int)mnb, o));
    mxb = max(mxb, (uint32_t)__shfl_xor((int)mxb, o));
    mna = min(mna, (uint32_t)__shfl_xor((int)mna, o));
    mxa = max(mxa, (uint32_t)__shfl_xor((int)mxa, o));
  }
  if ((threadIdx.x & 63) != 0) return;
  atomicAdd(&out->sum_trade_vol, tv);
  atomicAdd(&out->sum_trades, tr);
  atomicAdd(&out->sum_events, ev);
  atomicAdd(&out->sum_bid_vol, bv);
  atomicAdd(&out->sum_ask_vol, av);
  atomicMin(&out->min_bid, mnb);
  atomicMax(&out->max_bid, mxb);
  atomicMin(&out->min_ask, mna);
  atomicMax(&out->max_ask, mxa);
  if (blockIdx.x == 0 && threadIdx.x == 0) out->n_books = n_books;
}

// ---- trade egress: compact every book's retained records into one dense stream ----
struct OutTrade {  // == bk_trade (40 B), the record layout of the C ABI
  uint64_t t;
  uint32_t side_is_bid, price, vol, reserved;
  uint64_t active, passive;
};
// counts[b] = records retained for book b (since trade_base, at most trade_cap); then an exclusive scan by ONE
// workgroup: off[b] = sum of counts[0..b), off[n_books] = total.  B <= a few 10^5: a single 1024-thread block suffices.
__global__ __launch_bounds__(1024) void k_trade_scan(const uint32_t* state, uint32_t stride, uint32_t n_books,
                                                     uint32_t trade_cap, unsigned long long* off) {
  __shared__ unsigned long long part[1024];
  __shared__ unsigned long long carry;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_books; base += 1024) {
    const uint32_t b = base + tid;
    unsigned long long c = 0;
    if (b < n_books) {
      const uint32_t* h = state + (size_t)b * stride;
      const uint64_t n = mk64(h[H_TRADES_LO], h[H_TRADES_HI]) - mk64(h[H_TRADE_BASE_LO], h[H_TRADE_BASE_HI]);
      c = n < trade_cap ? n : trade_cap;
    }
    part[tid] = c;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
      const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    if (b < n_books) off[b] = carry + part[tid] - c;
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) off[n_books] = carry;
}
// one wave per book: copy its retained records to dense[off[b] ...] in the C ABI layout, then mark them consumed
// (trade_base = n_trades, as bk_clear_trades does)
__global__ __launch_bounds__(256) void k_trade_gather(uint32_t* state, uint32_t stride, uint32_t n_books,
                                                      uint32_t trade_cap, const DevTrade* trades,
                                                      const unsigned long long* off, OutTrade* dense) {
  const int lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_books) return;
  uint32_t* h = state + (size_t)b * stride;
  const uint32_t cnt = (uint32_t)(off[b + 1] - off[b]);
  const DevTrade* src = trades + (size_t)b * trade_cap;
  OutTrade* dst = dense + off[b];
  for (uint32_t i = lane; i < cnt; i += 64) {
    const DevTrade d = src[i];
    OutTrade o;
    o.t = mk64(d.t_lo, d.t_hi);
    o.side_is_bid = d.side_is_bid;
    o.price = d.price;
    o.vol = d.vol;
    o.reserved = 0;
    o.active = d.active;
    o.passive = d.passive;
    dst[i] = o;
  }
  if (lane == 0) {
    h[H_TRADE_BASE_LO] = h[H_TRADES_LO];
    h[H_TRADE_BASE_HI] = h[H_TRADES_HI];
  }
}

// self-test of the DPP reductions (used by tests on the GPU box)
__global__ void k_selftest_reduce(const uint32_t* in, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const uint32_t x = in[blockIdx.x * 64 + lane];
  const uint32_t mn = wave_umin(x), mx = wave_umax(x), sm = wave_add(x);
  const uint32_t s = sel(0xF0F0F0F0F0F0F0F0ull, 1u, 2u);
  if (lane == 0) {
    out[blockIdx.x * 4 + 0] = mn;
    out[blockIdx.x * 4 + 1] = mx;
    out[blockIdx.x * 4 + 2] = sm;
  }
  const uint32_t ssum = wave_add(s);
  if (lane == 0) out[blockIdx.x * 4 + 3] = ssum;
}
#endif  // BOURSE_AMD_FSM_UNIT

}  // namespace bkd
