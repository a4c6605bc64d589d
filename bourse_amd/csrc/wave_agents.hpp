// wave_agents.hpp — k_agents_wave: the RNG-serial half of a book-step (RandomAgents::update for every group +
// the Fisher-Yates shuffle of Env::step) with ONE WAVE PER BOOK and the book's RNG stream decoded 64 draws at a time.
//
// Why: the lane-per-book k_agents_fsm walks ~600 dependent draws per book-step at the ~6.5 clocks a lone wave gets per
// instruction — a 140 us floor per step however few books there are, which is what strong scaling over
// 8 GPUs (8 192 books per GPU) runs into.  Here the stream itself is parallel:
//
//   * xoroshiro128** is F2-linear: lane j keeps the generator state 4 j draws ahead of the block
//     start, produces its 4 draws independently, and the 64 lane states move to the next 256-draw block by T^256,
//     applied as the XOR of 32 entries of a nibble-indexed table (8 KB of LDS per workgroup; host_math.hpp builds it,
//     tests/cpp/host_math_test.cpp checks it against single steps).
//   * consumption order is data dependent (rejection sampling; an agent that holds an Active order cancels instead of
//     drawing side / tick / vol: ref crates/step_sim/src/agents/random_agent.rs:91-101).  Per 64-draw window the wave
//     builds ballot masks (activity hit, accept for range 2 / tick range / vol range), every lane p precomputes where
//     the stream would stand after a placement STARTING at p (three first-set-bit searches over a 128-draw look-ahead)
//     and what that order would be; a scalar walk then only visits the activity HITS (s_ff1 skips the runs of
//     inactive agents) and consumes one cancellation or one whole placement per iteration.
//   * the shuffle's acceptance test depends on how many earlier draws were accepted (the range i + 1 shrinks,
//     env.rs:121 / rand SliceRandom::shuffle): a fixed-point iteration over the window's accept mask — exact, because
//     lane 0's count is always right and correctness spreads left to right — then the swaps in order.
//   A placement whose draws run past the look-ahead (probability ~2^-60 at the shipped 64; forced in tests by a tiny
//   look-ahead) is resolved draw by draw on the scalar path.
//
// Output: the same per-book step batch k_agents_fsm writes (book_device.hpp "step batch layout"), consumed by
// k_step_batch unchanged.  tools/wave_decode_proto.py is the plain-Python model of this file (CPU test).
#pragma once
#include "book_device.hpp"

namespace bkd {

constexpr uint32_t WV_K = 4;                   // draws per lane per block
constexpr uint32_t WV_BLOCK = 64 * WV_K;       // 256 draws
constexpr uint32_t WV_RING = 2 * WV_BLOCK;     // generated draws kept in LDS
constexpr uint32_t WC_HDR = 64;                // cache record: 64 header dwords (lane i holds dword i) + 64 x uint4 states
constexpr uint32_t WC_STRIDE = WC_HDR + 256;   // dwords per book
constexpr uint32_t WC_MAGIC = 0x45564157u;     // "WAVE"
enum WcHdr : int { WC_S0_LO = 0, WC_S0_HI, WC_S1_LO, WC_S1_HI, WC_OFF, WC_TAG };
constexpr uint32_t WV_NONE = 0xFFFFu;          // "placement not resolvable inside the look-ahead"
constexpr uint32_t WV_ACTED = 0x10000u;        // marker bit of a lane's event word (above the 16 bits that are stored)

struct WaveArgs {
  const uint4* jt_block;  // T^256: 32 x 16 entries
  const uint4* jt_lane;   // T^(4 << b), b = 0..5: lane offsets when the cached lane states do not match the book's RNG
  uint32_t* wcache;       // [n_books][WC_STRIDE]: lane states of the block the book's RNG stands in + offset
  uint32_t lookahead;     // draws beyond the window a placement may use on the vector path (1..64)
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// T^n applied to the lane's state {s0_lo, s0_hi, s1_lo, s1_hi}: XOR of one table entry per state nibble
__device__ __forceinline__ uint4 wv_jump(const uint4* __restrict__ tab, uint4 s) {
  const uint32_t w[4] = {s.x, s.y, s.z, s.w};
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {  // two entries per step: one three-input XOR (v_bitop3_b32) per word instead of two XORs
      const uint32_t i0 = (w[d] >> (4 * k)) & 15u, i1 = (w[d] >> (4 * k + 4)) & 15u;
      const uint4 e0 = tab[(d * 8 + k) * 16 + i0], e1 = tab[(d * 8 + k + 1) * 16 + i1];
      acc.x = xor3(acc.x, e0.x, e1.x);
      acc.y = xor3(acc.y, e0.y, e1.y);
      acc.z = xor3(acc.z, e0.z, e1.z);
      acc.w = xor3(acc.w, e0.w, e1.w);
    }
  }
  return acc;
}

// index of the first set bit ABOVE position i in the 128-bit mask w3:w2:w1:w0 (all per lane); >= 128 if there is none.
// Branch-free: word j keeps its bits at positions >= n = i + 1 - the mask is all ones shifted left by n - 32 j clamped to
// [0, 32] (a 64-bit shift, so that 32 clears the word) - and the answer is the minimum over the four words of
// v_ffbl_b32 | 32 j (all ones for an empty word, with or without the OR).  ~25 vector instructions (round 2's form, through
// 64-bit halves and the compiler's guarded count-trailing-zeros: ~45).
__device__ __forceinline__ uint32_t first_above(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i) {
  const int32_t n = (int32_t)i + 1;
  auto keep = [](uint32_t w, int32_t t) {  // bits of w at positions >= t (t clamped to 0..32)
    const uint32_t sh = (uint32_t)max(0, min(t, 32));  // (v_med3_i32)
    return w & (uint32_t)(~0ull << sh);
  };
  uint32_t f0, f1, f2, f3;
  asm("v_ffbl_b32 %0, %1" : "=v"(f0) : "v"(keep(w0, n)));
  asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(keep(w1, n - 32)));
  asm("v_ffbl_b32 %0, %1" : "=v"(f2) : "v"(keep(w2, n - 64)));
  asm("v_ffbl_b32 %0, %1" : "=v"(f3) : "v"(keep(w3, n - 96)));
  return min(min(f0, f1 | 32u), min(f2 | 64u, f3 | 96u));
}

// The decode's searches (agents() below) use a cheaper form that only looks 64 positions ahead: the first set bit in
// [i + 1, min(i + 65, 128)), else a value >= 128 ("none").  A set bit further away is reported as none - the caller then
// treats the placement as not resolvable inside the look-ahead and resolves it draw by draw on the scalar path, so the
// result is exact either way (two accepted draws of one sample 64 draws apart: p < 2^-37 at the worst acceptance rate of
// 1/2).  One funnel shift of the 128-bit mask + one 64-bit first-set-bit search: ~19 vector instructions instead of ~45.
// Contract: for i < 127 the result is exact or >= 128; for i >= 127 (a chained search behind a "none") it is SOME value
// > i, i.e. still >= 128 - which is all the caller tests (q < lim <= 128).  That lets the i >= 127 guard and the
// "nothing found" select of round 3's form go: the count of trailing zeros is min(ffbl(w0), ffbl(w1) | 32), all ones when
// the window is empty (v_ffbl_b32 of 0 is -1, and -1 | 32 stays -1), and n + all ones wraps to i, caught by one compare.
// min(count of trailing zeros of x, cap), cap for x = 0 (per lane; v_ffbl_b32 of 0 is all ones, and all ones | 32 too)
__device__ __forceinline__ uint32_t ctz64_cap(uint64_t x, uint32_t cap) {
  uint32_t f0, f1;
  asm("v_ffbl_b32 %0, %1" : "=v"(f0) : "v"((uint32_t)x));
  asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"((uint32_t)(x >> 32)));
  return min(min(f0, f1 | 32u), cap);
}
__device__ __forceinline__ uint32_t first_above_near(uint64_t lo, uint64_t hi, uint32_t i) {
  const uint32_t n = i + 1u;
  const uint32_t sh = n & 63u;
  const bool in_lo = n < 64u;
  const uint64_t a = in_lo ? lo : hi;          // the word n lies in
  const uint64_t b1 = in_lo ? (hi << 1) : 0ull;  // the word above it, pre-shifted so that the funnel never shifts by 64
  const uint64_t w = (a >> sh) | (b1 << (63u - sh));
  const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  // (the instruction itself: the compiler's count-trailing-zeros guards the zero input with a compare and a select of its own)
  uint32_t f0, f1;
  asm("v_ffbl_b32 %0, %1" : "=v"(f0) : "v"(w0));
  asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(w1));
  const uint32_t c = min(f0, f1 | 32u);
  return c == 0xFFFFFFFFu ? 256u : n + c;
}

// The scalar walk over a window's activity hits (WaveDecoder::agents): shared text of its three forms.  WV_WALK_BEGIN ..
// [w = pack[p]; the live test, leaving SCC = agent holds an Active order] .. WV_WALK_REST
#define WV_WALK_BEGIN                                 \
  "s_mov_b32 m0, %[p]\n\t"                            \
  "s_or_b32 %[ag], %[ag], 0x10000\n\t"                \
  "s_or_b32 %[gm], %[gend], 0x10000\n\t"              \
  "1:\n\t"
#define WV_WALK_REST                                                                                              \
  "s_cbranch_scc0 2f\n\t"                                                                                         \
  "v_writelane_b32 %[agw], %[ag], m0\n\t"     /* holds an Active order: its cancellation (agent | marker) */       \
  "s_and_b32 %[t0], %[w], 0x7f\n\t"           /* distance to the next hit */                                       \
  "s_add_u32 %[ag], %[ag], %[t0]\n\t"                                                                             \
  "s_cmp_ge_u32 %[ag], %[gm]\n\t"                                                                                 \
  "s_cbranch_scc1 5f\n\t"                                                                                         \
  "s_add_u32 m0, m0, %[t0]\n\t"                                                                                   \
  "s_cmp_lt_u32 m0, 64\n\t"                                                                                       \
  "s_cbranch_scc1 1b\n\t"                                                                                         \
  "s_mov_b32 %[st], 0\n\t"                                                                                        \
  "s_branch 9f\n\t"                                                                                               \
  "2:\n\t"                                                                                                        \
  "s_cmp_lt_i32 %[w], 0\n\t"                  /* placement not resolvable in the look-ahead */                     \
  "s_cbranch_scc1 8f\n\t"                                                                                         \
  "s_or_b32 %[t0], %[ag], 0x8000\n\t"                                                                             \
  "v_writelane_b32 %[agw], %[t0], m0\n\t"                                                                         \
  "s_bfe_u32 %[t0], %[w], 0x70017\n\t"        /* agents the placement's path covers */                             \
  "s_add_u32 %[ag], %[ag], %[t0]\n\t"                                                                             \
  "s_cmp_ge_u32 %[ag], %[gm]\n\t"                                                                                 \
  "s_cbranch_scc1 6f\n\t"                                                                                         \
  "s_bfe_u32 m0, %[w], 0x80007\n\t"           /* next position */                                                  \
  "s_cmp_lt_u32 m0, 64\n\t"                                                                                       \
  "s_cbranch_scc1 1b\n\t"                                                                                         \
  "s_mov_b32 %[st], 0\n\t"                                                                                        \
  "s_branch 9f\n\t"                                                                                               \
  "5:\n\t"                                     /* group end behind a cancellation */                                \
  "s_sub_u32 %[ag], %[ag], %[t0]\n\t"                                                                             \
  "s_sub_u32 %[t0], %[gm], %[ag]\n\t"                                                                             \
  "s_add_u32 m0, m0, %[t0]\n\t"                                                                                   \
  "s_mov_b32 %[ag], %[gm]\n\t"                                                                                    \
  "s_mov_b32 %[st], 1\n\t"                                                                                        \
  "s_branch 9f\n\t"                                                                                               \
  "6:\n\t"                                     /* group end behind a placement: from f */                           \
  "s_sub_u32 %[ag], %[ag], %[t0]\n\t"                                                                             \
  "s_bfe_u32 %[t1], %[w], 0x8000f\n\t"                                                                            \
  "s_sub_u32 %[t0], %[gm], %[ag]\n\t"                                                                             \
  "s_add_u32 %[t0], %[t0], %[t1]\n\t"                                                                             \
  "s_sub_u32 m0, %[t0], 1\n\t"                                                                                    \
  "s_mov_b32 %[ag], %[gm]\n\t"                                                                                    \
  "s_mov_b32 %[st], 1\n\t"                                                                                        \
  "s_branch 9f\n\t"                                                                                               \
  "8:\n\t"                                                                                                        \
  "s_mov_b32 %[st], 2\n\t"                                                                                        \
  "9:\n\t"                                                                                                        \
  "s_mov_b32 %[p], m0\n\t"                                                                                        \
  "s_and_b32 %[ag], %[ag], 0xffff"

// The decoder state of one wave (= one book).  All pointers are wave-uniform; `pv` (new orders {price, vol} by pool
// slot) is global memory for the split pipeline (the step batch) and LDS for the fused kernel.
// MASKS: also keep the placing / side masks of the step's new orders (the fused kernel reads them; the split form's
// k_step_batch rebuilds them from the event words, so its decode does not pay two LDS atomics per placement for them).
#ifndef BOURSE_AMD_GEN_FAKE
#define BOURSE_AMD_GEN_FAKE 0
#endif
template <int R, bool MASKS = true>
struct WaveDecoder {
  const uint4* tab;    // LDS: T^256 table
  uint32_t* ring;      // LDS: the last WV_RING generated draws, ring[q & (WV_RING - 1)] = draw q of this launch's stream
  uint16_t* evl;       // LDS: this step's event list (event words, book_device.hpp EV_*)
  uint32_t* pm;        // LDS: placing-agents mask, 2 R words
  uint32_t* sm;        // LDS: bid-side mask of the placements, 2 R words
  uint2* pv;           // new orders by slot
  uint16_t* jarr;      // LDS: shuffle targets, jarr[i] = gen_range(0..i+1) of step i (64 R entries)
  uint4* wmask;        // LDS (R <= 2): 128 x 128-bit "steps that target this position" masks of the shuffle resolution
  uint32_t* co = nullptr;      // LDS (R > 2, optional): 64 R words {count << 16 | offset} of the bucketed resolution; without
  uint16_t* bucket = nullptr;  // them (+ 64 R u16) the swaps of a pool of more than 128 slots run one by one
  uint4* wmask2 = nullptr;     // LDS (R > 2, optional; needs co for the targets): 256 x 128-bit masks = 4 KB - up to 256 events are
                               // then resolved by the small pools' mask rule in TWO rounds instead of through the buckets
  uint4* wcs;          // global: the 64 lane states of the cache record
  int lane;
  uint4 cs;            // this lane's chunk-start state in the last generated block
  bool was_cached;     // load_cache found the record valid: finish() then stores the lane states only if the block changed
  uint32_t gen_end;    // draws generated so far (stream origin = start of the cached block)
  uint32_t pos;        // stream position: draws consumed so far
  BK_STAMP_FIELD

  // cached lane states valid for the book's RNG state (s0l..s1h)?  else lane j = T^(4 j) of it, by doubling
  __device__ __forceinline__ void load_cache(const uint32_t* wc, uint32_t s0l, uint32_t s0h, uint32_t s1l, uint32_t s1h,
                                             const uint4* jt_lane) {
    // (the record's six header words through the scalar cache: it was written by the previous launch, nothing in this
    // one writes it before finish())
    // (pools of <= 128 slots; the larger pools' kernels lose more to the extra live scalars than the lane reads cost:
    // book_device.hpp load_state_raw)
    cs = reinterpret_cast<const uint4*>(wc + WC_HDR)[lane];
    bool cached;
    if constexpr (R <= 2) {
      const bk_u32x8 cwc = sload_x8(wc);  // (explicit s_load: finish() stores these words at the end of the launch)
      pos = cwc[WC_OFF];
      cached = cwc[WC_TAG] == WC_MAGIC && cwc[WC_S0_LO] == s0l && cwc[WC_S0_HI] == s0h && cwc[WC_S1_LO] == s1l &&
               cwc[WC_S1_HI] == s1h && pos < WV_BLOCK;
    } else {
      const uint32_t wch = lane < 8 ? wc[lane] : 0u;  // (six words of the 256-byte header are in use: one 32-byte sector)
      pos = rdl(wch, WC_OFF);
      cached = rdl(wch, WC_TAG) == WC_MAGIC && rdl(wch, WC_S0_LO) == s0l && rdl(wch, WC_S0_HI) == s0h &&
               rdl(wch, WC_S1_LO) == s1l && rdl(wch, WC_S1_HI) == s1h && pos < WV_BLOCK;
    }
    if (!cached) {  // another pipeline (or a restore / a fresh env) moved the RNG
      cs = make_uint4(s0l, s0h, s1l, s1h);
      for (int b = 0; b < 6; ++b) {
        const uint4 j = wv_jump(jt_lane + b * 512, cs);
        const bool take = (lane >> b) & 1;
        cs.x = take ? j.x : cs.x;
        cs.y = take ? j.y : cs.y;
        cs.z = take ? j.z : cs.z;
        cs.w = take ? j.w : cs.w;
      }
      pos = 0;
    }
    was_cached = cached;
    gen_end = 0;
  }

  __device__ __forceinline__ void gen_block() {  // next 256 draws into the ring
    if (gen_end != 0) {
      // the stream position may still lie in the block being left when the launch ends (look-ahead): its chunk-start
      // states go to the cache record now (a fire-and-forget 1 KB store instead of four live registers)
      wcs[lane] = cs;
#if BOURSE_AMD_GEN_FAKE  // TIMING EXPERIMENT (results are wrong): what would the decode cost if generation were free?
      cs.x = cs.x * 0x9E3779B1u + 0x7F4A7C15u; cs.y ^= cs.x >> 7; cs.z += cs.y; cs.w ^= cs.z << 3;
#else
      cs = wv_jump(tab, cs);
#endif
    }
    RngLane t{cs.x, cs.y, cs.z, cs.w};
    uint4 x;
#if BOURSE_AMD_GEN_FAKE >= 2
    x.x = cs.x * 0x85EBCA6Bu; x.y = (cs.y ^ cs.x) * 0xC2B2AE35u; x.z = (cs.z + cs.x) * 0x27D4EB2Fu; x.w = (cs.w ^ cs.y) * 0x165667B1u;
    x.x ^= x.x >> 15; x.y ^= x.y >> 13; x.z ^= x.z >> 16; x.w ^= x.w >> 14;
#else
    x.x = t.next_u32();
    x.y = t.next_u32();
    x.z = t.next_u32();
    x.w = t.next_u32();
#endif
    reinterpret_cast<uint4*>(ring)[((gen_end >> 2) + lane) & (WV_RING / 4 - 1)] = x;
    gen_end += WV_BLOCK;
    wave_sync();
  }
  __device__ __forceinline__ void ensure(uint32_t upto) {
    while (gen_end < upto) gen_block();
  }

  // ================= agents.update: groups in declaration order (crates/macros/src/lib.rs:57-73) =================
  // livev: lane live_base + w holds bits [32 w, 32 w + 32) of the pool's live mask.  Returns the number of events.
  // lv0 / lv1: (R <= 2) the pool's live masks as two scalar pairs for the walk
  __device__ __forceinline__ uint32_t agents(const DevArgs& a, uint32_t lim, uint32_t livev, uint32_t live_base, uint64_t lv0,
                                             uint64_t lv1) {
    uint32_t n_ev = 0, ag = 0, gbase = 0;
    uint32_t lblk = 0xFFFFFFFFu, lw = 0;  // (R > 2) the cached live word of the walk and which 32 agents it covers
    for (uint32_t g = 0; g < a.n_groups; ++g) {
      const Group G = a.groups[g];
      const uint32_t gend = gbase + G.n;
      gbase = gend;
      while (ag < gend) {
        // ---- window [w0, w0 + 64) of the stream + 64 draws of look-ahead
        const uint32_t w0 = pos & ~63u;
        ensure(w0 + 128u);
        BK_STAMP(*this, 2, 0, lane);  // (diagnostic build) generation
        const uint32_t xc = ring[(w0 + lane) & (WV_RING - 1)], xn = ring[(w0 + 64u + lane) & (WV_RING - 1)];
        // p = gen::<f32>() < activity_rate (random_agent.rs:91-93) as an integer threshold (host_math.hpp)
        const uint64_t H = __ballot((xc >> 8) < G.thr);
        // UniformInt<u32>::sample_single: accept iff lo32(x * range) <= zone.  range 2: bit 30 clear.
        const uint64_t A2l = __ballot((xc & 0x40000000u) == 0u), A2h = __ballot((xn & 0x40000000u) == 0u);
        const uint64_t ATl = __ballot(xc * G.tick_rng <= G.tick_zone), ATh = __ballot(xn * G.tick_rng <= G.tick_zone);
        const uint64_t AVl = __ballot(xc * G.vol_rng <= G.vol_zone), AVh = __ballot(xn * G.vol_rng <= G.vol_zone);
        // lane p: the placement that starts if the activity draw at p hits and the agent holds no Active order:
        // side = [Ask, Bid].choose, tick, vol in that order (random_agent.rs:99-101)
        const uint32_t q1 = first_above_near(A2l, A2h, (uint32_t)lane);
        const uint32_t q2 = first_above_near(ATl, ATh, q1);
        const uint32_t q3 = first_above_near(AVl, AVh, q2);
        const uint32_t fpos = q3 < lim ? q3 + 1u : WV_NONE;
        const uint32_t x1 = ring[(w0 + q1) & (WV_RING - 1)], x2 = ring[(w0 + q2) & (WV_RING - 1)],
                       x3 = ring[(w0 + q3) & (WV_RING - 1)];
        const uint32_t fside = x1 >> 31;  // hi32(x * 2): 0 = Ask, 1 = Bid
        const uint32_t fprice = (G.tick_lo + __umulhi(x2, G.tick_rng)) * G.tick_size;
        const uint32_t fvol = G.vol_lo + __umulhi(x3, G.vol_rng);

        // ---- per lane p, both continuations of the walk from a hit at p, packed in one word so that the scalar walk
        // needs a single v_readlane per hit: after a cancellation (or any one-draw outcome) the next hit is the first
        // set bit of H above p; after a placement the stream stands at f = fpos(p) and the next hit is the first set
        // bit of H at or above f (f itself when the placement leaves the window).  Inactive agents in between are
        // skipped by position arithmetic: one agent per activity draw.
        //   bits 0..6 DISTANCE to the next hit after a cancellation (1..64) | 7..14 next position after a placement (<= 128) |
        //   15..22 f | 23..29 agents consumed by the placement path | 31 placement not resolvable in the look-ahead
        uint32_t pack;
        {
          // (count of trailing zeros capped at "the window's end": an empty remainder counts all ones and takes the cap)
          const uint64_t hc = (uint32_t)lane < 63u ? (H >> (lane + 1)) : 0ull;
          const uint32_t nC = (uint32_t)lane + 1u + ctz64_cap(hc, 63u - (uint32_t)lane);
          const uint32_t fq = fpos < 64u ? fpos : 0u;
          const uint64_t hp = H >> fq;
          const uint32_t nP = fpos < 64u ? fq + ctz64_cap(hp, 64u - fq) : fpos;
          const uint32_t dP = fpos < 64u ? 1u + nP - fpos : 1u;
          pack = (nC - (uint32_t)lane) | (fpos == WV_NONE ? 0x80000000u : ((nP << 7) | (fpos << 15) | (dP << 23)));
        }
        // ---- scalar walk over the activity hits of this window (p < 64 and ag < gend on entry).  Each acting lane
        // gets its event word (slot, EV_NEW, marker bit 16) by v_writelane; list positions follow from the lane order
        // at the flush below (events of a window are in stream order = agent order).
        uint32_t p = pos - w0;
        uint32_t agw = 0;
        bool slow = false;
        BK_STAMP(*this, 2, 1, lane);  // the window's masks, searches, placements and continuations
        {
          // to the first hit at or after p (or the end of the window / group)
          const uint64_t m = H >> p;
          const uint32_t d = m ? (uint32_t)__builtin_ctzll(m) : 64u - p;
          if (ag + d >= gend || m == 0) {
            const uint32_t adv = d < gend - ag ? d : gend - ag;
            p += adv;
            ag += adv;
          } else {
            p += d;
            ag += d;
            // The walk itself, hand-written (this loop is ~half of the kernel's scalar instructions, and under eight
            // waves' contention for the scalar port every one of them is a link of the wave's chain; the compiled form
            // spends 24-26 per hit on loop-exit flags and re-materialised booleans, round 2's 19-21, this one 14-17).
            // The position lives in m0 (lane select of the v_readlane / v_writelane, no copy), the agent index carries the
            // event word's marker bit (bit 16: no OR per event; `gm` = the group's end with the same bit), the word's
            // first field is the DISTANCE to the next hit.  One iteration:
            //   w = pack[p]; live bit of agent ag; cancellation -> event word ag | ACTED into lane p, p += w[6:0];
            //   placement -> (w[31]: not resolvable -> slow) event word ag | NEW | ACTED, p = w[14:7], agents += w[29:23];
            //   the group ends before the next hit -> position of agent `gend`'s draw from the run start (p + 1 or f).
            if constexpr (R <= 2) {
              // pools of <= 128 slots: the live masks sit in two SGPR pairs.  A walk that stays inside one 64-agent half
              // (always, with groups of 64) tests ONE mask chosen up front - one scalar instruction per hit; else the
              // mask is chosen per hit (three)
              uint32_t st, w, t0, t1, gm;
              if ((ag >> 6) == ((gend - 1u) >> 6)) {
                const uint64_t lm = (ag & 64u) ? lv1 : lv0;
                // (the live test IN FRONT of the lane read: a scalar instruction issued right behind a vector write of a
                // scalar register waits ~16 clocks for it, whether it reads it or not - the branch and the cancellation's
                // v_writelane do not)
                asm volatile(WV_WALK_BEGIN
                             "s_bitcmp1_b64 %[lm], %[ag]\n\t"              /* (the bit index is ag[5:0]) */
                             "v_readlane_b32 %[w], %[pack], m0\n\t"
                             WV_WALK_REST
                             : [st] "=&s"(st), [w] "=&s"(w), [t0] "=&s"(t0), [t1] "=&s"(t1), [gm] "=&s"(gm), [p] "+s"(p),
                               [ag] "+s"(ag), [agw] "+v"(agw)
                             : [pack] "v"(pack), [lm] "s"(lm), [gend] "s"(gend)
                             : "scc", "m0", "memory");
              } else {
                uint64_t lm;
                asm volatile(WV_WALK_BEGIN
                             "s_bitcmp1_b32 %[ag], 6\n\t"                  /* agent 64..127: the second mask */
                             "s_cselect_b64 %[lm], %[lv1], %[lv0]\n\t"
                             "s_bitcmp1_b64 %[lm], %[ag]\n\t"
                             "v_readlane_b32 %[w], %[pack], m0\n\t"
                             WV_WALK_REST
                             : [st] "=&s"(st), [w] "=&s"(w), [t0] "=&s"(t0), [t1] "=&s"(t1), [gm] "=&s"(gm), [p] "+s"(p),
                               [ag] "+s"(ag), [agw] "+v"(agw), [lm] "=&s"(lm)
                             : [pack] "v"(pack), [lv0] "s"(lv0), [lv1] "s"(lv1), [gend] "s"(gend)
                             : "scc", "m0", "memory");
              }
              slow = st == 2u;
            } else {
              // larger pools: the live word of the current 32 agents is CACHED in a scalar register across hits, windows
              // and groups (`lblk` = which; the masks do not change during the decode) and re-read from the vector
              // register only when the walk enters another block - round 2 read it for every hit: one more v_readlane and
              // a second vector-to-scalar hand-over on each hit's chain
              uint32_t st, w, t0, t1, gm;
              asm volatile(WV_WALK_BEGIN
                           "s_bfe_u32 %[t0], %[ag], 0xb0005\n\t"          /* agent / 32 (without the marker bit) */
                           "s_cmp_lg_u32 %[t0], %[lblk]\n\t"
                           "s_cbranch_scc1 7f\n\t"
                           "3:\n\t"
                           "s_bitcmp1_b32 %[lw], %[ag]\n\t"               /* (the bit index is ag[4:0]) */
                           "v_readlane_b32 %[w], %[pack], m0\n\t"
                           WV_WALK_REST
                           "\n\t"
                           "s_branch 4f\n\t"
                           "7:\n\t"
                           "s_mov_b32 %[lblk], %[t0]\n\t"
                           "s_add_u32 %[t0], %[t0], %[lb]\n\t"
                           "v_readlane_b32 %[lw], %[livev], %[t0]\n\t"
                           "s_branch 3b\n\t"
                           "4:\n\t"
                           "s_nop 0"
                           : [st] "=&s"(st), [w] "=&s"(w), [t0] "=&s"(t0), [t1] "=&s"(t1), [gm] "=&s"(gm), [p] "+s"(p),
                             [ag] "+s"(ag), [agw] "+v"(agw), [lblk] "+s"(lblk), [lw] "+s"(lw)
                           : [pack] "v"(pack), [livev] "v"(livev), [lb] "s"(live_base), [gend] "s"(gend)
                           : "scc", "m0", "memory");
              slow = st == 2u;
            }
          }
        }
        BK_STAMP(*this, 2, 2, lane);  // the walk
        // ---- the window's events, one lane each: list entry and the new order's fields
        {
          const bool acted = (agw & WV_ACTED) != 0, placed = (agw & EV_NEW) != 0;
          const uint64_t am = __ballot(acted);
          const uint32_t evi =
              n_ev + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
          if (acted) evl[evi] = (uint16_t)(agw | (placed ? (fside << 14) : 0u));
          if (placed) {
            const uint32_t slot = agw & EV_SLOT;
            pv[slot] = make_uint2(fprice, fvol);
            if constexpr (MASKS) {
              atomicOr(&pm[slot >> 5], 1u << (slot & 31u));
              atomicOr(&sm[slot >> 5], fside << (slot & 31u));
            }
          }
          n_ev += (uint32_t)__builtin_popcountll(am);
        }
        if (slow) {
          // slow path: the placement's draws run past the look-ahead; resolve it draw by draw (generating as needed:
          // the window's draws are already in registers, the ring may move on)
          uint32_t q = w0 + p + 1u, val[3];
          const uint32_t rng3[3] = {2u, G.tick_rng, G.vol_rng}, zone3[3] = {0x7FFFFFFFu, G.tick_zone, G.vol_zone};
#pragma unroll
          for (int sgi = 0; sgi < 3; ++sgi) {
            for (;;) {
              ensure(q + 1u);
              const uint32_t x = rfl(ring[q & (WV_RING - 1)]);
              ++q;
              const uint64_t mm = (uint64_t)x * rng3[sgi];
              if ((uint32_t)mm <= zone3[sgi]) {
                val[sgi] = (uint32_t)(mm >> 32);
                break;
              }
            }
          }
          if (lane == 0) {
            evl[n_ev] = (uint16_t)(ag | EV_NEW | (val[0] << 14));
            pv[ag] = make_uint2((G.tick_lo + val[1]) * G.tick_size, G.vol_lo + val[2]);
            if constexpr (MASKS) {
              atomicOr(&pm[ag >> 5], 1u << (ag & 31u));
              atomicOr(&sm[ag >> 5], val[0] << (ag & 31u));
            }
          }
          n_ev += 1;
          ag += 1;
          p = q - w0;
        }
        pos = w0 + p;
        BK_STAMP(*this, 2, 3, lane);  // the window's events out (+ the slow path)
      }
    }
    wave_sync();
    return n_ev;
  }

  // ============ transactions.shuffle(rng) (env.rs:121): for i in (1..n).rev() swap(i, gen_range(0..i+1)) ============
  __device__ __forceinline__ void shuffle(uint32_t n_ev) {
    uint32_t i = n_ev > 0 ? n_ev - 1u : 0u;
    while (i >= 1u) {
      const uint32_t w0 = pos & ~63u, p0 = pos - w0;
      ensure(w0 + 64u);
      const uint32_t x = ring[(w0 + lane) & (WV_RING - 1)];
      const uint64_t valid = ~0ull << p0;
      const bool is_valid = (uint32_t)lane >= p0;
      // draw p serves index i - (#accepted draws before p): iterate the accept mask to its (unique) fixed point
      uint64_t acc = valid;
      uint32_t ii = 0, jj = 0;
      for (;;) {
        const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc, 0u));
        ii = i - k;                                  // <= 0 (wrapped): past the end of the shuffle
        const uint32_t r = ii + 1u;                  // range i + 1
        const bool in_range = is_valid && k < i;     // ii >= 1
        const uint32_t zone = (r << __builtin_clz(r | 1u)) - 1u;
        const uint64_t mm = (uint64_t)x * r;
        jj = (uint32_t)(mm >> 32);
        const uint64_t nacc = __ballot(in_range && (uint32_t)mm <= zone);
        if (nacc == acc) break;
        acc = nacc;
      }
      const uint32_t cnt = (uint32_t)__builtin_popcountll(acc);
      if (R <= 2 || co != nullptr) {
        if (lane_bit(acc)) jarr[ii] = (uint16_t)jj;  // resolved in one go below
      } else {
        // the swaps, in stream order (uniform addresses: every lane reads the same pair and writes the same values)
        uint64_t todo = acc;
        while (todo) {
          const uint32_t l = (uint32_t)__builtin_ctzll(todo);
          todo &= todo - 1ull;
          const uint32_t si = rdl(ii, l), sj = rdl(jj, l);
          const uint16_t ea = evl[si], eb = evl[sj];
          wave_sync();
          evl[si] = eb;
          evl[sj] = ea;
          wave_sync();
        }
      }
      i -= cnt;  // cnt <= i by construction
      // draws consumed: up to the accepted draw that served index 1, else the whole window
      pos = w0 + ((i < 1u && acc) ? (64u - (uint32_t)__builtin_clzll(acc)) : 64u);
    }
    BK_STAMP(*this, 2, 4, lane);  // the shuffle's draws: acceptance fixed point, swap targets
#ifdef BOURSE_AMD_SKIP_RESOLUTION  // (timing experiment: the draws and their acceptance only - results are then wrong)
    return;
#endif
    if constexpr (R <= 2) {
      // All swap targets j_i are known: resolve the whole Fisher-Yates in parallel instead of n dependent LDS round
      // trips.  Steps run i = n-1 .. 1; the value that ends at position x was at position j_x just before step x, and a
      // position y holds, before step t, what the most recent earlier step s* = min{s > t : j_s = y} moved there - the
      // value position s* held before step s* - or its original entry if there is none.  One 128-bit mask per position
      // ("steps that target it"), a first-set-bit search per hop, every position chased by its own lane
      // (tools/wave_decode_proto.py checks the rule against sequential swaps).
      if (n_ev >= 2u) {
        wave_sync();
        wmask[lane] = make_uint4(0u, 0u, 0u, 0u);
        wmask[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();
        uint32_t* w32 = reinterpret_cast<uint32_t*>(wmask);
        uint32_t jx[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint32_t x = (uint32_t)lane + 64u * g;
          jx[g] = (x >= 1u && x < n_ev) ? jarr[x] : x;
          if (jx[g] != x) atomicOr(&w32[jx[g] * 4u + (x >> 5)], 1u << (x & 31u));  // a self-swap moves nothing
        }
        wave_sync();
        uint32_t val[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint32_t x = (uint32_t)lane + 64u * g;
          uint32_t y = jx[g], t = x;
          bool go = x < n_ev;
          while (__ballot(go)) {
            const uint4 m = wmask[y & 127u];
            const uint32_t sx = first_above(m.x, m.y, m.z, m.w, t);
            const bool hop = go && sx < 128u;
            y = hop ? sx : y;
            t = hop ? sx : t;
            go = hop;
          }
          val[g] = evl[y & (64u * R - 1u)];
        }
        wave_sync();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint32_t x = (uint32_t)lane + 64u * g;
          if (x < n_ev) evl[x] = (uint16_t)val[g];
        }
        wave_sync();
      }
    } else if (wmask2 != nullptr && co != nullptr && n_ev >= 2u && n_ev <= 256u) {
      // Up to 256 events of a larger pool by the mask rule above in TWO rounds (round 5: the bucketed form below - LDS atomics to
      // count, prefix and fill, a loop over the bucket per hop - was 33 us of C5 as written's 120 us decode launch).  Steps run
      // i = n-1 .. 1 and a step s < 128 only touches positions <= s, so the shuffle is (steps 127 .. 1) after (steps n-1 .. 128):
      //   round A: masks of the steps >= 128 per target position (bit s - 128); a position x >= 128 is final after its own step
      //            - chase from (j_x, x) -, a position x < 128 holds what the most recent of those steps moved there - chase
      //            from (x, "before step 127") -; the values are written back;
      //   round B: the small pools' rule on positions 0 .. 127 of that array.
      // (tools/wave_decode_proto.py two_round_resolution checks the rule against the swaps done in order.)
      uint4* WM = wmask2;
      uint32_t* w32 = reinterpret_cast<uint32_t*>(WM);
      wave_sync();
#pragma unroll
      for (int g = 0; g < 4; ++g) WM[lane + 64 * g] = make_uint4(0u, 0u, 0u, 0u);
      wave_sync();
      uint32_t jx[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        jx[g] = (x >= 1u && x < n_ev) ? jarr[x] : x;
        if (g >= 2 && jx[g] != x) atomicOr(&w32[jx[g] * 4u + ((x - 128u) >> 5)], 1u << (x & 31u));
      }
      wave_sync();
      uint32_t av[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        uint32_t y = g >= 2 ? jx[g] : x, t = g >= 2 ? x - 128u : 0xFFFFFFFFu;  // t: bit index of the step, -1 = before step 127
        bool go = x < n_ev;
        while (__ballot(go)) {
          const uint4 m = WM[y & 255u];
          const uint32_t sx = first_above(m.x, m.y, m.z, m.w, t);
          const bool hop = go && sx < 128u;
          y = hop ? sx + 128u : y;
          t = hop ? sx : t;
          go = hop;
        }
        av[g] = evl[y & (64u * R - 1u)];
      }
      wave_sync();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        if (x < n_ev) evl[x] = (uint16_t)av[g];
      }
      WM[lane] = make_uint4(0u, 0u, 0u, 0u);
      WM[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
      wave_sync();
      const uint32_t nb = n_ev < 128u ? n_ev : 128u;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        if (x < nb && jx[g] != x) atomicOr(&w32[jx[g] * 4u + (x >> 5)], 1u << (x & 31u));
      }
      wave_sync();
      uint32_t bv[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        uint32_t y = jx[g], t = x;
        bool go = x < nb;
        while (__ballot(go)) {
          const uint4 m = WM[y & 127u];
          const uint32_t sx = first_above(m.x, m.y, m.z, m.w, t);
          const bool hop = go && sx < 128u;
          y = hop ? sx : y;
          t = hop ? sx : t;
          go = hop;
        }
        bv[g] = evl[y & (64u * R - 1u)];
      }
      wave_sync();
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t x = (uint32_t)lane + 64u * g;
        if (x < nb) evl[x] = (uint16_t)bv[g];
      }
      wave_sync();
    } else if (co != nullptr && n_ev >= 2u) {
      // The same rule for up to 512 positions, where a mask per position would be 32 KB: the steps are BUCKETED by their
      // target instead (counting sort through LDS atomics: count, exclusive prefix, fill), and "the most recent earlier
      // step that targets y" is the smallest s > t in y's bucket - buckets hold ~1 step on average.  Replaces ~n
      // dependent LDS round trips (23 us for the 157 events of C5 as written) by a handful of wave-parallel passes.
      wave_sync();
#pragma unroll
      for (int r = 0; r < R; ++r) co[r * 64 + lane] = 0u;
      wave_sync();
      uint32_t jx[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t x = (uint32_t)lane + 64u * r;
        jx[r] = (x >= 1u && x < n_ev) ? jarr[x] : x;
        if (jx[r] != x) atomicAdd(&co[jx[r]], 0x10000u);  // a self-swap moves nothing
      }
      wave_sync();
      uint32_t run = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t c = co[r * 64 + lane] >> 16;
        uint32_t v = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t u = (uint32_t)__shfl_up((int)v, o);
          v += lane >= o ? u : 0u;
        }
        co[r * 64 + lane] = (c << 16) | (run + v - c);  // count | start of the bucket (advanced to its end by the fill)
        run += rdl(v, 63);
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t x = (uint32_t)lane + 64u * r;
        if (jx[r] != x) bucket[atomicAdd(&co[jx[r]], 1u) & 0xFFFFu] = (uint16_t)x;
      }
      wave_sync();
      uint32_t val[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t x = (uint32_t)lane + 64u * r;
        uint32_t y = jx[r], t = x;
        bool go = x < n_ev;
        while (__ballot(go)) {
          const uint32_t c = co[y & (64u * R - 1u)], e = c & 0xFFFFu, k = go ? c >> 16 : 0u;
          uint32_t best = 0xFFFFFFFFu;
          for (uint32_t i = 0; __ballot(i < k); ++i) {
            const uint32_t sx = i < k ? (uint32_t)bucket[(e - k + i) & (64u * R - 1u)] : 0u;
            best = (i < k && sx > t && sx < best) ? sx : best;
          }
          const bool hop = go && best != 0xFFFFFFFFu;
          y = hop ? best : y;
          t = hop ? best : t;
          go = hop;
        }
        val[r] = evl[y & (64u * R - 1u)];
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t x = (uint32_t)lane + 64u * r;
        if (x < n_ev) evl[x] = (uint16_t)val[r];
      }
      wave_sync();
    }
  }

  // The generator state at `pos` (= what the reference's RNG holds now), and the cache record for the next launch:
  // chunk-start states of the block `pos` lies in + the offset inside it.
  __device__ __forceinline__ void finish(uint32_t* wc, uint32_t& n0, uint32_t& n1, uint32_t& n2, uint32_t& n3) {
    const bool in_last = pos + WV_BLOCK >= gen_end;  // pos lies in the last generated block (or at its end)
    uint4 bs = cs;
    if (!in_last) bs = wcs[lane];  // the block before the last generated one: stored when it was left (same lane)
    uint32_t c2 = pos - (in_last ? gen_end - WV_BLOCK : gen_end - 2u * WV_BLOCK);
    if (gen_end == 0) c2 = pos;  // nothing generated (no agents, no events): the cached block start is still `cs`
    bool end_jump = false;
    if (c2 == WV_BLOCK) {        // exactly at the block end: the cache describes the next block
      bs = wv_jump(tab, bs);
      c2 = 0;
      end_jump = true;
    }
    RngLane t{bs.x, bs.y, bs.z, bs.w};
    const uint32_t nst = c2 & (WV_K - 1u);
    for (uint32_t s = 0; s < nst; ++s) (void)t.next_u32();
    const uint32_t src = c2 >> 2;
    n0 = rdl(t.a0, src), n1 = rdl(t.a1, src), n2 = rdl(t.b0, src), n3 = rdl(t.b1, src);
    uint32_t cw = 0;
    cw = lane == WC_S0_LO ? n0 : cw;
    cw = lane == WC_S0_HI ? n1 : cw;
    cw = lane == WC_S1_LO ? n2 : cw;
    cw = lane == WC_S1_HI ? n3 : cw;
    cw = lane == WC_OFF ? c2 : cw;
    cw = lane == WC_TAG ? WC_MAGIC : cw;
    if (lane < 8) wc[lane] = cw;
    // The 1 KB of lane states goes out only when the record in memory is not already this block's: the block the position
    // lies in is the one load_cache found (nothing written), or a block that was stored when generation left it (gen_block).
    // A host-driven step's shuffle (~60 draws of a 256-draw block, step_events.hpp) re-used its block three steps in four and
    // wrote the same kilobyte back every time (round 6: 1.36 x of k_step_events' algorithmic bytes at 8 192 books).
    if (!was_cached || end_jump || (in_last && gen_end > WV_BLOCK)) wcs[lane] = bs;
  }
};

// ==================================================================================
// Split form: k_agents_wave writes the step batch, k_step_batch (book_device.hpp) consumes it.
// ==================================================================================
// Compiled for SIX waves per SIMD at R <= 2 (80 VGPRs, no scratch): asked to fit eight the kernel takes 64 VGPRs + 32 B
// of scratch per lane and is slower wherever it runs (same box: 8 192 books 111.6 -> 114.6 M, 12 288: 126.5 -> 132.6 M,
// 16 384: 133.8 -> 134.8 M; seven waves 113.4 / 130.8 / 133.4, five 113.4 / 130.5 / 132.2, four 113.6 / 130.6 / 131.4).
// The larger pools never reached eight (R = 4: seven, R = 8: five) and keep the request.
#ifndef BOURSE_AMD_AW_OCC
#define BOURSE_AMD_AW_OCC(R) ((R) <= 2 ? 6 : 8)
#endif
// LDS of one workgroup of the wave-parallel decode (4 books): the T^256 table + per book the ring of generated draws, the
// step's event list, placing / side masks and the shuffle's targets (and buckets, pools of more than 128 slots)
template <int R>
struct WaveLds {
  uint4 tab[512];
  uint32_t ring[4][WV_RING];
  uint16_t evl[4][64 * R];
  uint16_t jarr[4][64 * R];
  uint16_t bucket[4][R > 2 ? 64 * R : 1];
};

// One book's RNG-serial half of a step: RandomAgents::update for every group + the shuffle, written as the book's step
// batch; the body of k_agents_wave.  hdr: lane i holds dword i of the book's header
// (RNG state, live masks at H_LIVE0 ..).
// `given` (k_step_decode): the RNG state and live masks handed over in registers by the event half that has just STORED them -
// the scalar cache would still hold the header as it was at kernel start.
struct WaveHdrScalars {
  uint32_t s0l, s0h, s1l, s1h;
  uint64_t lv0, lv1;
};
template <int R>
__device__ __forceinline__ void agents_wave_book(const DevArgs& a, const WaveArgs& wa, WaveLds<R>& L, int wv, uint32_t book, int lane,
                                                 uint32_t hdr, const WaveHdrScalars* given = nullptr) {
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t* bt = a.batch + (size_t)book * a.batch_stride;
  uint32_t* wc = wa.wcache + (size_t)book * WC_STRIDE;

  WaveDecoder<R, false> D;
  BK_STAMP_START(D, book);
  D.tab = L.tab;
  D.ring = L.ring[wv];
  D.evl = L.evl[wv];
  D.pm = nullptr;
  D.sm = nullptr;
  D.pv = reinterpret_cast<uint2*>(bt + BT_EV + 32 * R);
  D.jarr = L.jarr[wv];
  D.wmask = reinterpret_cast<uint4*>(L.ring[wv]);  // the generated draws are dead once the shuffle's windows are resolved
  if (R > 2) {  // (64 R count|offset words = 2 KB at R = 8: the ring's memory, like the masks of the small pools)
    static_assert(R <= 8, "the bucket words alias the 2 KB ring");
    D.co = L.ring[wv];
    D.bucket = L.bucket[wv];
  }
  D.wcs = reinterpret_cast<uint4*>(wc + WC_HDR);
  D.lane = lane;
  // the book's RNG state and live masks (header dwords H_LIVE0 + w hold bits [32 w, 32 w + 32) of the pool's live mask)
  // (the header's scalars through the scalar cache, as k_step_batch reads them: book_device.hpp load_state_scalars; the RNG
  // state is dwords 2..5, the live masks 32..35 - explicit s_loads, this kernel stores the RNG state at its end)
  bk_u32x8 h0 = {};
  bk_u32x4 hl = {};
  if (given) {
    D.load_cache(wc, given->s0l, given->s0h, given->s1l, given->s1h, wa.jt_lane);
    hl[0] = (uint32_t)given->lv0, hl[1] = (uint32_t)(given->lv0 >> 32), hl[2] = (uint32_t)given->lv1, hl[3] = (uint32_t)(given->lv1 >> 32);
  } else if constexpr (R <= 2) {
    h0 = sload_x8(st);
    asm volatile("s_load_dwordx4 %0, %1, 0x80\n\ts_waitcnt lgkmcnt(0)" : "=&s"(hl) : "s"(st) : "memory");
    D.load_cache(wc, h0[H_S0_LO], h0[H_S0_HI], h0[H_S1_LO], h0[H_S1_HI], wa.jt_lane);
  } else {
    D.load_cache(wc, rdl(hdr, H_S0_LO), rdl(hdr, H_S0_HI), rdl(hdr, H_S1_LO), rdl(hdr, H_S1_HI), wa.jt_lane);
  }
  const uint32_t lim = 64u + (wa.lookahead < 1u ? 1u : (wa.lookahead > 64u ? 64u : wa.lookahead));
  BK_STAMP(D, 1, 0, lane);  // lane-state cache in
  static_assert(H_LIVE0 == 32, "the live masks are read at byte offset 0x80 above");
  const uint64_t lv0 = R <= 2 ? mk64(hl[0], hl[1]) : 0ull;
  const uint64_t lv1 = R == 2 ? mk64(hl[2], hl[3]) : 0ull;
  const uint32_t n_ev = D.agents(a, lim, hdr, H_LIVE0, lv0, lv1);
  BK_STAMP(D, 1, 1, lane);  // agents.update: generation, windows, walk
  D.shuffle(n_ev);
  BK_STAMP(D, 1, 2, lane);  // shuffle

  // ---- publish: RNG state, lane-state cache, step batch
  uint32_t n0, n1, n2, n3;
  D.finish(wc, n0, n1, n2, n3);
  const uint32_t hv = lane == 0 ? n0 : (lane == 1 ? n1 : (lane == 2 ? n2 : n3));
  if (lane < 4) st[H_S0_LO + lane] = hv;
  uint32_t hb = 0;
  hb = lane == BT_NEV ? n_ev : hb;
  bt[lane] = hb;  // (k_step_batch rebuilds the placing / side masks from the event words)
  for (uint32_t k = lane; k < 32u * R; k += 64u) {
    const uint32_t lo = 2u * k < n_ev ? D.evl[2u * k] : 0u, hi = 2u * k + 1u < n_ev ? D.evl[2u * k + 1u] : 0u;
    bt[BT_EV + k] = lo | (hi << 16);
  }
  BK_STAMP(D, 1, 3, lane);  // publish
  BK_STAMP_COUNT(D, 1, lane);
}

template <int R>
__global__ __launch_bounds__(256, BOURSE_AMD_AW_OCC(R)) void k_agents_wave(DevArgs a, WaveArgs wa) {
  __shared__ WaveLds<R> L;
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);  // wave-uniform: the per-wave LDS regions get scalar base addresses
  for (int i = threadIdx.x; i < 512; i += 256) L.tab[i] = wa.jt_block[i];
  __syncthreads();
  const uint32_t book = rfl(a.book_begin + blockIdx.x * 4 + wv);
  if (book >= a.book_end) return;
  const uint32_t hdr = (a.state + (size_t)book * a.state_stride)[lane];
  agents_wave_book<R>(a, wa, L, wv, book, lane, hdr);
}

// ==================================================================================
// k_step_decode (OPT-IN, BOURSE_AMD_STEP_DECODE=1; round 4's experiment re-tried on round 5's kernels as VERDICT r4 item 2 asked):
// Env::step of step s (the body of k_step_batch) FOLLOWED BY the decode of step s + 1 (the body of k_agents_wave) for the same
// book in one launch - a part's inner steps are ONE launch instead of two, a wave goes from its events straight into its next
// decode.  Nothing is carried between the halves but the header words a decode reads (RNG state, live masks), taken from the
// registers the store has just written.  The level bins of the event half live in the ring's LDS (dead until the decode
// generates into it).  Measured: docs/EXPERIMENTS.md (round 4: neutral at 8 192 books, slower above; round 5: see there).
// (Round 4 also built k_run_split = the whole launch persistent with the book parked in its state block between the halves,
//  eight waves per SIMD, every book of a C4 shard resident (commit 0ff3c3c): 112.5 M against 118.0 M - removed.)
// ==================================================================================
#ifndef BOURSE_AMD_SD_OCC
#define BOURSE_AMD_SD_OCC(R) BOURSE_AMD_AW_OCC(R)
#endif
template <int R>
__global__ __launch_bounds__(256, BOURSE_AMD_SD_OCC(R)) void k_step_decode(DevArgs a, WaveArgs wa, uint64_t step_index, uint32_t write_last) {
  __shared__ WaveLds<R> L;
  static_assert(WV_RING >= (uint32_t)LDS_DW_PER_WAVE, "the level bins alias the ring");
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 512; i += 256) L.tab[i] = wa.jt_block[i];
  __syncthreads();
  const uint32_t book = rfl(a.book_begin + blockIdx.x * 4 + wv);
  if (book >= a.book_end) return;
  uint32_t hdr = 0;
  WaveHdrScalars hs{};
  {
    Book<R> B;
    Rng rng;
    step_batch_book<R, false, false>(a, book, lane, L.ring[wv], step_index, write_last, B, rng, a.hist_slot0);
    hs.s0l = rfl((uint32_t)rng.s0), hs.s0h = rfl((uint32_t)(rng.s0 >> 32));
    hs.s1l = rfl((uint32_t)rng.s1), hs.s1h = rfl((uint32_t)(rng.s1 >> 32));
    hs.lv0 = B.live[0];
    hs.lv1 = R >= 2 ? B.live[R >= 2 ? 1 : 0] : 0ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {  // (the larger pools' walk reads the live words from lanes H_LIVE0 .. of this register)
      hdr = wrl((uint32_t)B.live[r], H_LIVE0 + 2 * r, hdr);
      hdr = wrl((uint32_t)(B.live[r] >> 32), H_LIVE0 + 2 * r + 1, hdr);
    }
  }
  wave_sync();  // the bins' LDS becomes the ring
  agents_wave_book<R>(a, wa, L, wv, book, lane, hdr, &hs);
}

// ==================================================================================
// Fused form: n_steps x { agents.update (wave-parallel decode); Env::step } per book with the book in registers and the
// generated draws in LDS across all steps of the launch (sim_runner, runner.rs:53-68) - no per-step launches, no state
// round trips; for batches that fit the chip once or twice (8 192 books = one resident wave per book).
// ==================================================================================
template <int R>
__global__ __launch_bounds__(512, 6) void k_run_wave(DevArgs a, WaveArgs wa, uint64_t first_step, uint32_t n_steps) {
  constexpr int WPB = 8;                                       // books (waves) per workgroup
  constexpr int STAGE_DW = 128 * R > 256 ? 128 * R : 256;      // new orders {price, vol} by slot, then the level bins
  __shared__ uint4 tab[512];
  __shared__ uint32_t ring_s[WPB][WV_RING];
  __shared__ uint16_t evl_s[WPB][64 * R];
  __shared__ uint32_t pm_s[WPB][2 * R], sm_s[WPB][2 * R];
  __shared__ uint32_t stage_s[WPB][STAGE_DW];
  __shared__ uint16_t jarr_s[WPB][64 * R];
  __shared__ uint4 wmask_s[WPB][R <= 2 ? 128 : 1];  // the ring stays live across steps here: the masks get their own 2 KB
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 512; i += 512) tab[i] = wa.jt_block[i];
  __syncthreads();
  const uint32_t book = rfl(blockIdx.x * WPB + wv);
  if (book >= a.n_books) return;
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t* wc = wa.wcache + (size_t)book * WC_STRIDE;
  uint32_t* stage = stage_s[wv];

  Book<R> B;
  Rng rng;
  load_book<R>(B, rng, st, lane);
  WaveDecoder<R> D;
  D.tab = tab;
  D.ring = ring_s[wv];
  D.evl = evl_s[wv];
  D.pm = pm_s[wv];
  D.sm = sm_s[wv];
  D.pv = reinterpret_cast<uint2*>(stage);
  D.jarr = jarr_s[wv];
  D.wmask = wmask_s[wv];
  D.wcs = reinterpret_cast<uint4*>(wc + WC_HDR);
  D.lane = lane;
  D.load_cache(wc, (uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32), wa.jt_lane);
  const uint32_t lim = 64u + (wa.lookahead < 1u ? 1u : (wa.lookahead > 64u ? 64u : wa.lookahead));
  uint64_t all[R];
#pragma unroll
  for (int r = 0; r < R; ++r) all[r] = ~0ull;
  uint32_t last_ntr = 0, last_nev = 0;

  for (uint32_t s = 0; s < n_steps; ++s) {
    // ---------------- agents.update(env, rng) + the shuffle of Env::step ----------------
    uint32_t livev = 0;  // lane w: bits [32 w, 32 w + 32) of the pool's live mask
#pragma unroll
    for (int r = 0; r < R; ++r) {
      livev = wrl((uint32_t)B.live[r], 2 * r, livev);
      livev = wrl((uint32_t)(B.live[r] >> 32), 2 * r + 1, livev);
    }
    if (lane < 2 * R) {
      D.pm[lane] = 0;
      D.sm[lane] = 0;
    }
    wave_sync();
    const uint32_t n_ev = D.agents(a, lim, livev, 0u, R <= 2 ? mk64(rdl(livev, 0u), rdl(livev, 1u)) : 0ull,
                                   R == 2 ? mk64(rdl(livev, 2u), rdl(livev, 3u)) : 0ull);
    D.shuffle(n_ev);
    // ---------------- the step's new orders into the pool (create_order ids: dense, agent order) -------------
    uint32_t ev[R];
    uint32_t base = B.next_id;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t pend = mk64(rfl(D.pm[2 * r]), rfl(D.pm[2 * r + 1]));
      const uint64_t side = mk64(rfl(D.sm[2 * r]), rfl(D.sm[2 * r + 1]));
      ev[r] = D.evl[r * 64 + lane];
      const uint2 pvv = D.pv[r * 64 + lane];
      B.price[r] = sel(pend, pvv.x, B.price[r]);
      B.vol[r] = sel(pend, pvv.y, B.vol[r]);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
      B.id[r] = sel(pend, base + rank, B.id[r]);
      base += __builtin_popcountll(pend);
      B.bid[r] = (B.bid[r] & ~pend) | (side & pend);
      B.pend[r] = pend;  // handed to step_from_list, which clears it (the event words classify themselves: EV_NEW)
    }
    B.next_id = base;
    wave_sync();  // the staging area becomes the snapshot's level bins
    // ---------------- Env::step: events at t0 + k, clock, level-2 record, trades ----------------
    last_ntr = step_from_list<R, false, false, true>(B, a, book, lane, ev, n_ev, stage,
                                                     a.hist_cap ? (a.hist_slot0 + s) % a.hist_cap : 0u,
                                                     s + 1 == n_steps || a.hist_cap == 0, a.tick_div, all, last_nev);
    wave_sync();
  }
  uint32_t n0, n1, n2, n3;
  D.finish(wc, n0, n1, n2, n3);
  rng.s0 = mk64(n0, n1);
  rng.s1 = mk64(n2, n3);
  store_book<R>(B, rng, st, lane, first_step + n_steps, last_ntr, last_nev);
}

}  // namespace bkd
