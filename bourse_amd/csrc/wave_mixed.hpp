// wave_mixed.hpp — k_agents_mixed_wave: agents.update of an AgentSet of NoiseAgent / MomentumAgent members + the shuffle
// of Env::step with ONE WAVE PER BOOK and the book's xoroshiro128** stream decoded 64 draws at a time.
//
// Why: the lane-per-book members' update (k_agents_mixed_lanes) walks ~1 500 dependent draws per book-step on ONE wave
// per 64 books - at C5 as written (8 192 books x 256 momentum + 256 noise agents) that is 128 waves on a chip of 1 024
// SIMDs, 532 us per step, 11 % lane utilisation (profiles/r02/pmc_c5m.json).  The members' streams are regular between
// hits, so they decode like RandomAgents' (wave_agents.hpp), only simpler - NOTHING a trader draws depends on the book:
//
//   * common::cancel_live_orders (ref crates/step_sim/src/agents/common.rs:54-76): one f32 draw per Active order of the
//     member, in list (= order id) order.  The member's `orders` list is kept per book (pool slots in creation order,
//     book-major so that a wave reads 64 entries with one load); entry i of a 64-entry chunk is lane i's: live test, draw
//     index = ballot prefix count, keep / cancel, in-place compaction - 64 orders per iteration.  An entry whose order
//     died is dropped in its member's own pass; slots are only handed out if they were free at the START of the step, so
//     a stale entry can never meet a re-used slot (no "listed" masks as in the lane-per-book kernel).
//   * the traders' loop (noise_agent.rs:132-174, momentum_agent.rs:163-203): a trader's turn starting at stream position
//     q ends at F(q), a function of the draws at q.. only (threshold tests, gen_bool, the ziggurat's length).  Every lane
//     evaluates F for its own position of the 64-draw window, the positions actually visited are the orbit of the
//     window's entry point under F (pointer doubling, 5 rounds), and the visited lanes that hit place their orders ALL AT
//     ONCE: ziggurat, exp, tick rounding are per-lane f64 code across the window's orders instead of scalar code per order.
//   * order ids are dense in creation order (ballot prefix counts), the k-th order created in a step takes the k-th free
//     slot (a rank table in LDS), and is written straight into the book's pool block with its pend bit - what
//     k_step_batch<R, false, POOLPEND> (book_device.hpp) expects.
//   * the shuffle is wave_agents.hpp's, on the same stream (the lane states are handed over).
//
// Semantics restated from (paths relative to the reference repo): agents/common.rs:21-141, noise_agent.rs:127-176,
// momentum_agent.rs:146-208, rand 0.8.5 / rand_distr 0.4.3 sampling as in mixed_agents.hpp (PARITY UNPINNED against
// Rust, bit-exact against the oracle through pm_math.hpp).  Independent books only (assets == 1; markets keep the
// lane-per-book update); a RandomAgents member of such a set is walked draw by draw on the scalar path.
#pragma once
#include "mixed_agents.hpp"
#include "wave_agents.hpp"

#pragma clang fp contract(off)

namespace bkd {

// timing experiments only (scripts/mw_phase_times.sh): -DBOURSE_AMD_MW_SKIP=bits leaves phases out (results are then
// wrong): 1 the shuffle, 2 the traders' windows, 4 the cancel pass, 8 the window's order placement
#ifndef BOURSE_AMD_MW_SKIP
#define BOURSE_AMD_MW_SKIP 0
#endif
#ifndef BOURSE_AMD_TWO_ROUND
#define BOURSE_AMD_TWO_ROUND 1
#endif
constexpr uint32_t FLAG_DECODE_LOOKAHEAD = 256u;  // a ziggurat ran past the decode's look-ahead (p < 2^-90): flagged
constexpr uint32_t MW_RING = 512;                 // generated u64 draws kept in LDS
constexpr uint32_t MW_LOOK = 192;                 // a window's draws + look-ahead: positions [w0, w0 + MW_LOOK)
// books (waves) per workgroup: 8 (69.7 KB of LDS per workgroup since round 5's 64-entry price queue, two per CU = four waves
// per SIMD).  Until the event kernel's launches got 40 % shorter (the top-anchored key window, book_device.hpp
// keys_begin_wide) this kernel's occupancy did not matter - halving it cost 5 % -; since then the four parts' decode launches
// queue for LDS and halving it costs 13 %.  NINE waves per workgroup (76.8 KB, 18 waves per CU, 80 VGPRs) was tried: 36.0
// against 39.8 M - 228 workgroups per 2 048-book launch leave 28 CUs without one (docs/EXPERIMENTS.md).
#ifndef BOURSE_AMD_MW_WPB
#define BOURSE_AMD_MW_WPB 8
#endif
constexpr int MW_WPB = BOURSE_AMD_MW_WPB;

// per wave: the u64 ring | event list, free-slot table (u16 x 64 R each; the table's memory becomes the shuffle's swap
// targets) | orbit marks (72) + the pool's live words (16) + pad | deferred-price queue: 64 x f64 argument + 64 x u16
// {slot, side} - marks + live words + queue are 1 KB together, which becomes the shuffle's buckets
constexpr uint32_t MW_QCAP = 64;
static_assert(96 * 4 + MW_QCAP * 8 + MW_QCAP * 2 >= 64 * 8 * 2, "the shuffle's buckets (u16 x 512) alias marks + live words + queue");
constexpr uint32_t mw_wave_dwords(int R) { return 2 * MW_RING + 2 * 32 * R + 96 + 2 * MW_QCAP + MW_QCAP / 2; }
constexpr uint32_t MW_SHARED_DW = 2048 + 2 * 514 + 4;  // T^256 table, ziggurat x / f tables (257 doubles each), pad
constexpr size_t mixed_wave_lds_bytes(int R) { return (size_t)(MW_SHARED_DW + MW_WPB * mw_wave_dwords(R)) * 4; }

// xoroshiro128** on 32-bit halves, full 64-bit output word
__device__ __forceinline__ uint64_t rnglane_next_u64(RngLane& t) {
  const uint64_t s0 = mk64(t.a0, t.a1);
  uint64_t r = s0 * 5ull;
  r = (r << 7) | (r >> 57);
  r *= 9ull;
  t.advance();
  return r;
}

// The generator side of the decode: same block structure and cache record as WaveDecoder (wave_agents.hpp) - lane j
// holds the state 4 j draws into the 256-draw block - so the two kernels hand a book over through its record.
struct Stream64 {
  const uint4* tab;
  uint64_t* ring;  // ring[q & (MW_RING - 1)] = draw q of this launch's stream
  uint4* wcs;
  int lane;
  uint4 cs;
  uint32_t gen_end, pos;

  __device__ __forceinline__ void load_cache(const uint32_t* wc, uint32_t s0l, uint32_t s0h, uint32_t s1l, uint32_t s1h,
                                             const uint4* jt_lane) {
    const uint32_t wch = wc[lane];
    cs = reinterpret_cast<const uint4*>(wc + WC_HDR)[lane];
    pos = rdl(wch, WC_OFF);
    const bool cached = rdl(wch, WC_TAG) == WC_MAGIC && rdl(wch, WC_S0_LO) == s0l && rdl(wch, WC_S0_HI) == s0h &&
                        rdl(wch, WC_S1_LO) == s1l && rdl(wch, WC_S1_HI) == s1h && pos < WV_BLOCK;
    if (!cached) {
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
    gen_end = 0;
  }
  __device__ __forceinline__ void gen_block() {
    if (gen_end != 0) {
      wcs[lane] = cs;
      cs = wv_jump(tab, cs);
    }
    RngLane t{cs.x, cs.y, cs.z, cs.w};
    uint64_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = rnglane_next_u64(t);
    uint4* dst = reinterpret_cast<uint4*>(ring + ((gen_end + 4u * (uint32_t)lane) & (MW_RING - 1)));
    dst[0] = make_uint4((uint32_t)x[0], (uint32_t)(x[0] >> 32), (uint32_t)x[1], (uint32_t)(x[1] >> 32));
    dst[1] = make_uint4((uint32_t)x[2], (uint32_t)(x[2] >> 32), (uint32_t)x[3], (uint32_t)(x[3] >> 32));
    gen_end += WV_BLOCK;
    wave_sync();
  }
  __device__ __forceinline__ void ensure(uint32_t upto) {
    while (gen_end < upto) gen_block();
  }
  __device__ __forceinline__ uint64_t at(uint32_t q) const { return ring[q & (MW_RING - 1)]; }
};

// out[k] = the k-th slot of `mask` in slot order; returns the number of slots
template <int R>
__device__ __forceinline__ uint32_t rank_slots(const uint64_t (&mask)[R], uint16_t* out, int lane) {
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t lr = acc + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[r], 0u));
    if (lane_bit(mask[r])) out[lr] = (uint16_t)(64u * r + (uint32_t)lane);
    acc += (uint32_t)__builtin_popcountll(mask[r]);
  }
  wave_sync();
  return acc;
}

// the members' `orders` lists of the wave-per-book update: pool slots in creation (= id) order, per book
struct WaveLists {
  uint16_t* list;  // [n_books][MAX_MEMBERS][cap]
  uint32_t* len;   // [n_books][MAX_MEMBERS]
  uint32_t cap;    // entries per member = pool size
};

// rand_distr StandardNormal (256-layer ziggurat) read from the generated stream starting at q0; len = draws consumed.
// `lim`: first position NOT available; running into it sets `over` (the caller flags the book).
__device__ __forceinline__ double zig_from_stream(const Stream64& S, const double* zx, const double* zf, uint32_t q0,
                                                  uint32_t lim, uint32_t& len, bool& over) {
  uint32_t q = q0;
  double out = 0.0;
  for (;;) {
    if (q >= lim) {
      over = true;
      break;
    }
    const uint64_t bits = S.at(q++);
    const uint32_t i = (uint32_t)bits & 0xffu;
    const double u = pm::from_bits(0x4000000000000000ull | (bits >> 12)) - 3.0;
    const double x = u * zx[i];
    if (pm::fabs_(x) < zx[i + 1]) {
      out = x;
      break;
    }
    if (i == 0) {
      const double Rz = 3.654152885361008796;
      double xx = 1.0, yy = 0.0;
      bool bad = false;
      while (-2.0 * yy < xx * xx) {
        if (q + 2u > lim) {
          bad = true;
          break;
        }
        const double x_ = pm::from_bits(0x3FF0000000000000ull | (S.at(q) >> 12)) - (1.0 - 2.220446049250313e-16 / 2.0);
        const double y_ = pm::from_bits(0x3FF0000000000000ull | (S.at(q + 1u) >> 12)) - (1.0 - 2.220446049250313e-16 / 2.0);
        q += 2u;
        xx = pm::log(x_) / Rz;
        yy = pm::log(y_);
      }
      if (bad) {
        over = true;
        break;
      }
      out = (u < 0.0) ? xx - Rz : Rz - xx;
      break;
    }
    if (q >= lim) {
      over = true;
      break;
    }
    const double f = static_cast<double>(S.at(q++) >> 11) * (1.0 / 9007199254740992.0);
    const double lhs = zf[i + 1] + (zf[i] - zf[i + 1]) * f;
    if (lhs < pm::exp(-x * x / 2.0)) {
      out = x;
      break;
    }
  }
  len = q - q0;
  return out;
}

template <int R>
__global__ __launch_bounds__(64 * MW_WPB, MW_WPB > 8 ? 5 : 4) void k_agents_mixed_wave(DevArgs a, MixedArgs ma, WaveArgs wa, WaveLists wl) {
  extern __shared__ uint32_t mw_lds[];
  constexpr uint32_t SL = 64u * R;
  uint4* tab = reinterpret_cast<uint4*>(mw_lds);
  double* zx = reinterpret_cast<double*>(mw_lds + 2048);
  double* zf = zx + 257;
  const int lane = threadIdx.x & 63;
  const int wv = (int)rfl(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 512; i += 64 * MW_WPB) tab[i] = wa.jt_block[i];
  for (int i = threadIdx.x; i < 257; i += 64 * MW_WPB) {
    zx[i] = ZIG_NORM_X[i];
    zf[i] = ZIG_NORM_F[i];
  }
  __syncthreads();
  const uint32_t book = rfl(a.book_begin + blockIdx.x * MW_WPB + wv);
  if (book >= a.book_end) return;
  uint32_t* wbase = mw_lds + MW_SHARED_DW + (uint32_t)wv * mw_wave_dwords(R);
  uint64_t* ring = reinterpret_cast<uint64_t*>(wbase);
  uint16_t* evl = reinterpret_cast<uint16_t*>(wbase + 2 * MW_RING);
  uint16_t* freelist = evl + SL;  // the step's free slots in allocation order (then the shuffle's swap targets)
  uint32_t* mark = wbase + 2 * MW_RING + 2 * 32 * R;  // 72 dwords: the orbit's marks (index 64: left the window)
  uint32_t* lvw = mark + 72;                          // 16 dwords: the pool's live mask, 32 slots per word
  // Deferred limit prices: a window's orders are created at once, but exp() and the tick rounding - ~200 f64 instructions
  // - would run for the 2-3 lanes of every window that place one.  Those lanes create the order WITHOUT its price and
  // queue {exp argument, slot, side}; whenever 64 entries wait, all lanes price one each (same arithmetic on the same
  // operands, so the same bits).  A sell that might reach the u32::MAX clamp - the one case whose outcome (create_order's
  // Err: no id) changes what follows - keeps the in-line path.
  double* q_arg = reinterpret_cast<double*>(mark + 96);
  uint16_t* q_info = reinterpret_cast<uint16_t*>(mark + 96 + 2 * MW_QCAP);

  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t* bt = a.batch + (size_t)book * a.batch_stride;
  uint32_t* wc = wa.wcache + (size_t)book * WC_STRIDE;
  const uint32_t hdr = st[lane];
  uint64_t live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) live[r] = mk64(rdl(hdr, H_LIVE0 + 2 * r), rdl(hdr, H_LIVE0 + 2 * r + 1));
  {
    const uint32_t lw = (uint32_t)__shfl((int)hdr, (H_LIVE0 + lane) & 63);  // (all lanes: a shuffle reads active lanes only)
    if (lane < 16) lvw[lane] = lane < 2 * R ? lw : 0u;
    wave_sync();
  }
  Stream64 S;
  S.tab = tab;
  S.ring = ring;
  S.wcs = reinterpret_cast<uint4*>(wc + WC_HDR);
  S.lane = lane;
  S.load_cache(wc, rdl(hdr, H_S0_LO), rdl(hdr, H_S0_HI), rdl(hdr, H_S1_LO), rdl(hdr, H_S1_HI), wa.jt_lane);

  const uint32_t n_fixed = ma.n_fixed;
  const uint32_t next_id = rdl(hdr, H_NEXT_ID);
  uint32_t new_flags = 0, gflags = rdl(hdr, H_GFLAGS);
  uint32_t hdr_out = hdr;  // member state is patched into the header image lane by lane
  // OrderBook::mid_price (orderbook.rs:272-276) of the book the agents see: the touches of the last level-2 record
  // (updates only queue events, so the book is still the one that record describes)
  double mid;
  {
    const uint32_t* l2 = a.l2_last + (size_t)book * a.l2_width;
    const uint32_t bid = rfl(l2[1]), ask = rfl(l2[2]);
    mid = static_cast<double>(bid) + 0.5 * static_cast<double>(ask - bid);
  }
  // free slots of the dynamic region, in allocation order: free at the START of the step (see the header comment)
  uint32_t n_free;
  {
    uint64_t fr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t lo = n_fixed > 64u * r ? n_fixed - 64u * r : 0u;
      fr[r] = ~live[r] & (lo >= 64u ? 0ull : (~0ull << lo));
    }
    n_free = rank_slots<R>(fr, freelist, lane);
  }
  uint32_t n_ev = 0, n_created = 0;  // n_created: orders of Noise / Momentum members (each takes a free slot, in this order)
  uint32_t id_extra = 0;             // ids consumed by RandomAgents members (fixed slots): id = next_id + both counters

  for (uint32_t j = 0; j < ma.n_desc; ++j) {  // members in declaration order (crates/macros/src/lib.rs:57-73)
    const MixedDesc D = ma.descs[j];
    if (D.type == 0) {
      // ---- RandomAgents::update (random_agent.rs:85-119) as a MEMBER of such a set: fixed slots [slot_base, slot_base + n),
      // its draws taken one by one on the scalar path (uniform LDS reads of the generated stream).  Mixed sets are the
      // rare case and their RandomAgents members small; what matters is that the set as a whole stays on this kernel
      // instead of the lane-per-book update (a RandomAgents-only set has its own decode, wave_agents.hpp).
      auto draw = [&]() -> uint32_t {
        S.ensure(S.pos + 1u);
        const uint32_t x = rfl((uint32_t)S.at(S.pos));
        S.pos += 1u;
        return x;
      };
      auto below = [&](uint32_t range, uint32_t zone) -> uint32_t {  // UniformInt<u32>::sample_single
        for (;;) {
          const uint64_t mm = (uint64_t)draw() * range;
          if ((uint32_t)mm <= zone) return (uint32_t)(mm >> 32);
        }
      };
      for (uint32_t i = 0; i < D.n; ++i) {
        const uint32_t slot = D.slot_base + i;
        if ((draw() >> 8) < D.thr) {  // gen::<f32>() < activity_rate
          if (lane == 0) evl[n_ev] = (uint16_t)slot;
          n_ev += 1u;
          if (!((rfl(lvw[(slot >> 5) & 15u]) >> (slot & 31u)) & 1u)) {  // no Active order: side, tick, vol (:99-101)
            const uint32_t side = below(2u, 0x7FFFFFFFu);
            const uint32_t tick = D.tick_lo + below(D.tick_rng, D.tick_zone);
            const uint32_t vol = D.vol_lo + below(D.vol_rng, D.vol_zone);
            if (lane == 0) {
              uint32_t* p = st + HDR_DW + (slot >> 6) * (POOL_FIELDS * 64) + (slot & 63u);
              p[0 * 64] = tick * D.tick_size;
              p[1 * 64] = vol;
              p[2 * 64] = next_id + n_created + id_extra;
              p[4 * 64] = 4u | (side ? 2u : 0u);  // pending New, owner tag 0
            }
            id_extra += 1u;
          }  // else: its cancellation (the event kernel tells the two apart by the slot's pend bit)
        }
      }
      continue;
    }
    const uint32_t tag = j + 1;
    // ---- common::cancel_live_orders (common.rs:54-76): the list's Active orders in order, one f32 draw each; a draw
    // `> p_cancel` keeps the order, otherwise its cancellation is queued.  64 entries per iteration.
    uint16_t* my = wl.list + ((size_t)book * MAX_MEMBERS + j) * wl.cap;
    const uint32_t len = rfl(wl.len[(size_t)book * MAX_MEMBERS + j]);
    uint32_t keep_pos = 0;
    for (uint32_t c = 0; c < ((BOURSE_AMD_MW_SKIP & 4) ? 0u : len); c += 64u) {
      const uint32_t idx = c + (uint32_t)lane;
      const uint32_t slot = idx < len ? (uint32_t)my[idx] : 0u;
      const bool alive = idx < len && ((lvw[(slot >> 5) & 15u] >> (slot & 31u)) & 1u) != 0u;  // else: filled / cancelled meanwhile
      const uint64_t am = __ballot(alive);
      S.ensure(S.pos + 64u);
      const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
      const uint32_t x = (uint32_t)S.at(S.pos + rk);
      const bool keep = alive && (int32_t)(x >> 8) > D.keep_thr;
      const bool cancel = alive && !keep;
      const uint64_t km = __ballot(keep), cm = __ballot(cancel);
      if (cancel) evl[n_ev + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))] = (uint16_t)slot;
      // in-place compaction: the write position never passes the read position, and this chunk's entries are in registers
      if (keep) my[keep_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u))] = (uint16_t)slot;
      n_ev += (uint32_t)__builtin_popcountll(cm);
      keep_pos += (uint32_t)__builtin_popcountll(km);
      S.pos += (uint32_t)__builtin_popcountll(am);
    }
    // ---- the member's traders
    const bool noise_member = D.type == 1, noise = noise_member;
    double m = 0.0;
    uint64_t thr_l = 0, thr_m = 0;
    int sgn = 0;
    if (!noise) {  // MomentumAgent::update (momentum_agent.rs:146-162): this step's signal and order probabilities
      double p_market = 0.0;
      if ((gflags >> j) & 1u) {
        const double gm = pm::from_bits(mk64(rdl(hdr, H_GST + 4 * j), rdl(hdr, H_GST + 4 * j + 1)));
        const double gl = pm::from_bits(mk64(rdl(hdr, H_GST + 4 * j + 2), rdl(hdr, H_GST + 4 * j + 3)));
        m = uni(gm * (1.0 - D.decay) + D.decay * (mid - gl));
        p_market = uni(D.demand * pm::tanh(D.scale * m) / D.n_f);
      }
      thr_l = thr53(D.order_ratio * p_market);
      thr_m = thr53(p_market);
      thr_l = mk64(rfl((uint32_t)thr_l), rfl((uint32_t)(thr_l >> 32)));
      thr_m = mk64(rfl((uint32_t)thr_m), rfl((uint32_t)(thr_m >> 32)));
      sgn = (m > 0.0) ? 1 : ((m < 0.0) ? -1 : 0);
      const uint64_t mb = pm::to_bits(m), lb = pm::to_bits(mid);
      hdr_out = lane == H_GST + 4 * (int)j ? (uint32_t)mb : hdr_out;
      hdr_out = lane == H_GST + 4 * (int)j + 1 ? (uint32_t)(mb >> 32) : hdr_out;
      hdr_out = lane == H_GST + 4 * (int)j + 2 ? (uint32_t)lb : hdr_out;
      hdr_out = lane == H_GST + 4 * (int)j + 3 ? (uint32_t)(lb >> 32) : hdr_out;
      gflags |= 1u << j;
    }
    if ((BOURSE_AMD_MW_SKIP & 2) || (!noise && ((thr_l == 0 && thr_m == 0) || sgn == 0))) {
      S.pos += 2u * D.n;  // two threshold draws per trader, nobody can act (momentum_agent.rs:165,193)
      S.ensure(S.pos);    // (the position never runs ahead of the generated blocks: finish() locates it in the last two)
      if (lane == 0) wl.len[(size_t)book * MAX_MEMBERS + j] = keep_pos;
      continue;
    }
    // a sell at mid + exp(arg), rounded UP to the tick, stays below the u32::MAX clamp when arg < lnslack
    double lnslack;
    {
      const double slack = 4294967295.0 - mid - 2.0 * D.tick_f - 1.0;
      lnslack = uni(slack > 1.0 ? pm::log(slack) - 1e-9 : -1e300);
    }
    uint32_t qc = 0;  // queued prices
    auto drain = [&](uint32_t at_least) {
      while (qc >= at_least && qc > 0u) {
        wave_sync();
        const uint32_t take = qc < 64u ? qc : 64u, base = qc - take;
        if ((uint32_t)lane < take) {
          const double arg = q_arg[base + lane];
          const uint32_t info = q_info[base + lane], slot = info & 0x7FFFu;
          const double dist = pm::fabs_(pm::exp(arg));
          st[HDR_DW + (slot >> 6) * (POOL_FIELDS * 64) + (slot & 63u)] =
              (info & 0x8000u) ? round_price_down(mid - dist, D.tick_f) : round_price_up(mid + dist, D.tick_f);
        }
        qc = base;
        wave_sync();
      }
    };
    // (the traders' loop is instantiated per member kind: no per-lane selects between the two draw patterns)
    auto traders = [&](auto kind) {
    constexpr bool noise = decltype(kind)::value;
    uint32_t t = 0;
    while (t < D.n) {
      const uint32_t w0 = S.pos & ~63u, p0 = S.pos - w0, lim = w0 + MW_LOOK;
      S.ensure(lim);
      const uint32_t q = w0 + (uint32_t)lane;
      // ---- a trader's turn STARTING at q (every lane for its own position)
      const uint64_t xa = S.at(q);
      const bool hit_a = noise ? ((uint32_t)xa >> 8) < D.thr_limit : (xa >> 11) < thr_l;
      const uint32_t zstart = noise ? q + 2u : q + 1u;  // Noise: gen_bool(0.5) first (noise_agent.rs:135)
      uint32_t zlen = 0;
      double zval = 0.0;
      bool over = false;
      if (hit_a) zval = zig_from_stream(S, zx, zf, zstart, lim, zlen, over);
      const uint32_t qb = hit_a ? zstart + zlen : q + 1u;
      over = over || qb + 2u > lim;
      const uint64_t xb = S.at(qb);
      const bool hit_b = noise ? ((uint32_t)xb >> 8) < D.thr_market : (xb >> 11) < thr_m;
      const uint32_t f_end = qb + 1u + ((noise && hit_b) ? 1u : 0u);  // Noise: the market order's gen_bool (noise_agent.rs:163)
      // ---- the positions visited from p0: the orbit of p0 under lane -> f_end - w0 (a turn takes >= 2 draws: <= 32 hops)
      const uint32_t jk0 = min(f_end - w0, 64u);
      const uint32_t rem = D.n - t;
      uint64_t V = 0;
      uint32_t last = p0;
      {  // pointer doubling: 5 rounds of {marks through LDS, jump table squared by ds_bpermute}.  (A scalar walk - one
         // v_readlane per hop - was measured and is slower: 167 vs 147 us per full-batch launch at C5 as written.)
        uint32_t jk = jk0;
        bool vis = (uint32_t)lane == p0;
        mark[lane] = 0;  // (marks only ever get set inside a window: cleared once, not per round)
        wave_sync();
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          if (vis) mark[jk] = 1u;
          wave_sync();
          vis = vis || mark[lane] != 0u;
          const uint32_t jn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jk << 2), (int)jk);
          jk = jk < 64u ? jn : 64u;
        }
        V = __ballot(vis);
        if ((uint32_t)__builtin_popcountll(V) > rem) {  // the member's last trader sits inside this window
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(V >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)V, 0u));
          V = __ballot(vis && rk < rem);
        }
        last = 63u - (uint32_t)__builtin_clzll(V);
      }
      const bool vis = lane_bit(V);
      t += (uint32_t)__builtin_popcountll(V);
      S.pos = rdl(f_end, last);
      if (__ballot(vis && over)) new_flags |= FLAG_DECODE_LOOKAHEAD;
      // ---- the window's orders, all at once.  Limit: place_buy/sell_limit_order (common.rs:92-141)
      const bool do_a = vis && hit_a && !(BOURSE_AMD_MW_SKIP & 8);
      bool buy_a = sgn > 0, ok_a = false;
      uint32_t price_a = 0;
      const double arg_a = D.mu + D.sigma * zval;
      if (noise) buy_a = (S.at(q + 1u) >> 63) == 0ull;  // gen_bool(0.5): next_u64() < 2^63
      // a buy rounds DOWN from below the mid (clamped at 0), a sell below the clamp rounds to a multiple of the member's
      // tick, itself a multiple of the book's: create_order accepts both, whatever the price turns out to be
      const bool defer_a = do_a && (buy_a || arg_a < lnslack);
      if (do_a && !defer_a) {
        const double dist = pm::fabs_(pm::exp(arg_a));
        price_a = buy_a ? round_price_down(mid - dist, D.tick_f) : round_price_up(mid + dist, D.tick_f);
        // create_order's tick check (orderbook.rs:367-382): the reference `.unwrap()`s the Err (panics); flagged, and
        // like an Err nothing is created (see mixed_create)
        ok_a = price_a % a.tick_size == 0u;
      }
      ok_a = ok_a || defer_a;
      if (__ballot(do_a && !ok_a)) new_flags |= FLAG_PRICE_TICK;
      const bool do_b = vis && hit_b && !(BOURSE_AMD_MW_SKIP & 8);
      const bool buy_b = noise ? (S.at(qb + 1u) >> 63) == 0ull : sgn > 0;
      const uint64_t CA = __ballot(do_a && ok_a), CB = __ballot(do_b);
      const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(CA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)CA, 0u)) +
                              __builtin_amdgcn_mbcnt_hi((uint32_t)(CB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)CB, 0u));
      // Env::place_order: dense ids in creation order (orderbook.rs:363), a trader's limit order before its market order
      auto emit = [&](uint32_t k, bool bid, uint32_t price, uint32_t tg) -> uint32_t {
        if (k >= n_free) return 0xFFFFu;  // pool full: the id is consumed, the order and its event are dropped (flagged below)
        const uint32_t slot = freelist[k];
        uint32_t* p = st + HDR_DW + (slot >> 6) * (POOL_FIELDS * 64) + (slot & 63u);
        p[0 * 64] = price;
        p[1 * 64] = D.trade_vol;
        p[2 * 64] = next_id + id_extra + k;
        p[4 * 64] = 4u | (bid ? 2u : 0u) | (tg << 8);  // pending New
        evl[n_ev + (k - n_created)] = (uint16_t)slot;
        return slot;
      };
      if (do_a && ok_a) {
        const uint32_t slot = emit(n_created + before, buy_a, price_a, tag);
        // live_orders.push(order_id) (noise_agent.rs:158, momentum_agent.rs:188): behind the kept ones, in creation order
        const uint32_t li = keep_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(CA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)CA, 0u));
        if (slot != 0xFFFFu) my[li] = (uint16_t)slot;
      }
      {  // the orders whose price is still to come
        const bool qd = defer_a && n_created + before < n_free;
        const uint64_t qm = __ballot(qd);
        if (qc + (uint32_t)__builtin_popcountll(qm) > MW_QCAP) drain(1u);  // (no room for this window's: price what waits first)
        if (qd) {
          const uint32_t qi = qc + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
          q_arg[qi] = arg_a;
          q_info[qi] = (uint16_t)(freelist[n_created + before] | (buy_a ? 0x8000u : 0u));
        }
        qc += (uint32_t)__builtin_popcountll(qm);
        drain(MW_QCAP);
      }
      if (do_b) emit(n_created + before + ((do_a && ok_a) ? 1u : 0u), buy_b, buy_b ? 0xFFFFFFFFu : 0u, 0u);
      const uint32_t cnt = (uint32_t)__builtin_popcountll(CA) + (uint32_t)__builtin_popcountll(CB);
      const uint32_t room = n_free > n_created ? n_free - n_created : 0u;
      n_ev += cnt < room ? cnt : room;
      if (cnt > room) new_flags |= FLAG_POOL_OVERFLOW;
      // (a limit order dropped for want of a slot is not remembered: the slots run out for every later order too, so
      // the remembered ones are a prefix)
      {
        const uint32_t na = (uint32_t)__builtin_popcountll(CA);
        uint32_t got_a = na;
        if (cnt > room) {  // count the limit orders among the first `room` creations of the window
          const bool got = (do_a && ok_a) && (before < room);
          got_a = (uint32_t)__builtin_popcountll(__ballot(got));
        }
        keep_pos += got_a;
      }
      n_created += cnt;
    }
    };
    if (noise_member)
      traders(std::true_type{});
    else
      traders(std::false_type{});
    drain(1u);
    if (lane == 0) wl.len[(size_t)book * MAX_MEMBERS + j] = keep_pos;
  }
  wave_sync();

  // ---- transactions.shuffle(rng) (env.rs:121): wave_agents.hpp's decoder on the same stream.  Its ring holds 32-bit
  // draws: the low words of everything generated and not yet consumed move over (in place: all reads before any write)
  WaveDecoder<R> Dc;
  {
    uint32_t lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) lo[k] = (uint32_t)ring[k * 64 + lane];
    wave_sync();
    uint32_t* r32 = reinterpret_cast<uint32_t*>(ring);
#pragma unroll
    for (int k = 0; k < 8; ++k) r32[k * 64 + lane] = lo[k];
    wave_sync();
    Dc.tab = tab;
    Dc.ring = r32;
    Dc.evl = evl;
    Dc.pm = nullptr;
    Dc.sm = nullptr;
    Dc.pv = nullptr;
    Dc.jarr = freelist;
    Dc.wmask = reinterpret_cast<uint4*>(r32 + MW_RING);  // the upper half of the 64-bit ring's memory (2 KB), R <= 2 only
    if (R > 2) {  // large pools: the bucketed resolution's words there instead, its buckets in the price queue's memory
      Dc.co = r32 + MW_RING;
      Dc.bucket = reinterpret_cast<uint16_t*>(mark);  // (marks, live words and the price queue are dead here: 1 KB)
      // (the 64-bit ring's 4 KB: the draws are dead once the targets are known; -DBOURSE_AMD_TWO_ROUND=0: the buckets for every size)
      Dc.wmask2 = BOURSE_AMD_TWO_ROUND ? reinterpret_cast<uint4*>(r32) : nullptr;
    }
    Dc.wcs = S.wcs;
    Dc.lane = lane;
    Dc.cs = S.cs;
    Dc.gen_end = S.gen_end;
    Dc.pos = S.pos;
    Dc.was_cached = false;  // (finish() always stores the lane states: ~1 500 draws per step cross several blocks)
  }
  if (!(BOURSE_AMD_MW_SKIP & 1)) Dc.shuffle(n_ev);

  // ---- publish: RNG state + lane-state cache, member state, ids, cursor, flags; the step batch
  uint32_t n0, n1, n2, n3;
  Dc.finish(wc, n0, n1, n2, n3);
  hdr_out = lane == H_S0_LO ? n0 : hdr_out;
  hdr_out = lane == H_S0_HI ? n1 : hdr_out;
  hdr_out = lane == H_S1_LO ? n2 : hdr_out;
  hdr_out = lane == H_S1_HI ? n3 : hdr_out;
  hdr_out = lane == H_NEXT_ID ? next_id + n_created + id_extra : hdr_out;
  hdr_out = lane == H_FLAGS ? (hdr | new_flags) : hdr_out;
  hdr_out = lane == H_GFLAGS ? gflags : hdr_out;
  st[lane] = hdr_out;
  bt[lane] = lane == BT_NEV ? n_ev : 0u;
  for (uint32_t k = lane; k < 32u * R; k += 64u) {
    const uint32_t lo = 2u * k < n_ev ? evl[2u * k] : 0u, hi = 2u * k + 1u < n_ev ? evl[2u * k + 1u] : 0u;
    bt[BT_EV + k] = lo | (hi << 16);
  }
}

// (Re)build the wave-per-book lists from the pool after another pipeline (or a restore / a fresh set of agents): live
// slots tagged with the member, oldest order (smallest id) first.  One wave per book.
template <int R>
__global__ __launch_bounds__(256) void k_wave_lists_rebuild(DevArgs a, MixedArgs ma, WaveLists wl) {
  const int lane = threadIdx.x & 63;
  const uint32_t book = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (book >= a.n_books) return;
  const uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t id[R], meta[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    id[r] = st[HDR_DW + r * POOL_FIELDS * 64 + 2 * 64 + lane];
    meta[r] = st[HDR_DW + r * POOL_FIELDS * 64 + 4 * 64 + lane];
  }
  for (uint32_t j = 0; j < MAX_MEMBERS; ++j) {
    uint64_t mask[R];
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mask[r] = j < ma.n_desc ? __ballot((meta[r] & 1u) && ((meta[r] >> 8) & 0xFFu) == j + 1) : 0ull;
      any |= mask[r];
    }
    uint16_t* my = wl.list + ((size_t)book * MAX_MEMBERS + j) * wl.cap;
    uint32_t n = 0;
    while (any) {
      uint32_t m = 0xFFFFFFFFu;
#pragma unroll
      for (int r = 0; r < R; ++r) m = min(m, sel(mask[r], id[r], 0xFFFFFFFFu));
      const uint32_t idmin = wave_umin(m);
      int slot = 0;
      any = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint64_t hit = mask[r] & __ballot(id[r] == idmin);
        if (hit) slot = r * 64 + (int)__builtin_ctzll(hit);
        mask[r] &= ~hit;
        any |= mask[r];
      }
      if (lane == 0) my[n] = (uint16_t)slot;
      n += 1;
    }
    if (lane == 0) wl.len[(size_t)book * MAX_MEMBERS + j] = n;
  }
}

}  // namespace bkd
