// mixed_agents.hpp — gfx950 kernel for AgentSets that contain NoiseAgent / MomentumAgent members
// (SURVEY §8f rank 1), alone or together with RandomAgents groups, in declaration order.
//
//   k_run_mixed<R>: one wave per book, n_steps x { every member's update(env, rng); Env::step(rng) } with the
//   book in registers across the launch (like k_run_random).  The members' logic is strictly sequential in the
//   book's RNG stream, so it runs as wave-uniform control flow; the f64 price arithmetic (log-normal offsets around
//   the mid, rounding to ticks, tanh of the momentum) is evaluated redundantly on every lane.
//
// Restated semantics (paths relative to the reference repo):
//   agents::common::{cancel_live_orders, place_*_limit_order, round_price_*}  crates/step_sim/src/agents/common.rs:21-141
//   NoiseAgent::update      crates/step_sim/src/agents/noise_agent.rs:127-176
//   MomentumAgent::update   crates/step_sim/src/agents/momentum_agent.rs:146-208
//   rand / rand_distr sampling (Standard f64, Bernoulli(0.5), Open01, StandardNormal ziggurat, LogNormal): third-party,
//   restated — PARITY UNPINNED against Rust; exp/ln/tanh from pm_math.hpp so that the CPU oracle, which uses the
//   same restatement, can be matched bit-for-bit.
//
// Device representation: every order a member creates gets a pool slot at creation (pending until its New event is
// processed; released again if it never rests), RandomAgents members keep their fixed slot == agent mapping in
// [0, n_fixed); a member's `orders` list is "live slots tagged with the member", iterated in ascending order id
// (= the list's insertion order).
#pragma once
#include "book_device.hpp"
#include "pm_math.hpp"

// every f64 expression below must round exactly like the oracle's (no FMA contraction): file-scope switch
#pragma clang fp contract(off)

namespace bkd {

constexpr int MAX_MEMBERS = 4;
constexpr int H_GFLAGS = 30;  // bit j: member j has a last_price
constexpr int H_GST = 48;     // member j: dwords 48+4j.. = momentum (lo, hi), last_price (lo, hi)

struct MixedDesc {
  uint32_t type;  // 0 RandomAgents, 1 NoiseAgent, 2 MomentumAgent
  uint32_t n;
  uint32_t thr, tick_lo, tick_rng, tick_zone, vol_lo, vol_rng, vol_zone, tick_size;  // RandomAgents (see Group)
  uint32_t thr_limit, thr_market;  // NoiseAgent: (u32 >> 8) < thr  <=>  gen::<f32>() < p
  int32_t keep_thr;                // cancel_live_orders keeps an order iff (u32 >> 8) > keep_thr  <=>  gen::<f32>() > p_cancel
  uint32_t trade_vol;
  uint32_t slot_base;              // RandomAgents: first fixed slot
  uint32_t pad;
  double mu, sigma, decay, demand, scale, order_ratio, n_f, tick_f;
};
static_assert(sizeof(MixedDesc) == 128, "MixedDesc layout");

struct MixedArgs {
  const MixedDesc* descs;
  uint32_t n_desc;
  uint32_t n_fixed;  // pool slots reserved for RandomAgents members
  // MarketEnv mode (NoiseMarketAgent / MomentumMarketAgent / RandomMarketAgents members): the asset each member trades
  // and the fixed slots reserved in each asset's book
  uint32_t asset[MAX_MEMBERS];
  uint32_t n_fixed_a[MAX_ASSETS];
};

#define ZIG_TABLE_BEGIN(name) __constant__ const double name[257] = {
#define ZIG_TABLE_END };
#include "zig_norm_tables.inc"
#undef ZIG_TABLE_BEGIN
#undef ZIG_TABLE_END

// pin a wave-uniform 64-bit value to scalar registers (the f64 code around the generator otherwise drags the RNG
// arithmetic onto the vector unit: v_mad_u64_u32 chains instead of s_mul / s_lshl)
__device__ __forceinline__ uint64_t pin_scalar(uint64_t x) {
  return mk64(rfl((uint32_t)x), rfl((uint32_t)(x >> 32)));
}
__device__ __forceinline__ uint64_t next_u64(Rng& rng) {
  rng.s0 = pin_scalar(rng.s0);
  rng.s1 = pin_scalar(rng.s1);
  uint64_t r = rng.s0 * 5ull;
  r = (r << 7) | (r >> 57);
  r *= 9ull;
  const uint64_t t1 = rng.s1 ^ rng.s0;
  rng.s0 = ((rng.s0 << 24) | (rng.s0 >> 40)) ^ t1 ^ (t1 << 16);
  rng.s1 = (t1 << 37) | (t1 >> 27);
  return r;
}
// wave-uniform double out of the VALU back into scalar registers (keeps later control flow uniform)
__device__ __forceinline__ double uni(double x) {
  const uint64_t b = pm::to_bits(x);
  return pm::from_bits(mk64(rfl((uint32_t)b), rfl((uint32_t)(b >> 32))));
}
__device__ __forceinline__ double gen_f64(Rng& rng) {  // rand Standard f64
  return static_cast<double>(next_u64(rng) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double gen_open01(Rng& rng) {  // rand Open01 f64
  const double v = pm::from_bits(0x3FF0000000000000ull | (next_u64(rng) >> 12));
  return v - (1.0 - 2.220446049250313e-16 / 2.0);
}
// rand_distr StandardNormal: 256-layer ziggurat (utils.rs ziggurat, symmetric)
__device__ __forceinline__ double sample_standard_normal(Rng& rng) {
  for (;;) {
    const uint64_t bits = next_u64(rng);
    const uint32_t i = (uint32_t)bits & 0xffu;
    const double u = pm::from_bits(0x4000000000000000ull | (bits >> 12)) - 3.0;
    const double x = uni(u * ZIG_NORM_X[i]);
    if (pm::fabs_(x) < ZIG_NORM_X[i + 1]) return x;
    if (i == 0) {
      const double R = 3.654152885361008796;
      double xx = 1.0, yy = 0.0;
      while (-2.0 * yy < xx * xx) {
        const double x_ = gen_open01(rng);
        const double y_ = gen_open01(rng);
        xx = uni(pm::log(x_) / R);
        yy = uni(pm::log(y_));
      }
      return (u < 0.0) ? xx - R : R - xx;
    }
    const double lhs = uni(ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * gen_f64(rng));
    if (lhs < uni(pm::exp(-x * x / 2.0))) return x;
  }
}
__device__ __forceinline__ uint32_t price_from_f64(double p) {  // clamp(0, u32::MAX) as u32 (NaN -> 0)
  if (!(p == p)) return 0u;
  if (p < 0.0) p = 0.0;
  if (p > 4294967295.0) p = 4294967295.0;
  return static_cast<uint32_t>(p);
}
__device__ __forceinline__ uint32_t round_price_up(double p, double tick) {  // common.rs:21-25
  return price_from_f64(pm::ceil_(p / tick) * tick);
}
__device__ __forceinline__ uint32_t round_price_down(double p, double tick) {  // common.rs:36-40
  return price_from_f64(pm::floor_(p / tick) * tick);
}
// `gen::<f64>() < p` as an integer threshold on the 53 random bits: k * 2^-53 < p  <=>  k < ceil(p * 2^53)
__device__ __forceinline__ uint64_t thr53(double p) {
  if (!(p > 0.0)) return 0ull;
  if (p >= 1.0) return 9007199254740992ull;
  return static_cast<uint64_t>(static_cast<int64_t>(pm::ceil_(p * 9007199254740992.0)));
}

template <int R>
struct MixedCtx {
  uint32_t ev[R];     // event list (slot indices)
  uint32_t owner[R];  // member index + 1 of the limit order resting / pending in the slot, 0 = none
  uint32_t n_fixed;   // slots [0, n_fixed) belong to RandomAgents members; the rest is allocated dynamically
  uint32_t tick;      // the book's tick size (create_order's check)
  uint32_t n_ev;
};

// Env::place_order from a member: id assignment + New event; the order is parked in a free pool slot
template <int R>
__device__ __forceinline__ void mixed_create(Book<R>& B, MixedCtx<R>& C, int lane, bool is_bid, uint32_t price,
                                             uint32_t vol, uint32_t owner_tag) {
  // create_order's tick check (orderbook.rs:367-382); the reference `.unwrap()`s the Err (common.rs:107,140), i.e.
  // panics: flagged, and like an Err nothing is created (no id, no event).  Reachable when a log-normal offset drives
  // the price to the u32::MAX clamp on a book whose tick does not divide it.  Market orders carry no price.
  if (owner_tag != 0 && price % C.tick != 0) {
    B.flags |= FLAG_PRICE_TICK;
    return;
  }
  const uint32_t id = B.next_id;
  B.next_id += 1;  // create_order consumes the id (orderbook.rs:363)
  int slot = -1;
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    // dynamic slots of this register: index >= n_fixed (rebuilt here: 2R fewer live SGPRs than a mask array)
    const uint32_t lo = C.n_fixed > 64u * r ? C.n_fixed - 64u * r : 0u;
    const uint64_t dyn = lo >= 64u ? 0ull : (~0ull << lo);
    const uint64_t fr = ~(B.live[r] | B.pend[r]) & dyn;
    if (fr) slot = r * 64 + (int)__builtin_ctzll(fr);
  }
  if (slot < 0) {
    B.flags |= FLAG_POOL_OVERFLOW;  // reported, never silent: the order (and its event) is dropped
    return;
  }
  slot_write<R>(B.price, slot, price);
  slot_write<R>(B.vol, slot, vol);
  slot_write<R>(B.id, slot, id);
  slot_write<R>(C.owner, slot, owner_tag);
  mask_set<R>(B.bid, slot, is_bid);
  mask_set<R>(B.pend, slot, true);
  slot_write<R>(C.ev, C.n_ev, (uint32_t)slot);
  C.n_ev += 1;
  (void)lane;
}

// common::cancel_live_orders (common.rs:54-76): the member's Active orders in list (= id) order, one f32 draw each;
// `draw > p_cancel` keeps the order, otherwise its cancellation is queued
template <int R>
__device__ __forceinline__ void mixed_cancel_live(Book<R>& B, MixedCtx<R>& C, Rng& rng, int lane, uint32_t tag,
                                                  int32_t keep_thr) {
  uint64_t mask[R];
  uint64_t any = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mask[r] = B.live[r] & __ballot(C.owner[r] == tag);
    any |= mask[r];
  }
  while (any) {
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < R; ++r) m = min(m, sel(mask[r], B.id[r], 0xFFFFFFFFu));
    const uint32_t idmin = wave_umin(m);
    int slot = 0;
    any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t hit = mask[r] & __ballot(B.id[r] == idmin);
      if (hit) slot = r * 64 + (int)__builtin_ctzll(hit);
      mask[r] &= ~hit;
      any |= mask[r];
    }
    const uint32_t x = rng.next_u32();
    if (!((int32_t)(x >> 8) > keep_thr)) {  // not kept -> env.cancel_order(id)
      slot_write<R>(C.ev, C.n_ev, (uint32_t)slot);
      C.n_ev += 1;
    }
  }
  (void)lane;
}

// member state that lives across steps: momentum / last mid price per member (momentum_agent.rs:108-117)
struct MixedState {
  uint32_t gflags;
  double g_mom[MAX_MEMBERS], g_last[MAX_MEMBERS];
};
__device__ __forceinline__ void mixed_load_state(MixedState& S, const uint32_t* st, int lane) {
  const uint32_t hdr = st[lane];
  S.gflags = rdl(hdr, H_GFLAGS);
#pragma unroll
  for (int j = 0; j < MAX_MEMBERS; ++j) {
    S.g_mom[j] = pm::from_bits(mk64(rdl(hdr, H_GST + 4 * j), rdl(hdr, H_GST + 4 * j + 1)));
    S.g_last[j] = pm::from_bits(mk64(rdl(hdr, H_GST + 4 * j + 2), rdl(hdr, H_GST + 4 * j + 3)));
  }
}
// after store_book: member state into the header, owner tags (+ live / bid / pend bits) into the pool's meta words
template <int R>
__device__ __forceinline__ void mixed_store_state(const MixedState& S, const Book<R>& B, const MixedCtx<R>& C,
                                                  uint32_t* st, int lane) {
  uint32_t h2 = st[lane];
  auto put = [&](int idx, uint32_t v) { h2 = (lane == idx) ? v : h2; };
  put(H_GFLAGS, S.gflags);
#pragma unroll
  for (int j = 0; j < MAX_MEMBERS; ++j) {
    const uint64_t mb = pm::to_bits(S.g_mom[j]), lb = pm::to_bits(S.g_last[j]);
    put(H_GST + 4 * j, (uint32_t)mb);
    put(H_GST + 4 * j + 1, (uint32_t)(mb >> 32));
    put(H_GST + 4 * j + 2, (uint32_t)lb);
    put(H_GST + 4 * j + 3, (uint32_t)(lb >> 32));
  }
  st[lane] = h2;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t* p = st + HDR_DW + r * POOL_FIELDS * 64 + 4 * 64;
    p[lane] = (lane_bit(B.live[r]) ? 1u : 0u) | (lane_bit(B.bid[r]) ? 2u : 0u) | (lane_bit(B.pend[r]) ? 4u : 0u) |
              (C.owner[r] << 8);
  }
}
template <int R>
__device__ __forceinline__ void mixed_load_ctx(MixedCtx<R>& C, const uint32_t* st, const MixedArgs& ma, int lane) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    C.ev[r] = 0;
    C.owner[r] = (st[HDR_DW + r * POOL_FIELDS * 64 + 4 * 64 + lane] >> 8) & 0xFFu;
  }
  C.n_fixed = ma.n_fixed;
  C.n_ev = 0;
}

// One step's agents.update(env, rng) for every member in declaration order (crates/macros/src/lib.rs:57-73) followed
// by transactions.shuffle(rng) (env.rs:121): fills C.ev / C.n_ev and parks the new orders in the pool.
template <int R>
__device__ __forceinline__ void mixed_update_and_shuffle(Book<R>& B, MixedCtx<R>& C, Rng& rng, const MixedArgs& ma,
                                                         MixedState& S, int lane) {
  C.n_ev = 0;
  // OrderBook::mid_price (orderbook.rs:272-276) of the book as the agents see it (updates only queue events)
  double mid;
  {
    uint32_t mb = 0u, mk = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mb = max(mb, sel(B.live[r] & B.bid[r], B.price[r], 0u));
      mk = min(mk, sel(B.live[r] & ~B.bid[r], B.price[r], 0xFFFFFFFFu));
    }
    const uint32_t bid = wave_umax(mb), ask = wave_umin(mk);
    mid = static_cast<double>(bid) + 0.5 * static_cast<double>(ask - bid);
  }
  for (uint32_t j = 0; j < ma.n_desc; ++j) {
    const MixedDesc D = ma.descs[j];
    if (D.type == 0) {
      // ---- RandomAgents::update (random_agent.rs:85-119), fixed slots [slot_base, slot_base + n)
      for (uint32_t i = 0; i < D.n; ++i) {
        const uint32_t n = D.slot_base + i;
        const uint32_t x = rng.next_u32();
        if ((x >> 8) < D.thr) {
          slot_write<R>(C.ev, C.n_ev, n);
          C.n_ev += 1;
          if (!mask_test<R>(B.live, n)) {
            const uint32_t side = rng.below(2u, 0x7FFFFFFFu);
            const uint32_t tick = D.tick_lo + rng.below(D.tick_rng, D.tick_zone);
            const uint32_t vol = D.vol_lo + rng.below(D.vol_rng, D.vol_zone);
            slot_write<R>(B.price, n, tick * D.tick_size);
            slot_write<R>(B.vol, n, vol);
            slot_write<R>(B.id, n, B.next_id);
            B.next_id += 1;
            mask_set<R>(B.bid, n, side != 0);
            mask_set<R>(B.pend, n, true);
          }
        }
      }
    } else {
      const uint32_t tag = j + 1;
      mixed_cancel_live<R>(B, C, rng, lane, tag, D.keep_thr);
      if (D.type == 1) {
        // ---- NoiseAgent::update (noise_agent.rs:127-176)
        for (uint32_t t = 0; t < D.n; ++t) {
          if ((rng.next_u32() >> 8) < D.thr_limit) {                // gen::<f32>() < p_limit
            const bool buy = next_u64(rng) < 0x8000000000000000ull;  // gen_bool(0.5)
            const double dist = pm::fabs_(uni(pm::exp(D.mu + D.sigma * sample_standard_normal(rng))));
            const uint32_t price =
                rfl(buy ? round_price_down(mid - dist, D.tick_f) : round_price_up(mid + dist, D.tick_f));
            mixed_create<R>(B, C, lane, buy, price, D.trade_vol, tag);
          }
          if ((rng.next_u32() >> 8) < D.thr_market) {                // gen::<f32>() < p_market
            const bool buy = next_u64(rng) < 0x8000000000000000ull;
            mixed_create<R>(B, C, lane, buy, buy ? 0xFFFFFFFFu : 0u, D.trade_vol, 0u);
          }
        }
      } else {
        // ---- MomentumAgent::update (momentum_agent.rs:146-208)
        double m = 0.0, p_market = 0.0;
        if ((S.gflags >> j) & 1u) {
          double gm = S.g_mom[0], gl = S.g_last[0];
#pragma unroll
          for (int q = 1; q < MAX_MEMBERS; ++q) {
            gm = ((uint32_t)q == j) ? S.g_mom[q] : gm;
            gl = ((uint32_t)q == j) ? S.g_last[q] : gl;
          }
          m = uni(gm * (1.0 - D.decay) + D.decay * (mid - gl));
          p_market = uni(D.demand * pm::tanh(D.scale * m) / D.n_f);
        }
        uint64_t thr_l, thr_m;
        {
          const double p_limit = D.order_ratio * p_market;
          thr_l = thr53(p_limit);
          thr_m = thr53(p_market);
          thr_l = mk64(rfl((uint32_t)thr_l), rfl((uint32_t)(thr_l >> 32)));
          thr_m = mk64(rfl((uint32_t)thr_m), rfl((uint32_t)(thr_m >> 32)));
        }
        const int sgn = (m > 0.0) ? 1 : ((m < 0.0) ? -1 : 0);
        for (uint32_t t = 0; t < D.n; ++t) {
          if ((next_u64(rng) >> 11) < thr_l) {  // gen::<f64>() < p_limit
            if (sgn != 0) {
              const double dist = pm::fabs_(uni(pm::exp(D.mu + D.sigma * sample_standard_normal(rng))));
              const uint32_t price =
                  rfl(sgn > 0 ? round_price_down(mid - dist, D.tick_f) : round_price_up(mid + dist, D.tick_f));
              mixed_create<R>(B, C, lane, sgn > 0, price, D.trade_vol, tag);
            }
          }
          if ((next_u64(rng) >> 11) < thr_m) {  // gen::<f64>() < p_market
            if (sgn != 0) mixed_create<R>(B, C, lane, sgn > 0, sgn > 0 ? 0xFFFFFFFFu : 0u, D.trade_vol, 0u);
          }
        }
#pragma unroll
        for (int q = 0; q < MAX_MEMBERS; ++q) {
          if ((uint32_t)q == j) {
            S.g_mom[q] = m;
            S.g_last[q] = mid;
          }
        }
        S.gflags |= 1u << j;
      }
    }
  }
  // ---- Env::step's shuffle (env.rs:121)
  for (uint32_t i = C.n_ev; i-- > 1;) {
    const uint32_t jx = rng.below(i + 1);
    const uint32_t ai = slot_read<R>(C.ev, i), aj = slot_read<R>(C.ev, jx);
    slot_write<R>(C.ev, i, aj);
    slot_write<R>(C.ev, jx, ai);
  }
}

// fused: n_steps x { members' update + shuffle; event loop + snapshot } with the book in registers (small batches)
template <int R>
__global__ __launch_bounds__(256) void k_run_mixed(DevArgs a, MixedArgs ma, uint64_t first_step, uint32_t n_steps) {
  __shared__ uint32_t lds[4][LDS_DW_PER_WAVE];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t book = rfl(blockIdx.x * 4 + wv);
  if (book >= a.n_books) return;
  uint32_t* st = a.state + (size_t)book * a.state_stride;

  Book<R> B;
  Rng rng;
  load_book<R>(B, rng, st, lane);
  MixedCtx<R> C;
  MixedState S;
  mixed_load_state(S, st, lane);
  mixed_load_ctx<R>(C, st, ma, lane);
  C.tick = a.tick_size;
  uint32_t last_ntr = 0, last_nev = 0;
  for (uint32_t s = 0; s < n_steps; ++s) {
    mixed_update_and_shuffle<R>(B, C, rng, ma, S, lane);
    last_ntr = step_from_list<R>(B, a, book, lane, C.ev, C.n_ev, lds[wv],
                                 a.hist_cap ? (a.hist_slot0 + s) % a.hist_cap : 0u, s + 1 == n_steps || a.hist_cap == 0,
                                 a.tick_div, B.pend, last_nev);
  }
  store_book<R>(B, rng, st, lane, first_step + n_steps, last_ntr, last_nev);
  mixed_store_state<R>(S, B, C, st, lane);
}

// split: one step's members' update + shuffle per book; the new orders stay parked in the pool (pend bit in the
// stored meta word), the shuffled event list goes to the book's step batch for k_step_batch<R, false, true>.
template <int R>
__global__ __launch_bounds__(256) void k_agents_mixed(DevArgs a, MixedArgs ma) {
  const int lane = threadIdx.x & 63;
  const uint32_t book = rfl(a.book_begin + blockIdx.x * 4 + (threadIdx.x >> 6));
  if (book >= a.book_end) return;
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  uint32_t* bt = a.batch + (size_t)book * a.batch_stride;
  Book<R> B;
  Rng rng;
  load_book<R>(B, rng, st, lane);
  MixedCtx<R> C;
  MixedState S;
  mixed_load_state(S, st, lane);
  mixed_load_ctx<R>(C, st, ma, lane);
  C.tick = a.tick_size;
  mixed_update_and_shuffle<R>(B, C, rng, ma, S, lane);
  // the step counter / last-step figures are k_step_batch's to write: keep the header's values
  const uint32_t hdr = st[lane];
  store_book<R>(B, rng, st, lane, mk64(rdl(hdr, H_STEPS_LO), rdl(hdr, H_STEPS_HI)), rdl(hdr, H_LAST_NTRADES),
                rdl(hdr, H_LAST_NEVENTS));
  mixed_store_state<R>(S, B, C, st, lane);
  if (lane == 0) bt[BT_NEV] = C.n_ev;
#pragma unroll
  for (int r = 0; r < R; ++r) reinterpret_cast<uint16_t*>(bt + BT_EV)[r * 64 + lane] = (uint16_t)C.ev[r];
}

// ==================================================================================
// Lane-per-book members' update (the split pipeline's default for these AgentSets): 64 books per wave, every lane runs
// its book's agents.update + shuffle as ordinary per-lane code — the RNG stream, the ziggurat / exp / tanh arithmetic and
// the order creation are all SIMT across books instead of scalar code on one wave per book.  What a lane needs of its
// book: the touch prices (mid price) from the last level-2 record, the pool's live masks (one 64-byte line of the
// header), and the member's `orders` list, kept in creation order in an aux buffer laid out [member][entry][book] so
// that the lanes' accesses coalesce.  New orders are written straight into free pool slots with their pend bit; the
// event kernel is k_step_batch<R, false, true>.
//
// A slot counts as free when it is neither live, nor pending, nor still referenced by some member's list (`inl` masks):
// a dead order's entry is purged at its member's next update, so "live bit set" alone identifies a list entry's order.
// ==================================================================================
struct MixedLists {
  uint16_t* list;  // [MAX_MEMBERS][cap][n_books]: pool slots of the member's orders, oldest first
  uint32_t* len;   // [MAX_MEMBERS][n_books]
  uint32_t* inl;   // [2R][n_books]: slots referenced by the lists, 32 per word
  uint32_t cap;    // entries per member = pool size
  uint32_t n_books;
  uint32_t n_units;  // lists are per market (= book when assets == 1)
};

struct LaneRng {  // xoroshiro128** per lane
  uint64_t s0, s1;
  __device__ __forceinline__ uint64_t next_u64() {
    uint64_t r = s0 * 5ull;
    r = (r << 7) | (r >> 57);
    r *= 9ull;
    const uint64_t t1 = s1 ^ s0;
    s0 = ((s0 << 24) | (s0 >> 40)) ^ t1 ^ (t1 << 16);
    s1 = (t1 << 37) | (t1 >> 27);
    return r;
  }
  __device__ __forceinline__ uint32_t next_u32() { return (uint32_t)next_u64(); }
  __device__ __forceinline__ uint32_t below(uint32_t range, uint32_t zone) {
    for (;;) {
      const uint64_t m = (uint64_t)next_u32() * range;
      if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
  }
  __device__ __forceinline__ double f64() { return static_cast<double>(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  __device__ __forceinline__ double open01() {
    const double v = pm::from_bits(0x3FF0000000000000ull | (next_u64() >> 12));
    return v - (1.0 - 2.220446049250313e-16 / 2.0);
  }
  // rand_distr StandardNormal (256-layer ziggurat), as above; zx / zf = the tables (staged in LDS by the caller: a
  // lane's index is its own, and a table miss would be a full memory round trip in a latency-bound kernel)
  __device__ __forceinline__ double std_normal(const double* zx, const double* zf) {
    for (;;) {
      const uint64_t bits = next_u64();
      const uint32_t i = (uint32_t)bits & 0xffu;
      const double u = pm::from_bits(0x4000000000000000ull | (bits >> 12)) - 3.0;
      const double x = u * zx[i];
      if (pm::fabs_(x) < zx[i + 1]) return x;
      if (i == 0) {
        const double Rz = 3.654152885361008796;
        double xx = 1.0, yy = 0.0;
        while (-2.0 * yy < xx * xx) {
          const double x_ = open01();
          const double y_ = open01();
          xx = pm::log(x_) / Rz;
          yy = pm::log(y_);
        }
        return (u < 0.0) ? xx - Rz : Rz - xx;
      }
      const double lhs = zf[i + 1] + (zf[i] - zf[i + 1]) * f64();
      if (lhs < pm::exp(-x * x / 2.0)) return x;
    }
  }
};

// MKT: the lane owns a MARKET (books [b*M, b*M + M)): one RNG stream, one event queue; member j trades asset
// ma.asset[j] (NoiseMarketAgent / MomentumMarketAgent / RandomMarketAgents: noise_agent.rs:281-339,
// momentum_agent.rs:328-396, random_agent.rs:204-247); event entries carry the asset in bits 12..14.
constexpr int MLQ_CAP = 128;  // deferred-price queue of k_agents_mixed_lanes: drained at 64, at most 64 more per agent turn
                               // (2 x 80.9 KB of LDS still fit a CU at R = 8)
constexpr size_t mixed_lanes_lds_bytes(int R, bool mkt) {
  return (size_t)(32 * R * 64 + 2 * (2 * R * 64) + 2 * (mkt ? MAX_ASSETS * 64 : 64)) * 4 + 2 * 257 * sizeof(double) +
         MLQ_CAP * (2 * sizeof(double) + 4);
}

template <int R, bool MKT>
__global__ __launch_bounds__(64) void k_agents_mixed_lanes(DevArgs a, MixedArgs ma, MixedLists ml) {
  // dynamic LDS (up to ~82 KB at R = 8; MI355X allows 160 KB per workgroup), see mixed_lanes_lds_bytes():
  //   event list of lane l: list[k * 64 + l] (u16) | live / listed masks of the open book, word w of lane l at
  //   [w * 64 + l] | per-asset allocation cursors (MKT) | ziggurat tables.  Everything a lane looks up per order lives
  //   here: with one wave per SIMD every global round trip is exposed latency.
  extern __shared__ uint32_t smem[];
  uint16_t* list = reinterpret_cast<uint16_t*>(smem);
  uint32_t* lds_live = smem + 32 * R * 64;
  uint32_t* lds_inl = lds_live + 2 * R * 64;
  uint32_t* cur_w = lds_inl + 2 * R * 64;
  uint32_t* cur_c = cur_w + (MKT ? MAX_ASSETS * 64 : 64);
  double* zx = reinterpret_cast<double*>(cur_c + (MKT ? MAX_ASSETS * 64 : 64));
  double* zf = zx + 257;
  // Deferred limit prices (not MKT): a member's turn draws and decides for the 64 books in lockstep, but only the few
  // lanes whose agent places an order need exp() and the tick rounding - 150 f64 instructions at 5-9 % lane
  // utilisation, 60 % of this kernel's vector work (profiles/r02/pmc_c5m.json).  Those lanes create the order without a
  // price and queue {exp argument, mid, book lane, slot}; whenever 64 entries are waiting, ALL lanes take one each.
  // Same arithmetic on the same operands, so the same bits.  Orders that might reach the u32::MAX clamp (the one case
  // whose outcome - Err, nothing created - changes what is drawn next) keep the in-line path.
  double* q_arg = zf + 257;
  double* q_mid = q_arg + MLQ_CAP;
  uint32_t* q_info = reinterpret_cast<uint32_t*>(q_mid + MLQ_CAP);
  const int lane = threadIdx.x;
  for (int i = lane; i < 257; i += 64) {
    zx[i] = ZIG_NORM_X[i];
    zf[i] = ZIG_NORM_F[i];
  }
  __syncthreads();
  const uint32_t b = a.book_begin + blockIdx.x * 64 + lane;  // book, or market when MKT
  if (b >= a.book_end) return;
  const uint32_t M = MKT ? a.assets : 1u;
  uint32_t* st0 = a.state + (size_t)b * M * a.state_stride;
  uint32_t* bt = a.batch + (size_t)b * M * a.batch_stride;
  const size_t NB = ml.n_books, NU = ml.n_units;
  const uint64_t act = __builtin_amdgcn_ballot_w64(true);  // the lanes with a book (all 64 but in the last workgroup)
  const uint32_t n_act = __builtin_popcountll(act);
  const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));

  LaneRng rng;
  {
    const uint2 x0 = *reinterpret_cast<const uint2*>(st0 + H_S0_LO);
    const uint2 x1 = *reinterpret_cast<const uint2*>(st0 + H_S1_LO);
    rng.s0 = mk64(x0.x, x0.y);
    rng.s1 = mk64(x1.x, x1.y);
  }
  uint32_t n_ev = 0;
  // the book the current member trades on
  uint32_t* st = st0;
  uint32_t bk = b * M, asset = 0, n_fixed = MKT ? ma.n_fixed_a[0] : ma.n_fixed;
  uint32_t next_id = 0, new_flags = 0;
  // slot allocation cursor: word `wcur` of the occupancy (live | listed | allocated this step), lowest free bit first
  uint32_t wcur = 0, cw = 0xFFFFFFFFu;
  // stage the open book's live masks (one 64-byte line of its header) and listed masks in LDS / write the latter back
  auto stage_masks = [&]() {
#pragma unroll
    for (int q = 0; q < (2 * R) / 4; ++q) {
      const uint4 v = *reinterpret_cast<const uint4*>(st + H_LIVE0 + 4 * q);
      lds_live[(4 * q + 0) * 64 + lane] = v.x;
      lds_live[(4 * q + 1) * 64 + lane] = v.y;
      lds_live[(4 * q + 2) * 64 + lane] = v.z;
      lds_live[(4 * q + 3) * 64 + lane] = v.w;
    }
    if (R == 1) {
      lds_live[lane] = st[H_LIVE0];
      lds_live[64 + lane] = st[H_LIVE0 + 1];
    }
#pragma unroll
    for (int w = 0; w < 2 * R; ++w) lds_inl[w * 64 + lane] = ml.inl[(size_t)w * NB + bk];
  };
  auto unstage_masks = [&]() {
#pragma unroll
    for (int w = 0; w < 2 * R; ++w) ml.inl[(size_t)w * NB + bk] = lds_inl[w * 64 + lane];
  };
  auto load_word = [&](uint32_t w) -> uint32_t {
    uint32_t v = lds_live[w * 64 + lane] | lds_inl[w * 64 + lane];
    if (n_fixed > 32u * w) v |= (n_fixed - 32u * w >= 32u) ? 0xFFFFFFFFu : ((1u << (n_fixed - 32u * w)) - 1u);
    return v;
  };
  if (MKT) {
    for (uint32_t as = 0; as < M; ++as) cur_w[as * 64 + lane] = 0xFFFFFFFFu;  // not opened yet
  } else {
    stage_masks();
    wcur = n_fixed >> 5;
    if (wcur < 2u * R) cw = load_word(wcur);
    next_id = st[H_NEXT_ID];
  }
  auto open_book = [&](uint32_t as) {  // MKT: switch to the member's asset
    asset = as;
    st = st0 + (size_t)as * a.state_stride;
    bk = b * M + as;
    n_fixed = ma.n_fixed_a[as];
    next_id = st[H_NEXT_ID];
    new_flags = 0;
    stage_masks();
    wcur = cur_w[as * 64 + lane];
    cw = cur_c[as * 64 + lane];
    if (wcur == 0xFFFFFFFFu) {
      wcur = n_fixed >> 5;
      cw = wcur < 2u * R ? load_word(wcur) : 0xFFFFFFFFu;
    }
  };
  auto close_book = [&]() {
    unstage_masks();
    st[H_NEXT_ID] = next_id;
    if (new_flags) st[H_FLAGS] |= new_flags;
    cur_w[asset * 64 + lane] = wcur;
    cur_c[asset * 64 + lane] = cw;
  };
  auto pool_ptr = [&](uint32_t slot, int field) -> uint32_t* {
    return st + HDR_DW + (slot >> 6) * (POOL_FIELDS * 64) + field * 64 + (slot & 63u);
  };
  // The step's event list holds 64 * R entries per unit.  One book can never queue more (every event refers to its own
  // pool slot); a MARKET's joint queue can, when its books together keep more than one pool's worth of orders in play:
  // the event is then dropped and the book flagged (BK_FLAG_EVENT_OVERFLOW) - never written past the list.
  auto event_room = [&]() -> bool {
    if (n_ev < 64u * R) return true;
    new_flags |= FLAG_EVENT_OVERFLOW;
    return false;
  };
  auto push_event = [&](uint32_t slot) {
    list[n_ev * 64 + lane] = (uint16_t)(slot | (asset << 12));
    n_ev += 1;
  };
  // Env::place_order from a member: id + New event; returns the slot (or 0xFFFF when the pool is full: flagged)
  auto create = [&](bool is_bid, uint32_t price, uint32_t vol, uint32_t tag, bool deferred = false) -> uint32_t {
    // create_order's tick check (orderbook.rs:367-382): the reference `.unwrap()`s the Err, i.e. panics — flagged, and
    // like an Err nothing is created (see mixed_create).  Market orders (tag 0 here) carry no price.  (A deferred price
    // is a tick multiple below the clamp by construction.)
    if (!deferred && tag != 0 && price % (MKT ? a.asset_tick[asset] : a.tick_size) != 0) {
      new_flags |= FLAG_PRICE_TICK;
      return 0xFFFFu;
    }
    if (!event_room()) return 0xFFFFu;
    const uint32_t id = next_id;
    next_id += 1;  // create_order consumes the id (orderbook.rs:363)
    while (cw == 0xFFFFFFFFu && wcur < 2u * R) {
      wcur += 1;
      if (wcur < 2u * R) cw = load_word(wcur);
    }
    if (wcur >= 2u * R) {
      new_flags |= FLAG_POOL_OVERFLOW;  // reported, never silent: the order (and its event) is dropped
      return 0xFFFFu;
    }
    const uint32_t bit = __builtin_ctz(~cw);
    cw |= 1u << bit;
    const uint32_t slot = wcur * 32u + bit;
    if (!deferred) *pool_ptr(slot, 0) = price;
    *pool_ptr(slot, 1) = vol;
    *pool_ptr(slot, 2) = id;
    *pool_ptr(slot, 4) = 4u | (is_bid ? 2u : 0u) | (tag << 8);  // pending New
    push_event(slot);
    return slot;
  };
  // ---- deferred limit prices (see q_arg above)
  const uint32_t wave_b0 = a.book_begin + blockIdx.x * 64;
  // limit order at mid -/+ exp(arg): in line when a sell might reach the clamp (or in a market), otherwise created
  // without its price and handed to `queue_turn` below
  uint32_t qc = 0;  // entries waiting (wave-uniform)
  bool pend_q = false;
  double pend_arg = 0.0;
  uint32_t pend_info = 0;
  auto place_limit = [&](bool buy, double arg, double mid, double lnslack, const MixedDesc& D, uint32_t tag) -> uint32_t {
    if (MKT || !(buy || arg < lnslack)) {
      const double dist = pm::fabs_(pm::exp(arg));
      const uint32_t price = buy ? round_price_down(mid - dist, D.tick_f) : round_price_up(mid + dist, D.tick_f);
      return create(buy, price, D.trade_vol, tag);
    }
    const uint32_t slot = create(buy, 0u, D.trade_vol, tag, true);
    if (slot != 0xFFFFu) {
      pend_q = true;
      pend_arg = arg;
      pend_info = (uint32_t)lane | (slot << 6) | (buy ? 0x8000u : 0u);
    }
    return slot;
  };
  // all lanes, 64 (or, with `all`, whatever is left) queued orders: one each
  auto drain = [&](bool all, double tick_f) {
    if (MKT) return;
    while (qc >= (all ? 1u : 64u)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t take = qc < 64u ? qc : 64u, base = qc - take;
      for (uint32_t e = my_rank; e < take; e += n_act) {
        const double arg = q_arg[base + e], mid_e = q_mid[base + e];
        const uint32_t info = q_info[base + e], slot = (info >> 6) & 0x1FFu;
        const double dist = pm::fabs_(pm::exp(arg));
        const uint32_t price = (info & 0x8000u) ? round_price_down(mid_e - dist, tick_f) : round_price_up(mid_e + dist, tick_f);
        uint32_t* sb = a.state + (size_t)(wave_b0 + (info & 63u)) * a.state_stride;
        sb[HDR_DW + (slot >> 6) * (POOL_FIELDS * 64) + (slot & 63u)] = price;
      }
      qc = base;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  // at the end of an agent's turn, where the lanes are together again: the orders placed in it join the queue
  auto queue_turn = [&](double mid, double tick_f) {
    if (MKT) return;
    const uint64_t w = __builtin_amdgcn_ballot_w64(pend_q);
    if (w == 0) return;
    if (pend_q) {
      const uint32_t q = qc + __builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0u));
      q_arg[q] = pend_arg;
      q_mid[q] = mid;
      q_info[q] = pend_info;
    }
    pend_q = false;
    qc += __builtin_popcountll(w);
    drain(false, tick_f);
  };
  // a sell at mid + exp(arg), rounded UP to the tick, stays below the u32::MAX clamp when arg < lnslack
  auto clamp_bound = [&](double mid, double tick_f) -> double {
    const double slack = 4294967295.0 - mid - 2.0 * tick_f - 1.0;
    return slack > 1.0 ? pm::log(slack) - 1e-9 : -1e300;
  };

  for (uint32_t j = 0; j < ma.n_desc; ++j) {  // members in declaration order (crates/macros/src/lib.rs:57-73)
    const MixedDesc D = ma.descs[j];
    if (MKT) open_book(ma.asset[j]);
    // OrderBook::mid_price (orderbook.rs:272-276): the touches of the book's last level-2 record (updates only queue
    // events, so the book is still the one that record describes)
    double mid;
    {
      const uint32_t* l2 = a.l2_last + (size_t)bk * a.l2_width;
      const uint32_t bid = l2[1], ask = l2[2];
      mid = static_cast<double>(bid) + 0.5 * static_cast<double>(ask - bid);
    }
    if (D.type == 0) {
      // ---- RandomAgents::update (random_agent.rs:85-119), fixed slots [slot_base, slot_base + n)
      uint32_t lw = 0;
      for (uint32_t i = 0; i < D.n; ++i) {
        const uint32_t n = D.slot_base + i;
        if (i == 0 || (n & 31u) == 0) lw = lds_live[(n >> 5) * 64 + lane];
        const uint32_t x = rng.next_u32();
        if ((x >> 8) < D.thr && event_room()) {
          push_event(n);
          if (!((lw >> (n & 31u)) & 1u)) {
            const uint32_t side = rng.below(2u, 0x7FFFFFFFu);
            const uint32_t tick = D.tick_lo + rng.below(D.tick_rng, D.tick_zone);
            const uint32_t vol = D.vol_lo + rng.below(D.vol_rng, D.vol_zone);
            *pool_ptr(n, 0) = tick * D.tick_size;
            *pool_ptr(n, 1) = vol;
            *pool_ptr(n, 2) = next_id;
            *pool_ptr(n, 4) = 4u | (side ? 2u : 0u);
            next_id += 1;
          }
        }
      }
      if (MKT) close_book();
      continue;
    }
    // ---- common::cancel_live_orders (common.rs:56-75): Active orders of the list in order, one f32 draw each
    const uint32_t tag = j + 1;
    uint16_t* my = ml.list + (size_t)j * ml.cap * NU + b;
    uint32_t len = ml.len[(size_t)j * NU + b], keep = 0;
    {
      // entries are fetched eight at a time BEFORE any of them is processed: the list is compacted in place (writes
      // never pass the read position), and a load issued after a store to the same array would wait for it
      uint32_t lw = 0, lwi = 0xFFFFFFFFu;
      for (uint32_t i0 = 0; i0 < len; i0 += 8) {
        uint32_t ent[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ent[q] = (i0 + q < len) ? my[(size_t)(i0 + q) * NU] : 0xFFFFu;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t slot = ent[q];
          if (slot == 0xFFFFu) continue;
          if ((slot >> 5) != lwi) {
            lwi = slot >> 5;
            lw = lds_live[lwi * 64 + lane];
          }
          if (!((lw >> (slot & 31u)) & 1u)) {  // filled or cancelled meanwhile: forget it, the slot becomes allocatable
            atomicAnd(&lds_inl[(slot >> 5) * 64 + lane], ~(1u << (slot & 31u)));  // ds_and, nothing to wait for
            continue;
          }
          const uint32_t x = rng.next_u32();
          if ((int32_t)(x >> 8) > D.keep_thr) {  // gen::<f32>() > p_cancel: kept
            my[(size_t)keep * NU] = (uint16_t)slot;
            keep += 1;
          } else if (event_room()) {  // env.cancel_order(id); stays live (and unallocatable) until the event is processed
            push_event(slot);
            atomicAnd(&lds_inl[(slot >> 5) * 64 + lane], ~(1u << (slot & 31u)));  // ds_and, nothing to wait for
          } else {  // no room for the cancellation (flagged): the order stays the member's
            my[(size_t)keep * NU] = (uint16_t)slot;
            keep += 1;
          }
        }
      }
    }
    auto remember = [&](uint32_t slot) {  // live_orders.push(order_id)
      if (slot == 0xFFFFu) return;
      my[(size_t)keep * NU] = (uint16_t)slot;
      keep += 1;
      atomicOr(&lds_inl[(slot >> 5) * 64 + lane], 1u << (slot & 31u));
    };
    const double lnslack = MKT ? 0.0 : clamp_bound(mid, D.tick_f);
    if (D.type == 1) {
      // ---- NoiseAgent::update (noise_agent.rs:127-176)
      for (uint32_t t = 0; t < D.n; ++t) {
        if ((rng.next_u32() >> 8) < D.thr_limit) {                   // gen::<f32>() < p_limit
          const bool buy = rng.next_u64() < 0x8000000000000000ull;   // gen_bool(0.5)
          remember(place_limit(buy, D.mu + D.sigma * rng.std_normal(zx, zf), mid, lnslack, D, tag));
        }
        if ((rng.next_u32() >> 8) < D.thr_market) {                  // gen::<f32>() < p_market
          const bool buy = rng.next_u64() < 0x8000000000000000ull;
          create(buy, buy ? 0xFFFFFFFFu : 0u, D.trade_vol, 0u);
        }
        queue_turn(mid, D.tick_f);
      }
      drain(true, D.tick_f);
    } else {
      // ---- MomentumAgent::update (momentum_agent.rs:146-208)
      double m = 0.0, p_market = 0.0;
      const uint32_t gflags = st[H_GFLAGS];
      if ((gflags >> j) & 1u) {
        const double gm = pm::from_bits(mk64(st[H_GST + 4 * j], st[H_GST + 4 * j + 1]));
        const double gl = pm::from_bits(mk64(st[H_GST + 4 * j + 2], st[H_GST + 4 * j + 3]));
        m = gm * (1.0 - D.decay) + D.decay * (mid - gl);
        p_market = D.demand * pm::tanh(D.scale * m) / D.n_f;
      }
      const uint64_t thr_l = thr53(D.order_ratio * p_market), thr_m = thr53(p_market);
      const int sgn = (m > 0.0) ? 1 : ((m < 0.0) ? -1 : 0);
      for (uint32_t t = 0; t < D.n; ++t) {
        if ((rng.next_u64() >> 11) < thr_l) {  // gen::<f64>() < p_limit
          if (sgn != 0) remember(place_limit(sgn > 0, D.mu + D.sigma * rng.std_normal(zx, zf), mid, lnslack, D, tag));
        }
        if ((rng.next_u64() >> 11) < thr_m) {  // gen::<f64>() < p_market
          if (sgn != 0) create(sgn > 0, sgn > 0 ? 0xFFFFFFFFu : 0u, D.trade_vol, 0u);
        }
        queue_turn(mid, D.tick_f);
      }
      drain(true, D.tick_f);
      const uint64_t mb = pm::to_bits(m), lb = pm::to_bits(mid);
      st[H_GST + 4 * j] = (uint32_t)mb;
      st[H_GST + 4 * j + 1] = (uint32_t)(mb >> 32);
      st[H_GST + 4 * j + 2] = (uint32_t)lb;
      st[H_GST + 4 * j + 3] = (uint32_t)(lb >> 32);
      st[H_GFLAGS] = gflags | (1u << j);
    }
    ml.len[(size_t)j * NU + b] = keep;
    if (MKT) close_book();
  }

  // ---- transactions.shuffle(rng) (env.rs:121)
  {
    uint32_t i = n_ev > 1 ? n_ev - 1 : 0;
    while (i != 0) {
      const uint32_t rg = i + 1;
      const uint32_t jx = rng.below(rg, (rg << __builtin_clz(rg)) - 1u);
      const uint16_t ai = list[i * 64 + lane], aj = list[jx * 64 + lane];
      list[i * 64 + lane] = aj;
      list[jx * 64 + lane] = ai;
      --i;
    }
  }
  for (uint32_t as = 0; as < M; ++as) {  // every book of a market carries a copy of the market's RNG state
    uint32_t* h = st0 + (size_t)as * a.state_stride;
    *reinterpret_cast<uint2*>(h + H_S0_LO) = make_uint2((uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32));
    *reinterpret_cast<uint2*>(h + H_S1_LO) = make_uint2((uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32));
  }
  if (!MKT) {
    unstage_masks();
    st[H_NEXT_ID] = next_id;
    if (new_flags) st[H_FLAGS] |= new_flags;
  }
  bt[BT_NEV] = n_ev;
  for (uint32_t k = 0; k < n_ev; k += 2) {
    const uint32_t lo = list[k * 64 + lane];
    const uint32_t hi = (k + 1 < n_ev) ? list[(k + 1) * 64 + lane] : 0u;
    bt[BT_EV + (k >> 1)] = lo | (hi << 16);
  }
}

// (Re)build the members' lists from the pool after the wave-per-book kernels (or a restore) have run: live slots
// tagged with the member, oldest order first.  One wave per book.
template <int R>
__global__ __launch_bounds__(256) void k_mixed_lists_rebuild(DevArgs a, MixedArgs ma, MixedLists ml) {
  const int lane = threadIdx.x & 63;
  const uint32_t book = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (book >= a.n_books) return;
  const uint32_t* st = a.state + (size_t)book * a.state_stride;
  const size_t NB = ml.n_books, NU = ml.n_units;
  const uint32_t unit = book / a.assets, my_asset = book - unit * a.assets;  // assets == 1: unit == book
  uint32_t id[R], meta[R];
  uint64_t listed[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    id[r] = st[HDR_DW + r * POOL_FIELDS * 64 + 2 * 64 + lane];
    meta[r] = st[HDR_DW + r * POOL_FIELDS * 64 + 4 * 64 + lane];
    listed[r] = 0;
  }
  for (uint32_t j = 0; j < ma.n_desc; ++j) {
    if (a.assets > 1 && ma.asset[j] != my_asset) continue;  // the member trades another asset of the market
    uint64_t mask[R];
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mask[r] = __ballot((meta[r] & 1u) && ((meta[r] >> 8) & 0xFFu) == j + 1);
      any |= mask[r];
      listed[r] |= mask[r];
    }
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
      if (lane == 0) ml.list[((size_t)j * ml.cap + n) * NU + unit] = (uint16_t)slot;
      n += 1;
    }
    if (lane == 0) ml.len[(size_t)j * NU + unit] = n;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (lane == 0) {
      ml.inl[(size_t)(2 * r) * NB + book] = (uint32_t)listed[r];
      ml.inl[(size_t)(2 * r + 1) * NB + book] = (uint32_t)(listed[r] >> 32);
    }
  }
}

// pm_math.hpp routines on the device, element-wise (tests only)
__global__ void k_selftest_math(int op, const double* in, double* out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  out[i] = op == 0 ? pm::exp(x) : (op == 1 ? pm::log(x) : pm::tanh(x));
}

}  // namespace bkd
