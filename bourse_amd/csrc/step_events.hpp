// step_events.hpp - k_step_events: Env::step (crates/step_sim/src/env.rs:116-135) of every book over the instructions a HOST or
// device-resident agent layer submitted (place / cancel / modify: crates/order_book/src/orderbook.rs:583-611, 622-644, 743-772)
// - the event-by-event loop (any step) and, round 5, the keyed form (step_events_keyed) for steps without modifications.
// Moved out of book_device.hpp in round 5: the step's shuffle (env.rs:121) borrows the wave-parallel Fisher-Yates of the
// agent pipelines' decode (wave_agents.hpp WaveDecoder::shuffle).
#pragma once
#include "book_device.hpp"
#include "wave_agents.hpp"

namespace bkd {

// ==================================================================================
// Kernel 2: host-driven order flow.  One Env::step per book over an uploaded event batch
// (New / Cancellation / Modify), shuffled on the device with the book's RNG.
// One wave per workgroup; the shuffle permutation lives in LDS.
// ==================================================================================
constexpr uint32_t EV_LDS_CAP = 8192;  // events per book per step (u16 permutation in LDS)

template <int R>
__device__ __forceinline__ int find_live_by_id(const Book<R>& B, uint32_t id) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t m = B.live[r] & __ballot(B.id[r] == id);
    if (m) return r * 64 + (int)__builtin_ctzll(m);
  }
  return -1;
}
template <int R>
__device__ __forceinline__ int find_free_slot(const Book<R>& B) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t m = ~B.live[r];
    if (m) return r * 64 + (int)__builtin_ctzll(m);
  }
  return -1;
}

// A new order into pool slot n (k_step_events): all four fields and both masks from ONE per-lane predicate per pool register -
// a vector compare whose result is the slot's bit (the ballot), four selects, three scalar mask operations.  slot_write /
// mask_set do the same through a wave-uniform branch per register and field: ~56 scalar-port instructions per order at
// R = 4 against 12, on the kernel's binding port (round 5, docs/EXPERIMENTS.md).
template <int R>
__device__ __forceinline__ void insert_order(Book<R>& B, int lane, uint32_t n, uint32_t price, uint32_t vol, uint32_t id,
                                             uint32_t seq, bool is_bid) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool mine = (uint32_t)lane + 64u * (uint32_t)r == n;
    const uint64_t M = __ballot(mine);
    B.price[r] = mine ? price : B.price[r];
    B.vol[r] = mine ? vol : B.vol[r];
    B.id[r] = mine ? id : B.id[r];
    B.seq[r] = mine ? seq : B.seq[r];
    B.live[r] |= M;
    B.bid[r] = is_bid ? (B.bid[r] | M) : (B.bid[r] & ~M);
  }
}

// ----------------------------------------------------------------------------------
// k_step_events on the KEYED loop (round 5).  The host-driven step addressed its orders by id, event by event: a slot search
// per new order, an id search per cancellation, the two-reduction match, 6-11 single-lane log stores - ~250 instructions
// per event, most of them on the scalar port.  A step WITHOUT modifications (the numpy API's instructions have none:
// step_sim_numpy.rs:255-268), of at most one event per pool slot's worth (n_ev <= 64 R), on a trading book, is instead
// turned into the slot-addressed form the decode kernels produce and run on the same assembly loops as k_step_batch:
//   * every new order gets a pool slot that is FREE WHEN THE STEP BEGINS (the i-th new order in event order the i-th free
//     slot: any free slot is as good as another, the result does not depend on it) and its fields go into the pool there,
//     pending - so no slot is used twice in a step, and one spare free slot serves as the target of cancellations that
//     find no order (already gone, or never placed: a no-op in the reference, orderbook.rs:622-644);
//   * a cancellation's id is looked up ONCE, among the orders live now and this step's new ones (a cancellation that comes
//     before its order's placement clears a key that is still 0; one that comes after it clears the order: as the reference);
//   * the event words carry slot | EV_NEW | EV_BID and the compare value, as key_event_words builds them;
//   * the ORDER LOG is rebuilt afterwards, one pool LANE per order (vector stores, 64 orders per instruction), from four
//     per-slot facts collected in LDS: the order's arrival position, the first cancellation after it, the last trade that
//     took from it while it rested, and what was left of it on arrival (its volume minus what it took as the aggressor:
//     the compact trade records are scattered into those words before each flush).
// MODIFICATIONS (round 6; orderbook.rs:743-772, 656-723).  What a modification does depends on the state of
// its order WHEN ITS EVENT IS PROCESSED (Active?  is the new volume below what is left of it?), so it cannot be turned into
// list entries up front.  The assembly loops take a range of positions [k, n): the list is cut at every modification, the
// loop runs up to it, and the modification itself is a few lines of C++ between two statements, on the real state:
//   * order not Active (its key lane is 0: filled, cancelled, not yet placed, or no such order) or neither field given: nothing;
//   * volume only and below the order's: the volume in its pool lane, priority kept (reduce_order_vol);
//   * otherwise replace_order: the trade buffer is flushed (its records' prices are read from the pool at the flush), the
//     order leaves the book (key := 0), its slot takes the new price / volume, and the event word AT THAT POSITION becomes a
//     New of that slot with the new price's compare value - the loop then matches it and rests the remainder with a fresh
//     arrival stamp, exactly replace_order's re-match and re-insert with key time = now (:699-721).  The order keeps its slot
//     and id, so later cancellations / modifications of the same step find it where the set-up looked it up.
//   The order log: the rebuild below also visits the slots a modification touched; price and priority key of a replaced order
//   are written at the modification (the rebuild leaves those fields of such a slot alone).
// Returns false - nothing that matters changed: only fields of FREE pool slots - when the step is not of that form (an unknown
// id, a volume of 0 anywhere, fewer free slots than new orders + 1, prices - a modification's new one included - or arrival
// stamps outside the key window): the caller runs the event-by-event loop.  Market orders run on the `m` forms of the loops
// (round 6: also on the hand-written ones of the small pools).  LDS (dynamic, `perm`): 12 x 64 R bytes (ev_keyed_lds_bytes).
// ----------------------------------------------------------------------------------
#ifndef BOURSE_AMD_EV_KEYED
#define BOURSE_AMD_EV_KEYED 1
#endif
// measurement build (profiles/r06/ab_ev_mods.txt): 0 = round 5's rule - a step with a modification, or with a market order on a
// pool of <= 128 slots, is not of the keyed form
#ifndef BOURSE_AMD_EV_KEYED_MODS
#define BOURSE_AMD_EV_KEYED_MODS 1
#endif
constexpr uint32_t ev_keyed_lds_bytes(int R) { return 12u * 64u * (uint32_t)R; }
// MKT (the kernel's): the book belongs to a market of several assets (MarketEnv, market_env.rs:110-121).  The market's queue
// holds every asset's events at their global positions; the other assets' stay in this book's list as events that do nothing,
// so that the positions - the time stamps - are the market's; n_own = this book's events.  (A template parameter, and the
// single-asset text left exactly as it was: the run-time form cost the single-asset kernel 1.5 %, a first templated form that
// simplified its expressions 4 % - the register allocation of this kernel sits on an edge: profiles/r05/ab_ev_markets*.txt.)
template <int R, bool MKT, bool MODS = true>
// perm: the shuffled positions of THESE n_ev events (a whole step's, or one chunk of a longer queue's: then t0 is the chunk's
// first time stamp); wk: 12 x 64 R bytes of work area - the same bytes as perm for a whole step (the permutation is consumed
// on the way), behind the permutation for a chunked one (the later chunks still need theirs).  n_own is ADDED to.
__device__ __forceinline__ bool step_events_keyed(Book<R>& B, const DevArgs& a, uint32_t book, uint64_t t0, int lane,
                                                  uint32_t n_ev, uint32_t e0, const uint16_t* perm, uint16_t* wk, const LogCtx& lg,
                                                  uint32_t asset, uint32_t& n_own, uint32_t* bins) {
  constexpr uint32_t S = 64u * R;
  uint16_t* rank2ev = wk + S;       // bytes [2S, 4S): event position of the i-th new order
  uint16_t* ev2slot = wk + 2u * S;  // bytes [4S, 6S): pool slot of the new order at an event position
  uint32_t* W0 = reinterpret_cast<uint32_t*>(wk + 4u * S);  // bytes [8S, 12S): first cancellation after arrival << 16 | arrival position + 1
  uint32_t* W1 = reinterpret_cast<uint32_t*>(wk);           // bytes [0, 4S), once the lists above are consumed: last passive trade's position + 1
  uint32_t* W2 = reinterpret_cast<uint32_t*>(wk + 2u * S);  // bytes [4S, 8S): a new order's volume minus what it took as the aggressor
  // ---- the events in shuffled order, one per lane
  uint32_t eww[R], eid[R], evq[R];
  // (a modification is known by its KIND in its lane of eww - 0xFF in lanes past the queue's end and in other assets' lanes -
  // not by a third set of wave masks: at 512 slots the kernel is over its scalar registers as it is, and the spilled masks of a
  // first version took 65 536 books x 48 clean instructions from 0.52 to 0.60 ms per launch)
  uint64_t is_new[R], is_can[R];
  uint64_t any_mod = 0;
  bool bad = false;
  uint32_t xmin = 0xFFFFFFFFu, xmax = 0u;  // the modifications' new prices: part of the key window
#pragma unroll
  for (int re = 0; re < R; ++re) {
    // (pools of <= 256 slots: list registers past the queue's end - all but the first at the ingress scripts' 48 events - cost
    // one uniform branch each; at 512 slots the branches take the kernel from 8 to 60 B of scratch at its 96 registers and
    // cost 9 %, same box: profiles/r05/device_ingress_rate_guards.txt)
    if constexpr (R <= 4) {
      is_new[re] = is_can[re] = 0ull;
      eww[re] = 0xFFu;
      eid[re] = evq[re] = 0u;
      if (n_ev <= (uint32_t)re * 64u) continue;
    }
    const uint32_t pos = (uint32_t)(re * 64 + lane);
    const bool valid = pos < n_ev;
    uint4 rec = make_uint4(0xFFu, 0u, 0u, 0u);
    if (valid) rec = a.ev[e0 + perm[pos]];
    const uint32_t kind = rec.x & 0xFFu;
    if constexpr (MKT) {
      const bool own = valid && ((rec.x >> 16) & 0xFFu) == asset;
      is_new[re] = __ballot(own && kind == 0u);
      is_can[re] = __ballot(own && kind == 1u);
      any_mod |= __ballot(own && kind == 2u);
      rec.x = own ? rec.x : 0xFFu;
      bad |= own && (kind > 2u || (kind == 0u && rec.w == 0u) || (kind == 2u && (rec.x & 0x400u) && rec.w == 0u) ||
                     (a.ev_len && kind != 0u && rec.y >= B.next_id));
    } else {
      is_new[re] = __ballot(valid && kind == 0u);
      is_can[re] = __ballot(valid && kind == 1u);
      any_mod |= __ballot(valid && kind == 2u);
      bad |= valid && (kind > 2u || ((rec.x >> 16) & 0xFFu) != 0u || (kind == 0u && rec.w == 0u) ||
                       (kind == 2u && (rec.x & 0x400u) && rec.w == 0u) || (a.ev_len && kind != 0u && rec.y >= B.next_id));
    }
    if ((rec.x & 0x2FFu) == 0x202u) {  // a modification that carries a price (kind 2, has-price bit)
      xmin = min(xmin, rec.z);
      xmax = max(xmax, rec.z);
    }
    eww[re] = rec.x;
    eid[re] = rec.y;
    evq[re] = rec.w;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) bad |= lane_bit(B.live[r]) && B.vol[r] == 0u;
  if (__ballot(bad)) return false;
  if constexpr (!MODS) {  // (this instantiation leaves modifications to the event-by-event loop: see k_step_events)
    if (any_mod) return false;
    any_mod = 0;
    xmin = 0xFFFFFFFFu;
    xmax = 0u;
  }
  uint32_t own_cnt = n_ev;
  if constexpr (MKT) {  // (this book's events: News, Cancellations and Modifications of its asset)
    own_cnt = 0;
#pragma unroll
    for (int re = 0; re < R; ++re) own_cnt += (uint32_t)__builtin_popcountll(__ballot((eww[re] & 0xFFu) <= 2u));
  }
  // ---- slots for the new orders
  uint32_t n_new = 0;
#pragma unroll
  for (int re = 0; re < R; ++re) {
    if (R <= 4 && !is_new[re]) continue;
    const uint32_t rank = n_new + __builtin_amdgcn_mbcnt_hi((uint32_t)(is_new[re] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)is_new[re], 0u));
    if (lane_bit(is_new[re])) rank2ev[rank] = (uint16_t)(re * 64 + lane);
    n_new += __builtin_popcountll(is_new[re]);
  }
  wave_sync();
  uint64_t newm[R], live0[R], mkt = 0;
  uint32_t s_nop = 0xFFFFFFFFu, nf = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    live0[r] = B.live[r];
    const uint64_t freem = ~(B.live[r] | B.pend[r]);
    const uint32_t fr = nf + __builtin_amdgcn_mbcnt_hi((uint32_t)(freem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)freem, 0u));
    const bool fre = lane_bit(freem), mine = fre && fr < n_new;
    const uint64_t nopm = __ballot(fre && fr == n_new);
    if (nopm) s_nop = (uint32_t)r * 64u + (uint32_t)__builtin_ctzll(nopm);
    newm[r] = __ballot(mine);
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    uint32_t e = 0;
    if (mine) {
      e = rank2ev[fr];
      ev2slot[e] = (uint16_t)(r * 64 + lane);
      rec = a.ev[e0 + perm[e]];
    }
    B.price[r] = mine ? rec.z : B.price[r];
    B.vol[r] = mine ? rec.w : B.vol[r];
    B.id[r] = mine ? rec.y : B.id[r];
    const bool bidl = (rec.x >> 8) & 1u;
    B.bid[r] = (B.bid[r] & ~newm[r]) | __ballot(mine && bidl);
    mkt |= __ballot(mine && rec.z == (bidl ? 0xFFFFFFFFu : 0u));
    W0[r * 64 + lane] = 0xFFFF0000u | (mine ? e + 1u : 0u);
    nf += __builtin_popcountll(freem);
  }
  if (s_nop == 0xFFFFFFFFu) return false;  // fewer than n_new + 1 free slots
  // (round 6: the small pools' hand-written loops have their `m` forms too - event_asm.hpp events_key_r1m / _r2m)
  if (!BOURSE_AMD_EV_KEYED_MODS && R <= 2 && mkt) return false;
  wave_sync();
  // ---- the cancellations' and the modifications' slots: one id search each, among the orders live now and this step's new ones
  uint32_t evs[R];
  if (!BOURSE_AMD_EV_KEYED_MODS && any_mod) return false;
#pragma unroll
  for (int re = 0; re < R; ++re) {
    if constexpr (R <= 4) {
      evs[re] = s_nop;
      if (n_ev <= (uint32_t)re * 64u) continue;
    }
    uint32_t cs = s_nop;
    const bool ismod = (eww[re] & 0xFFu) == 2u;
    uint64_t m = is_can[re] | (any_mod ? __ballot(ismod) : 0ull);
    while (m) {
      const uint32_t l = (uint32_t)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint32_t id = rdl(eid[re], l);
      uint32_t slot = s_nop;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint64_t hit = (live0[r] | newm[r]) & __ballot(B.id[r] == id);
        if (hit) slot = (uint32_t)r * 64u + (uint32_t)__builtin_ctzll(hit);
      }
      cs = wrl(slot, l, cs);
    }
    const uint32_t pos = (uint32_t)(re * 64 + lane);
    if (lane_bit(is_can[re]) && cs != s_nop) {
      const uint32_t arr = W0[cs] & 0xFFFFu;  // (the arrival halves are final: written above, behind the barrier)
      if (pos + 1u > arr) atomicMin(&W0[cs], (pos << 16) | arr);
    }
    evs[re] = lane_bit(is_new[re]) ? ((uint32_t)ev2slot[pos & (S - 1u)] | EV_NEW | (((eww[re] >> 8) & 1u) ? EV_BID : 0u)) : cs;
    // a modification keeps {slot, EV_MOD, has-price, has-volume} in its event word, its new volume in evq, and - once the key
    // window is known, below - its new price's FIELD in the word's upper half; the price itself is read again from the
    // record here (the permutation is still intact; the id in eid is not needed any more)
    if (any_mod) {
      if (ismod) {
        eid[re] = a.ev[e0 + perm[pos]].z;
        evs[re] = cs | EV_MOD | (eww[re] & (EV_MOD_P | EV_MOD_V));
      }
    }
  }
  // ---- keys, event words, the loop
  KeyState<R> K;
  if (!keys_begin<R, true>(B, newm, n_ev, K, xmin, xmax)) {  // (the permutation is still intact for the caller's loop)
    // prices that span more than the key window: the top-anchored window with the far-low bids saturated (book_device.hpp
    // keys_begin_wide, built for MomentumAgent's bids at price 0) - an external agent's far-away bids, round 6.  Not with
    // modifications (a replacement could move a bid across the window's edge after the guard was evaluated), and not in the 512-slot
    // kernel (its register budget: see k_step_events).  bins: the snapshot's level bins, not in use yet.
    if constexpr (BOURSE_AMD_KEYED_WIDE && R <= 4) {
      if (any_mod || !(keys_begin_wide<R, true>(B, newm, n_ev, K, evs, bins, lane) ||
                       keys_begin_wide_high<R, true>(B, newm, n_ev, K, evs, bins, lane)))  // (asks far ABOVE the book: the mirror window)
        return false;
    } else {
      return false;
    }
  }
  uint32_t evw[R];
  key_event_words<R>(K, evs, n_ev, evw);
  if (any_mod) {
#pragma unroll
    for (int re = 0; re < R; ++re)
      if (evw[re] & EV_MOD) evw[re] = (evw[re] & 0xFFFFu) | ((eid[re] - K.pbase) << 16);
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < R; ++r) {  // (perm, rank2ev, ev2slot are consumed: their bytes now hold W1, W2)
    W1[r * 64 + lane] = 0u;
    W2[r * 64 + lane] = lane_bit(newm[r]) ? B.vol[r] : 0u;
  }
  wave_sync();
  const uint32_t nev = rfl(n_ev);
  // the list is cut at the modifications: [k, kend) runs on the assembly loop, a modification is handled between two statements
  // alo <= best ask key, bhi >= best bid key: kept across this step's statements (event_asm.hpp EK_R2M_STMT)
  uint32_t alo = 0x80000000u, bhi = 0x7FFFFFFFu;
  uint32_t k = 0, kend = any_mod ? 0u : nev;
  uint32_t post = 0xFFFFFFFFu, post_p = 0;  // the slot (and its old price) of the replacement whose New event the last range was
  // pool slots a modification changed / replaced in this step: ONE vector register - bit r of a lane's word = slot (r, lane)
  // changed, bit 8 + r = replaced (not 2 R wave masks: scalar registers are what this kernel is short of)
  uint32_t mflags = 0;
  auto mflag_set = [&](uint32_t sl, uint32_t base) {
    mflags = (uint32_t)lane == (sl & 63u) ? (mflags | (1u << (base + (sl >> 6)))) : mflags;
  };
  auto scatter_and_flush = [&]() {
    if (B.tr_n) {
      B.trade_vol += wave_add((uint32_t)lane < B.tr_n ? B.tr_vol : 0u);
      if (lg.base && (uint32_t)lane < B.tr_n) {  // the log's facts of these trades
        const uint32_t kk = B.tr_k & 0x7FFFFFFFu;
        atomicMax(&W1[B.tr_pas & (S - 1u)], kk + 1u);
      }
      if (lg.base) {
        const uint32_t as = pool_gather<R>(evw, (B.tr_k & 0x7FFFFFFFu) < S ? (B.tr_k & 0x7FFFFFFFu) : 0u) & EV_SLOT & (S - 1u);
        if ((uint32_t)lane < B.tr_n) atomicSub(&W2[as], B.tr_vol);
      }
    }
    flush_trades_compact<R>(B, a, book, t0, lane, evw);
  };
  if constexpr (!MODS) {  // no modification reaches this instantiation: round 5's plain loop (statement, flush, again)
    for (;;) {
      uint32_t full;
      if constexpr (R == 1)
        full = events_key_r1m(0u, k, nev, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol[0], K.key[0], evw[0], B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
      else if constexpr (R == 2)
        full = events_key_r2m(0u, k, nev, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol[0], B.vol[1], K.key[0], K.key[1], evw[0], evw[1], B.tr_k,
                              B.tr_vol, B.tr_pas, alo, bhi);
      else if constexpr (R == 4)
        full = events_key_r4x(0u, k, nev, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
      else
        full = events_key_r8x(0u, k, nev, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
      // (scatter_and_flush()'s text, in line: through the lambda this instantiation takes 32 B of scratch at 512 slots instead of 8)
      if (B.tr_n) {
        B.trade_vol += wave_add((uint32_t)lane < B.tr_n ? B.tr_vol : 0u);
        if (lg.base && (uint32_t)lane < B.tr_n) {
          const uint32_t kk = B.tr_k & 0x7FFFFFFFu;
          atomicMax(&W1[B.tr_pas & (S - 1u)], kk + 1u);
        }
        if (lg.base) {
          const uint32_t as = pool_gather<R>(evw, (B.tr_k & 0x7FFFFFFFu) < S ? (B.tr_k & 0x7FFFFFFFu) : 0u) & EV_SLOT & (S - 1u);
          if ((uint32_t)lane < B.tr_n) atomicSub(&W2[as], B.tr_vol);
        }
      }
      flush_trades_compact<R>(B, a, book, t0, lane, evw);
      if (!full) break;
    }
  } else {
  for (;;) {
    if (any_mod && k == kend) {
      if (post != 0xFFFFFFFFu) {
        // ---- the replaced order's New event has run: did it come to rest? (replace_order's tail, orderbook.rs:699-721)
        const bool rested = slot_read<R>(K.key, post) != 0u;
        if (!rested) {  // Filled in the re-match: for the rebuild, a dead order of volume 0 whose last trade was this event
          slot_write<R>(B.vol, post, 0u);
          if (lane == 0) W1[post] = k;  // (position + 1 of the event just processed; no earlier record is pending: flushed before it)
        }
        if (lg.base) {
          const uint32_t id = slot_read<R>(B.id, post);
          if (id >= lg.cap) {
            B.flags |= FLAG_ORDER_LOG_FULL;
          } else if (lane == 0) {
            DevOrderLog* e = lg.base + id;
            const uint64_t tk = t0 + (k - 1u);
            e->price = slot_read<R>(B.price, post);
            const uint32_t arr = W0[post] & 0xFFFFu;
            if (rested) {  // re-keyed: (new price, now)
              e->key_price = e->price;
              e->key_lo = (uint32_t)tk;
              e->key_hi = (uint32_t)(tk >> 32);
            } else if (arr != 0u && !((rdl(mflags, post & 63u) >> (8u + (post >> 6))) & 1u)) {
              // an order of THIS step keeps the key it rested with on arrival (the rebuild leaves a replaced slot's key alone)
              const uint64_t ta = t0 + (arr - 1u);
              e->key_price = post_p;
              e->key_lo = (uint32_t)ta;
              e->key_hi = (uint32_t)(ta >> 32);
            }
          }
        }
        mflag_set(post, 8u);
        wave_sync();
        post = 0xFFFFFFFFu;
      }
      if (k >= nev) break;
      // the next modification at or behind k
      uint32_t km = nev;
#pragma unroll
      for (int re = R - 1; re >= 0; --re) {
        uint64_t m = __ballot((evw[re] & (EV_NEW | EV_MOD)) == EV_MOD && (uint32_t)(re * 64 + lane) < nev);
        const uint32_t base = 64u * (uint32_t)re;
        if (k > base) m = (k - base >= 64u) ? 0ull : (m & (~0ull << (k - base)));
        if (m) km = base + (uint32_t)__builtin_ctzll(m);
      }
      kend = km;
      if (km == k) {
        // ---- the modification at position k (orderbook.rs:743-772)
        const uint32_t w = slot_read<R>(evw, k), sl = w & EV_SLOT & (S - 1u);
        const bool has_p = (w & EV_MOD_P) != 0u, has_v = (w & EV_MOD_V) != 0u;
        const uint32_t keyv = slot_read<R>(K.key, sl);
        k += 1;
        kend = k;  // (nothing to run: the boundary code above looks for the next modification)
        if (keyv == 0u || !(has_p || has_v)) continue;  // not Active (or no such order: the spare slot's key is 0) / (None, None)
        const uint32_t cur_v = slot_read<R>(B.vol, sl), cur_p = slot_read<R>(B.price, sl), nv_in = slot_read<R>(evq, k - 1u);
        mflag_set(sl, 0u);
        if (!has_p && nv_in < cur_v) {  // reduce_order_vol: in place, priority kept
          slot_write<R>(B.vol, sl, nv_in);
          continue;
        }
        // replace_order (:679-723): out of the book, new price / volume, then the same slot's New event at this position
        const bool is_bid = (int32_t)keyv > 0;
        const uint32_t np = has_p ? K.pbase + (w >> 16) : cur_p, nv = has_v ? nv_in : cur_v;
        if (np != cur_p) scatter_and_flush();  // (the buffered records' prices are read from the pool: before this slot's changes)
        slot_write<R>(K.key, sl, 0u);
        slot_write<R>(B.price, sl, np);
        slot_write<R>(B.vol, sl, nv);
        slot_write<R>(evq, k - 1u, nv);
        const uint32_t pf = np - K.pbase;
        slot_write<R>(evw, k - 1u, sl | EV_NEW | (is_bid ? EV_BID : 0u) | ((is_bid ? 0x8000u | pf : pf) << 16));
        post = sl;
        post_p = cur_p;
        k -= 1;  // run [k, k + 1)
        kend = k + 1u;
      }
      if (k == kend) continue;
    }
    uint32_t full;
    if constexpr (R == 1)
      full = events_key_r1m(0u, k, kend, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol[0], K.key[0], evw[0], B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
    else if constexpr (R == 2)
      full = events_key_r2m(0u, k, kend, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol[0], B.vol[1], K.key[0], K.key[1], evw[0], evw[1], B.tr_k,
                            B.tr_vol, B.tr_pas, alo, bhi);
    else if constexpr (R == 4)
      full = events_key_r4x(0u, k, kend, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
    else
      full = events_key_r8x(0u, k, kend, 0xFFFFFFFFu, B.tr_n, K.sq, B.vol, K.key, evw, evq, B.tr_k, B.tr_vol, B.tr_pas, alo, bhi);
    scatter_and_flush();
    if (!any_mod && !full) break;
  }
  }
  keys_end<R>(B, K);
#pragma unroll
  for (int r = 0; r < R; ++r) B.pend[r] = 0;
  n_own += own_cnt;
  if (!lg.base) return true;
  // ---- the order log, one pool lane per order touched in this step
  wave_sync();
  uint64_t over = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t w0 = W0[r * 64 + lane], arr = w0 & 0xFFFFu, c = w0 >> 16, f = W1[r * 64 + lane], rem = W2[r * 64 + lane];
    const bool alive = lane_bit(B.live[r]), was = lane_bit(live0[r]), isnew = arr != 0u, bidl = lane_bit(B.bid[r]);
    const bool rp = (mflags >> (8 + r)) & 1u;  // replaced in this step: price and key were written at the modification
    const bool touched = isnew || (was && (f != 0u || !alive)) || ((mflags >> r) & 1u);
    const uint32_t id = B.id[r], price = B.price[r];
    const bool market = price == (bidl ? 0xFFFFFFFFu : 0u);
    // a new order that did not come to rest: a market order, or filled on arrival
    const bool norest = isnew && !rp && (market || rem == 0u);
    uint32_t status, vol;
    uint64_t end = ~0ull, key_t = 0ull;
    const uint64_t t_arr = t0 + (arr - 1u);
    if (norest) {
      status = rem == 0u ? 2u : 3u;  // Filled / the market order's remainder Cancelled (orderbook.rs:521-524,:564-567)
      vol = rem;
      end = t_arr;
    } else {
      key_t = t_arr;
      vol = B.vol[r];
      status = alive ? 1u : (vol == 0u ? 2u : 3u);
      if (!alive) end = t0 + (vol == 0u ? (uint64_t)(f - 1u) : (uint64_t)c);
    }
    const bool fits = id < lg.cap;
    over |= __ballot(touched && !fits);
    if (touched && fits) {
      uint32_t* e = reinterpret_cast<uint32_t*>(lg.base + id);
      e[1] = vol;
      if (isnew || !alive) {
        e[0] = status;
        e[6] = (uint32_t)end;
        e[7] = (uint32_t)(end >> 32);
      }
      if (isnew) {
        e[4] = (uint32_t)t_arr;
        e[5] = (uint32_t)(t_arr >> 32);
      }
      if (isnew && !rp) {
        e[2] = price;
        e[3] = price;
        e[8] = (uint32_t)key_t;
        e[9] = (uint32_t)(key_t >> 32);
      }
    }
  }
  if (over) B.flags |= FLAG_ORDER_LOG_FULL;
  return true;
}

// timing experiments only (scripts/ev_phase_times.sh): -DBOURSE_AMD_EV_SKIP=bits leaves phases of k_step_events out (results
// are then wrong): 1 the shuffle's swaps, 2 the order-log writes, 4 matching
#ifndef BOURSE_AMD_EV_SKIP
#define BOURSE_AMD_EV_SKIP 0
#endif
#ifndef BOURSE_AMD_EV_OCC
#define BOURSE_AMD_EV_OCC(R) ((R) <= 4 ? 8 : 5)
#endif
// (the 512-slot form WITH the keyed modifications: four waves per SIMD - 128 VGPRs, no scratch - instead of five with 116 B of
// scratch per lane; profiles/r06/ab_ev_mods.txt)
#ifndef BOURSE_AMD_EV_OCC_M
#define BOURSE_AMD_EV_OCC_M(R, MODS) (((R) == 8 && (MODS)) ? 4 : BOURSE_AMD_EV_OCC(R))
#endif
#ifndef BOURSE_AMD_EV_WAVE_SHUFFLE
#define BOURSE_AMD_EV_WAVE_SHUFFLE 1
#endif
// dynamic LDS of k_step_events beside the long queues' permutation: the wave-parallel shuffle's lists + ring, then the keyed
// form's lists and per-slot words in the same bytes
constexpr uint32_t ev_lds_bytes(int R) {
  const uint32_t S = 64u * (uint32_t)R, sh = 6u * S + WV_RING * 4u, ky = ev_keyed_lds_bytes(R);
  return sh > ky ? sh : ky;
}
template <int R, bool MKT = false, bool CHUNKS = false, bool MODS = true>
// (MKT: launched for the books of markets with more than one asset, bk_config.assets > 1 - only the keyed form differs)
// (MODS = false: modifications stay on the event-by-event loop - the 512-slot kernel is on the edge of its registers (8 B of
// scratch per lane without the modification code, ~100 B with either half of it, and 65 536 books x 48 clean instructions went from
// 0.52 to 0.60 ms per launch), so the host launches the form with it only once the env has SEEN a modification: bk_modify_order, or
// k_ingest's flag word in mapped host memory - bourse_amd.hip launch_events; smaller pools always run MODS = true)
// (CHUNKS: launched when a queue of this step is longer than the pool - the keyed form then runs chunk by chunk, in a loop
// around its one call site; as a run-time loop in the ONE kernel it cost the ordinary launch 2 %: 32 B more scratch at R = 4)
// (eight waves per SIMD for pools of <= 256 slots, five for 512: 8 192 books - the C4 shard, the ingress rate scripts - are then ONE residency round;
// at the 69 VGPRs the compiler took for R = 4 a seventh of the waves ran as a second round)
__global__ __launch_bounds__(64, BOURSE_AMD_EV_OCC_M(R, MODS)) void k_step_events(DevArgs a, WaveArgs wa, uint64_t step_index, uint32_t wave_shuffle_min,
                                                                          uint32_t lds_bytes /* the launch's dynamic LDS */) {
  __shared__ uint32_t bins[LDS_DW_PER_WAVE];
  // the shuffle permutation: dynamic LDS sized by the host to this step's longest queue (<= EV_LDS_CAP entries), so
  // that quiet steps do not pay 16 KB of LDS per one-wave workgroup in occupancy
  extern __shared__ uint16_t perm[];
  const int lane = threadIdx.x;
  const uint32_t book = blockIdx.x;
  // MarketEnv mode: the event queue belongs to the MARKET (book / assets); every book of the market shuffles it with
  // its copy of the market's RNG stream (identical results) and processes its own asset's events at their global
  // positions t0 + k (market_env.rs:110-121).  assets == 1: market == book.
  const uint32_t mkt = book / a.assets, asset = book - mkt * a.assets;
  uint32_t* st = a.state + (size_t)book * a.state_stride;
  Book<R> B;
  Rng rng;
  load_book<R>(B, rng, st, lane);
  const uint64_t step_size = mk64(a.step_lo, a.step_hi);
  const uint32_t e0 = a.ev_len ? mkt * a.ev_stride : a.ev_off[mkt];
  const uint32_t n_ev = a.ev_len ? a.ev_len[mkt] : a.ev_off[mkt + 1] - e0;
  uint32_t n_own = 0;
  LogCtx lg{(a.order_log && !(BOURSE_AMD_EV_SKIP & 2)) ? a.order_log + (size_t)book * a.log_cap : nullptr, a.log_cap};

  const uint64_t t0 = B.t;
  B.trade_vol = 0;
  const uint64_t trades_before = B.n_trades;
  if (step_size != 0 && (uint64_t)n_ev >= step_size) B.flags |= FLAG_STEP_SIZE;
  for (uint32_t j = lane; j < n_ev; j += 64) perm[j] = (uint16_t)j;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // shuffle (env.rs:121): for i in (1..n).rev() swap(i, gen_range(0..i + 1)).  Round 5: the kernel is bound by the CU's ONE
  // scalar port (9.9 k scalar-port instructions per book-step, 43 % of the wave-cycles waiting for issue; docs/EXPERIMENTS.md), and
  // this loop was a quarter of them - the generator's 64-bit arithmetic (~22 scalar instructions per draw), the rejection
  // test, the swap's lane-0 masking.  The SAME arithmetic now runs on the vector unit, which has the slots: the state as
  // four 32-bit halves in vector registers (RngLane - every lane computes the same value), the swap as plain LDS reads and
  // writes by all lanes (same address, same value), one vector-to-scalar hand-over per draw for the accept test.
  // Round 5, second half: once the keyed form had halved the rest of the kernel, these n - 1 DEPENDENT draws (~35 vector
  // instructions each, every lane computing the same value) were a quarter of its instructions.  A queue of at most one event
  // per pool slot is therefore shuffled by the decode's wave-parallel Fisher-Yates (wave_agents.hpp WaveDecoder::shuffle: 64
  // draws per window from the book's cached lane states, the acceptance fixed point, all swaps resolved at once); the T^256
  // table is read from global memory (one block change every few steps), the draws, targets and bucket words use the LDS
  // the keyed form takes over afterwards (a market's books each shuffle their copy of the market's stream, from their own
  // cache record).  Longer queues keep the loop below.
  const bool wave_shuffle = BOURSE_AMD_EV_WAVE_SHUFFLE && !BOURSE_AMD_EV_SKIP && wa.wcache != nullptr && (MKT || a.assets == 1u) && n_ev >= 2u &&
                            n_ev >= wave_shuffle_min && n_ev <= 64u * R;
  if (wave_shuffle) {
    constexpr uint32_t S = 64u * R;
    uint32_t* wc = wa.wcache + (size_t)book * WC_STRIDE;
    WaveDecoder<R, false> D;
    BK_STAMP_START(D, book);
    D.tab = wa.jt_block;
    D.evl = perm;
    D.pm = nullptr;
    D.sm = nullptr;
    D.pv = nullptr;
    D.jarr = perm + S;                                       // bytes [2S, 4S)
    D.ring = reinterpret_cast<uint32_t*>(perm + 3u * S);     // bytes [6S, 6S + 2048)
    D.wmask = reinterpret_cast<uint4*>(D.ring);              // (the draws are dead once the windows are resolved)
    if (R > 2) {
      static_assert(64u * R * 4u <= WV_RING * 4u, "the bucket words alias the ring");
      D.co = D.ring;
      D.bucket = perm + 2u * S;                              // bytes [4S, 6S)
    }
    D.wcs = reinterpret_cast<uint4*>(wc + WC_HDR);
    D.lane = lane;
    D.load_cache(wc, (uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32), wa.jt_lane);
    D.shuffle(n_ev);
    uint32_t n0, n1, n2, n3;
    D.finish(wc, n0, n1, n2, n3);
    rng.s0 = mk64(n0, n1);
    rng.s1 = mk64(n2, n3);
  } else {
    RngLane v{(uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32)};
    asm volatile("" : "+v"(v.a0), "+v"(v.a1), "+v"(v.b0), "+v"(v.b1));  // (uniform values: keep the compiler from moving them back to the scalar unit)
    for (uint32_t i = (BOURSE_AMD_EV_SKIP & 1) ? 0u : n_ev; i-- > 1;) {
      const uint32_t range = i + 1u, zone = (range << __builtin_clz(range)) - 1u;  // UniformInt<u32>::sample_single (App. B.3)
      uint32_t j;
      for (;;) {
        const uint32_t x = v.next_u32();
        const uint32_t lo = x * range;
        j = __umulhi(x, range);
        if (rfl((uint32_t)(lo <= zone))) break;
      }
      const uint32_t pi = perm[i], pj = perm[j];
      perm[i] = (uint16_t)pj;
      perm[j] = (uint16_t)pi;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    rng.s0 = mk64(rfl(v.a0), rfl(v.a1));
    rng.s1 = mk64(rfl(v.b0), rfl(v.b1));
  }
  // (the keyed form first: see step_events_keyed; the event-by-event loop below is the general case)
  // (the keyed form runs on the assembly loops only: a -DBOURSE_AMD_ASM_EVENTS=0 / -DBOURSE_AMD_ASM_R48=0 build steps event by event)
  constexpr bool has_asm = BOURSE_AMD_ASM_EVENTS && (R <= 2 || BOURSE_AMD_ASM_R48);
  // A queue of more than one event per pool slot (round 6) runs the keyed form CHUNK by chunk: 64 R events at a time, each
  // chunk with its own slot assignment and key window - an order placed by one chunk rests (or is gone) for the next, the time
  // stamps carry the chunk's first position - when the launch's LDS holds the work area behind the permutation (`lds_bytes`,
  // sized by the host to the longest queue).  A chunk that is not of the keyed form hands the REST of the step to the loop below.
  constexpr uint32_t S_ = 64u * R;
  bool keyed = false;
  uint32_t done = 0;  // events already processed (whole chunks)
  if constexpr (!CHUNKS) {
    keyed = BOURSE_AMD_EV_KEYED && has_asm && !(BOURSE_AMD_EV_SKIP & ~1) && (MKT || a.assets == 1u) && B.trading && n_ev != 0u && n_ev <= S_ &&
            step_events_keyed<R, MKT, MODS>(B, a, book, t0, lane, n_ev, e0, perm, perm, lg, asset, n_own, bins);
    done = keyed ? n_ev : 0u;
  } else if (BOURSE_AMD_EV_KEYED && has_asm && !(BOURSE_AMD_EV_SKIP & ~1) && (MKT || a.assets == 1u) && B.trading && n_ev != 0u &&
             (n_ev <= S_ || 2u * ((n_ev + 63u) & ~63u) + ev_keyed_lds_bytes(R) <= lds_bytes)) {
    uint16_t* wk = n_ev <= S_ ? perm : perm + ((n_ev + 63u) & ~63u);  // (one chunk: the work area takes the permutation's bytes over)
    keyed = true;
    while (done < n_ev) {
      const uint32_t len = n_ev - done < S_ ? n_ev - done : S_;
      if (!step_events_keyed<R, MKT, MODS>(B, a, book, t0 + done, lane, len, e0, perm + done, wk, lg, asset, n_own, bins)) {
        keyed = false;
        break;
      }
      done += len;
    }
  }
  uint4 evr = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t k = done; k < n_ev; ++k) {
    // the 16-byte records of 64 shuffled positions are fetched at once, one per lane (one memory round trip per 64
    // events instead of one per event), then broadcast one by one
    if ((k & 63u) == 0u && k + lane < n_ev) evr = a.ev[e0 + perm[k + lane]];
    const uint32_t w = rdl(evr.x, k & 63u);
    if (((w >> 16) & 0xFFu) != asset) continue;  // another asset's event
    ++n_own;
    const uint32_t id = rdl(evr.y, k & 63u);
    const uint32_t ep = rdl(evr.z, k & 63u);
    const uint32_t evv = rdl(evr.w, k & 63u);
    const uint32_t kind = w & 0xFFu;
    const uint64_t tk = t0 + k;
    // an id that was never created: the reference panics while processing (orderbook.rs:642); the host-driven path refuses
    // the step before uploading anything, the device-resident ingress (no host in the loop) flags the book and drops it
    if (a.ev_len && kind != 0 && id >= B.next_id) {
      B.flags |= FLAG_UNKNOWN_ORDER;
      continue;
    }
    if (kind == 0) {
      // ---- New: place_order (orderbook.rs:583-611); a fresh id is New by construction
      const bool is_bid = (w >> 8) & 1u;
      const uint32_t p = ep;
      uint32_t v = evv;
      const bool market = is_bid ? (p == 0xFFFFFFFFu) : (p == 0u);
      bool filled = false;
      uint32_t status = 1;  // Active
      uint64_t end = ~0ull;
      if (B.trading && !(BOURSE_AMD_EV_SKIP & 4)) {
        filled = match<R>(B, a, book, t0, lane, k, is_bid, p, v, id, lg);
        if (filled) {
          status = 2;
          end = tk;
        } else if (market) {
          status = 3;  // unfilled market remainder -> Cancelled (:521-524,:564-567)
          end = tk;
        }
      } else if (market) {
        status = 4;  // Rejected (:526-529)
        end = tk;
      }
      if (!market && !filled) {
        const int n = find_free_slot<R>(B);
        if (n < 0) {
          B.flags |= FLAG_POOL_OVERFLOW;
        } else {
          insert_order<R>(B, lane, (uint32_t)n, p, v, id, B.seq_ctr, is_bid);
          B.seq_ctr += 1;
        }
      }
      // key: provisional (price, 0) from create_order (orderbook.rs:388-391) unless the order rests (:501-505)
      log_write(lg, B.flags, lane, id, status, v, p, tk, end, true, true, status == 1 ? tk : 0ull);
    } else if (kind == 1) {
      // ---- Cancellation (orderbook.rs:622-644): only an Active order changes
      // (one pass over the pool registers: the order's bit, if it is live, from a per-lane compare; its volume and price
      // picked by that lane - no slot index, no per-register scalar selects; ids are unique, so at most one bit is set)
      uint64_t any = 0;
      uint32_t vsel = 0, psel = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint64_t hit = B.live[r] & __ballot(B.id[r] == id);
        B.live[r] &= ~hit;
        any |= hit;
        vsel = lane_bit(hit) ? B.vol[r] : vsel;
        psel = lane_bit(hit) ? B.price[r] : psel;
      }
      if (any) {
        const uint32_t l = (uint32_t)__builtin_ctzll(any);
        log_write(lg, B.flags, lane, id, 3, rdl(vsel, l), rdl(psel, l), 0, tk, false);
      }
    } else {
      // ---- Modify (orderbook.rs:743-772, 656-723): only an Active order changes
      const int n = find_live_by_id<R>(B, id);
      const bool has_p = (w >> 9) & 1u, has_v = (w >> 10) & 1u;
      if (n >= 0 && (has_p || has_v)) {
        const uint32_t cur_v = slot_read<R>(B.vol, n);
        const uint32_t cur_p = slot_read<R>(B.price, n);
        if (!has_p && evv < cur_v) {
          // reduce in place, priority kept (reduce_order_vol)
          slot_write<R>(B.vol, n, evv);
          log_write(lg, B.flags, lane, id, 1, evv, cur_p, 0, ~0ull, false);
        } else {
          // replace_order: remove, re-match at the new price, re-insert with a new time stamp
          const bool is_bid = mask_test<R>(B.bid, n);
          const uint32_t np = has_p ? ep : cur_p;
          uint32_t nv = has_v ? evv : cur_v;
          mask_set<R>(B.live, n, false);
          bool filled = false;
          if (B.trading) filled = match<R>(B, a, book, t0, lane, k, is_bid, np, nv, id, lg);
          if (!filled) {
            slot_write<R>(B.price, n, np);
            slot_write<R>(B.vol, n, nv);
            slot_write<R>(B.seq, n, B.seq_ctr);
            B.seq_ctr += 1;
            mask_set<R>(B.live, n, true);
            log_write(lg, B.flags, lane, id, 1, nv, np, 0, ~0ull, false, true, tk);  // re-keyed (orderbook.rs:699-721)
          } else {
            log_write(lg, B.flags, lane, id, 2, 0, np, 0, tk, false);
          }
        }
      }
    }
  }
  B.n_events += n_own;
  B.t = t0 + step_size;
  snapshot<R>(B, a, book, lane, bins, a.hist_slot0, B.flags, true, a.asset_div[asset]);
  flush_trades<R>(B, a, book, t0, lane);
  store_book<R>(B, rng, st, lane, step_index + 1, (uint32_t)(B.n_trades - trades_before), n_own, keyed ? 1u : 0u);
}


}  // namespace bkd
