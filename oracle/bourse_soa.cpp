// bourse_soa.cpp — batched structure-of-arrays CPU implementation of the hot path (TEST / MEASUREMENT INFRASTRUCTURE).
//
// "many books, SoA, fixed ladder, identical results to the oracle per book; multi-threaded over
// books.  This is both the design prototype for the kernel and the timed CPU baseline" — the second CPU figure of
// bench.py's `cpu_baseline` (kind "soa"), beside the literal map-based oracle (kind "port"), so that the gain of the
// DATA STRUCTURE is separated from the gain of the GPU.  Never linked into or imported by the product (bourse_amd/).
// tests/test_soa_cpu.py checks it equal to the oracle (level-2 history, trade records, RNG states) on every bench shape.
//
// Same semantics as oracle/bourse_oracle.cpp, restated from the reference (paths relative to the reference repo):
//   sim_runner               crates/step_sim/src/runner.rs:46-69      one RNG per book: agents.update, then env.step
//   RandomAgents::update     crates/step_sim/src/agents/random_agent.rs:85-119
//   Env::step                crates/step_sim/src/env.rs:116-135        shuffle, events at t0 + k, clock, L2 record
//   place / match / cancel   crates/order_book/src/orderbook.rs:429-487,495-611,622-644,843-870
//   level_2_data             orderbook.rs:229-264,314-324              levels at touch -/+ i * tick, gaps (0, 0)
//   RNG                      rand 0.8.5 / rand_xoshiro 0.6.0
// Data structure (instead of the reference's two BTreeMaps per side): RandomAgents prices lie on a bounded grid, so a
// book is a direct-mapped LADDER of price levels (index = (price - p_min) / tick), each level an intrusive FIFO
// (doubly linked through the order pool, so a cancellation unlinks in O(1)) with its (volume, count) aggregate, plus one
// non-empty-level bitmap per side (best price = first / last set bit).  An agent holds at most one resting order
// (SURVEY App. A.15), so the pool slot IS the agent index.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

struct Rng {  // Xoroshiro128StarStar, next_u32 = low half of next_u64 (App. B.1)
  uint64_t s0, s1;
  inline uint32_t next_u32() {
    uint64_t r = s0 * 5ull;
    r = (r << 7) | (r >> 57);
    r *= 9ull;
    const uint64_t t = s1 ^ s0;
    s0 = ((s0 << 24) | (s0 >> 40)) ^ t ^ (t << 16);
    s1 = (t << 37) | (t >> 27);
    return static_cast<uint32_t>(r);
  }
  inline uint32_t below(uint32_t range, uint32_t zone) {  // UniformInt<u32>::sample_single (App. B.3)
    for (;;) {
      const uint64_t m = static_cast<uint64_t>(next_u32()) * range;
      if (static_cast<uint32_t>(m) <= zone) return static_cast<uint32_t>(m >> 32);
    }
  }
};
inline uint32_t zone_of(uint32_t range) { return (range << __builtin_clz(range)) - 1u; }
inline void seed_from_u64(uint64_t seed, uint64_t& s0, uint64_t& s1) {  // SplitMix64 (App. B.2)
  uint64_t x = seed;
  auto next = [&x]() {
    x += 0x9e3779b97f4a7c15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  s0 = next();
  s1 = next();
}

struct Group {
  uint32_t n, thr, tick_lo, tick_rng, tick_zone, vol_lo, vol_rng, vol_zone, tick_size;
};
struct Trade {  // == bk_trade / the oracle's trade record layout (40 B)
  uint64_t t;
  uint32_t side_is_bid, price, vol, reserved;
  uint64_t active, passive;
};

constexpr uint16_t NIL = 0xFFFF;
constexpr int MAX_W = 4096, MAX_A = 4096;

struct Shape {  // shared by all books
  uint32_t A = 0, W = 0, words = 0, levels = 0, tick = 0, p_min = 0;
  uint64_t step_size = 0, start = 0;
  std::vector<Group> groups;
};

struct Book {
  Rng rng;
  uint64_t t;
  uint32_t next_id = 0, seq = 0;
  // pool: slot = agent
  std::vector<uint32_t> price, vol, id;
  std::vector<uint16_t> nxt, prv, lvl;
  std::vector<uint8_t> live, bid;
  // ladder
  // one ladder per side, index = side * W + level (a zero-volume order rests without matching, so a bid can sit at or
  // above the best ask: the sides must not share queues)
  std::vector<uint16_t> head, tail;
  std::vector<uint32_t> lvol, lcnt;          // level aggregates
  std::vector<uint64_t> bid_mask, ask_mask;  // non-empty levels per side
  uint32_t bid_vol = 0, ask_vol = 0;         // side totals (wrapping u32 like the reference)
  std::vector<Trade> trades;
  uint64_t n_trades = 0, n_events = 0;
  std::vector<uint32_t> ev;                  // this step's queue: slot | new << 16 | bid << 17
};

struct Many {
  Shape sh;
  std::vector<std::unique_ptr<Book>> books;
  uint64_t seed_base = 0;
  uint64_t steps_done = 0, hist_cap = 0;
  std::vector<uint32_t> hist;  // ring [hist_cap][B][5 + 4 L]
  bool keep_trades = true;
};

inline int best_ask_level(const Book& b, uint32_t words) {  // lowest non-empty ask level or -1
  for (uint32_t w = 0; w < words; ++w)
    if (b.ask_mask[w]) return static_cast<int>(w * 64 + __builtin_ctzll(b.ask_mask[w]));
  return -1;
}
inline int best_bid_level(const Book& b, uint32_t words) {  // highest non-empty bid level or -1
  for (int w = static_cast<int>(words) - 1; w >= 0; --w)
    if (b.bid_mask[w]) return w * 64 + 63 - __builtin_clzll(b.bid_mask[w]);
  return -1;
}

inline void unlink(Book& b, uint32_t s, uint32_t W) {  // remove_order (side.rs:77-82): level aggregates, FIFO links, bitmap
  const uint32_t L = b.lvl[s], X = (b.bid[s] ? W : 0u) + L;
  const uint16_t p = b.prv[s], n = b.nxt[s];
  if (p != NIL) b.nxt[p] = n; else b.head[X] = n;
  if (n != NIL) b.prv[n] = p; else b.tail[X] = p;
  b.lvol[X] -= b.vol[s];
  b.lcnt[X] -= 1;
  if (b.bid[s]) b.bid_vol -= b.vol[s]; else b.ask_vol -= b.vol[s];
  if (b.lcnt[X] == 0) {
    if (b.bid[s]) b.bid_mask[L >> 6] &= ~(1ull << (L & 63)); else b.ask_mask[L >> 6] &= ~(1ull << (L & 63));
  }
  b.live[s] = 0;
}

// match_bid / match_ask + match_orders (orderbook.rs:429-487, 843-870); returns remaining volume
inline uint32_t match(Many& m, Book& b, bool agg_bid, uint32_t p, uint32_t v, uint32_t agg_id, uint64_t t, uint32_t& trade_vol) {
  const Shape& sh = m.sh;
  while (v > 0) {
    const int L = agg_bid ? best_ask_level(b, sh.words) : best_bid_level(b, sh.words);
    if (L < 0) break;
    const uint32_t best = sh.p_min + static_cast<uint32_t>(L) * sh.tick;
    if (agg_bid ? (p < best) : (p > best)) break;  // inclusive crossing test (:430 / :463)
    const uint32_t X = (agg_bid ? 0u : sh.W) + static_cast<uint32_t>(L);  // the passive side's ladder
    const uint32_t s = b.head[X];                    // oldest order at the touch: strict price-time priority
    const uint32_t tv = v < b.vol[s] ? v : b.vol[s];
    v -= tv;
    trade_vol += tv;
    if (m.keep_trades) b.trades.push_back(Trade{t, b.bid[s], best, tv, 0u, agg_id, b.id[s]});
    b.n_trades += 1;
    if (b.vol[s] == tv) {
      unlink(b, s, sh.W);  // passive Filled: removed with its remaining volume (== tv), orderbook.rs:444
      b.vol[s] = 0;
    } else {         // partial fill: remove_vol (side.rs:93-96)
      b.vol[s] -= tv;
      b.lvol[X] -= tv;
      if (b.bid[s]) b.bid_vol -= tv; else b.ask_vol -= tv;
    }
  }
  return v;
}

void step_book(Many& m, Book& b, uint32_t* rec /* this step's L2 record or nullptr */) {
  const Shape& sh = m.sh;
  // ---- agents.update: groups in declaration order, agents in index order (random_agent.rs:85-119)
  b.ev.clear();
  uint32_t a = 0;
  for (const Group& G : sh.groups) {
    for (uint32_t i = 0; i < G.n; ++i, ++a) {
      const uint32_t x = b.rng.next_u32();
      if ((x >> 8) >= G.thr) continue;  // p = gen::<f32>() < activity_rate as an integer threshold
      if (b.live[a]) {                  // holds an Active order: queue its cancellation (:95-97)
        b.ev.push_back(a);
      } else {                          // side, tick, vol in that order (:99-101); id = orders.len() (orderbook.rs:363)
        const uint32_t side = b.rng.below(2u, 0x7FFFFFFFu);
        const uint32_t tick = G.tick_lo + b.rng.below(G.tick_rng, G.tick_zone);
        const uint32_t vol = G.vol_lo + b.rng.below(G.vol_rng, G.vol_zone);
        b.price[a] = tick * G.tick_size;
        b.vol[a] = vol;
        b.id[a] = b.next_id++;
        b.bid[a] = static_cast<uint8_t>(side);
        b.ev.push_back(a | 0x10000u);
      }
    }
  }
  // ---- Env::step (env.rs:116-135)
  const uint32_t n = static_cast<uint32_t>(b.ev.size());
  for (uint32_t i = n; i-- > 1;) {  // transactions.shuffle(rng): SliceRandom (App. B.4)
    const uint32_t j = b.rng.below(i + 1u, zone_of(i + 1u));
    const uint32_t tmp = b.ev[i];
    b.ev[i] = b.ev[j];
    b.ev[j] = tmp;
  }
  const uint64_t t0 = b.t;
  uint32_t trade_vol = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t e = b.ev[k], s = e & 0xFFFFu;
    if (!(e & 0x10000u)) {  // Cancellation (orderbook.rs:622-644): only an Active order changes
      if (b.live[s]) unlink(b, s, sh.W);
      continue;
    }
    // New: place_order (orderbook.rs:583-611), trading enabled
    const bool is_bid = b.bid[s];
    const uint32_t p = b.price[s], v0 = b.vol[s];
    const uint32_t v = match(m, b, is_bid, p, v0, b.id[s], t0 + k, trade_vol);
    const bool filled = v0 != 0 && v == 0;  // a zero-volume order never matches and rests (:430)
    b.vol[s] = v;
    if (!filled) {  // rest the remainder at the tail of its level (key (price, t), t unique per event: FIFO)
      const uint32_t L = (p - sh.p_min) / sh.tick, X = (is_bid ? sh.W : 0u) + L;
      b.lvl[s] = static_cast<uint16_t>(L);
      b.nxt[s] = NIL;
      b.prv[s] = b.tail[X];
      if (b.tail[X] != NIL) b.nxt[b.tail[X]] = static_cast<uint16_t>(s); else b.head[X] = static_cast<uint16_t>(s);
      b.tail[X] = static_cast<uint16_t>(s);
      b.lvol[X] += v;
      b.lcnt[X] += 1;
      if (is_bid) {
        b.bid_vol += v;
        b.bid_mask[L >> 6] |= 1ull << (L & 63);
      } else {
        b.ask_vol += v;
        b.ask_mask[L >> 6] |= 1ull << (L & 63);
      }
      b.live[s] = 1;
    }
  }
  b.n_events += n;
  b.t = t0 + sh.step_size;
  if (!rec) return;
  // ---- level-2 record (rust/src/step_sim_numpy.rs:351-368 layout)
  const int lb = best_bid_level(b, sh.words), la = best_ask_level(b, sh.words);
  rec[0] = trade_vol;
  rec[1] = lb < 0 ? 0u : sh.p_min + static_cast<uint32_t>(lb) * sh.tick;
  rec[2] = la < 0 ? 0xFFFFFFFFu : sh.p_min + static_cast<uint32_t>(la) * sh.tick;
  rec[3] = b.ask_vol;
  rec[4] = b.bid_vol;
  for (uint32_t i = 0; i < sh.levels; ++i) {
    uint32_t* q = rec + 5 + 4 * i;
    const int Lb = lb - static_cast<int>(i), La = la + static_cast<int>(i);
    const bool hb = lb >= 0 && Lb >= 0 && (b.bid_mask[Lb >> 6] >> (Lb & 63) & 1ull);
    const bool ha = la >= 0 && La < static_cast<int>(sh.W) && (b.ask_mask[La >> 6] >> (La & 63) & 1ull);
    q[0] = hb ? b.lvol[sh.W + Lb] : 0u;
    q[1] = hb ? b.lcnt[sh.W + Lb] : 0u;
    q[2] = ha ? b.lvol[La] : 0u;
    q[3] = ha ? b.lcnt[La] : 0u;
  }
}

void init_book(const Many& m, Book& b, uint64_t seed, uint32_t trade_reserve) {
  const Shape& sh = m.sh;
  seed_from_u64(seed, b.rng.s0, b.rng.s1);
  b.t = sh.start;
  b.price.assign(sh.A, 0);
  b.vol.assign(sh.A, 0);
  b.id.assign(sh.A, 0);
  b.nxt.assign(sh.A, NIL);
  b.prv.assign(sh.A, NIL);
  b.lvl.assign(sh.A, 0);
  b.live.assign(sh.A, 0);
  b.bid.assign(sh.A, 0);
  b.head.assign(2 * sh.W, NIL);
  b.tail.assign(2 * sh.W, NIL);
  b.lvol.assign(2 * sh.W, 0);
  b.lcnt.assign(2 * sh.W, 0);
  b.bid_mask.assign(sh.words, 0);
  b.ask_mask.assign(sh.words, 0);
  b.ev.reserve(sh.A);
  b.trades.reserve(trade_reserve);
}

template <class F>
void parallel_books(size_t B, int n_threads, F&& f) {
  if (n_threads <= 1) {
    f(size_t{0}, B);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(f, B * t / n_threads, B * (t + 1) / n_threads);
  for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// groups: n_groups rows of {n, tick_lo, tick_hi, vol_lo, vol_hi, tick_size, thr} (thr = the activity threshold on
// u32 >> 8, i.e. ceil(rate * 2^24): the caller passes the same integer the device uses).  Returns nullptr if the shape
// is outside what the ladder covers (tick grid, window > 4096 levels, > 4096 agents).
// Books are CONSTRUCTED by the threads that will step them (first-touch placement, thread-local allocator arenas).
void* soa_new(uint32_t n_books, uint64_t seed_base, uint64_t start, uint32_t tick, uint64_t step_size, uint32_t levels,
              int n_groups, const uint32_t* groups, uint64_t hist_cap, uint32_t trade_reserve, int keep_trades, int n_threads) {
  auto m = std::make_unique<Many>();
  Shape& sh = m->sh;
  sh.levels = levels;
  sh.tick = tick;
  sh.step_size = step_size;
  sh.start = start;
  uint32_t pmin = 0xFFFFFFFFu, pmax = 0;
  for (int g = 0; g < n_groups; ++g) {
    const uint32_t* r = groups + 7 * g;
    Group G{r[0], r[6], r[1], r[2] - r[1], 0, r[3], r[4] - r[3], 0, r[5]};
    if (r[2] <= r[1] || r[4] <= r[3] || tick == 0 || G.tick_size % tick != 0 || r[1] == 0) return nullptr;
    G.tick_zone = zone_of(G.tick_rng);
    G.vol_zone = zone_of(G.vol_rng);
    if (static_cast<uint64_t>(r[2] - 1) * G.tick_size >= 0xFFFFFFFFull) return nullptr;
    if (G.n) {
      pmin = std::min(pmin, r[1] * G.tick_size);
      pmax = std::max(pmax, (r[2] - 1) * G.tick_size);
    }
    sh.A += G.n;
    sh.groups.push_back(G);
  }
  if (sh.A == 0) {
    pmin = 0;
    pmax = 0;
  }
  sh.p_min = pmin;
  sh.W = (pmax - pmin) / tick + 1;
  sh.words = (sh.W + 63) / 64;
  if (sh.W > MAX_W || sh.A > MAX_A) return nullptr;
  m->seed_base = seed_base;
  m->hist_cap = hist_cap;
  m->keep_trades = keep_trades != 0;
  m->books.resize(n_books);
  const size_t Wd = 5 + 4 * static_cast<size_t>(levels);
  m->hist.resize(static_cast<size_t>(hist_cap) * n_books * Wd);
  Many* mp = m.get();
  parallel_books(n_books, n_threads, [mp, trade_reserve](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; ++b) {
      mp->books[b] = std::make_unique<Book>();
      init_book(*mp, *mp->books[b], mp->seed_base + b, trade_reserve);
    }
  });
  return m.release();
}
void soa_free(void* h) { delete static_cast<Many*>(h); }

// n_steps of sim_runner's loop for every book; books statically partitioned over n_threads (book-major inside a thread)
int soa_run(void* h, uint64_t n_steps, int n_threads) {
  Many& m = *static_cast<Many*>(h);
  const size_t B = m.books.size(), Wd = 5 + 4 * static_cast<size_t>(m.sh.levels);
  const uint64_t first = m.steps_done;
  parallel_books(B, n_threads, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; ++b)
      for (uint64_t s = 0; s < n_steps; ++s) {
        uint32_t* rec = m.hist_cap ? m.hist.data() + (((first + s) % m.hist_cap) * B + b) * Wd : nullptr;
        step_book(m, *m.books[b], rec);
      }
  });
  m.steps_done += n_steps;
  return 0;
}
uint64_t soa_steps_done(void* h) { return static_cast<Many*>(h)->steps_done; }
// the last n retained steps' level-2 records: out[n][B][5 + 4 L]
void soa_history(void* h, uint64_t first_step, uint64_t n, uint32_t* out) {
  Many& m = *static_cast<Many*>(h);
  const size_t B = m.books.size(), Wd = 5 + 4 * static_cast<size_t>(m.sh.levels);
  for (uint64_t s = 0; s < n; ++s)
    std::memcpy(out + s * B * Wd, m.hist.data() + ((first_step + s) % m.hist_cap) * B * Wd, B * Wd * 4);
}
void soa_trade_counts(void* h, uint64_t* out) {
  Many& m = *static_cast<Many*>(h);
  for (size_t b = 0; b < m.books.size(); ++b) out[b] = m.books[b]->n_trades;
}
void soa_event_counts(void* h, uint64_t* out) {
  Many& m = *static_cast<Many*>(h);
  for (size_t b = 0; b < m.books.size(); ++b) out[b] = m.books[b]->n_events;
}
void soa_rng_states(void* h, uint64_t* out) {
  Many& m = *static_cast<Many*>(h);
  for (size_t b = 0; b < m.books.size(); ++b) {
    out[2 * b] = m.books[b]->rng.s0;
    out[2 * b + 1] = m.books[b]->rng.s1;
  }
}
uint64_t soa_trades_retained(void* h, uint32_t book) { return static_cast<Many*>(h)->books[book]->trades.size(); }
void soa_trades(void* h, uint32_t book, void* out) {
  const auto& v = static_cast<Many*>(h)->books[book]->trades;
  if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(Trade));
}
void soa_clear_trades(void* h) {
  for (auto& b : static_cast<Many*>(h)->books) b->trades.clear();
}

}  // extern "C"
