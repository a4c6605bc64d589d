// bourse_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
// See bourse_oracle.hpp for scope, parity status and usage restrictions.
// Every function cites the reference lines it restates (paths relative to
// /root/reference).  Written to be obviously correct, never optimised: the
// data structures are the reference's (two ordered maps per side, append-only
// order and trade vectors).
#include "bourse_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cstring>

namespace orc {

// ===========================================================================
// RNG — third-party arithmetic, restated (PARITY UNPINNED, see header)
// ===========================================================================
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

// rand_xoshiro 0.6.0: `seed_from_u64` for Xoroshiro128StarStar is overridden to
// `from_splitmix!`: build SplitMix64{x: seed}, `from_rng` it, i.e. fill the
// 16-byte seed with two little-endian next_u64 outputs; s0 = first, s1 = second.
// SplitMix64::next_u64: x += 0x9e3779b97f4a7c15; z = x; z = (z^(z>>30))*0xbf58476d1ce4e5b9;
// z = (z^(z>>27))*0x94d049bb133111eb; return z^(z>>31).
Rng Rng::seed_from_u64(uint64_t seed) {
  uint64_t x = seed;
  auto splitmix = [&x]() {
    x += 0x9e3779b97f4a7c15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  Rng r;
  r.s0 = splitmix();
  r.s1 = splitmix();
  // from_seed maps an all-zero seed to a fixed non-zero one (deal_with_zero_seed!);
  // two consecutive SplitMix64 outputs cannot both be zero, so it never triggers here.
  return r;
}

// xoroshiro128** (Blackman & Vigna), as in rand_xoshiro 0.6.0 xoroshiro128starstar.rs:
// r = rotl(s0*5, 7)*9; s1 ^= s0; s0 = rotl(s0,24) ^ s1 ^ (s1<<16); s1 = rotl(s1,37).
uint64_t Rng::next_u64() {
  const uint64_t r = rotl64(s0 * 5ull, 7) * 9ull;
  uint64_t t1 = s1 ^ s0;
  s0 = rotl64(s0, 24) ^ t1 ^ (t1 << 16);
  s1 = rotl64(t1, 37);
  return r;
}

// rand_xoshiro 0.6.0: the StarStar generators' next_u32 is `self.next_u64() as u32`.
uint32_t Rng::next_u32() { return static_cast<uint32_t>(next_u64()); }

// rand 0.8.5 distributions/float.rs, Standard for f32: take a u32, keep the top
// 24 bits, scale by 2^-24  ->  [0,1).  Call site: ref random_agent.rs:91.
float Rng::gen_f32() {
  const uint32_t v = next_u32() >> 8;
  return static_cast<float>(v) * (1.0f / 16777216.0f);
}

// rand 0.8.5 distributions/uniform.rs, UniformInt<u32>::sample_single(lo, hi)
// = sample_single_inclusive(lo, hi-1): range = hi-lo; zone = (range << lz(range)) - 1;
// loop { v = next_u32; (hi_w, lo_w) = v wmul range; if lo_w <= zone return lo + hi_w }.
// Call sites: ref random_agent.rs:100-101.
uint32_t Rng::gen_range_u32(uint32_t lo, uint32_t hi) {
  assert(lo < hi);
  const uint32_t range = hi - lo;  // (hi-1) - lo + 1
  const uint32_t zone = (range << __builtin_clz(range)) - 1u;
  for (;;) {
    const uint32_t v = next_u32();
    const uint64_t m = static_cast<uint64_t>(v) * static_cast<uint64_t>(range);
    const uint32_t hi_w = static_cast<uint32_t>(m >> 32);
    const uint32_t lo_w = static_cast<uint32_t>(m);
    if (lo_w <= zone) return lo + hi_w;
  }
}

// rand 0.8.5 seq/mod.rs gen_index: for ubound <= u32::MAX, gen_range(0..ubound as u32).
uint32_t Rng::gen_index(uint32_t ubound) { return gen_range_u32(0, ubound); }

// rand 0.8.5 SliceRandom::shuffle: for i in (1..len).rev() { swap(i, gen_index(i+1)) }.
// Call site: ref crates/step_sim/src/env.rs:121.
template <class T>
static void shuffle(std::vector<T>& v, Rng& rng) {
  for (size_t i = v.size(); i-- > 1;) {
    const uint32_t j = rng.gen_index(static_cast<uint32_t>(i + 1));
    std::swap(v[i], v[j]);
  }
}

// ===========================================================================
// OrderBookSide — ref: crates/order_book/src/side.rs:54-143
// ===========================================================================
void OrderBookSide::insert_order(const OrderKey& key, OrderId idx, Vol v) {  // side.rs:54-66
  orders[{key.price_key, key.t}] = idx;  // BTreeMap::insert overwrites on equal key (App. A.9 hazard)
  auto it = volumes.find(key.price_key);
  if (it != volumes.end()) {
    it->second.first += v;
    it->second.second += 1;
  } else {
    volumes.emplace(key.price_key, std::make_pair(v, OrderCount{1}));
  }
  vol += v;
}

void OrderBookSide::remove_order(const OrderKey& key, Vol v) {  // side.rs:75-84
  orders.erase({key.price_key, key.t});
  auto it = volumes.find(key.price_key);
  assert(it != volumes.end());  // .unwrap()
  it->second.first -= v;
  it->second.second -= 1;
  if (it->second.second == 0) volumes.erase(it);
  vol -= v;
}

void OrderBookSide::remove_vol(Price price_key, Vol v) {  // side.rs:93-96
  auto it = volumes.find(price_key);
  assert(it != volumes.end());
  it->second.first -= v;
  vol -= v;
}

Price OrderBookSide::best_price() const {  // side.rs:99-104
  return orders.empty() ? PRICE_MAX : orders.begin()->first.first;
}

std::pair<Vol, OrderCount> OrderBookSide::best_vol_and_orders() const {  // side.rs:107-112
  return volumes.empty() ? std::make_pair(Vol{0}, OrderCount{0}) : volumes.begin()->second;
}

Vol OrderBookSide::best_vol() const {  // side.rs:115-120
  return volumes.empty() ? 0 : volumes.begin()->second.first;
}

std::optional<OrderId> OrderBookSide::best_order_idx() const {  // side.rs:128-130
  if (orders.empty()) return std::nullopt;
  return orders.begin()->second;
}

std::pair<Vol, OrderCount> OrderBookSide::vol_and_orders_at_price(Price price_key) const {  // side.rs:138-143
  auto it = volumes.find(price_key);
  return it == volumes.end() ? std::make_pair(Vol{0}, OrderCount{0}) : it->second;
}

// ===========================================================================
// OrderBook — ref: crates/order_book/src/orderbook.rs
// ===========================================================================
OrderBook::OrderBook(Nanos start_time, Price tick, bool trading_, int levels_)  // orderbook.rs:158-171
    : t(start_time), tick_size(tick), trading(trading_), levels(levels_) {
  assert(tick > 0);
}

std::vector<std::pair<Vol, OrderCount>> OrderBook::ask_levels() const {  // orderbook.rs:229-236
  const Price start = bid_ask().second;
  std::vector<std::pair<Vol, OrderCount>> out(levels);
  for (int i = 0; i < levels; ++i)
    out[i] = ask_side.vol_and_orders_at_price(start + static_cast<Price>(i) * tick_size);  // wrapping_add
  return out;
}

std::vector<std::pair<Vol, OrderCount>> OrderBook::bid_levels() const {  // orderbook.rs:257-264
  const Price start = bid_ask().first;
  std::vector<std::pair<Vol, OrderCount>> out(levels);
  for (int i = 0; i < levels; ++i)
    out[i] = bid_side.vol_and_orders_at_price(start - static_cast<Price>(i) * tick_size);  // wrapping_sub
  return out;
}

double OrderBook::mid_price() const {  // orderbook.rs:272-276
  auto [bid, ask] = bid_ask();
  const Price spread = ask - bid;
  return static_cast<double>(bid) + 0.5 * static_cast<double>(spread);
}

Level2Data OrderBook::level_2_data() const {  // orderbook.rs:314-324
  Level2Data d;
  auto [b, a] = bid_ask();
  d.bid_price = b;
  d.ask_price = a;
  d.bid_vol = bid_vol();
  d.ask_vol = ask_vol();
  d.bid_price_levels = bid_levels();
  d.ask_price_levels = ask_levels();
  return d;
}

int OrderBook::create_order(Side side, Vol vol, TraderId trader, std::optional<Price> price,
                            OrderId* out_id) {  // orderbook.rs:356-396
  const OrderId order_id = orders.size();  // current_order_id, :327-329
  Order o;
  o.side = side;
  o.status = Status::New;
  o.arr_time = t;
  o.end_time = NANOS_MAX;  // types.rs:142
  o.vol = vol;
  o.start_vol = vol;
  o.trader_id = trader;
  o.order_id = order_id;
  if (price.has_value()) {
    if (*price % tick_size != 0) return ORC_PRICE_NOT_TICK_MULTIPLE;  // :367-372 / :377-382
    o.price = *price;
  } else {
    o.price = (side == Side::Bid) ? PRICE_MAX : 0;  // types.rs:168 / :221 (market sentinels)
  }
  const OrderKey key = (side == Side::Bid) ? get_bid_key(0, o.price) : get_ask_key(0, o.price);  // :388-391
  orders.push_back(OrderEntry{o, key});
  if (out_id) *out_id = order_id;
  return ORC_OK;
}

// ref: orderbook.rs:843-870
static Vol match_orders(Nanos t, Order& agg, Order& pass, std::vector<Trade>& trades) {
  const Vol trade_vol = std::min(agg.vol, pass.vol);
  agg.vol -= trade_vol;
  pass.vol -= trade_vol;
  trades.push_back(Trade{t, pass.side, pass.price, trade_vol, agg.order_id, pass.order_id});
  if (pass.vol == 0) {
    pass.end_time = t;
    pass.status = Status::Filled;
  }
  if (agg.vol == 0) {
    agg.end_time = t;
    agg.status = Status::Filled;
  }
  return trade_vol;
}

void OrderBook::match_bid(OrderEntry& e) {  // orderbook.rs:429-454
  while ((e.order.vol > 0) & (e.order.price >= ask_side.best_price())) {
    auto next = ask_side.s.best_order_idx();
    if (!next.has_value()) break;
    OrderEntry& m = orders[*next];
    const Vol tv = match_orders(t, e.order, m.order, trades);
    trade_vol += tv;
    if (m.order.status == Status::Filled)
      ask_side.s.remove_order(m.key, tv);
    else
      ask_side.s.remove_vol(m.key.price_key, tv);
  }
}

void OrderBook::match_ask(OrderEntry& e) {  // orderbook.rs:462-487
  while ((e.order.vol > 0) & (e.order.price <= bid_side.best_price())) {
    auto next = bid_side.s.best_order_idx();
    if (!next.has_value()) break;
    OrderEntry& m = orders[*next];
    const Vol tv = match_orders(t, e.order, m.order, trades);
    trade_vol += tv;
    if (m.order.status == Status::Filled)
      bid_side.s.remove_order(m.key, tv);
    else
      bid_side.s.remove_vol(m.key.price_key, tv);
  }
}

void OrderBook::place_bid_limit(OrderEntry& e) {  // orderbook.rs:495-505
  if (trading) match_bid(e);
  if (e.order.status != Status::Filled) {
    const OrderKey key{Side::Bid, e.key.price_key, t};
    e.key = key;
    bid_side.s.insert_order(key, e.order.order_id, e.order.vol);
  }
}

void OrderBook::place_bid_market(OrderEntry& e) {  // orderbook.rs:517-531
  if (trading) {
    match_bid(e);
    if (e.order.status != Status::Filled) {
      e.order.status = Status::Cancelled;
      e.order.end_time = t;
    }
  } else {
    e.order.status = Status::Rejected;
    e.order.end_time = t;
  }
}

void OrderBook::place_ask_limit(OrderEntry& e) {  // orderbook.rs:538-548
  if (trading) match_ask(e);
  if (e.order.status != Status::Filled) {
    const OrderKey key{Side::Ask, e.key.price_key, t};
    e.key = key;
    ask_side.s.insert_order(key, e.order.order_id, e.order.vol);
  }
}

void OrderBook::place_ask_market(OrderEntry& e) {  // orderbook.rs:560-574
  if (trading) {
    match_ask(e);
    if (e.order.status != Status::Filled) {
      e.order.status = Status::Cancelled;
      e.order.end_time = t;
    }
  } else {
    e.order.status = Status::Rejected;
    e.order.end_time = t;
  }
}

void OrderBook::place_order(OrderId id) {  // orderbook.rs:583-611
  OrderEntry e = orders[id];  // copy out (:584)
  if (e.order.status != Status::New) return;
  e.order.status = Status::Active;
  e.order.arr_time = t;
  if (e.order.side == Side::Bid) {
    if (e.order.price == PRICE_MAX)
      place_bid_market(e);
    else
      place_bid_limit(e);
  } else {
    if (e.order.price == 0)
      place_ask_market(e);
    else
      place_ask_limit(e);
  }
  orders[id] = e;  // write back (:610)
}

int OrderBook::cancel_order(OrderId id) {  // orderbook.rs:622-644
  if (id >= orders.size()) return ORC_UNKNOWN_ORDER_ID;  // panic! at :642
  OrderEntry& e = orders[id];
  if (e.order.status == Status::Active) {
    e.order.status = Status::Cancelled;
    e.order.end_time = t;
    if (e.key.side == Side::Bid)
      bid_side.s.remove_order(e.key, e.order.vol);
    else
      ask_side.s.remove_order(e.key, e.order.vol);
  }
  return ORC_OK;
}

void OrderBook::reduce_order_vol(OrderEntry& e, Vol reduce_vol) {  // orderbook.rs:656-667
  e.order.vol -= reduce_vol;
  if (e.key.side == Side::Bid)
    bid_side.s.remove_vol(e.key.price_key, reduce_vol);
  else
    ask_side.s.remove_vol(e.key.price_key, reduce_vol);
}

void OrderBook::replace_order(OrderEntry& e, Price new_price, Vol new_vol) {  // orderbook.rs:679-723
  if (e.key.side == Side::Bid)
    bid_side.s.remove_order(e.key, e.order.vol);
  else
    ask_side.s.remove_order(e.key, e.order.vol);
  e.order.vol = new_vol;
  e.order.price = new_price;
  if (trading) {
    if (e.key.side == Side::Bid)
      match_bid(e);
    else
      match_ask(e);
  }
  if (e.order.status != Status::Filled) {
    if (e.key.side == Side::Bid) {
      const OrderKey key = get_bid_key(t, new_price);
      e.key = key;
      bid_side.s.insert_order(key, e.order.order_id, e.order.vol);
    } else {
      const OrderKey key = get_ask_key(t, new_price);
      e.key = key;
      ask_side.s.insert_order(key, e.order.order_id, e.order.vol);
    }
  }
}

int OrderBook::modify_order(OrderId id, std::optional<Price> new_price,
                            std::optional<Vol> new_vol) {  // orderbook.rs:743-772
  if (id >= orders.size()) return ORC_UNKNOWN_ORDER_ID;  // index panic at :749
  OrderEntry e = orders[id];
  if (e.order.status == Status::Active) {
    if (!new_price && !new_vol) {
      // (None, None): no-op
    } else if (!new_price && new_vol) {
      const Vol v = *new_vol;
      if (v < e.order.vol) {
        reduce_order_vol(e, e.order.vol - v);
      } else {
        replace_order(e, e.order.price, v);
      }
    } else if (new_price && !new_vol) {
      replace_order(e, *new_price, e.order.vol);
    } else {
      replace_order(e, *new_price, *new_vol);
    }
  }
  orders[id] = e;
  return ORC_OK;
}

int OrderBook::process_event(const Event& ev) {  // orderbook.rs:782-792
  switch (ev.kind) {
    case Event::NewOrder:
      if (ev.order_id >= orders.size()) return ORC_UNKNOWN_ORDER_ID;
      place_order(ev.order_id);
      return ORC_OK;
    case Event::Cancellation:
      return cancel_order(ev.order_id);
    case Event::Modify:
      return modify_order(ev.order_id, ev.new_price, ev.new_vol);
  }
  return ORC_OK;
}

// ===========================================================================
// Level2DataRecords — ref: crates/step_sim/src/data.rs:26-56
// ===========================================================================
Level2DataRecords::Level2DataRecords(int n_)
    : n(n_), bid_vols_at_levels(n_), ask_vols_at_levels(n_), bid_orders_at_levels(n_),
      ask_orders_at_levels(n_) {}

void Level2DataRecords::append_record(const Level2Data& r) {  // data.rs:44-56
  bid_prices.push_back(r.bid_price);
  ask_prices.push_back(r.ask_price);
  bid_vols.push_back(r.bid_vol);
  ask_vols.push_back(r.ask_vol);
  for (int i = 0; i < n; ++i) {
    bid_vols_at_levels[i].push_back(r.bid_price_levels[i].first);
    bid_orders_at_levels[i].push_back(r.bid_price_levels[i].second);
    ask_vols_at_levels[i].push_back(r.ask_price_levels[i].first);
    ask_orders_at_levels[i].push_back(r.ask_price_levels[i].second);
  }
}

// ===========================================================================
// Env — ref: crates/step_sim/src/env.rs
// ===========================================================================
Env::Env(Nanos start_time, Price tick_size, Nanos step_size_, bool trading, int levels)  // env.rs:84-95
    : step_size(step_size_),
      order_book(start_time, tick_size, trading, levels),
      level_2_data(order_book.level_2_data()),
      level_2_data_records(levels) {}

int Env::step(Rng& rng) {  // env.rs:116-135
  const Nanos start_time = order_book.t;
  order_book.trade_vol = 0;  // reset_trade_vol
  std::vector<Event> txs;
  txs.swap(transactions);  // mem::take
  shuffle(txs, rng);
  int rc = ORC_OK;
  for (size_t i = 0; i < txs.size(); ++i) {
    order_book.t = start_time + static_cast<Nanos>(i);
    const int r = order_book.process_event(txs[i]);
    if (r != ORC_OK) {  // the reference panics here; report and stop
      rc = r;
      break;
    }
  }
  order_book.t = start_time + step_size;
  level_2_data = order_book.level_2_data();
  level_2_data_records.append_record(level_2_data);
  trade_vols.push_back(order_book.trade_vol);
  return rc;
}

int Env::place_order(Side side, Vol vol, TraderId trader, std::optional<Price> price,
                     OrderId* out_id) {  // env.rs:166-176
  OrderId id = 0;
  const int rc = order_book.create_order(side, vol, trader, price, &id);
  if (rc != ORC_OK) return rc;  // `?`: nothing created, nothing queued
  transactions.push_back(Event{Event::NewOrder, id, std::nullopt, std::nullopt});
  if (out_id) *out_id = id;
  return ORC_OK;
}

void Env::cancel_order(OrderId id) {  // env.rs:189-191
  transactions.push_back(Event{Event::Cancellation, id, std::nullopt, std::nullopt});
}

void Env::modify_order(OrderId id, std::optional<Price> new_price,
                       std::optional<Vol> new_vol) {  // env.rs:208-219
  transactions.push_back(Event{Event::Modify, id, new_price, new_vol});
}

// ===========================================================================
// RandomAgents — ref: crates/step_sim/src/agents/random_agent.rs:67-120
// ===========================================================================
RandomAgents::RandomAgents(size_t n, Price tlo, Price thi, Vol vlo, Vol vhi, Price tick, float rate)
    : orders(n, std::nullopt), tick_lo(tlo), tick_hi(thi), vol_lo(vlo), vol_hi(vhi), tick_size(tick),
      activity_rate(rate) {}

void RandomAgents::update(Env& env, Rng& rng) {  // random_agent.rs:85-119
  for (size_t n = 0; n < orders.size(); ++n) {
    std::optional<OrderId>& slot = orders[n];
    const float p = rng.gen_f32();  // :91
    if (p < activity_rate) {
      if (slot.has_value() &&
          env.order_book.orders[*slot].order.status == Status::Active) {  // :95 -> env.rs:288
        env.cancel_order(*slot);                                          // :96
        slot = std::nullopt;                                              // :97
      } else {
        // [Side::Ask, Side::Bid].choose(rng): index 0 = Ask, 1 = Bid (:99)
        const Side side = rng.gen_index(2) == 0 ? Side::Ask : Side::Bid;
        const Price tick = rng.gen_range_u32(tick_lo, tick_hi);  // :100
        const Vol vol = rng.gen_range_u32(vol_lo, vol_hi);       // :101
        OrderId id = 0;
        const int rc = env.place_order(side, vol, static_cast<TraderId>(n), tick * tick_size, &id);  // :103-109
        assert(rc == ORC_OK);  // .unwrap()
        (void)rc;
        slot = id;
      }
    }
    // inactive: keep the held id (:116)
  }
}

// ---------------------------------------------------------------------------
// Market / MarketEnv / RandomMarketAgents
// ---------------------------------------------------------------------------
Market::Market(Nanos start_time, const std::vector<Price>& tick_sizes, bool trading, int levels) {  // market.rs:74-81
  for (Price tk : tick_sizes) order_books.emplace_back(start_time, tk, trading, levels);
}
void Market::set_time(Nanos t) {
  for (OrderBook& b : order_books) b.t = t;
}
void Market::set_trading(bool on) {
  for (OrderBook& b : order_books) b.trading = on;
}
void Market::reset_trade_vols() {
  for (OrderBook& b : order_books) b.trade_vol = 0;
}
int Market::process_event(const MarketEvent& e) {  // market.rs:343-354: dispatch on order_id.0
  OrderBook& b = order_books[e.asset];
  return b.process_event(Event{e.kind, e.order_id, e.new_price, e.new_vol});
}

MarketEnv::MarketEnv(Nanos start_time, const std::vector<Price>& tick_sizes, Nanos step_size_, bool trading, int levels)
    : step_size(step_size_), market(start_time, tick_sizes, trading, levels), trade_vols(tick_sizes.size()) {
  for (const OrderBook& b : market.order_books) {  // market_env.rs:84-94
    level_2_data.push_back(b.level_2_data());
    level_2_data_records.emplace_back(levels);
  }
}

int MarketEnv::step(Rng& rng) {  // market_env.rs:110-132
  const Nanos start_time = market.get_time();
  market.reset_trade_vols();
  std::vector<MarketEvent> txs;
  txs.swap(transactions);
  shuffle(txs, rng);  // ONE shuffle over the events of all assets (:114)
  int rc = ORC_OK;
  for (size_t i = 0; i < txs.size(); ++i) {
    market.set_time(start_time + static_cast<Nanos>(i));  // every book's clock (:117-118)
    const int r = market.process_event(txs[i]);
    if (r != ORC_OK) {
      rc = r;
      break;
    }
  }
  market.set_time(start_time + step_size);
  for (size_t a = 0; a < market.order_books.size(); ++a) {  // :124-131
    level_2_data[a] = market.order_books[a].level_2_data();
    level_2_data_records[a].append_record(level_2_data[a]);
    trade_vols[a].push_back(market.order_books[a].trade_vol);
  }
  return rc;
}

int MarketEnv::place_order(uint32_t asset, Side side, Vol vol, TraderId trader, std::optional<Price> price,
                           OrderId* out_id) {  // market_env.rs:163-176
  OrderId id = 0;
  const int rc = market.order_books[asset].create_order(side, vol, trader, price, &id);
  if (rc != ORC_OK) return rc;
  transactions.push_back(MarketEvent{Event::NewOrder, asset, id, std::nullopt, std::nullopt});
  if (out_id) *out_id = id;
  return ORC_OK;
}
void MarketEnv::cancel_order(uint32_t asset, OrderId id) {  // :189-191
  transactions.push_back(MarketEvent{Event::Cancellation, asset, id, std::nullopt, std::nullopt});
}
void MarketEnv::modify_order(uint32_t asset, OrderId id, std::optional<Price> new_price,
                             std::optional<Vol> new_vol) {  // :207-218
  transactions.push_back(MarketEvent{Event::Modify, asset, id, new_price, new_vol});
}

RandomMarketAgents::RandomMarketAgents(uint32_t asset_, size_t n, Price tlo, Price thi, Vol vlo, Vol vhi, Price tick,
                                       float rate)
    : asset(asset_), orders(n), tick_lo(tlo), tick_hi(thi), vol_lo(vlo), vol_hi(vhi), tick_size(tick),
      activity_rate(rate) {}

void RandomMarketAgents::update(MarketEnv& env, Rng& rng) {  // random_agent.rs:204-247: RandomAgents on one asset
  for (size_t n = 0; n < orders.size(); ++n) {
    std::optional<OrderId>& slot = orders[n];
    const float p = rng.gen_f32();  // :215
    if (p < activity_rate) {
      if (slot.has_value() &&
          env.market.order_books[asset].orders[*slot].order.status == Status::Active) {  // :219
        env.cancel_order(asset, *slot);
        slot = std::nullopt;
      } else {
        const Side side = rng.gen_index(2) == 0 ? Side::Ask : Side::Bid;  // :223
        const Price tick = rng.gen_range_u32(tick_lo, tick_hi);           // :224
        const Vol vol = rng.gen_range_u32(vol_lo, vol_hi);                // :225
        OrderId id = 0;
        const int rc = env.place_order(asset, side, vol, static_cast<TraderId>(n), tick * tick_size, &id);
        assert(rc == ORC_OK);  // .unwrap() :235
        (void)rc;
        slot = id;
      }
    }
  }
}

}  // namespace orc
