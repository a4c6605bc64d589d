// bourse_oracle_capi.cpp — C ABI over the CPU ORACLE, for ctypes (tests/, smoke(),
// bench.py cpu_baseline only — see bourse_oracle.hpp).  Test infrastructure.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>

#include "bourse_oracle.hpp"
#include "bourse_oracle_agents.hpp"

using namespace orc;

namespace {

struct OrcEnv {  // mirrors the PyO3 pyclass: Env + its own RNG (ref rust/src/step_sim.rs:55-75)
  Env env;
  Rng rng;
  OrcEnv(uint64_t seed, Nanos start, Price tick, Nanos step, bool trading, int levels)
      : env(start, tick, step, trading, levels), rng(Rng::seed_from_u64(seed)) {}
};

struct OrcAgents {  // an AgentSet: members updated in declaration order (ref crates/macros/src/lib.rs:57-73)
  std::vector<std::unique_ptr<AgentBase>> groups;
  void update(Env& env, Rng& rng) {
    for (auto& g : groups) g->update(env, rng);
  }
};

struct OrcAgentDesc {  // one AgentSet member; mirrored by pyoracle.AGENT_DESC_DTYPE (96 bytes)
  uint32_t type;       // 0 RandomAgents, 1 NoiseAgent, 2 MomentumAgent
  uint32_t n;
  uint32_t tick_lo, tick_hi, vol_lo, vol_hi;  // RandomAgents ranges
  uint32_t tick_size;
  float rate;          // RandomAgents activity_rate
  uint32_t trader_start;
  float p_limit, p_market, p_cancel;
  uint32_t trade_vol;
  uint32_t pad;
  double mu, sigma, decay, demand, scale, order_ratio;
};
static_assert(sizeof(OrcAgentDesc) == 104, "layout");

void add_from_desc(OrcAgents& a, const OrcAgentDesc& d) {
  if (d.type == 0) {
    a.groups.push_back(std::make_unique<RandomAgentsBox>(
        RandomAgents(d.n, d.tick_lo, d.tick_hi, d.vol_lo, d.vol_hi, d.tick_size, d.rate)));
  } else if (d.type == 1) {
    a.groups.push_back(std::make_unique<NoiseAgent>(
        d.trader_start, static_cast<uint16_t>(d.n),
        NoiseAgentParams{d.tick_size, d.p_limit, d.p_market, d.p_cancel, d.trade_vol, d.mu, d.sigma}));
  } else {
    a.groups.push_back(std::make_unique<MomentumAgent>(
        d.trader_start, static_cast<uint16_t>(d.n),
        MomentumParams{d.tick_size, d.p_cancel, d.trade_vol, d.decay, d.demand, d.scale, d.order_ratio, d.mu, d.sigma}));
  }
}

struct OrcOrderRec {  // PyOrder layout, ref rust/src/types.rs:17-31
  uint8_t side_is_bid;
  uint8_t status;
  uint8_t pad[6];
  uint64_t arr_time, end_time;
  uint32_t vol, start_vol, price, trader_id;
  uint64_t order_id;
};
static_assert(sizeof(OrcOrderRec) == 48, "layout");

struct OrcTradeRec {  // PyTrade layout, ref rust/src/types.rs:4-15
  uint64_t t;
  uint32_t side_is_bid;
  uint32_t price;
  uint32_t vol;
  uint32_t pad;
  uint64_t active_id, passive_id;
};
static_assert(sizeof(OrcTradeRec) == 40, "layout");

std::optional<uint32_t> opt(int has, uint32_t v) { return has ? std::optional<uint32_t>(v) : std::nullopt; }

// numpy `level_2_data` layout, ref rust/src/step_sim_numpy.rs:351-368 (with L levels):
// [trade_vol, bid_price, ask_price, ask_vol, bid_vol, {bid_vol_i, bid_n_i, ask_vol_i, ask_n_i}...]
void pack_l2(const Level2Data& d, Vol trade_vol, int levels, uint32_t* out) {
  out[0] = trade_vol;
  out[1] = d.bid_price;
  out[2] = d.ask_price;
  out[3] = d.ask_vol;
  out[4] = d.bid_vol;
  for (int i = 0; i < levels; ++i) {
    out[5 + 4 * i + 0] = d.bid_price_levels[i].first;
    out[5 + 4 * i + 1] = d.bid_price_levels[i].second;
    out[5 + 4 * i + 2] = d.ask_price_levels[i].first;
    out[5 + 4 * i + 3] = d.ask_price_levels[i].second;
  }
}

void fill_order(const Order& o, OrcOrderRec* r) {
  std::memset(r, 0, sizeof(*r));
  r->side_is_bid = side_to_bool(o.side);
  r->status = static_cast<uint8_t>(o.status);
  r->arr_time = o.arr_time;
  r->end_time = o.end_time;
  r->vol = o.vol;
  r->start_vol = o.start_vol;
  r->price = o.price;
  r->trader_id = o.trader_id;
  r->order_id = o.order_id;
}

void fill_trade(const Trade& t, OrcTradeRec* r) {
  std::memset(r, 0, sizeof(*r));
  r->t = t.t;
  r->side_is_bid = side_to_bool(t.side);
  r->price = t.price;
  r->vol = t.vol;
  r->active_id = t.active_order_id;
  r->passive_id = t.passive_order_id;
}

// One independent simulated market: Env + agents + RNG, as sim_runner owns them
// (ref crates/step_sim/src/runner.rs:46-69).
struct ManyBook {
  Env env;
  OrcAgents agents;
  Rng rng;
  ManyBook(Nanos start, Price tick, Nanos step, bool trading, int levels, uint64_t seed)
      : env(start, tick, step, trading, levels), rng(Rng::seed_from_u64(seed)) {}
};

struct OrcMany {
  int levels;
  std::vector<std::unique_ptr<ManyBook>> books;
};

}  // namespace

extern "C" {

// ------------------------------------------------------------------ RNG ----
void orc_rng_seed(uint64_t seed, uint64_t* st) {
  Rng r = Rng::seed_from_u64(seed);
  st[0] = r.s0;
  st[1] = r.s1;
}
uint64_t orc_rng_next_u64(uint64_t* st) {
  Rng r{st[0], st[1]};
  uint64_t v = r.next_u64();
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
uint32_t orc_rng_next_u32(uint64_t* st) {
  Rng r{st[0], st[1]};
  uint32_t v = r.next_u32();
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
float orc_rng_f32(uint64_t* st) {
  Rng r{st[0], st[1]};
  float v = r.gen_f32();
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
uint32_t orc_rng_range(uint64_t* st, uint32_t lo, uint32_t hi) {
  Rng r{st[0], st[1]};
  uint32_t v = r.gen_range_u32(lo, hi);
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
void orc_rng_shuffle_u32(uint64_t* st, uint32_t* v, uint64_t n) {
  Rng r{st[0], st[1]};
  for (uint64_t i = n; i-- > 1;) {
    uint32_t j = r.gen_index(static_cast<uint32_t>(i + 1));
    std::swap(v[i], v[j]);
  }
  st[0] = r.s0;
  st[1] = r.s1;
}

// ------------------------------------------------- OrderBookSide (unit KATs) --
// kind: 0 = raw OrderBookSide, 1 = AskSide, 2 = BidSide (ref side.rs:36-43,148-152)
struct OrcSide { int kind; OrderBookSide s; };
void* orc_side_new(int kind) { return new OrcSide{kind, {}}; }
void orc_side_free(void* p) { delete static_cast<OrcSide*>(p); }
static OrderKey side_key(OrcSide* x, uint64_t t, uint32_t price) {
  if (x->kind == 2) return get_bid_key(t, price);
  if (x->kind == 1) return get_ask_key(t, price);
  return OrderKey{Side::Ask, price, t};
}
void orc_side_insert(void* p, uint64_t t, uint32_t price, uint64_t idx, uint32_t vol) {
  auto* x = static_cast<OrcSide*>(p);
  x->s.insert_order(side_key(x, t, price), idx, vol);
}
void orc_side_remove_order(void* p, uint64_t t, uint32_t price, uint32_t vol) {
  auto* x = static_cast<OrcSide*>(p);
  x->s.remove_order(side_key(x, t, price), vol);
}
void orc_side_remove_vol(void* p, uint32_t price, uint32_t vol) {
  auto* x = static_cast<OrcSide*>(p);
  x->s.remove_vol(side_key(x, 0, price).price_key, vol);
}
uint32_t orc_side_vol(void* p) { return static_cast<OrcSide*>(p)->s.vol; }
uint32_t orc_side_best_price(void* p) {
  auto* x = static_cast<OrcSide*>(p);
  return x->kind == 2 ? PRICE_MAX - x->s.best_price() : x->s.best_price();
}
void orc_side_best_vol_and_orders(void* p, uint32_t* out2) {
  auto v = static_cast<OrcSide*>(p)->s.best_vol_and_orders();
  out2[0] = v.first;
  out2[1] = v.second;
}
// returns 1 and writes *out if the side is non-empty
int orc_side_best_order_idx(void* p, uint64_t* out) {
  auto v = static_cast<OrcSide*>(p)->s.best_order_idx();
  if (!v) return 0;
  *out = *v;
  return 1;
}
void orc_side_vol_and_orders_at_price(void* p, uint32_t price, uint32_t* out2) {
  auto* x = static_cast<OrcSide*>(p);
  auto v = x->s.vol_and_orders_at_price(side_key(x, 0, price).price_key);
  out2[0] = v.first;
  out2[1] = v.second;
}

// ------------------------------------------------ immediate-mode OrderBook --
void* orc_book_new(uint64_t start, uint32_t tick, int trading, int levels) {
  return new OrderBook(start, tick, trading != 0, levels);
}
void orc_book_free(void* b) { delete static_cast<OrderBook*>(b); }
void orc_book_set_time(void* b, uint64_t t) { static_cast<OrderBook*>(b)->t = t; }
uint64_t orc_book_get_time(void* b) { return static_cast<OrderBook*>(b)->t; }
void orc_book_set_trading(void* b, int on) { static_cast<OrderBook*>(b)->trading = on != 0; }
int orc_book_create_order(void* b, int bid, uint32_t vol, uint32_t trader, int has_price, uint32_t price,
                          uint64_t* out_id) {
  return static_cast<OrderBook*>(b)->create_order(side_from_bool(bid != 0), vol, trader, opt(has_price, price),
                                                  out_id);
}
int orc_book_place_order(void* b, uint64_t id) {
  auto* ob = static_cast<OrderBook*>(b);
  if (id >= ob->orders.size()) return ORC_UNKNOWN_ORDER_ID;
  ob->place_order(id);
  return ORC_OK;
}
// create_and_place_order, ref orderbook.rs:411-421
int orc_book_create_and_place(void* b, int bid, uint32_t vol, uint32_t trader, int has_price, uint32_t price,
                              uint64_t* out_id) {
  auto* ob = static_cast<OrderBook*>(b);
  uint64_t id = 0;
  int rc = ob->create_order(side_from_bool(bid != 0), vol, trader, opt(has_price, price), &id);
  if (rc != ORC_OK) return rc;
  ob->place_order(id);
  if (out_id) *out_id = id;
  return ORC_OK;
}
int orc_book_cancel(void* b, uint64_t id) { return static_cast<OrderBook*>(b)->cancel_order(id); }
int orc_book_modify(void* b, uint64_t id, int has_price, uint32_t price, int has_vol, uint32_t vol) {
  return static_cast<OrderBook*>(b)->modify_order(id, opt(has_price, price), opt(has_vol, vol));
}
void orc_book_bid_ask(void* b, uint32_t* out2) {
  auto [bid, ask] = static_cast<OrderBook*>(b)->bid_ask();
  out2[0] = bid;
  out2[1] = ask;
}
uint32_t orc_book_bid_vol(void* b) { return static_cast<OrderBook*>(b)->bid_vol(); }
uint32_t orc_book_ask_vol(void* b) { return static_cast<OrderBook*>(b)->ask_vol(); }
uint32_t orc_book_trade_vol(void* b) { return static_cast<OrderBook*>(b)->trade_vol; }
void orc_book_best_bid_vol_and_orders(void* b, uint32_t* out2) {  // orderbook.rs:249-251
  auto v = static_cast<OrderBook*>(b)->bid_side.s.best_vol_and_orders();
  out2[0] = v.first;
  out2[1] = v.second;
}
void orc_book_best_ask_vol_and_orders(void* b, uint32_t* out2) {  // orderbook.rs:221-223
  auto v = static_cast<OrderBook*>(b)->ask_side.s.best_vol_and_orders();
  out2[0] = v.first;
  out2[1] = v.second;
}
double orc_book_mid_price(void* b) { return static_cast<OrderBook*>(b)->mid_price(); }
// out = [bid_price, ask_price, bid_vol, ask_vol, bid_levels (vol,n)*L, ask_levels (vol,n)*L]
void orc_book_level2(void* b, uint32_t* out) {
  auto* ob = static_cast<OrderBook*>(b);
  Level2Data d = ob->level_2_data();
  out[0] = d.bid_price;
  out[1] = d.ask_price;
  out[2] = d.bid_vol;
  out[3] = d.ask_vol;
  for (int i = 0; i < ob->levels; ++i) {
    out[4 + 2 * i] = d.bid_price_levels[i].first;
    out[4 + 2 * i + 1] = d.bid_price_levels[i].second;
    out[4 + 2 * ob->levels + 2 * i] = d.ask_price_levels[i].first;
    out[4 + 2 * ob->levels + 2 * i + 1] = d.ask_price_levels[i].second;
  }
}
uint64_t orc_book_n_orders(void* b) { return static_cast<OrderBook*>(b)->orders.size(); }
uint64_t orc_book_n_trades(void* b) { return static_cast<OrderBook*>(b)->trades.size(); }
int orc_book_order_status(void* b, uint64_t id, uint8_t* out) {
  auto* ob = static_cast<OrderBook*>(b);
  if (id >= ob->orders.size()) return ORC_UNKNOWN_ORDER_ID;
  *out = static_cast<uint8_t>(ob->orders[id].order.status);
  return ORC_OK;
}
void orc_book_get_orders(void* b, void* out, uint64_t first, uint64_t n) {
  auto* ob = static_cast<OrderBook*>(b);
  auto* r = static_cast<OrcOrderRec*>(out);
  for (uint64_t i = 0; i < n; ++i) fill_order(ob->orders[first + i].order, r + i);
}
void orc_book_get_trades(void* b, void* out, uint64_t first, uint64_t n) {
  auto* ob = static_cast<OrderBook*>(b);
  auto* r = static_cast<OrcTradeRec*>(out);
  for (uint64_t i = 0; i < n; ++i) fill_trade(ob->trades[first + i], r + i);
}

// OrderEntry.key of orders [first, first+n) (orderbook.rs:34-39): side (1 = Bid), price_key, t
void orc_book_get_keys(void* b, uint64_t first, uint64_t n, uint8_t* side_is_bid, uint32_t* price_key, uint64_t* t) {
  auto* ob = static_cast<OrderBook*>(b);
  for (uint64_t i = 0; i < n; ++i) {
    const OrderKey& k = ob->orders[first + i].key;
    side_is_bid[i] = side_to_bool(k.side);
    price_key[i] = k.price_key;
    t[i] = k.t;
  }
}
// TryFrom<OrderBookState> (orderbook.rs:891-918): orders + keys + trades copied as given, the two sides rebuilt by
// inserting every Active order under its stored key.
void* orc_book_from_state(uint64_t t, uint32_t tick, uint32_t trade_vol, int trading, int levels, uint64_t n_orders,
                          const void* orders, const uint8_t* key_bid, const uint32_t* key_price, const uint64_t* key_t,
                          uint64_t n_trades, const void* trades) {
  auto* ob = new OrderBook(t, tick, trading != 0, levels);
  ob->trade_vol = trade_vol;
  const auto* ro = static_cast<const OrcOrderRec*>(orders);
  for (uint64_t i = 0; i < n_orders; ++i) {
    Order o{side_from_bool(ro[i].side_is_bid != 0), static_cast<Status>(ro[i].status), ro[i].arr_time, ro[i].end_time,
            ro[i].vol, ro[i].start_vol, ro[i].price, ro[i].trader_id, ro[i].order_id};
    OrderKey k{side_from_bool(key_bid[i] != 0), key_price[i], key_t[i]};
    ob->orders.push_back(OrderEntry{o, k});
    if (o.status == Status::Active) {
      if (o.side == Side::Bid)
        ob->bid_side.s.insert_order(k, o.order_id, o.vol);
      else
        ob->ask_side.s.insert_order(k, o.order_id, o.vol);
    }
  }
  const auto* rt = static_cast<const OrcTradeRec*>(trades);
  for (uint64_t i = 0; i < n_trades; ++i)
    ob->trades.push_back(Trade{rt[i].t, side_from_bool(rt[i].side_is_bid != 0), rt[i].price, rt[i].vol, rt[i].active_id,
                               rt[i].passive_id});
  return ob;
}

// ---------------------------------------------------------------- Env ------
void* orc_env_new(uint64_t seed, uint64_t start, uint32_t tick, uint64_t step, int trading, int levels) {
  return new OrcEnv(seed, start, tick, step, trading != 0, levels);
}
void orc_env_free(void* e) { delete static_cast<OrcEnv*>(e); }
void* orc_env_book(void* e) { return &static_cast<OrcEnv*>(e)->env.order_book; }
void orc_env_rng_state(void* e, uint64_t* st) {
  auto* x = static_cast<OrcEnv*>(e);
  st[0] = x->rng.s0;
  st[1] = x->rng.s1;
}
int orc_env_place_order(void* e, int bid, uint32_t vol, uint32_t trader, int has_price, uint32_t price,
                        uint64_t* out_id) {
  return static_cast<OrcEnv*>(e)->env.place_order(side_from_bool(bid != 0), vol, trader, opt(has_price, price),
                                                  out_id);
}
void orc_env_cancel_order(void* e, uint64_t id) { static_cast<OrcEnv*>(e)->env.cancel_order(id); }
void orc_env_modify_order(void* e, uint64_t id, int has_price, uint32_t price, int has_vol, uint32_t vol) {
  static_cast<OrcEnv*>(e)->env.modify_order(id, opt(has_price, price), opt(has_vol, vol));
}
// StepEnvNumpy.submit_instructions as ONE native call (ref rust/src/step_sim_numpy.rs:233-275: the loop over the six arrays runs
// in Rust there): action 1 = new limit order, 2 = cancellation, anything else nothing (:266); the first price that is not a
// multiple of the tick size stops the batch - earlier elements stay queued (:255-268).  out_ids[i] = the id element i created,
// else u64::MAX.  Returns 0, or 1 for that price error with *applied = elements applied before it.
int orc_env_submit_instructions(void* e, uint64_t n, const uint32_t* action, const uint8_t* side, const uint32_t* vol,
                                const uint32_t* trader, const uint32_t* price, const uint64_t* order_id, uint64_t* out_ids,
                                uint64_t* applied) {
  auto& env = static_cast<OrcEnv*>(e)->env;
  for (uint64_t i = 0; i < n; ++i) {
    if (out_ids) out_ids[i] = ~0ull;
    if (action[i] == 1u) {
      uint64_t id = 0;
      if (int rc = env.place_order(side_from_bool((side[i] & 1u) != 0), vol[i], trader[i], opt(1, price[i]), &id)) {
        if (applied) *applied = i;
        return rc;
      }
      if (out_ids) out_ids[i] = id;
    } else if (action[i] == 2u) {
      env.cancel_order(order_id[i]);
    } else if (action[i] == 0x80000003u) {
      // BK_ACTION_MODIFY (include/bourse_amd.h): the library's one extension of the action codes - Env::modify_order
      // (env.rs:208-219), side bit 1 = has price, bit 2 = has volume; the reference's own submit_instructions ignores it (:266)
      env.modify_order(order_id[i], opt((side[i] & 2u) != 0, price[i]), opt((side[i] & 4u) != 0, vol[i]));
    }
  }
  if (applied) *applied = n;
  return 0;
}
uint64_t orc_env_n_transactions(void* e) { return static_cast<OrcEnv*>(e)->env.transactions.size(); }
// kinds of queued events (0 New, 1 Cancellation, 2 Modify) — for the agent structure tests
void orc_env_transaction_kinds(void* e, uint8_t* out) {
  auto& tx = static_cast<OrcEnv*>(e)->env.transactions;
  for (size_t i = 0; i < tx.size(); ++i) out[i] = tx[i].kind;
}
int orc_env_step(void* e) {  // StepEnv.step, ref rust/src/step_sim.rs:200-203
  auto* x = static_cast<OrcEnv*>(e);
  return x->env.step(x->rng);
}
// numpy level_2_data layout: snapshot + LIVE trade_vol (ref step_sim_numpy.rs:351-368)
void orc_env_level2(void* e, uint32_t* out) {
  auto* x = static_cast<OrcEnv*>(e);
  pack_l2(x->env.level_2_data, x->env.order_book.trade_vol, x->env.order_book.levels, out);
}
uint64_t orc_env_n_steps(void* e) { return static_cast<OrcEnv*>(e)->env.trade_vols.size(); }
// history in the same layout: out[step][5+4L], trade_vol = recorded per-step value
void orc_env_history(void* e, uint32_t* out) {
  auto* x = static_cast<OrcEnv*>(e);
  const auto& r = x->env.level_2_data_records;
  const int L = r.n;
  const size_t T = x->env.trade_vols.size();
  const size_t W = 5 + 4 * static_cast<size_t>(L);
  for (size_t s = 0; s < T; ++s) {
    uint32_t* o = out + s * W;
    o[0] = x->env.trade_vols[s];
    o[1] = r.bid_prices[s];
    o[2] = r.ask_prices[s];
    o[3] = r.ask_vols[s];
    o[4] = r.bid_vols[s];
    for (int i = 0; i < L; ++i) {
      o[5 + 4 * i + 0] = r.bid_vols_at_levels[i][s];
      o[5 + 4 * i + 1] = r.bid_orders_at_levels[i][s];
      o[5 + 4 * i + 2] = r.ask_vols_at_levels[i][s];
      o[5 + 4 * i + 3] = r.ask_orders_at_levels[i][s];
    }
  }
}

// ------------------------------------------------------------- agents ------
void* orc_agents_new() { return new OrcAgents(); }
void orc_agents_free(void* a) { delete static_cast<OrcAgents*>(a); }
void orc_agents_add_random(void* a, uint64_t n, uint32_t tick_lo, uint32_t tick_hi, uint32_t vol_lo,
                           uint32_t vol_hi, uint32_t tick_size, float rate) {
  static_cast<OrcAgents*>(a)->groups.push_back(
      std::make_unique<RandomAgentsBox>(RandomAgents(n, tick_lo, tick_hi, vol_lo, vol_hi, tick_size, rate)));
}
void orc_agents_add_desc(void* a, const void* desc) {
  add_from_desc(*static_cast<OrcAgents*>(a), *static_cast<const OrcAgentDesc*>(desc));
}
// mutate a NoiseAgent's probabilities (the reference's tests do `agents.params.p_limit = 0.0`): which 0 p_limit, 1 p_market, 2 p_cancel
void orc_agents_set_noise_prob(void* a, int g, int which, float v) {
  auto* n = dynamic_cast<NoiseAgent*>(static_cast<OrcAgents*>(a)->groups[g].get());
  if (!n) return;
  if (which == 0) n->params.p_limit = v;
  if (which == 1) n->params.p_market = v;
  if (which == 2) n->params.p_cancel = v;
}
// ids currently held by a Noise/Momentum member (its `orders` list)
uint64_t orc_agents_order_list(void* a, int g, uint64_t* out, uint64_t cap) {
  AgentBase* b = static_cast<OrcAgents*>(a)->groups[g].get();
  const std::vector<OrderId>* v = nullptr;
  if (auto* n = dynamic_cast<NoiseAgent*>(b)) v = &n->orders;
  if (auto* m = dynamic_cast<MomentumAgent*>(b)) v = &m->orders;
  if (!v) return 0;
  for (size_t i = 0; i < v->size() && i < cap; ++i) out[i] = (*v)[i];
  return v->size();
}
// held ids of group g: out[i] = id or u64::MAX for None
void orc_agents_held_ids(void* a, int g, uint64_t* out) {
  auto& grp = dynamic_cast<RandomAgentsBox&>(*static_cast<OrcAgents*>(a)->groups[g]).inner;
  for (size_t i = 0; i < grp.orders.size(); ++i) out[i] = grp.orders[i].value_or(ORDER_ID_MAX);
}
// agents.update(env, rng) with the env's own RNG — one call of the AgentSet
void orc_agents_update(void* a, void* e) {
  auto* x = static_cast<OrcEnv*>(e);
  static_cast<OrcAgents*>(a)->update(x->env, x->rng);
}
// sim_runner body (ref runner.rs:53-68) continuing from an explicit RNG state;
// st = seed_from_u64(seed) for a fresh run.  The env's own RNG is not touched.
int orc_sim_run(void* e, void* a, uint64_t* st, uint64_t n_steps) {
  auto* x = static_cast<OrcEnv*>(e);
  auto* ag = static_cast<OrcAgents*>(a);
  Rng rng{st[0], st[1]};
  int rc = ORC_OK;
  for (uint64_t s = 0; s < n_steps; ++s) {
    ag->update(x->env, rng);
    int r = x->env.step(rng);
    if (r != ORC_OK) rc = r;
  }
  st[0] = rng.s0;
  st[1] = rng.s1;
  return rc;
}

// ---------------------------------------------- many independent books -----
// B independent (Env, agents, rng) triples, book b seeded seed_base + b.
// groups: n_groups rows of {n, tick_lo, tick_hi, vol_lo, vol_hi, tick_size, rate_bits(f32)}
// Threads that CONSTRUCT the books of the next orc_many_new* call (default 1).  With n > 1 the books are built by the same
// static partition orc_many_run(.., n) steps them with: every book's maps and vectors then live in the allocator arena of
// (and are first touched by) the thread that works on them, instead of all in the main thread's arena - the timed CPU
// baseline of bench.py uses it; results do not depend on it.
static int g_build_threads = 1;
void orc_many_set_build_threads(int n) { g_build_threads = n < 1 ? 1 : n; }
}  // extern "C"
template <class F>
static void build_books(OrcMany* m, uint32_t n_books, F&& make) {
  m->books.resize(n_books);
  const int nt = std::min<int>(g_build_threads, std::max<uint32_t>(1u, n_books));
  auto work = [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; ++b) m->books[b] = make(static_cast<uint32_t>(b));
  };
  if (nt == 1) {
    work(0, n_books);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back(work, static_cast<size_t>(n_books) * t / nt, static_cast<size_t>(n_books) * (t + 1) / nt);
  for (auto& t : th) t.join();
}
extern "C" {

void* orc_many_new(uint32_t n_books, uint64_t seed_base, uint64_t start, uint32_t tick, uint64_t step,
                   int trading, int levels, int n_groups, const uint32_t* groups) {
  auto* m = new OrcMany();
  m->levels = levels;
  build_books(m, n_books, [&](uint32_t b) {
    auto bk = std::make_unique<ManyBook>(start, tick, step, trading != 0, levels, seed_base + b);
    for (int g = 0; g < n_groups; ++g) {
      const uint32_t* r = groups + 7 * g;
      float rate;
      std::memcpy(&rate, &r[6], 4);
      bk->agents.groups.push_back(
          std::make_unique<RandomAgentsBox>(RandomAgents(r[0], r[1], r[2], r[3], r[4], r[5], rate)));
    }
    return bk;
  });
  return m;
}
// same with an arbitrary AgentSet given as n_desc OrcAgentDesc records
void* orc_many_new_mixed(uint32_t n_books, uint64_t seed_base, uint64_t start, uint32_t tick, uint64_t step,
                         int trading, int levels, int n_desc, const void* descs) {
  auto* m = new OrcMany();
  m->levels = levels;
  const OrcAgentDesc* d = static_cast<const OrcAgentDesc*>(descs);
  build_books(m, n_books, [&](uint32_t b) {
    auto bk = std::make_unique<ManyBook>(start, tick, step, trading != 0, levels, seed_base + b);
    for (int g = 0; g < n_desc; ++g) add_from_desc(bk->agents, d[g]);
    return bk;
  });
  return m;
}
void orc_many_free(void* m) { delete static_cast<OrcMany*>(m); }

// Run n_steps of sim_runner's loop for every book, books statically partitioned
// over n_threads host threads.
int orc_many_run(void* mp, uint64_t n_steps, int n_threads) {
  auto* m = static_cast<OrcMany*>(mp);
  const size_t B = m->books.size();
  if (n_threads < 1) n_threads = 1;
  std::atomic<int> rc{ORC_OK};
  auto work = [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; ++b) {
      ManyBook& k = *m->books[b];
      for (uint64_t s = 0; s < n_steps; ++s) {
        k.agents.update(k.env, k.rng);
        int r = k.env.step(k.rng);
        if (r != ORC_OK) rc = r;
      }
    }
  };
  if (n_threads == 1) {
    work(0, B);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
      size_t lo = B * t / n_threads, hi = B * (t + 1) / n_threads;
      th.emplace_back(work, lo, hi);
    }
    for (auto& t : th) t.join();
  }
  return rc;
}
uint64_t orc_many_n_steps(void* mp) {
  auto* m = static_cast<OrcMany*>(mp);
  return m->books.empty() ? 0 : m->books[0]->env.trade_vols.size();
}
void* orc_many_book(void* mp, uint32_t b) { return &static_cast<OrcMany*>(mp)->books[b]->env.order_book; }
void orc_many_rng_state(void* mp, uint32_t b, uint64_t* st) {
  auto& k = *static_cast<OrcMany*>(mp)->books[b];
  st[0] = k.rng.s0;
  st[1] = k.rng.s1;
}
// out[step - first_step][book][5+4L] for steps [first_step, first_step + n)
void orc_many_history(void* mp, uint64_t first_step, uint64_t n, uint32_t* out) {
  auto* m = static_cast<OrcMany*>(mp);
  const int L = m->levels;
  const size_t W = 5 + 4 * static_cast<size_t>(L);
  const size_t B = m->books.size();
  for (size_t b = 0; b < B; ++b) {
    const Env& env = m->books[b]->env;
    const auto& r = env.level_2_data_records;
    for (uint64_t s = 0; s < n; ++s) {
      const size_t ss = first_step + s;
      uint32_t* o = out + (s * B + b) * W;
      o[0] = env.trade_vols[ss];
      o[1] = r.bid_prices[ss];
      o[2] = r.ask_prices[ss];
      o[3] = r.ask_vols[ss];
      o[4] = r.bid_vols[ss];
      for (int i = 0; i < L; ++i) {
        o[5 + 4 * i + 0] = r.bid_vols_at_levels[i][ss];
        o[5 + 4 * i + 1] = r.bid_orders_at_levels[i][ss];
        o[5 + 4 * i + 2] = r.ask_vols_at_levels[i][ss];
        o[5 + 4 * i + 3] = r.ask_orders_at_levels[i][ss];
      }
    }
  }
}
void orc_many_trade_counts(void* mp, uint64_t* out) {
  auto* m = static_cast<OrcMany*>(mp);
  for (size_t b = 0; b < m->books.size(); ++b) out[b] = m->books[b]->env.order_book.trades.size();
}
void orc_many_order_counts(void* mp, uint64_t* out) {
  auto* m = static_cast<OrcMany*>(mp);
  for (size_t b = 0; b < m->books.size(); ++b) out[b] = m->books[b]->env.order_book.orders.size();
}

// ---------------------------------------------- many independent MARKETS -----
// n_markets x (MarketEnv, RandomMarketAgents groups, RNG) as market_sim_runner owns them (runner.rs:108-131);
// market m seeded seed_base + m.  groups: rows of {asset, n, tick_lo, tick_hi, vol_lo, vol_hi, tick_size, rate_bits}
struct OneMarket {
  MarketEnv env;
  std::vector<std::unique_ptr<MarketAgentBase>> agents;  // a MarketAgentSet: members updated in declaration order
  Rng rng;
  OneMarket(Nanos start, const std::vector<Price>& ticks, Nanos step, bool trading, int levels, uint64_t seed)
      : env(start, ticks, step, trading, levels), rng(Rng::seed_from_u64(seed)) {}
};
struct OrcMarkets {
  int levels, assets;
  std::vector<std::unique_ptr<OneMarket>> mk;
};

void* orc_mkts_new(uint32_t n_markets, uint64_t seed_base, uint64_t start, uint32_t assets, const uint32_t* tick_sizes,
                   uint64_t step, int trading, int levels, int n_groups, const uint32_t* groups) {
  auto* m = new OrcMarkets();
  m->levels = levels;
  m->assets = static_cast<int>(assets);
  const std::vector<Price> ticks(tick_sizes, tick_sizes + assets);
  for (uint32_t i = 0; i < n_markets; ++i) {
    auto k = std::make_unique<OneMarket>(start, ticks, step, trading != 0, levels, seed_base + i);
    for (int g = 0; g < n_groups; ++g) {
      const uint32_t* r = groups + 8 * g;
      float rate;
      std::memcpy(&rate, &r[7], 4);
      k->agents.push_back(std::make_unique<RandomMarketAgentsBox>(
          RandomMarketAgents(r[0], r[1], r[2], r[3], r[4], r[5], r[6], rate)));
    }
    m->mk.push_back(std::move(k));
  }
  return m;
}
// the same with any mix of market members: n_desc OrcAgentDesc records + the asset each member trades
void* orc_mkts_new_mixed(uint32_t n_markets, uint64_t seed_base, uint64_t start, uint32_t assets,
                         const uint32_t* tick_sizes, uint64_t step, int trading, int levels, int n_desc,
                         const void* descs, const uint32_t* member_asset) {
  auto* m = new OrcMarkets();
  m->levels = levels;
  m->assets = static_cast<int>(assets);
  const std::vector<Price> ticks(tick_sizes, tick_sizes + assets);
  const OrcAgentDesc* d = static_cast<const OrcAgentDesc*>(descs);
  for (uint32_t i = 0; i < n_markets; ++i) {
    auto k = std::make_unique<OneMarket>(start, ticks, step, trading != 0, levels, seed_base + i);
    for (int g = 0; g < n_desc; ++g) {
      const OrcAgentDesc& x = d[g];
      const uint32_t as = member_asset[g];
      if (x.type == 0)
        k->agents.push_back(std::make_unique<RandomMarketAgentsBox>(
            RandomMarketAgents(as, x.n, x.tick_lo, x.tick_hi, x.vol_lo, x.vol_hi, x.tick_size, x.rate)));
      else if (x.type == 1)
        k->agents.push_back(std::make_unique<NoiseMarketAgent>(
            as, x.trader_start, static_cast<uint16_t>(x.n),
            NoiseAgentParams{x.tick_size, x.p_limit, x.p_market, x.p_cancel, x.trade_vol, x.mu, x.sigma}));
      else
        k->agents.push_back(std::make_unique<MomentumMarketAgent>(
            as, x.trader_start, static_cast<uint16_t>(x.n),
            MomentumParams{x.tick_size, x.p_cancel, x.trade_vol, x.decay, x.demand, x.scale, x.order_ratio, x.mu, x.sigma}));
    }
    m->mk.push_back(std::move(k));
  }
  return m;
}
void orc_mkts_free(void* m) { delete static_cast<OrcMarkets*>(m); }
int orc_mkts_run(void* mp, uint64_t n_steps, int n_threads) {
  auto* m = static_cast<OrcMarkets*>(mp);
  const size_t B = m->mk.size();
  if (n_threads < 1) n_threads = 1;
  std::atomic<int> rc{ORC_OK};
  auto work = [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; ++b) {
      OneMarket& k = *m->mk[b];
      for (uint64_t s = 0; s < n_steps; ++s) {
        for (auto& a : k.agents) a->update(k.env, k.rng);  // MarketAgentSet: fields in order
        int r = k.env.step(k.rng);
        if (r != ORC_OK) rc = r;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(work, B * t / n_threads, B * (t + 1) / n_threads);
  for (auto& t : th) t.join();
  return rc;
}
// host-driven MarketEnv calls on market `mi`
int orc_mkts_place(void* mp, uint32_t mi, uint32_t asset, int bid, uint32_t vol, uint32_t trader, int has_price,
                   uint32_t price, uint64_t* out_id) {
  OrderId id = 0;
  const int rc = static_cast<OrcMarkets*>(mp)->mk[mi]->env.place_order(
      asset, side_from_bool(bid != 0), vol, trader, has_price ? std::optional<Price>(price) : std::nullopt, &id);
  if (out_id) *out_id = id;
  return rc;
}
void orc_mkts_cancel(void* mp, uint32_t mi, uint32_t asset, uint64_t id) {
  static_cast<OrcMarkets*>(mp)->mk[mi]->env.cancel_order(asset, id);
}
void orc_mkts_modify(void* mp, uint32_t mi, uint32_t asset, uint64_t id, int has_p, uint32_t p, int has_v, uint32_t v) {
  static_cast<OrcMarkets*>(mp)->mk[mi]->env.modify_order(asset, id, has_p ? std::optional<Price>(p) : std::nullopt,
                                                         has_v ? std::optional<Vol>(v) : std::nullopt);
}
int orc_mkts_step(void* mp) {  // one MarketEnv::step on every market, each with its own RNG
  int rc = ORC_OK;
  for (auto& k : static_cast<OrcMarkets*>(mp)->mk) {
    const int r = k->env.step(k->rng);
    if (r != ORC_OK) rc = r;
  }
  return rc;
}
void orc_mkts_set_trading(void* mp, int on) {
  for (auto& k : static_cast<OrcMarkets*>(mp)->mk) k->env.market.set_trading(on != 0);
}
uint64_t orc_mkts_n_steps(void* mp) {
  auto* m = static_cast<OrcMarkets*>(mp);
  return m->mk.empty() ? 0 : m->mk[0]->env.trade_vols[0].size();
}
void* orc_mkts_book(void* mp, uint32_t mi, uint32_t asset) {
  return &static_cast<OrcMarkets*>(mp)->mk[mi]->env.market.order_books[asset];
}
void orc_mkts_rng_state(void* mp, uint32_t mi, uint64_t* st) {
  auto& k = *static_cast<OrcMarkets*>(mp)->mk[mi];
  st[0] = k.rng.s0;
  st[1] = k.rng.s1;
}
// out[step - first_step][market * assets + asset][5+4L]
void orc_mkts_history(void* mp, uint64_t first_step, uint64_t n, uint32_t* out) {
  auto* m = static_cast<OrcMarkets*>(mp);
  const int L = m->levels;
  const size_t W = 5 + 4 * static_cast<size_t>(L), A = m->assets, B = m->mk.size() * A;
  for (size_t b = 0; b < B; ++b) {
    const MarketEnv& env = m->mk[b / A]->env;
    const auto& r = env.level_2_data_records[b % A];
    for (uint64_t s = 0; s < n; ++s) {
      const size_t ss = first_step + s;
      uint32_t* o = out + (s * B + b) * W;
      o[0] = env.trade_vols[b % A][ss];
      o[1] = r.bid_prices[ss];
      o[2] = r.ask_prices[ss];
      o[3] = r.ask_vols[ss];
      o[4] = r.bid_vols[ss];
      for (int i = 0; i < L; ++i) {
        o[5 + 4 * i + 0] = r.bid_vols_at_levels[i][ss];
        o[5 + 4 * i + 1] = r.bid_orders_at_levels[i][ss];
        o[5 + 4 * i + 2] = r.ask_vols_at_levels[i][ss];
        o[5 + 4 * i + 3] = r.ask_orders_at_levels[i][ss];
      }
    }
  }
}

// ---- sampling and portable-math hooks for unit tests
double orc_rng_f64(uint64_t* st) {
  Rng r{st[0], st[1]};
  double v = gen_f64(r);
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
double orc_rng_std_normal(uint64_t* st) {
  Rng r{st[0], st[1]};
  double v = sample_standard_normal(r);
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
double orc_rng_lognormal(uint64_t* st, double mu, double sigma) {
  Rng r{st[0], st[1]};
  double v = LogNormal{mu, sigma}.sample(r);
  st[0] = r.s0;
  st[1] = r.s1;
  return v;
}
double orc_pm_exp(double x) { return pm::exp(x); }
double orc_pm_log(double x) { return pm::log(x); }
double orc_pm_tanh(double x) { return pm::tanh(x); }
uint32_t orc_round_price_up(double p, double tick) { return round_price_up(p, tick); }
uint32_t orc_round_price_down(double p, double tick) { return round_price_down(p, tick); }

int orc_version() { return 3; }

}  // extern "C"
