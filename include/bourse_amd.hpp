// bourse_amd.hpp — C++17 host-side mirror of the reference's Rust surface over the C ABI (bourse_amd.h).
//
// The reference's host language is Rust; the build image has no cargo/rustc, so the compiled-language mirror of
// `bourse_de::Env`, the `Agent` trait and `sim_runner` is this header (INTEGRATION.md shows the equivalent Rust shim).
// Names, argument meaning and error behaviour follow the reference (paths relative to the reference repository):
//   Env            crates/step_sim/src/env.rs:58-295        -> bourse_amd::Env (one book of a ManyEnv)
//   OrderError     crates/order_book/src/orderbook.rs:127-142 -> bourse_amd::OrderError (the Result's Err, thrown)
//   Agent          crates/step_sim/src/agents/mod.rs:46-55   -> bourse_amd::Agent
//   sim_runner     crates/step_sim/src/runner.rs:46-69       -> bourse_amd::sim_runner
// Header-only; link libbourse_amd.so.  No CPU fallback: without a GPU the ManyEnv constructor throws.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bourse_amd.h"

namespace bourse_amd {

using OrderId = uint64_t;  // usize
using Nanos = uint64_t;
using Price = uint32_t;
using Vol = uint32_t;
using TraderId = uint32_t;
using OrderCount = uint32_t;
enum class Side : uint8_t { Bid, Ask };                                   // types.rs:26-47
enum class Status : uint8_t { New, Active, Filled, Cancelled, Rejected };  // types.rs:51-75

struct OrderError : std::runtime_error {  // OrderError::PriceError { price, tick_size }
  using std::runtime_error::runtime_error;
};
struct Error : std::runtime_error {  // everything the reference would panic on / device capacities / HIP errors
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc == BK_OK) return;
  if (rc == BK_PRICE_NOT_TICK_MULTIPLE) throw OrderError(bk_last_error());
  throw Error(rc, bk_last_error());
}

struct Trade {  // types.rs:103-118
  Nanos t;
  Side side;
  Price price;
  Vol vol;
  OrderId active_order_id, passive_order_id;
};
struct Order {  // types.rs:79-99
  Side side;
  Status status;
  Nanos arr_time, end_time;
  Vol vol, start_vol;
  Price price;
  TraderId trader_id;
  OrderId order_id;
};
struct Level2Data {  // types.rs:272-285 with LEVELS a run-time value
  Price bid_price, ask_price;
  Vol bid_vol, ask_vol;
  std::vector<std::pair<Vol, OrderCount>> bid_price_levels, ask_price_levels;
};

class ManyEnv;

// One book of a ManyEnv with the method set of `bourse_de::Env`.  A cheap view: copy freely.
class Env {
 public:
  Env(bk_env* h, uint32_t book, uint32_t levels) : h_(h), book_(book), levels_(levels) {}
  // Env::place_order (env.rs:166-176): Err(PriceError) -> throws OrderError, nothing created or queued
  OrderId place_order(Side side, Vol vol, TraderId trader_id, std::optional<Price> price) {
    uint64_t id = 0;
    check(bk_place_order(h_, book_, side == Side::Bid, vol, trader_id, price.has_value(), price.value_or(0), &id));
    return id;
  }
  void cancel_order(OrderId id) { check(bk_cancel_order(h_, book_, id)); }  // env.rs:189-191
  void modify_order(OrderId id, std::optional<Price> new_price, std::optional<Vol> new_vol) {  // env.rs:208-219
    check(bk_modify_order(h_, book_, id, new_price.has_value(), new_price.value_or(0), new_vol.has_value(),
                          new_vol.value_or(0)));
  }
  Status order_status(OrderId id) const {  // env.rs:288-290
    uint8_t s = 0;
    check(bk_order_status(h_, book_, id, &s));
    return static_cast<Status>(s);
  }
  Level2Data level_2_data() const {  // env.rs:293-295 (the end-of-step snapshot)
    std::vector<uint32_t> w(5 + 4 * levels_);
    check(bk_level2(h_, book_, 1, w.data()));
    Level2Data d{w[1], w[2], w[4], w[3], {}, {}};
    for (uint32_t i = 0; i < levels_; ++i) {
      d.bid_price_levels.emplace_back(w[5 + 4 * i], w[6 + 4 * i]);
      d.ask_price_levels.emplace_back(w[7 + 4 * i], w[8 + 4 * i]);
    }
    return d;
  }
  Vol last_trade_vol() const {  // get_trade_vols().last()
    std::vector<uint32_t> w(5 + 4 * levels_);
    check(bk_level2(h_, book_, 1, w.data()));
    return w[0];
  }
  std::vector<Trade> get_trades() const {  // env.rs:277-280
    uint64_t total = 0, base = 0;
    check(bk_trade_count(h_, book_, &total, &base));
    std::vector<bk_trade> raw(total - base);
    if (!raw.empty()) check(bk_get_trades(h_, book_, base, raw.size(), raw.data()));
    std::vector<Trade> out;
    for (const bk_trade& t : raw)
      out.push_back(Trade{t.t, t.side_is_bid ? Side::Bid : Side::Ask, t.price, t.vol, t.active_order_id, t.passive_order_id});
    return out;
  }
  std::vector<Order> get_orders() const {  // env.rs:262-264
    uint64_t n = 0;
    check(bk_order_count(h_, book_, &n));
    std::vector<bk_order> raw(n);
    if (n) check(bk_get_orders(h_, book_, 0, n, raw.data()));
    std::vector<Order> out;
    for (const bk_order& o : raw)
      out.push_back(Order{o.side_is_bid ? Side::Bid : Side::Ask, static_cast<Status>(o.status), o.arr_time, o.end_time, o.vol,
                          o.start_vol, o.price, o.trader_id, o.order_id});
    return out;
  }
  Nanos time() const {
    uint64_t t = 0;
    check(bk_time(h_, book_, &t));
    return t;
  }
  uint32_t book() const { return book_; }

 private:
  bk_env* h_;
  uint32_t book_, levels_;
};

// B independent `Env`s stepped in lockstep on one MI355X: `Env::new(start_time, tick_size, step_size, trading)` per book,
// book b seeded seed + book_offset + b.  `env(b)` is book b's Env.
class ManyEnv {
 public:
  ManyEnv(uint32_t n_books, uint64_t seed, Nanos start_time, Price tick_size, Nanos step_size, bool trading = true,
          uint32_t levels = 10, uint32_t max_live_orders = 128, uint32_t max_orders = 1 << 14,
          uint32_t trade_capacity = 1 << 14, uint32_t history_capacity = 0, int device = 0, uint64_t book_offset = 0) {
    bk_config c{};
    c.n_books = n_books;
    c.levels = levels;
    c.start_time = start_time;
    c.tick_size = tick_size;
    c.trading = trading;
    c.step_size = step_size;
    c.seed = seed;
    c.book_offset = book_offset;
    c.max_live_orders = max_live_orders;
    c.max_orders = max_orders;
    c.trade_capacity = trade_capacity;
    c.history_capacity = history_capacity;
    c.device = device;
    check(bk_env_create(&c, &h_));
    n_books_ = n_books;
    levels_ = levels;
  }
  ~ManyEnv() { bk_env_destroy(h_); }
  ManyEnv(const ManyEnv&) = delete;
  ManyEnv& operator=(const ManyEnv&) = delete;

  Env env(uint32_t book) { return Env(h_, book, levels_); }
  uint32_t n_books() const { return n_books_; }
  void step() { check(bk_step(h_)); }                        // Env::step (env.rs:116-135) for every book
  void enable_trading() { check(bk_enable_trading(h_, 1)); }  // env.rs:138-145
  void disable_trading() { check(bk_enable_trading(h_, 0)); }
  // on-device order flow: sim_runner's loop with RandomAgents groups (runner.rs:53-68)
  void set_random_agents(const std::vector<bk_random_agents>& groups) {
    check(bk_set_random_agents(h_, static_cast<uint32_t>(groups.size()), groups.data()));
  }
  void run(uint64_t n_steps) {
    check(bk_run(h_, n_steps));
    check(bk_env_sync(h_));
  }
  bk_env* handle() { return h_; }

 private:
  bk_env* h_ = nullptr;
  uint32_t n_books_ = 0, levels_ = 10;
};

// `Agent::update(&mut self, env: &mut Env, rng: &mut R)` (agents/mod.rs:46-55).  The device owns each book's
// xoroshiro128** stream and spends it on the shuffle only; a host-side agent draws from its own generator `Rng`,
// exactly as the reference's Python agents draw from numpy's (src/bourse/step_sim/runner.py:100).
template <class Rng>
struct Agent {
  virtual ~Agent() = default;
  virtual void update(Env& env, Rng& rng) = 0;
};

// `sim_runner(env, agents, seed, n_steps, _)` (runner.rs:46-69) for every book: agents[b] is book b's AgentSet (its
// members in declaration order), rngs[b] its host generator.
template <class Rng>
void sim_runner(ManyEnv& many, std::vector<std::vector<Agent<Rng>*>>& agents, std::vector<Rng>& rngs, uint64_t n_steps) {
  for (uint64_t s = 0; s < n_steps; ++s) {
    for (uint32_t b = 0; b < many.n_books(); ++b) {
      Env e = many.env(b);
      for (Agent<Rng>* a : agents[b]) a->update(e, rngs[b]);
    }
    many.step();
  }
}

}  // namespace bourse_amd
