// C++ host-side use of include/bourse_amd.hpp: user-defined agents written against `Env` (place / cancel / order_status,
// as a Rust `impl Agent` would be) drive 24 books on the GPU; the same agents drive the CPU oracle's `orc::Env`s with the
// same per-book shuffle seeds; level-2 records, trades and orders must agree exactly.
// Build: g++ -std=c++17 -Iinclude -Ioracle tests/cpp/env_mirror.cpp -Lbourse_amd/csrc -lbourse_amd -Loracle -lbourse_oracle
// Exit 0 = parity, 77 = no GPU.  (tests/ may link the oracle; the product never does.)
#include <cstdio>
#include <memory>
#include <random>

#include "bourse_amd.hpp"
#include "bourse_oracle.hpp"

namespace ba = bourse_amd;

// A momentum-free "ping-pong" market maker + a noise taker, both written once against a minimal env concept.
template <class EnvT, class SideT>
struct Maker {
  uint32_t trader;
  std::vector<uint64_t> mine;
  template <class Rng, class StatusFn>
  void update_impl(EnvT& env, Rng& rng, StatusFn is_active) {
    std::vector<uint64_t> keep;
    for (uint64_t id : mine) {
      if (!is_active(id)) continue;
      if (rng() % 4 == 0)
        env_cancel(env, id);
      else
        keep.push_back(id);
    }
    mine.swap(keep);
    const uint32_t mid = 100 + rng() % 5;
    mine.push_back(env_place(env, true, 10 + rng() % 20, trader, mid - 1 - rng() % 3));
    mine.push_back(env_place(env, false, 10 + rng() % 20, trader, mid + 1 + rng() % 3));
    if (rng() % 3 == 0) env_place_market(env, rng() % 2, 5 + rng() % 30, trader + 100);
  }
  // the two back-ends
  static void env_cancel(ba::Env& e, uint64_t id) { e.cancel_order(id); }
  static uint64_t env_place(ba::Env& e, bool bid, uint32_t vol, uint32_t tr, uint32_t p) {
    return e.place_order(bid ? ba::Side::Bid : ba::Side::Ask, vol, tr, p);
  }
  static uint64_t env_place_market(ba::Env& e, bool bid, uint32_t vol, uint32_t tr) {
    return e.place_order(bid ? ba::Side::Bid : ba::Side::Ask, vol, tr, std::nullopt);
  }
  static void env_cancel(orc::Env& e, uint64_t id) { e.cancel_order(id); }
  static uint64_t env_place(orc::Env& e, bool bid, uint32_t vol, uint32_t tr, uint32_t p) {
    orc::OrderId id = 0;
    e.place_order(bid ? orc::Side::Bid : orc::Side::Ask, vol, tr, p, &id);
    return id;
  }
  static uint64_t env_place_market(orc::Env& e, bool bid, uint32_t vol, uint32_t tr) {
    orc::OrderId id = 0;
    e.place_order(bid ? orc::Side::Bid : orc::Side::Ask, vol, tr, std::nullopt, &id);
    return id;
  }
};

struct GpuMaker : ba::Agent<std::mt19937>, Maker<ba::Env, ba::Side> {
  explicit GpuMaker(uint32_t t) { trader = t; }
  void update(ba::Env& env, std::mt19937& rng) override {
    update_impl(env, rng, [&](uint64_t id) { return env.order_status(id) == ba::Status::Active; });
  }
};

#define EXPECT(c)                                                  \
  do {                                                             \
    if (!(c)) {                                                    \
      std::fprintf(stderr, "line %d: %s\n", __LINE__, #c);         \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main() {
  const uint32_t B = 24, T = 40, L = 10;
  std::unique_ptr<ba::ManyEnv> many;
  try {
    many = std::make_unique<ba::ManyEnv>(B, 77, 0, 1, 1000, true, L, 256, 4096, 8192, 0);
  } catch (const ba::Error& e) {
    if (e.code == BK_NO_DEVICE) {
      std::printf("no HIP device: %s\n", e.what());
      return 77;
    }
    throw;
  }
  // error behaviour of Env::place_order: Err(PriceError) -> OrderError, nothing queued
  {
    ba::ManyEnv t2(1, 0, 0, 2, 1000);
    bool threw = false;
    try {
      t2.env(0).place_order(ba::Side::Bid, 1, 0, 11);
    } catch (const ba::OrderError& e) {
      threw = std::string(e.what()).find("Price 11 was not a multiple of tick-size 2") != std::string::npos;
    }
    EXPECT(threw);
  }
  std::vector<std::vector<ba::Agent<std::mt19937>*>> agents(B);
  std::vector<std::mt19937> rngs;
  std::vector<GpuMaker> makers;
  makers.reserve(2 * B);
  for (uint32_t b = 0; b < B; ++b) {
    rngs.emplace_back(1000 + b);
    makers.emplace_back(1);
    makers.emplace_back(2);
    agents[b] = {&makers[2 * b], &makers[2 * b + 1]};
  }
  ba::sim_runner(*many, agents, rngs, T);

  // the same agents on the oracle (one orc::Env + shuffle RNG per book, seed 77 + b)
  for (uint32_t b = 0; b < B; ++b) {
    orc::Env env(0, 1, 1000, true, L);
    orc::Rng shuffle_rng = orc::Rng::seed_from_u64(77 + b);
    std::mt19937 rng(1000 + b);
    Maker<orc::Env, orc::Side> m1{1, {}}, m2{2, {}};
    for (uint32_t s = 0; s < T; ++s) {
      auto active = [&](uint64_t id) { return env.order_book.orders[id].order.status == orc::Status::Active; };
      m1.update_impl(env, rng, active);
      m2.update_impl(env, rng, active);
      env.step(shuffle_rng);
    }
    ba::Env g = many->env(b);
    const ba::Level2Data d = g.level_2_data();
    EXPECT(d.bid_price == env.level_2_data.bid_price && d.ask_price == env.level_2_data.ask_price);
    EXPECT(d.bid_vol == env.level_2_data.bid_vol && d.ask_vol == env.level_2_data.ask_vol);
    for (uint32_t i = 0; i < L; ++i) {
      EXPECT(d.bid_price_levels[i] == env.level_2_data.bid_price_levels[i]);
      EXPECT(d.ask_price_levels[i] == env.level_2_data.ask_price_levels[i]);
    }
    const auto tr = g.get_trades();
    EXPECT(tr.size() == env.order_book.trades.size());
    for (size_t i = 0; i < tr.size(); ++i) {
      const orc::Trade& o = env.order_book.trades[i];
      EXPECT(tr[i].t == o.t && tr[i].price == o.price && tr[i].vol == o.vol && tr[i].active_order_id == o.active_order_id &&
             tr[i].passive_order_id == o.passive_order_id && (tr[i].side == ba::Side::Bid) == (o.side == orc::Side::Bid));
    }
    const auto od = g.get_orders();
    EXPECT(od.size() == env.order_book.orders.size());
    for (size_t i = 0; i < od.size(); ++i) {
      const orc::Order& o = env.order_book.orders[i].order;
      EXPECT(static_cast<int>(od[i].status) == static_cast<int>(o.status) && od[i].vol == o.vol && od[i].price == o.price &&
             od[i].arr_time == o.arr_time && od[i].end_time == o.end_time && od[i].trader_id == o.trader_id);
    }
    EXPECT(g.time() == static_cast<uint64_t>(T) * 1000);
  }
  std::printf("env_mirror: ok (%u books x %u steps, user-defined C++ agents, parity with the oracle)\n", B, T);
  return 0;
}
