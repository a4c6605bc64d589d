// bourse_oracle_agents.cpp — CPU ORACLE (test infrastructure): see bourse_oracle_agents.hpp.
#include "bourse_oracle_agents.hpp"

namespace orc {

namespace {
#define ZIG_TABLE_BEGIN(name) const double name[257] = {
#define ZIG_TABLE_END };
#include "zig_norm_tables.inc"
#undef ZIG_TABLE_BEGIN
#undef ZIG_TABLE_END
constexpr double ZIG_NORM_R = 3.654152885361008796;
}  // namespace

// rand 0.8.5 distributions/float.rs, Standard for f64: 53 random bits scaled by 2^-53 -> [0, 1)
double gen_f64(Rng& rng) { return static_cast<double>(rng.next_u64() >> 11) * (1.0 / 9007199254740992.0); }

// rand 0.8.5 distributions/bernoulli.rs: p_int = (p * 2^64) as u64 = 2^63 for p = 0.5; sample = next_u64() < p_int
// (call sites: ref noise_agent.rs:135,163)
bool gen_bool_half(Rng& rng) { return rng.next_u64() < 0x8000000000000000ull; }

// rand 0.8.5 Open01 for f64: (u64 >> 12) into [1,2) minus (1 - eps/2)  ->  (0, 1)
double gen_open01(Rng& rng) {
  const uint64_t fraction = rng.next_u64() >> 12;
  const double v = pm::from_bits(0x3FF0000000000000ull | fraction);
  return v - (1.0 - 2.220446049250313e-16 / 2.0);
}

// rand_distr 0.4.3 normal.rs StandardNormal + utils.rs ziggurat(symmetric = true)
double sample_standard_normal(Rng& rng) {
  for (;;) {
    const uint64_t bits = rng.next_u64();
    const unsigned i = static_cast<unsigned>(bits & 0xff);
    // (bits >> 12).into_float_with_exponent(1) - 3.0  ->  [-1, 1)
    const double u = pm::from_bits(0x4000000000000000ull | (bits >> 12)) - 3.0;
    const double x = u * ZIG_NORM_X[i];
    if (pm::fabs_(x) < ZIG_NORM_X[i + 1]) return x;
    if (i == 0) {  // zero_case: sample the tail by hand
      double xx = 1.0, yy = 0.0;
      while (-2.0 * yy < xx * xx) {
        const double x_ = gen_open01(rng);
        const double y_ = gen_open01(rng);
        xx = pm::log(x_) / ZIG_NORM_R;
        yy = pm::log(y_);
      }
      return (u < 0.0) ? xx - ZIG_NORM_R : ZIG_NORM_R - xx;
    }
    // f1 + U * (f0 - f1) < pdf(x), pdf(x) = exp(-x^2 / 2)
    if (ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * gen_f64(rng) < pm::exp(-x * x / 2.0)) return x;
  }
}

// rand_distr 0.4.3: LogNormal::sample = Normal{mu, sigma}.sample(rng).exp(); Normal::sample = mean + std_dev * z
double LogNormal::sample(Rng& rng) const { return pm::exp(mu + sigma * sample_standard_normal(rng)); }

// ref common.rs:21-25 — ceil to a tick multiple, clamp to [0, u32::MAX], saturating cast
Price round_price_up(double p, double tick_size) {
  p = pm::ceil_(p / tick_size) * tick_size;
  if (!(p == p)) return 0;  // NaN as u32 == 0
  if (p < 0.0) p = 0.0;
  if (p > 4294967295.0) p = 4294967295.0;
  return static_cast<Price>(p);
}
// ref common.rs:36-40
Price round_price_down(double p, double tick_size) {
  p = pm::floor_(p / tick_size) * tick_size;
  if (!(p == p)) return 0;
  if (p < 0.0) p = 0.0;
  if (p > 4294967295.0) p = 4294967295.0;
  return static_cast<Price>(p);
}

// The single-asset agents and their multi-asset twins run the same update against "the book I trade on": a small
// access adapter (Env, or one asset of a MarketEnv) keeps one restatement of the logic for both
// (ref common.rs:54-141 vs :156-258 differ only in `env.place_order(asset, ..)` / MarketOrderId).
namespace {
struct EnvAccess {
  Env& env;
  const OrderBook& book() const { return env.order_book; }
  void cancel(OrderId id) { env.cancel_order(id); }
  // The reference `.unwrap()`s create_order's Result (common.rs:107,140): an Err (a limit price that the u32::MAX clamp
  // left off the tick grid) is a panic there.  Here, as on the device, such an order simply does not exist: no id, no
  // event, nothing remembered by the agent (nullopt).
  std::optional<OrderId> place(Side side, Vol vol, TraderId trader, std::optional<Price> price) {
    OrderId id = 0;
    if (env.place_order(side, vol, trader, price, &id) != 0) return std::nullopt;
    return id;
  }
};
struct MarketAccess {
  MarketEnv& env;
  uint32_t asset;
  const OrderBook& book() const { return env.market.order_books[asset]; }
  void cancel(OrderId id) { env.cancel_order(asset, id); }
  std::optional<OrderId> place(Side side, Vol vol, TraderId trader, std::optional<Price> price) {
    OrderId id = 0;
    if (env.place_order(asset, side, vol, trader, price, &id) != 0) return std::nullopt;
    return id;
  }
};

// ref common.rs:54-76 (/ :156-175): filter Active, then partition by `gen::<f32>() > p_cancel` (kept) in list order
template <class A>
std::vector<OrderId> cancel_live(A acc, Rng& rng, const std::vector<OrderId>& orders, float p_cancel) {
  std::vector<OrderId> live, to_cancel;
  for (OrderId id : orders) {
    if (acc.book().orders[id].order.status != Status::Active) continue;
    if (rng.gen_f32() > p_cancel)
      live.push_back(id);
    else
      to_cancel.push_back(id);
  }
  for (OrderId id : to_cancel) acc.cancel(id);
  return live;
}
// ref common.rs:92-108 / :124-141 (/ :197-258)
template <class A>
std::optional<OrderId> place_buy_limit(A acc, Rng& rng, const LogNormal& d, double mid, double tick, Vol vol, TraderId trader) {
  const double dist = pm::fabs_(d.sample(rng));
  return acc.place(Side::Bid, vol, trader, round_price_down(mid - dist, tick));
}
template <class A>
std::optional<OrderId> place_sell_limit(A acc, Rng& rng, const LogNormal& d, double mid, double tick, Vol vol, TraderId trader) {
  const double dist = pm::fabs_(d.sample(rng));
  return acc.place(Side::Ask, vol, trader, round_price_up(mid + dist, tick));
}

template <class A>
void noise_update(NoiseAgent& g, A acc, Rng& rng) {  // noise_agent.rs:127-176 (/ :281-339)
  std::vector<OrderId> live = cancel_live(acc, rng, g.orders, g.params.p_cancel);
  const double mid = acc.book().mid_price();
  for (TraderId trader : g.trader_ids) {
    if (rng.gen_f32() < g.params.p_limit) {
      const bool buy = gen_bool_half(rng);
      const std::optional<OrderId> id =
          buy ? place_buy_limit(acc, rng, g.price_dist, mid, g.tick_size, g.params.trade_vol, trader)
              : place_sell_limit(acc, rng, g.price_dist, mid, g.tick_size, g.params.trade_vol, trader);
      if (id) live.push_back(*id);
    }
    if (rng.gen_f32() < g.params.p_market) {
      const bool buy = gen_bool_half(rng);
      acc.place(buy ? Side::Bid : Side::Ask, g.params.trade_vol, trader, std::nullopt);
    }
  }
  g.orders = live;
}

template <class A>
void momentum_update(MomentumAgent& g, A acc, Rng& rng) {  // momentum_agent.rs:146-208 (/ :328-396)
  std::vector<OrderId> live = cancel_live(acc, rng, g.orders, g.params.p_cancel);
  const double mid = acc.book().mid_price();
  double m = 0.0, p_market = 0.0;
  if (g.has_last_price) {
    m = g.momentum * (1.0 - g.params.decay) + g.params.decay * (mid - g.last_price);
    p_market = g.params.demand * pm::tanh(g.params.scale * m) / g.n;
  }
  const double p_limit = g.params.order_ratio * p_market;
  for (TraderId trader : g.trader_ids) {
    if (gen_f64(rng) < p_limit) {
      std::optional<OrderId> id;
      if (m > 0.0)
        id = place_buy_limit(acc, rng, g.price_dist, mid, g.tick_size, g.params.trade_vol, trader);
      else if (m < 0.0)
        id = place_sell_limit(acc, rng, g.price_dist, mid, g.tick_size, g.params.trade_vol, trader);
      if (id) live.push_back(*id);
    }
    if (gen_f64(rng) < p_market) {
      if (m > 0.0)
        acc.place(Side::Bid, g.params.trade_vol, trader, std::nullopt);
      else if (m < 0.0)
        acc.place(Side::Ask, g.params.trade_vol, trader, std::nullopt);
    }
  }
  g.momentum = m;
  g.last_price = mid;
  g.has_last_price = true;
  g.orders = live;
}
}  // namespace

std::vector<OrderId> cancel_live_orders(Env& env, Rng& rng, const std::vector<OrderId>& orders, float p_cancel) {
  return cancel_live(EnvAccess{env}, rng, orders, p_cancel);
}

NoiseAgent::NoiseAgent(TraderId agent_id_start, uint16_t n_agents, NoiseAgentParams p)  // noise_agent.rs:110-123
    : tick_size(static_cast<double>(p.tick_size)), price_dist{p.price_dist_mu, p.price_dist_sigma}, params(p) {
  for (TraderId t = agent_id_start; t < agent_id_start + n_agents; ++t) trader_ids.push_back(t);
}
void NoiseAgent::update(Env& env, Rng& rng) { noise_update(*this, EnvAccess{env}, rng); }
void NoiseMarketAgent::update(MarketEnv& env, Rng& rng) { noise_update(core, MarketAccess{env, asset}, rng); }

MomentumAgent::MomentumAgent(TraderId agent_id_start, uint16_t n_agents, MomentumParams p)  // momentum_agent.rs:128-142
    : price_dist{p.price_dist_mu, p.price_dist_sigma}, n(static_cast<double>(n_agents)),
      tick_size(static_cast<double>(p.tick_size)), params(p) {
  for (TraderId t = agent_id_start; t < agent_id_start + n_agents; ++t) trader_ids.push_back(t);
}
void MomentumAgent::update(Env& env, Rng& rng) { momentum_update(*this, EnvAccess{env}, rng); }
void MomentumMarketAgent::update(MarketEnv& env, Rng& rng) { momentum_update(core, MarketAccess{env, asset}, rng); }

}  // namespace orc
