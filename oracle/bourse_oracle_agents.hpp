// bourse_oracle_agents.hpp — CPU ORACLE (test infrastructure): NoiseAgent, MomentumAgent and the rand /
// rand_distr sampling they use.  See bourse_oracle.hpp for the usage restrictions.
//
// PARITY STATUS: "PARITY UNPINNED" against the Rust reference, twice over:
//   * the sampling arithmetic is third-party (rand 0.8.5 Standard<f64>/Bernoulli/Open01, rand_distr 0.4.3
//     StandardNormal ziggurat + LogNormal), restated from the crates' published algorithms; the ziggurat tables are
//     regenerated with the crates' recipe (tools/gen_zig_tables.py reproduces the remembered constants);
//   * exp / ln / tanh come from oracle/pm_math.hpp (portable, <= 2 ulp from libm) instead of the platform libm so
//     that the HIP path can be compared bit-for-bit; against Rust + libm this is statistical parity only.
// The agents' LOGIC is a literal restatement; each function cites its reference lines.
#pragma once
#include <memory>
#include <vector>

#include "bourse_oracle.hpp"
#include "pm_math.hpp"

namespace orc {

// ---- rand 0.8.5 / rand_distr 0.4.3 sampling (restated) -------------------------------------------------
double gen_f64(Rng& rng);                 // Standard: (next_u64 >> 11) * 2^-53
bool gen_bool_half(Rng& rng);             // Bernoulli(0.5): next_u64 < 2^63
double gen_open01(Rng& rng);              // Open01<f64>
double sample_standard_normal(Rng& rng);  // StandardNormal via ziggurat (256 layers)
struct LogNormal {                        // LogNormal::new(mu, sigma): exp(mu + sigma * N)
  double mu, sigma;
  double sample(Rng& rng) const;
};

// ---- agents::common (ref crates/step_sim/src/agents/common.rs) ----------------------------------------
Price round_price_up(double p, double tick_size);    // :21-25
Price round_price_down(double p, double tick_size);  // :36-40
std::vector<OrderId> cancel_live_orders(Env& env, Rng& rng, const std::vector<OrderId>& orders, float p_cancel);  // :54-76

struct AgentBase {
  virtual ~AgentBase() = default;
  virtual void update(Env& env, Rng& rng) = 0;  // Agent::update, agents/mod.rs:46-55
};

struct RandomAgentsBox : AgentBase {
  RandomAgents inner;
  explicit RandomAgentsBox(RandomAgents r) : inner(std::move(r)) {}
  void update(Env& env, Rng& rng) override { inner.update(env, rng); }
};

// ref crates/step_sim/src/agents/noise_agent.rs:24-177
struct NoiseAgentParams {
  Price tick_size;
  float p_limit, p_market, p_cancel;
  Vol trade_vol;
  double price_dist_mu, price_dist_sigma;
};
struct NoiseAgent : AgentBase {
  double tick_size;
  LogNormal price_dist;
  std::vector<OrderId> orders;
  std::vector<TraderId> trader_ids;
  NoiseAgentParams params;
  NoiseAgent(TraderId agent_id_start, uint16_t n_agents, NoiseAgentParams p);
  void update(Env& env, Rng& rng) override;
};

// ref crates/step_sim/src/agents/momentum_agent.rs:24-209
struct MomentumParams {
  Price tick_size;
  float p_cancel;
  Vol trade_vol;
  double decay, demand, scale, order_ratio, price_dist_mu, price_dist_sigma;
};
struct MomentumAgent : AgentBase {
  LogNormal price_dist;
  std::vector<OrderId> orders;
  std::vector<TraderId> trader_ids;
  bool has_last_price = false;
  double last_price = 0.0, momentum = 0.0, n, tick_size;
  MomentumParams params;
  MomentumAgent(TraderId agent_id_start, uint16_t n_agents, MomentumParams p);
  void update(Env& env, Rng& rng) override;
};

// ---- multi-asset twins (ref noise_agent.rs:226-340 NoiseMarketAgent, momentum_agent.rs:282-397 MomentumMarketAgent,
// common.rs:156-258 *_market helpers): the same update addressed to one asset of a MarketEnv, drawing from the MARKET's
// RNG; orders are MarketOrderId = (asset, id) on that asset's book.
struct MarketAgentBase {
  virtual ~MarketAgentBase() = default;
  virtual void update(MarketEnv& env, Rng& rng) = 0;  // MarketAgent::update, agents/mod.rs:162-170
};
struct RandomMarketAgentsBox : MarketAgentBase {
  RandomMarketAgents inner;
  explicit RandomMarketAgentsBox(RandomMarketAgents r) : inner(std::move(r)) {}
  void update(MarketEnv& env, Rng& rng) override { inner.update(env, rng); }
};
struct NoiseMarketAgent : MarketAgentBase {
  uint32_t asset;
  NoiseAgent core;  // parameters, trader ids and the `orders` list (ids on the asset's book)
  NoiseMarketAgent(uint32_t asset_, TraderId agent_id_start, uint16_t n_agents, NoiseAgentParams p)
      : asset(asset_), core(agent_id_start, n_agents, p) {}
  void update(MarketEnv& env, Rng& rng) override;
};
struct MomentumMarketAgent : MarketAgentBase {
  uint32_t asset;
  MomentumAgent core;
  MomentumMarketAgent(uint32_t asset_, TraderId agent_id_start, uint16_t n_agents, MomentumParams p)
      : asset(asset_), core(agent_id_start, n_agents, p) {}
  void update(MarketEnv& env, Rng& rng) override;
};

}  // namespace orc
