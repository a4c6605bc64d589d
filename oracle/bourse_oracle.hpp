// bourse_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A literal C++17 restatement of the reference's single-book limit-order-book
// step simulator: `bourse_book::OrderBook` + `bourse_de::Env::step` + the
// `RandomAgents` order-flow generator, with the third-party RNG arithmetic
// (rand 0.8.5 / rand_core 0.6.4 / rand_xoshiro 0.6.0, pinned in the
// reference's Cargo.lock:623-669) restated from their published algorithms.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// use anything in this directory.  Nothing under bourse_amd/ links, imports or
// executes it.
//
// PARITY STATUS
//   * Matching engine / Env::step / L2 snapshot / numpy layouts: PINNED by the
//     reference's own known-answer tests (re-expressed in tests/test_oracle_kat.py,
//     every vector cited to its reference file:line) and by running the
//     reference's own Python test-suite against this oracle in the build
//     container (oracle/run_reference_pytests.py).
//   * Everything that depends on RNG output (shuffle order, RandomAgents
//     samples): "PARITY UNPINNED".  The Rust reference cannot be compiled here
//     (no cargo/rustc) and none of its tests asserts an RNG-dependent value.
//     The generator recurrence itself is pinned by xoroshiro128**'s published
//     known-answer vector (tests/test_oracle_rng.py); seed_from_u64 /
//     gen_range / shuffle / choose / f32 follow the crates' algorithms as
//     restated below and are the specification for the HIP path.
//     To pin them: integration/rust/pin_rng (real crates) must print exactly
//     tests/golden/rng_pin_expected.txt (this oracle, tools/gen_rng_pin.py).
//
// All `ref:` citations are paths relative to /root/reference.
#pragma once
#include <cstdint>
#include <map>
#include <optional>
#include <utility>
#include <vector>

namespace orc {

// ref: crates/order_book/src/types.rs:6-22 (scalar aliases)
using OrderId = uint64_t;  // usize on the 64-bit reference targets
using Nanos = uint64_t;
using Price = uint32_t;
using Vol = uint32_t;
using TraderId = uint32_t;
using OrderCount = uint32_t;
constexpr Price PRICE_MAX = 0xFFFFFFFFu;
constexpr Nanos NANOS_MAX = ~0ull;
constexpr OrderId ORDER_ID_MAX = ~0ull;  // usize::MAX

// ref: types.rs:26-47 — Side; bool conversion: true = Bid.
enum class Side : uint8_t { Bid = 0, Ask = 1 };
inline bool side_to_bool(Side s) { return s == Side::Bid; }
inline Side side_from_bool(bool b) { return b ? Side::Bid : Side::Ask; }

// ref: types.rs:51-75 — Status and its u8 mapping.
enum class Status : uint8_t { New = 0, Active = 1, Filled = 2, Cancelled = 3, Rejected = 4 };

// ref: types.rs:79-99
struct Order {
  Side side;
  Status status;
  Nanos arr_time;
  Nanos end_time;
  Vol vol;
  Vol start_vol;
  Price price;
  TraderId trader_id;
  OrderId order_id;
};

// ref: types.rs:103-118
struct Trade {
  Nanos t;
  Side side;
  Price price;
  Vol vol;
  OrderId active_order_id;
  OrderId passive_order_id;
};

// ref: types.rs:8 — OrderKey = (Side, u32 price_key, u64 t)
struct OrderKey {
  Side side;
  uint32_t price_key;
  Nanos t;
};

// ref: types.rs:229-249 — Event<ID>
struct Event {
  enum Kind : uint8_t { NewOrder = 0, Cancellation = 1, Modify = 2 } kind;
  OrderId order_id;
  std::optional<Price> new_price;
  std::optional<Vol> new_vol;
};

// ref: types.rs:272-285 — Level2Data<N>, N made a run-time value here.
struct Level2Data {
  Price bid_price = 0, ask_price = PRICE_MAX;
  Vol bid_vol = 0, ask_vol = 0;
  std::vector<std::pair<Vol, OrderCount>> bid_price_levels, ask_price_levels;
};

// ---------------------------------------------------------------------------
// RNG (third-party; restated — see header note "PARITY UNPINNED")
// ---------------------------------------------------------------------------
struct Rng {
  uint64_t s0, s1;
  // rand_xoshiro 0.6.0 Xoroshiro128StarStar::seed_from_u64 = SplitMix64 fill
  // (call sites: ref crates/step_sim/src/runner.rs:53, rust/src/step_sim.rs:73)
  static Rng seed_from_u64(uint64_t seed);
  uint64_t next_u64();
  uint32_t next_u32();                           // low 32 bits of next_u64
  float gen_f32();                               // rand Standard: (u32 >> 8) * 2^-24
  uint32_t gen_range_u32(uint32_t lo, uint32_t hi);  // UniformInt::sample_single, lo < hi
  uint32_t gen_index(uint32_t ubound);           // seq::gen_index for len <= u32::MAX
};

// ---------------------------------------------------------------------------
// One side of the book — ref: crates/order_book/src/side.rs:36-143
// ---------------------------------------------------------------------------
struct OrderBookSide {
  Vol vol = 0;
  std::map<Price, std::pair<Vol, OrderCount>> volumes;     // BTreeMap<Price,(Vol,OrderCount)>
  std::map<std::pair<Price, Nanos>, OrderId> orders;       // BTreeMap<(Price,Nanos),OrderId>

  void insert_order(const OrderKey& key, OrderId idx, Vol v);
  void remove_order(const OrderKey& key, Vol v);
  void remove_vol(Price price_key, Vol v);
  Price best_price() const;
  std::pair<Vol, OrderCount> best_vol_and_orders() const;
  Vol best_vol() const;
  std::optional<OrderId> best_order_idx() const;
  std::pair<Vol, OrderCount> vol_and_orders_at_price(Price price_key) const;
};

// ref: side.rs:148-222 (BidSide) / :152,224-291 (AskSide)
struct BidSide {
  OrderBookSide s;
  Price best_price() const { return PRICE_MAX - s.best_price(); }
  std::pair<Vol, OrderCount> vol_and_orders_at_price(Price p) const {
    return s.vol_and_orders_at_price(PRICE_MAX - p);
  }
};
struct AskSide {
  OrderBookSide s;
  Price best_price() const { return s.best_price(); }
  std::pair<Vol, OrderCount> vol_and_orders_at_price(Price p) const {
    return s.vol_and_orders_at_price(p);
  }
};
// ref: side.rs:300-313
inline OrderKey get_bid_key(Nanos t, Price price) { return {Side::Bid, PRICE_MAX - price, t}; }
inline OrderKey get_ask_key(Nanos t, Price price) { return {Side::Ask, price, t}; }

// ---------------------------------------------------------------------------
// OrderBook — ref: crates/order_book/src/orderbook.rs
// ---------------------------------------------------------------------------
struct OrderEntry {  // ref: orderbook.rs:34-39
  Order order;
  OrderKey key;
};

enum OracleStatus : int {
  ORC_OK = 0,
  ORC_PRICE_NOT_TICK_MULTIPLE = 1,  // OrderError::PriceError, orderbook.rs:127-142
  ORC_UNKNOWN_ORDER_ID = 2,         // panic at orderbook.rs:642 / index panic :338,:749
};

struct OrderBook {
  Nanos t;
  Price tick_size;
  Vol trade_vol = 0;
  AskSide ask_side;
  BidSide bid_side;
  std::vector<OrderEntry> orders;
  std::vector<Trade> trades;
  bool trading;
  int levels;  // LEVELS const generic made run-time

  OrderBook(Nanos start_time, Price tick, bool trading_, int levels_);

  Vol ask_vol() const { return ask_side.s.vol; }
  Vol bid_vol() const { return bid_side.s.vol; }
  std::pair<Price, Price> bid_ask() const { return {bid_side.best_price(), ask_side.best_price()}; }
  std::vector<std::pair<Vol, OrderCount>> ask_levels() const;
  std::vector<std::pair<Vol, OrderCount>> bid_levels() const;
  double mid_price() const;
  Level2Data level_2_data() const;

  int create_order(Side side, Vol vol, TraderId trader, std::optional<Price> price, OrderId* out_id);
  void place_order(OrderId id);
  int cancel_order(OrderId id);
  int modify_order(OrderId id, std::optional<Price> new_price, std::optional<Vol> new_vol);
  int process_event(const Event& e);

 private:
  void match_bid(OrderEntry& e);
  void match_ask(OrderEntry& e);
  void place_bid_limit(OrderEntry& e);
  void place_bid_market(OrderEntry& e);
  void place_ask_limit(OrderEntry& e);
  void place_ask_market(OrderEntry& e);
  void reduce_order_vol(OrderEntry& e, Vol reduce_vol);
  void replace_order(OrderEntry& e, Price new_price, Vol new_vol);
};

// ---------------------------------------------------------------------------
// Level2DataRecords — ref: crates/step_sim/src/data.rs:9-56
// ---------------------------------------------------------------------------
struct Level2DataRecords {
  int n;
  std::vector<Price> bid_prices, ask_prices;
  std::vector<Vol> bid_vols, ask_vols;
  std::vector<std::vector<Vol>> bid_vols_at_levels, ask_vols_at_levels;
  std::vector<std::vector<OrderCount>> bid_orders_at_levels, ask_orders_at_levels;
  explicit Level2DataRecords(int n_);
  void append_record(const Level2Data& r);
};

// ---------------------------------------------------------------------------
// Env — ref: crates/step_sim/src/env.rs:58-295
// ---------------------------------------------------------------------------
struct Env {
  Nanos step_size;
  OrderBook order_book;
  std::vector<Vol> trade_vols;
  std::vector<Event> transactions;
  Level2Data level_2_data;
  Level2DataRecords level_2_data_records;

  Env(Nanos start_time, Price tick_size, Nanos step_size_, bool trading, int levels);
  int step(Rng& rng);
  int place_order(Side side, Vol vol, TraderId trader, std::optional<Price> price, OrderId* out_id);
  void cancel_order(OrderId id);
  void modify_order(OrderId id, std::optional<Price> new_price, std::optional<Vol> new_vol);
};

// ---------------------------------------------------------------------------
// RandomAgents — ref: crates/step_sim/src/agents/random_agent.rs:48-120
// ---------------------------------------------------------------------------
struct RandomAgents {
  std::vector<std::optional<OrderId>> orders;
  Price tick_lo, tick_hi;
  Vol vol_lo, vol_hi;
  Price tick_size;
  float activity_rate;
  RandomAgents(size_t n, Price tlo, Price thi, Vol vlo, Vol vhi, Price tick, float rate);
  void update(Env& env, Rng& rng);
};

// ---------------------------------------------------------------------------
// Market — ref: crates/order_book/src/market.rs:59-356.  ASSETS order books sharing one clock; an order is
// addressed by MarketOrderId = (asset, per-book order id) (types.rs:20-22).
// ---------------------------------------------------------------------------
struct MarketEvent {  // Event<MarketOrderId>
  Event::Kind kind;
  uint32_t asset;
  OrderId order_id;
  std::optional<Price> new_price;
  std::optional<Vol> new_vol;
};

struct Market {
  std::vector<OrderBook> order_books;
  Market(Nanos start_time, const std::vector<Price>& tick_sizes, bool trading, int levels);  // market.rs:74-81
  Nanos get_time() const { return order_books[0].t; }                                        // :103-105
  void set_time(Nanos t);                                                                     // :113-117
  void set_trading(bool on);                                                                  // :120-134
  void reset_trade_vols();                                                                    // :142-146
  int process_event(const MarketEvent& e);                                                    // :343-354
};

// ---------------------------------------------------------------------------
// MarketEnv — ref: crates/step_sim/src/market_env.rs:46-340
// ---------------------------------------------------------------------------
struct MarketEnv {
  Nanos step_size;
  Market market;
  std::vector<std::vector<Vol>> trade_vols;  // per asset
  std::vector<MarketEvent> transactions;     // ONE queue for all assets
  std::vector<Level2Data> level_2_data;
  std::vector<Level2DataRecords> level_2_data_records;

  MarketEnv(Nanos start_time, const std::vector<Price>& tick_sizes, Nanos step_size_, bool trading, int levels);
  int step(Rng& rng);  // market_env.rs:110-132
  int place_order(uint32_t asset, Side side, Vol vol, TraderId trader, std::optional<Price> price, OrderId* out_id);
  void cancel_order(uint32_t asset, OrderId id);
  void modify_order(uint32_t asset, OrderId id, std::optional<Price> new_price, std::optional<Vol> new_vol);
};

// ---------------------------------------------------------------------------
// RandomMarketAgents — ref: crates/step_sim/src/agents/random_agent.rs:164-248
// ---------------------------------------------------------------------------
struct RandomMarketAgents {
  uint32_t asset;
  std::vector<std::optional<OrderId>> orders;
  Price tick_lo, tick_hi;
  Vol vol_lo, vol_hi;
  Price tick_size;
  float activity_rate;
  RandomMarketAgents(uint32_t asset_, size_t n, Price tlo, Price thi, Vol vlo, Vol vhi, Price tick, float rate);
  void update(MarketEnv& env, Rng& rng);
};

}  // namespace orc
