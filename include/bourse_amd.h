/* bourse_amd.h — C ABI of the MI355X many-book limit-order-book step simulator.
 *
 * Drop-in boundary for the reference's `bourse_de::Env` hot path (plain pointers and
 * sizes, no torch / C++ types).  One `bk_env` owns B >= 1 INDEPENDENT books on one GPU;
 * book b behaves exactly like one reference `Env` driven by its own
 * `Xoroshiro128StarStar::seed_from_u64(seed + book_offset + b)`.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository).  INTEGRATION.md shows the Rust `extern "C"` / ctypes bindings.
 *
 * Threads: a `bk_env` is used by ONE host thread at a time (like `&mut Env`); DISTINCT envs may be driven from distinct
 * threads concurrently (every call selects the env's device; bk_last_error() is per thread; the process-wide part streams
 * are created under a lock) - tests/test_gpu_parity.py test_envs_driven_from_concurrent_host_threads.
 *
 * Status codes: every function returns one of these.
 */
#ifndef BOURSE_AMD_H
#define BOURSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum bk_status {
  BK_OK = 0,
  BK_PRICE_NOT_TICK_MULTIPLE = 1, /* OrderError::PriceError, crates/order_book/src/orderbook.rs:127-142 */
  BK_UNKNOWN_ORDER_ID = 2,        /* panic! at orderbook.rs:642 / index panic :338,:749 */
  BK_CAPACITY = 3,                /* a fixed device capacity was exceeded (cannot occur in the reference) */
  BK_STEP_SIZE_EXCEEDED = 4,      /* #events in a step >= step_size: (price,t) key collision hazard, SURVEY App. A.9 */
  BK_INVALID_ARGUMENT = 5,
  BK_HIP_ERROR = 6,               /* a HIP runtime call failed; bk_last_error() has the text */
  BK_NO_DEVICE = 7                /* no usable GPU: the product path never falls back to a CPU */
};

/* per-book sticky flag bits reported by bk_book_flags() */
#define BK_FLAG_POOL_OVERFLOW 1u   /* live-order pool full: an order that should rest was dropped */
#define BK_FLAG_TRADE_OVERFLOW 2u  /* trade buffer full: records dropped (counts stay exact) */
#define BK_FLAG_STEP_SIZE 4u       /* a step queued >= step_size events */
#define BK_FLAG_ORDER_LOG_FULL 8u  /* order id beyond the order-log capacity */
#define BK_FLAG_UNKNOWN_ORDER 16u  /* cancel/modify of an id that was never created */
#define BK_FLAG_HIST_OVERFLOW 32u  /* RESERVED, never set (the L2 history is a ring; reading an overwritten step is an
                                     * error of the reader): the bit keeps its place so that the others keep theirs */
#define BK_FLAG_PRICE_TICK 64u     /* a Noise/Momentum agent's limit price (clamped to u32::MAX) was not a tick multiple:
                                     * the reference panics here (`.unwrap()`, common.rs:107,140); the order is not created */
#define BK_FLAG_DECODE_LOOKAHEAD 256u /* the wave-parallel decode of a Noise/Momentum member met a ziggurat rejection loop
                                     * longer than its 128-draw look-ahead (p < 2^-90): results of that book are suspect */
#define BK_FLAG_EVENT_OVERFLOW 128u /* a MARKET with Noise/Momentum members queued more than max_live_orders events in one
                                     * step (its books share one queue of that size): the excess events were dropped */

typedef struct bk_env bk_env;

/* Env::<LEVELS>::new(start_time, tick_size, step_size, trading) — crates/step_sim/src/env.rs:84-95,
 * plus the sizes a fixed-capacity device implementation needs.  Zero-initialise, then set. */
typedef struct bk_config {
  uint32_t n_books;         /* B >= 1 independent books on this GPU */
  uint32_t levels;          /* LEVELS: L2 depth per side (reference default 10; 16/32/64 in the benchmark configs) */
  uint64_t start_time;
  uint32_t tick_size;       /* > 0 (assert at orderbook.rs:159) */
  uint32_t trading;         /* bool */
  uint64_t step_size;
  uint64_t seed;            /* book b is seeded seed + book_offset + b (identity for B = 1) */
  uint64_t book_offset;     /* global index of this GPU's first book (multi-GPU sharding) */
  uint32_t max_live_orders; /* live-order pool per book, rounded up to 64, 128, 256 or 512 (0 = 128; more is refused) */
  uint32_t max_orders;      /* order-log capacity per book for the host-driven path (0 = no log) */
  uint32_t trade_capacity;  /* trade records retained per book between bk_clear_trades() calls */
  uint32_t history_capacity;/* L2 history ring: the last N steps are retained (0 = keep only the latest record) */
  int32_t device;           /* HIP device ordinal */
  uint32_t assets;          /* MarketEnv<ASSETS> mode (crates/step_sim/src/market_env.rs:46-132): `assets` consecutive books
                             * form a market sharing one clock, ONE RNG stream and ONE shuffled event queue; book =
                             * market * assets + asset; market m is seeded seed + book_offset + m (book_offset counted in
                             * markets).  0 or 1 = independent books (Env).  <= 8, must divide n_books. */
} bk_config;

/* RandomAgents::new(n_agents, tick_range, vol_range, tick_size, activity_rate)
 * — crates/step_sim/src/agents/random_agent.rs:67-81 */
typedef struct bk_random_agents {
  uint32_t n_agents;
  uint32_t tick_lo, tick_hi; /* tick_range (half-open) */
  uint32_t vol_lo, vol_hi;   /* vol_range (half-open) */
  uint32_t tick_size;        /* the AGENTS' tick size: price = tick * tick_size */
  float activity_rate;
} bk_random_agents;

/* One member of a derive(AgentSet) struct (crates/macros/src/lib.rs:57-73): RandomAgents, NoiseAgent
 * (NoiseAgent::new(agent_id_start, n_agents, NoiseAgentParams), crates/step_sim/src/agents/noise_agent.rs:24-123) or
 * MomentumAgent (MomentumAgent::new(agent_id_start, n_agents, MomentumParams), momentum_agent.rs:24-142). */
enum bk_agent_type { BK_AGENT_RANDOM = 0, BK_AGENT_NOISE = 1, BK_AGENT_MOMENTUM = 2 };
typedef struct bk_agent_desc {
  uint32_t type;             /* bk_agent_type */
  uint32_t n_agents;
  uint32_t tick_lo, tick_hi; /* RandomAgents tick_range */
  uint32_t vol_lo, vol_hi;   /* RandomAgents vol_range */
  uint32_t tick_size;        /* the member's tick_size parameter */
  float activity_rate;       /* RandomAgents */
  uint32_t agent_id_start;   /* Noise/Momentum: first trader id */
  float p_limit, p_market;   /* NoiseAgentParams */
  float p_cancel;            /* Noise/Momentum */
  uint32_t trade_vol;        /* Noise/Momentum */
  uint32_t reserved;
  double price_dist_mu, price_dist_sigma;       /* Noise/Momentum: LogNormal(mu, sigma) */
  double decay, demand, scale, order_ratio;     /* MomentumParams */
} bk_agent_desc;

/* Trade — crates/order_book/src/types.rs:103-118; tuple order of PyTrade, rust/src/types.rs:4-15 */
typedef struct bk_trade {
  uint64_t t;
  uint32_t side_is_bid; /* side of the PASSIVE order */
  uint32_t price;
  uint32_t vol;
  uint32_t reserved;
  uint64_t active_order_id;
  uint64_t passive_order_id;
} bk_trade;

/* Order — crates/order_book/src/types.rs:79-99; tuple order of PyOrder, rust/src/types.rs:17-31 */
typedef struct bk_order {
  uint8_t side_is_bid;
  uint8_t status; /* 0 New, 1 Active, 2 Filled, 3 Cancelled, 4 Rejected (types.rs:65-75) */
  uint8_t reserved[6];
  uint64_t arr_time;
  uint64_t end_time;
  uint32_t vol;
  uint32_t start_vol;
  uint32_t price;
  uint32_t trader_id;
  uint64_t order_id;
} bk_order;

/* whole-shard market statistics, modelled on Market's array-valued queries
 * (crates/order_book/src/market.rs:137-216); the unit all-gathered across GPUs (64 bytes) */
typedef struct bk_stats {
  uint64_t n_books;
  uint64_t sum_trade_vol;  /* sum over books of the last step's trade volume */
  uint64_t sum_trades;     /* cumulative trade count over books */
  uint64_t sum_events;     /* cumulative processed events over books */
  uint64_t sum_bid_vol, sum_ask_vol;
  uint32_t min_bid, max_bid, min_ask, max_ask; /* over books with a non-empty side; 0xFFFFFFFF/0 if none */
} bk_stats;

const char* bk_last_error(void);
int bk_device_count(int* out);
/* HIP_VERSION the library was compiled against / hipRuntimeGetVersion() of the runtime this process bound it to (no
 * reference counterpart: diagnostics - a Python process may run the library on PyTorch's bundled runtime, bourse_amd/_lib.py) */
int bk_hip_versions(int* built, int* runtime);

/* ------------------------------------------------------------------ lifetime */
int bk_env_create(const bk_config* cfg, bk_env** out);   /* Env::new, env.rs:84-95 (x B) */
void bk_env_destroy(bk_env* env);
int bk_env_set_stream(bk_env* env, void* hip_stream);    /* run on the caller's hipStream_t (NULL = default) */
int bk_env_sync(bk_env* env);                            /* wait for queued device work */

/* ------------------------------------------- host-driven order flow (per book) */
/* Env::place_order, env.rs:166-176: tick check + id assignment now, New event queued for the next step */
int bk_place_order(bk_env* env, uint32_t book, int bid, uint32_t vol, uint32_t trader_id, int has_price,
                   uint32_t price, uint64_t* out_order_id);
/* Env::cancel_order, env.rs:189-191 */
int bk_cancel_order(bk_env* env, uint32_t book, uint64_t order_id);
/* Env::modify_order, env.rs:208-219 */
int bk_modify_order(bk_env* env, uint32_t book, uint64_t order_id, int has_price, uint32_t new_price, int has_vol,
                    uint32_t new_vol);
/* StepEnvNumpy.submit_instructions, rust/src/step_sim_numpy.rs:233-275: action 0 none / 1 new limit / 2 cancel; every
 * other value is a no-op, as the reference's `_ => Ok(OrderId::MAX)` (:266) - on the host entries AND on the device
 * entry below.  The one extension, the same on all three entries, is BK_ACTION_MODIFY (a code far outside the
 * reference's range, so a reference-written batch can never mean it): Env::modify_order (env.rs:208-219) on
 * order_id[i], side bit 1 = has a new price (price[i]), side bit 2 = has a new volume (vol[i]).
 * out_ids[i] = new id or UINT64_MAX.  Stops at the first bad price (earlier elements stay queued) and
 * returns BK_PRICE_NOT_TICK_MULTIPLE with *n_done = index of the offending element. */
#define BK_ACTION_MODIFY 0x80000003u
int bk_submit_instructions(bk_env* env, uint32_t book, size_t n, const uint32_t* action, const uint8_t* side,
                           const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                           const uint64_t* order_id, uint64_t* out_ids, size_t* n_done);
/* The same for every book in one call: book b's instructions are elements [book_offsets[b], book_offsets[b+1]) of the
 * arrays (CSR over n_books + 1 offsets).  *n_done = global index reached.  Large batches are spread over host threads
 * (one contiguous range of books each; BOURSE_AMD_HOST_THREADS overrides the count); on an error, books after the failing
 * one may already be queued. */
int bk_submit_instructions_csr(bk_env* env, const uint64_t* book_offsets, const uint32_t* action, const uint8_t* side,
                               const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                               const uint64_t* order_id, uint64_t* out_ids, size_t* n_done);
int bk_enable_trading(bk_env* env, int enabled);         /* Env::{enable,disable}_trading, env.rs:138-145 */
/* Env::step, env.rs:116-135, for every book: shuffle + process the queued events, snapshot L2 */
int bk_step(bk_env* env);

/* ------------------------------------------------------------- device-resident instruction ingress
 * For agent layers that already run on the GPU: `submit_instructions` (rust/src/step_sim_numpy.rs:233-275) and the
 * host half of Env::place_order / cancel_order / modify_order (crates/step_sim/src/env.rs:166-219) for EVERY book in one
 * kernel launch, the six SoA arrays taken as DEVICE pointers - nothing of a step passes through host memory.
 * bk_device_ingress_enable: switches a fresh env to this flow (before any order; an env runs one flow: the per-order
 *   host entries and bk_run are refused from then on) and allocates the per-book (per-market) event queues of
 *   `queue_capacity` events per step (1..8192).
 * bk_submit_instructions_device: book b's instructions are elements [book_offsets_dev[b], book_offsets_dev[b+1]) of the
 *   arrays (all in device memory, u64 offsets over n_books + 1).  action 0 = none, 1 = new limit order, 2 = cancel,
 *   anything else a no-op (as the reference's, :266), BK_ACTION_MODIFY = Env::modify_order (the extension of
 *   bk_submit_instructions: side bit 1 = has price, bit 2 = has volume; bit 0 is the bid flag of a new order) - the same
 *   arrays mean the same on the host and on the device entry.  Per book exactly the reference's semantics: ids are dense in element order (an exclusive prefix
 *   sum over action == 1 on the device), the first price that is not a multiple of the book's tick size stops THAT
 *   book's batch - earlier elements stay created and queued (:255-268) - and the other books are unaffected.
 *   out_ids_dev (optional, one u64 per element): the created order's id, u64::MAX otherwise; untouched from the failing
 *   element on.  status_dev (optional, 2 u32 per book): {bk_status code - BK_OK, BK_PRICE_NOT_TICK_MULTIPLE, BK_CAPACITY
 *   (queue full / id space) -, number of the book's elements applied}.  Asynchronous on the env's stream.
 *   A cancel / modify of an id that was never created (the reference panics while processing, orderbook.rs:642; the
 *   host-driven bk_step refuses the step) is dropped at the step and the book flagged BK_FLAG_UNKNOWN_ORDER.
 *   All seven input arrays must be non-null (a step with no instruction for any book: do not call).
 * bk_step_async: Env::step over the device-resident queues without waiting (bk_step on such an env = this + a wait).
 * Readers (bk_get_orders, bk_order_status, bk_get_trades, bk_history ...) work as on the host-driven flow. */
int bk_device_ingress_enable(bk_env* env, uint32_t queue_capacity);
int bk_submit_instructions_device(bk_env* env, const uint64_t* book_offsets_dev, const uint32_t* action_dev,
                                  const uint8_t* side_dev, const uint32_t* vol_dev, const uint32_t* trader_dev,
                                  const uint32_t* price_dev, const uint64_t* order_id_dev, uint64_t* out_ids_dev,
                                  uint32_t* status_dev);
int bk_step_async(bk_env* env);
/* HOST arrays through the device ingress (a BaseNumpyAgent-style caller: src/bourse/step_sim/agents/base_agent.py:67-116
 * returns host numpy arrays, runner.py:103-112 passes them to submit_instructions, rust/src/step_sim_numpy.rs:233-275).
 * Same arrays, same per-book semantics as bk_submit_instructions_device, but the pointers are HOST memory: the library
 * stages them in pinned memory (host threads), uploads them on a copy stream of its own, runs the ingest kernel on the
 * env's stream and brings ids and per-book status back on a second copy stream.  Asynchronous: returns a ticket at
 * once; two tickets are in flight at most (the upload of one under the step kernel of the other), and
 * bk_submit_result(ticket) waits for that ticket only.  A ticket's results stay readable until two more submits (also
 * across a batch that outgrows the staging: the results move with it).
 *   bk_ingress_staging: OPTIONAL zero-copy - pinned arrays (>= min_elements each; book_offsets n_books + 1) that the NEXT
 *     bk_submit_instructions_host call will upload from: fill them in place and pass these very pointers (any array
 *     passed from elsewhere is copied as usual).  Valid until that submit returns.
 *   bk_submit_result: out_ids (optional, book_offsets[n_books] u64: the created order's id, UINT64_MAX otherwise -
 *     also from a book's failing element on), status (optional, 2 u32 per book: {bk_status code,
 *     elements of the book's batch applied}), first_failed_book (optional: lowest book whose code is not BK_OK, UINT32_MAX
 *     if none - the Python layer raises the reference's ValueError for it). */
typedef struct bk_ingress_arrays {
  uint64_t capacity;       /* elements each instruction array holds */
  uint64_t* book_offsets;  /* n_books + 1 */
  uint32_t* action;
  uint8_t* side;
  uint32_t* vol;
  uint32_t* trader_id;
  uint32_t* price;
  uint64_t* order_id;
} bk_ingress_arrays;
int bk_ingress_staging(bk_env* env, uint64_t min_elements, bk_ingress_arrays* out);
int bk_submit_instructions_host(bk_env* env, const uint64_t* book_offsets, const uint32_t* action, const uint8_t* side,
                                const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                                const uint64_t* order_id, uint64_t* out_ticket);
int bk_submit_result(bk_env* env, uint64_t ticket, uint64_t* out_ids, uint32_t* status, uint32_t* first_failed_book);
/* The same without a copy: pointers INTO the library's pinned staging (ids: book_offsets[n_books] u64; status: 2 u32 per book),
 * valid until two more submits.  For a loop that only looks at the ids (3 MB per step at 8 192 books x 48 instructions). */
int bk_submit_result_view(bk_env* env, uint64_t ticket, const uint64_t** out_ids, const uint32_t** status,
                          uint32_t* first_failed_book);
/* Env::order_status / Env::order, env.rs:283-290 (needs max_orders > 0) */
int bk_order_status(bk_env* env, uint32_t book, uint64_t order_id, uint8_t* out_status);
int bk_order_count(bk_env* env, uint32_t book, uint64_t* out);
int bk_get_orders(bk_env* env, uint32_t book, uint64_t first, uint64_t n, bk_order* out); /* Env::get_orders */
/* OrderEntry.key (crates/order_book/src/orderbook.rs:34-39) of orders [first, first+n): the price and the time the
 * order's priority key (side, price_key, t) was last set with — what OrderBook::save_json serialises beside each order. */
int bk_get_order_keys(bk_env* env, uint32_t book, uint64_t first, uint64_t n, uint32_t* key_price, uint64_t* key_time);
/* OrderBook::load_json -> TryFrom<OrderBookState> (orderbook.rs:827-918): replace one book's state with a snapshot
 * (clock, trade volume, all orders listed by id with their keys, all trades); Active orders re-enter the book in key
 * order and the level-2 record is rebuilt.  Nothing may be queued for the book (its market). */
int bk_load_book(bk_env* env, uint32_t book, uint64_t t, uint32_t trade_vol, uint64_t n_orders, const bk_order* orders,
                 const uint32_t* key_price, const uint64_t* key_time, uint64_t n_trades, const bk_trade* trades);

/* ---------------------------------------------------- on-device order flow */
/* An AgentSet of RandomAgents groups, identical for every book, updated in declaration order
 * (crates/macros/src/lib.rs:57-73).  Sum of n_agents <= max_live_orders. */
int bk_set_random_agents(bk_env* env, uint32_t n_groups, const bk_random_agents* groups);
/* Market mode.  Market::new(start_time, tick_size: [Price; ASSETS], trading) — crates/order_book/src/market.rs:74-81:
 * per-asset tick sizes (default: cfg.tick_size for every asset); call before anything else. */
int bk_set_tick_sizes(bk_env* env, uint32_t n_assets, const uint32_t* tick_sizes);
/* A MarketAgentSet of RandomMarketAgents groups (RandomMarketAgents::new(asset, n_agents, tick_range, vol_range,
 * tick_size, activity_rate), random_agent.rs:185-201), identical for every market, updated in declaration order with
 * the market's RNG (market_sim_runner, runner.rs:108-131).  assets[g] = the asset group g trades (NULL: all 0).
 * Sum of n_agents <= max_live_orders.  bk_run() then steps every market; host-driven orders use the per-book calls with
 * book = market * assets + asset (MarketEnv::place_order / cancel_order / modify_order, market_env.rs:163-218). */
int bk_set_random_market_agents(bk_env* env, uint32_t n_groups, const bk_random_agents* groups,
                                const uint32_t* assets);
/* Any mix of built-in members (at most 4 when a Noise/Momentum member is present), identical for every book.
 * Noise/Momentum members price orders with f64 log-normal offsets: their outputs match the CPU oracle bit for bit
 * but only statistically match a Rust build (third-party sampling + libm). */
int bk_set_agents(bk_env* env, uint32_t n_members, const bk_agent_desc* members);
/* Market mode: a MarketAgentSet of RandomMarketAgents / NoiseMarketAgent / MomentumMarketAgent members
 * (random_agent.rs:164-247, noise_agent.rs:226-340, momentum_agent.rs:282-397); member i trades asset assets[i] of every
 * market and draws from the market's RNG in declaration order. */
int bk_set_market_agents(bk_env* env, uint32_t n_members, const bk_agent_desc* members, const uint32_t* assets);
/* sim_runner's loop body n_steps times for every book: agents.update(env, rng); env.step(rng)
 * (crates/step_sim/src/runner.rs:53-68), sharing each book's RNG between agents and shuffle. */
int bk_run(bk_env* env, uint64_t n_steps);
/* Warm-up without side effects: n_steps of this env's own kernels on its own books (same agents, pipeline and streams),
 * then state blocks, level-2 records and the step counter are put back; the scratch steps write no history slot and no
 * trade record.  For a short bk_run after host-side work: the GPU's clocks fall within milliseconds of idling and take
 * ~15 ms of load to come back, and a pipeline's first launch pays one-off set-up.  Asynchronous like bk_run.  (No
 * counterpart in the reference: a CPU has no launch set-up to hide.) */
int bk_warm(bk_env* env, uint64_t n_steps);
/* One env runs ONE of the two order flows: once bk_run has stepped it with on-device agents, bk_place_order /
 * bk_cancel_order / bk_modify_order / bk_submit_instructions* / bk_step return BK_INVALID_ARGUMENT (host order ids would
 * restart at 0 and collide with the agents'); and bk_run refuses an env that holds host-placed orders. */

/* ------------------------------------------------------------------ readers */
/* Level-2 record width in u32: 5 + 4*levels, laid out as StepEnvNumpy.level_2_data
 * (rust/src/step_sim_numpy.rs:351-368): [trade_vol, bid_price, ask_price, ask_vol, bid_vol,
 * {bid_vol_i, bid_n_i, ask_vol_i, ask_n_i} for i < levels]. */
uint32_t bk_l2_width(const bk_env* env);
/* Env::level_2_data (end-of-step snapshot) for books [first, first+n): out[n][width] */
int bk_level2(bk_env* env, uint32_t first_book, uint32_t n_books, uint32_t* out);
/* Level2DataRecords + trade_vols (data.rs:9-57, env.rs:64,134): out[n_steps][n_books][width] for retained steps */
int bk_history_len(bk_env* env, uint64_t* first_step, uint64_t* n_steps);
int bk_history(bk_env* env, uint64_t first_step, uint64_t n_steps, uint32_t first_book, uint32_t n_books,
               uint32_t* out);
int bk_clear_history(bk_env* env);
/* Streaming egress: the history buffer is a ring of history_capacity steps; this queues the
 * device-to-host copy of retained steps on `copy_stream`, ordered after the work queued on the env so far, and returns
 * at once, so the next bk_run() overlaps the copy.  Keep history_capacity >= 2 x the chunk being copied; `out` should
 * come from bk_pinned_alloc().  Wait with bk_stream_sync(copy_stream). */
int bk_history_copy_async(bk_env* env, uint64_t first_step, uint64_t n_steps, uint32_t first_book, uint32_t n_books,
                          uint32_t* out, void* copy_stream);
int bk_stream_create(void** out);
int bk_stream_sync(void* stream);
int bk_stream_destroy(void* stream);
int bk_pinned_alloc(uint64_t nbytes, void** out);
int bk_pinned_free(void* p);
/* OrderBook::get_trades, orderbook.rs:800-802 */
int bk_trade_count(bk_env* env, uint32_t book, uint64_t* total, uint64_t* first_retained);
int bk_trade_counts(bk_env* env, uint64_t* totals /* [n_books] */);
int bk_get_trades(bk_env* env, uint32_t book, uint64_t first, uint64_t n, bk_trade* out);
int bk_clear_trades(bk_env* env);
/* Trade egress at scale (Env::get_trades, env.rs:277-280, for every book at once): bk_trades_compact gathers all
 * retained records into one dense device stream in the bk_trade layout with CSR offsets (book b owns
 * [offsets[b], offsets[b+1])) and marks them consumed (like bk_clear_trades); *out_total = number of records.
 * bk_trades_compact_copy_async copies that stream (records: out_total x bk_trade, offsets: n_books + 1) to the host on
 * `copy_stream`, ordered after the compaction (NULL: the env's stream, synchronous), so it overlaps the next bk_run.
 * A book whose records overflowed trade_capacity keeps its BK_FLAG_TRADE_OVERFLOW flag: nothing is dropped silently. */
int bk_trades_compact(bk_env* env, uint64_t* out_total);
int bk_trades_compact_copy_async(bk_env* env, bk_trade* records, uint64_t* offsets, void* copy_stream);
int bk_time(bk_env* env, uint32_t book, uint64_t* out);          /* OrderBook::get_time */
int bk_set_time(bk_env* env, uint32_t book, uint64_t t);         /* OrderBook::set_time, orderbook.rs:183-185 */
int bk_trade_vol(bk_env* env, uint32_t book, uint32_t* out);     /* OrderBook::get_trade_vol (live) */
int bk_steps_done(bk_env* env, uint64_t* out);
int bk_book_flags(bk_env* env, uint32_t* out /* [n_books] */);   /* sticky BK_FLAG_* bits */
/* clear the bits of `mask` in every book's sticky flags (the caller has seen and handled them) */
int bk_clear_flags(bk_env* env, uint32_t mask);
/* what a strict caller polls after a step: OR of every book's flags, and the largest number of trade records any book
 * retains (towards trade_capacity) - two words instead of n_books */
int bk_flags_summary(bk_env* env, uint32_t* flags_or, uint64_t* max_retained_trades);
int bk_rng_state(bk_env* env, uint32_t book, uint64_t out_state[2]);
/* resting (Active) orders of one book in price-time priority per side: bids first, then asks */
int bk_live_orders(bk_env* env, uint32_t book, uint32_t cap, bk_order* out, uint32_t* n_out);

/* ------------------------------------------------------ stats / multi-GPU */
/* reduce this shard's books into one 64-byte record on the device.  out_host != NULL: waits and copies the record to the
 * host.  out_host == NULL: fully asynchronous - the reduction is queued on the env's stream behind the stepping kernels
 * and never synchronises the host (the record is then read on the device through bk_stats_device_ptr, e.g. by an RCCL
 * all-gather on the same stream). */
int bk_stats_compute(bk_env* env, bk_stats* out_host);
/* device address of the 64-byte record (for an RCCL all-gather issued by the caller) */
int bk_stats_device_ptr(bk_env* env, void** out);
/* Device pointer of the latest level-2 records, u32[n_books][bk_l2_width()] (Env::level_2_data of every book), for
 * on-device consumers: the optional per-book L1 all-gather of SURVEY §8e (ii) reads its 9 leading words per book. */
int bk_level2_device_ptr(bk_env* env, void** out);

/* ------------------------------------------------------------- measurement */
/* accumulate HIP-event timings of the step kernels: on = 0 off, N >= 1 time the kernels of every Nth step */
int bk_profile_enable(bk_env* env, int on);
int bk_profile_read(bk_env* env, double* total_ms, uint64_t* n_launches, int reset);
/* per kernel: kind 0 the fused kernels (k_run_random / k_run_wave / k_run_mixed), 1 the agents kernel of a split pipeline
 * (k_agents_fsm / k_agents_wave / k_agents_mixed_*), 2 k_step_batch, 3 k_step_events (host-driven flow) */
int bk_profile_read_kind(bk_env* env, int kind, double* total_ms, uint64_t* n_launches);
/* bk_run kernel pipeline: 0 auto (by shape and batch size), 1 fused (one wave per book, all phases),
 * 2 split (RNG-serial phases one lane per book + event phase one wave per book), 3 split with the members of an AgentSet
 * decoded one wave per book (older form of 4), 4 / 5 below.  Results are identical; only speed differs. */
int bk_set_pipeline(bk_env* env, int mode);
/* Mode 4 on an AgentSet with Noise / Momentum members (independent books): the members' update one WAVE per book with
 * their stream decoded 64 draws at a time (k_agents_mixed_wave) + the event kernel; auto from 512 books.  A RandomAgents
 * member of such a set is walked on that kernel's scalar path (same results).  Markets (assets > 1): as mode 2.
 * Modes 4 ("wave_split") and 5 ("wave") on RandomAgents books: the RNG-serial phases run one WAVE per
 * book with the book's xoroshiro stream decoded 64 draws at a time (jump-ahead lane states + ballot/prefix resolution) -
 * as a kernel of its own in front of the event kernel (4), or fused with the event phase in one persistent kernel that
 * keeps the book in registers across all steps of a bk_run (5).  bk_set_wave_options: look-ahead of the decode's vector
 * path (1..64 draws, default 64; smaller values push placements onto its scalar slow path - a test knob) and the number
 * of parts mode 4 cuts the batch in (0 = default). */
int bk_set_wave_options(bk_env* env, uint32_t lookahead, int parts);
/* DEPRECATED, always *out = 0; kept only so that clients built against rounds 1-3 still link.  (Those rounds rolled an
 * auto-selected launch of the lane-per-book members' update back when it overflowed a pool the other kernels still fit,
 * and counted the roll-backs here; the auto rule has not picked that pipeline since round 3.) */
int bk_pipeline_fallbacks(bk_env* env, uint64_t* out);
/* the pipeline bk_run will use: *split = 0 fused / 1 split (lane-per-book agents) / 2 wave_split / 3 wave; *n_parts =
 * contiguous book parts launched on separate streams */
int bk_get_pipeline(bk_env* env, int* split, int* n_parts);
uint64_t bk_state_bytes_per_book(const bk_env* env);
/* split-pipeline geometry: the batch is cut in min(n_parts, books / min_part) contiguous parts, each on its own HIP
 * stream (defaults 4 - one per hardware queue - and 4096; n_parts in 1..8, min_part >= 64).  Results never depend on it. */
int bk_set_split_parts(bk_env* env, int n_parts, uint32_t min_part);
int bk_get_split_parts(bk_env* env, int* n_parts, uint32_t* min_part); /* the two settings as stored */
/* orders created so far in every book by the on-device agents: OrderBook::current_order_id / orders.len()
 * (crates/order_book/src/orderbook.rs:327-329), totals[n_books] */
int bk_order_counts(bk_env* env, uint64_t* totals /* [n_books] */);
/* A diagnostic of the host-driven / ingress step (bk_step, bk_step_async: Env::step over submitted instructions,
 * crates/step_sim/src/env.rs:116-135): how many of each book's steps ran on the keyed event loop - steps without
 * modifications whose events fit one per pool slot, on a trading book, with prices and arrival stamps inside the key window
 * (bourse_amd/csrc/step_events.hpp step_events_keyed); the others ran the event-by-event loop.  Results never depend on it. */
int bk_event_steps_keyed(bk_env* env, uint64_t* counts /* [n_books] */);

/* ------------------------------------------------------ checkpoint / resume */
/* Dump / restore the complete simulation state of an on-device-order-flow env (pool, clock, counters, per-book
 * RNG).  The image starts with a versioned header (magic, shape, levels, assets, a hash of the installed agent set):
 * bk_checkpoint_load refuses an image taken from a differently shaped env or with different agents.  The reference has no counterpart: its Env/agents/RNG are not serialisable (only OrderBook JSON snapshots,
 * crates/order_book/src/orderbook.rs:811-832).  A restored env continues bit-identically. */
uint64_t bk_checkpoint_bytes(const bk_env* env);
int bk_checkpoint_save(bk_env* env, void* out, uint64_t nbytes);
int bk_checkpoint_load(bk_env* env, const void* in, uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* BOURSE_AMD_H */
