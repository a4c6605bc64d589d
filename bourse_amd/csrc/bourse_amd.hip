// bourse_amd.hip — C ABI (include/bourse_amd.h) over the gfx950 kernels in book_device.hpp.
//
// Host side of the drop-in boundary: owns the device state of B independent books, keeps the
// host-visible half of `bourse_de::Env` (order creation, id assignment, the per-step event
// queue — crates/step_sim/src/env.rs:166-219) and launches the step kernels.  There is no CPU
// execution path: without a usable GPU every entry point fails with BK_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bourse_amd.h"
#include "book_device.hpp"
// k_agents_fsm is compiled in its own unit (fsm_unit.hip: another machine-scheduler strategy); only declared here
namespace bkd {
extern template __global__ void k_agents_fsm<1>(DevArgs);
extern template __global__ void k_agents_fsm<2>(DevArgs);
extern template __global__ void k_agents_fsm<4>(DevArgs);
extern template __global__ void k_agents_fsm<8>(DevArgs);
}  // namespace bkd
#include "host_math.hpp"
#include "host_pool.hpp"
#include "mixed_agents.hpp"
#include "wave_agents.hpp"
#include "wave_mixed.hpp"
#include "step_events.hpp"

using namespace bkd;

static_assert(sizeof(DevStats) == sizeof(bk_stats), "bk_stats layout");
static_assert(sizeof(OutTrade) == sizeof(bk_trade), "bk_trade layout");
static_assert(sizeof(bk_config) == 72 && sizeof(bk_random_agents) == 28 && sizeof(bk_trade) == 40 &&
                  sizeof(bk_order) == 48,
              "C ABI struct layout (mirrored by bourse_amd/_lib.py)");
static_assert(sizeof(bk_agent_desc) == 104, "bk_agent_desc layout (mirrored by bourse_amd/_lib.py)");
static_assert(sizeof(DevTrade) == 32 && sizeof(DevOrderLog) == 48 && sizeof(uint4) == 16, "device record layout");

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* c = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
  return h;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fail(BK_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e));               \
  } while (0)

// AgentSets of Noise / Momentum members on independent books: from this many books the members' update runs one WAVE per
// book with the stream decoded 64 draws at a time (k_agents_mixed_wave, wave_mixed.hpp) in front of the event kernel
constexpr uint32_t MIXED_WAVE_MIN_BOOKS = 512;
// The auto rule for RandomAgents books, derived from the SHAPE (pool registers R = pool / 64) instead of a book count
// swept at one shape (scripts/shape_sweep.py, profiles/r03/shape_sweep.txt: C2 R = 1, C3 R = 2, a 256-slot pool R = 4,
// C5 R = 8, 1 024 .. 65 536 books):
//   * `wave` (k_run_wave: decode + events fused, book in registers across the launch) while the batch fits the chip in
//     ONE residency round of that kernel - asked of the runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor x 8 books
//     per workgroup x CUs: 6 144 books at R <= 2, 4 096 at R = 4, 2 048 at R = 8); one book more and a second round at a
//     fraction of the occupancy costs more than the split form's launches (C3: 103 M at 6 144, 94 M at 7 168 books).
//     64-slot pools are the exception: their book-step is so short that the persistent kernel wins up to the lane split's
//     take-over (C2: 195 M vs 170 M at 8 192, 216 vs 206 at 16 384);
//   * `wave_split` (k_agents_wave + k_step_batch, three parts) from there;
//   * `split` (lane-per-book k_agents_fsm + k_step_batch, four parts) from lane_split_min_books(R): its 125 us chain
//     per step needs that many books to be hidden.  Crossovers re-measured at the end of round 4, after the decode and
//     both event loops got faster (profiles/r04/shape_sweep_crossovers.txt, twice: the second sweep after the decode's last
//     trims and the event waves' priority rule): 26 k / 26.4 - 27.9 k (two boxes) / 25.3 k / 26.5 k books for R = 1, 2, 4, 8 (round 3: 23 k /
//     24.5 k / 18 k / 24.5 k - the 256-slot pools' wave_split gained most: 53 -> 76 M).
// behind the wave-parallel decode the event waves run at priority 1 from this many books (book_device.hpp k_step_batch).  Re-swept
// at the end of round 4: pools of <= 128 slots gain from 8 192 books now (132.5 -> 135.3 M there, +1 % at 12 288; round 3: -2 %
// at 8 192), the 512-slot pools still lose below 16 384 (C5 stand-in 32.0 -> 31.4 M at 8 192)
constexpr uint32_t wave_step_prio_books(int R) { return R <= 2 ? 8192u : 16384u; }
constexpr uint32_t lane_split_min_books(int R) { return R == 4 ? 25600u : (R == 2 ? 27648u : 26624u); }

struct HostOrder {  // immutable half of an order, fixed at create_order (orderbook.rs:356-396)
  uint8_t bid;
  uint32_t start_vol, price, trader;
  uint64_t create_time;
};
struct HostEvent {
  uint32_t word, id, price, vol;
};
static_assert(sizeof(HostEvent) == 16, "HostEvent is uploaded as one uint4 per event");
struct BookHost {
  std::vector<HostOrder> orders;
  std::vector<HostEvent> queue;
  std::vector<DevOrderLog> log_cache;
  uint64_t n_uploaded = 0;  // orders whose New event has reached the device
  uint64_t time_offset = 0;  // set by bk_set_time: book time = start_time + steps_done * step_size + time_offset (wrapping)
  bool log_fresh = false;
  uint64_t mirror_epoch = ~0ull;  // device-resident ingress: `orders` mirrors the device's records as of this ingest epoch
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
  }
};

}  // namespace

struct bk_env {
  bk_config cfg{};
  int R = 1;
  uint32_t M = 1;  // books per market (MarketEnv<ASSETS>): cfg.assets, 1 = independent books
  uint32_t asset_tick[MAX_ASSETS] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t W = 0, stride = 0;
  hipStream_t stream = nullptr;
  DevBuf<uint32_t> state, l2_last, hist, ev_off, batch;
  DevBuf<uint4> ev;               // this step's events, 16 B each (HostEvent layout), CSR by book / market
  HostEvent* ev_stage = nullptr;  // pinned staging of the same (uploaded at link speed)
  uint32_t* off_stage = nullptr;
  uint32_t batch_stride = 0;
  // 0 auto, 1 fused (k_run_random), 2 split (k_agents_fsm + k_step_batch), 3 split with wave-per-book AgentSet members,
  // 4 wave_split (k_agents_wave + k_step_batch), 5 wave (k_run_wave: wave-parallel decode + events, persistent)
  int pipeline = 0;
  DevBuf<uint32_t> warm_snap;         // bk_warm: state + L2 copy of the scratch steps
  bool warming = false;               // bk_warm's scratch steps: no history slots, no trade records
  DevBuf<uint4> jump_tabs;      // k_agents_wave: T^256 (block jump) then T^(4 << b), b = 0..5 (lane offsets): 7 x 8 KB
  DevBuf<uint32_t> wcache;      // k_agents_wave: per-book lane states of the RNG block in progress
  uint32_t wave_lookahead = 64;
  uint32_t stagger_us = ~0u;    // parts of a split launch start i x stagger_us apart; ~0 = default rule, 0 = by events
  int wave_parts = 0;           // 0 = as the lane split (n_parts / min_part)
  bool wave_ok() const { return !n_mixed && M == 1 && !groups.empty(); }  // RandomAgents on independent books
  // AgentSets with Noise / Momentum members on independent books: wave-parallel decode of the members' update
  bool wl_valid = false;    // the wave-per-book members' lists (wl_list) describe the pools as of steps_done
  DevBuf<uint16_t> wl_list;    // [n_books][MAX_MEMBERS][pool]: k_agents_mixed_wave's lists, book-major
  DevBuf<uint32_t> wl_len;
  bool mw_attr_set = false;
  bool use_mixed_wave() const {
    return n_mixed && M == 1 &&
           (pipeline == 4 || (pipeline == 0 && cfg.n_books >= MIXED_WAVE_MIN_BOOKS));
  }
  uint32_t fused_resident = 0;   // books one residency round of k_run_wave<R> holds on this device (0 = not asked yet)
  uint32_t wave_fused_max() const {
    if (R == 1) return lane_split_min_books(1) - 1u;
    const uint32_t res = fused_resident ? fused_resident : (R == 8 ? 2048u : 6144u);
    // (256-slot pools: a round holds 6 144 books like the 128-slot ones, but the split form is already ahead at 5 120 -
    // 37.7 vs 33.8 M - and level at 4 096)
    // (512-slot pools: two workgroups per CU fit since round 4 - 4 096 books - but the split form is 4 % ahead there: 25.4 vs 24.4 M)
    return R >= 8 ? std::min(res, 2048u) : (R >= 4 ? std::min(res, 4096u) : res);
  }
  bool use_wave() const {        // split form: k_agents_wave + k_step_batch
    return wave_ok() && (pipeline == 4 || (pipeline == 0 && cfg.n_books > wave_fused_max() && cfg.n_books < lane_split_min_books(R)));
  }
  bool use_wave_fused() const {  // persistent fused form: k_run_wave
    return wave_ok() && (pipeline == 5 || (pipeline == 0 && cfg.n_books <= wave_fused_max()));
  }
  int wave_split_parts() const {
    if (wave_parts > 0)  // set explicitly (tests, sweeps): any batch of >= 64 books per part
      return static_cast<int>(std::max(1u, std::min(static_cast<uint32_t>(wave_parts), cfg.n_books / 64u)));
    // One part per hardware queue (four) once a part holds 2 048 books.  Re-swept in round 4, after the event loops got
    // faster (scripts/exp_c5p.sh): C5 as written 32.3 / 32.8 / 34.0 / 21.9 M in 2 / 3 / 4 / 5 parts (round 3: two parts), C5
    // stand-in 26.1 / 27.0 / 19.3 M in 3 / 4 / 5, the C3 shards 116.7 / 117.7 M (8 192 books) and 140.0 / 139.7 M (16 384) in
    // 3 / 4; a fifth part shares a queue and halves the rate.
    return static_cast<int>(std::max(1u, std::min(4u, cfg.n_books / 2048u)));
  }
  // THE pipeline choice: the one function bk_run launches from and bk_get_pipeline reports from (they duplicated the rule
  // until round 4).  `pipeline` is the caller's request (0 auto); a request the env's agents cannot take (e.g. "wave" for
  // a market) falls back as the comments say.
  enum PlanKind {
    PL_FUSED_RANDOM,  // k_run_random: one wave per book, all phases, n_steps per launch (also: no agents = plain steps)
    PL_FUSED_WAVE,    // k_run_wave: wave-parallel decode + events, persistent
    PL_SPLIT_LANES,   // k_agents_fsm (one lane per book / market) + k_step_batch
    PL_SPLIT_WAVE,    // k_agents_wave (one wave per book, stream decoded 64 draws at a time) + k_step_batch
    PL_MIXED_FUSED,   // k_run_mixed: AgentSet members, fused
    PL_MIXED_WAVE,    // k_agents_mixed_wave + k_step_batch<POOLPEND>
    PL_MIXED_LANES,   // k_agents_mixed_lanes + k_step_batch<POOLPEND> (markets' only pipeline; on request otherwise)
    PL_MIXED_WPB,     // k_agents_mixed (one wave per book, scalar) + k_step_batch<POOLPEND> (mode 3, on request)
  };
  struct Plan {
    PlanKind kind;
    int parts;
  };
  Plan plan() const {
    if (n_mixed) {
      if (use_mixed_wave()) return {PL_MIXED_WAVE, wave_split_parts()};
      if (pipeline == 2 || M > 1) return {PL_MIXED_LANES, parts()};
      if (pipeline == 3) return {PL_MIXED_WPB, parts()};
      return {PL_MIXED_FUSED, 1};
    }
    if (use_wave_fused()) return {PL_FUSED_WAVE, 1};
    if (use_wave()) return {PL_SPLIT_WAVE, wave_split_parts()};
    // (auto with RandomAgents on independent books never gets here below lane_split_min_books: the wave forms take it)
    if ((pipeline >= 2 && pipeline != 5) || M > 1 || (pipeline == 0 && wave_ok())) return {PL_SPLIT_LANES, parts()};
    return {PL_FUSED_RANDOM, 1};
  }
  // split pipeline: the batch is cut into n_parts contiguous parts, each on its own stream and started one
  // k_agents_fsm apart, so the latency-bound lane-per-book kernel of one part runs under the issue-bound
  // wave-per-book kernel of another.
  int n_parts = 4;  // one per hardware queue (part_streams)
  uint32_t min_part = 4096;  // books (markets) per part below which the batch is cut in fewer parts
  static constexpr int MAX_PARTS = 8;
  hipStream_t part_stream[MAX_PARTS] = {};
  hipEvent_t ev_fork = nullptr, ev_first[MAX_PARTS] = {}, ev_join[MAX_PARTS] = {};
  DevBuf<DevTrade> trades;
  DevBuf<DevOrderLog> order_log;
  DevBuf<DevStats> stats;
  DevBuf<OutTrade> tr_dense;           // bk_trades_compact: dense record stream + CSR offsets
  DevBuf<unsigned long long> tr_off;
  uint64_t tr_total = 0;
  DevBuf<MixedDesc> mixed_descs;  // AgentSets with Noise/Momentum members (k_run_mixed)
  uint32_t n_mixed = 0, n_fixed = 0;
  // lane-per-book members' update (k_agents_mixed_lanes): the members' order lists, [member][entry][book]
  DevBuf<uint16_t> ml_list;
  DevBuf<uint32_t> ml_len, ml_inl;
  bool lds_attr_set = false;  // hipFuncAttributeMaxDynamicSharedMemorySize applied on this env's device
  DevBuf<uint64_t> gather_buf;  // n_books u64: gather_header()
  bool fsm_attr_set = false;  // same for k_agents_fsm (its LDS is dynamic: book_device.hpp)
  bool ml_valid = false;  // the lists describe the pool as of steps_done (false after a wave-per-book launch / restore)
  uint32_t member_asset[MAX_MEMBERS] = {0, 0, 0, 0};
  uint32_t n_fixed_a[MAX_ASSETS] = {0, 0, 0, 0, 0, 0, 0, 0};
  int parts() const {
    const uint32_t units = cfg.n_books / M;
    return static_cast<int>(std::max(1u, std::min(static_cast<uint32_t>(n_parts), units / min_part)));
  }
  MixedLists lists() const {
    return MixedLists{ml_list.p, ml_len.p, ml_inl.p, static_cast<uint32_t>(R) * 64u, cfg.n_books, cfg.n_books / M};
  }
  MixedArgs margs() const {
    MixedArgs ma{};
    ma.descs = mixed_descs.p;
    ma.n_desc = n_mixed;
    ma.n_fixed = n_fixed;
    for (int i = 0; i < MAX_MEMBERS; ++i) ma.asset[i] = member_asset[i];
    for (int i = 0; i < MAX_ASSETS; ++i) ma.n_fixed_a[i] = n_fixed_a[i];
    return ma;
  }
  std::vector<BookHost> books;
  std::unique_ptr<HostPool> pool;  // host threads for large host-driven batches (created on first use)
  HostPool& host_pool() {
    if (!pool) {
      unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
      if (const char* e = std::getenv("BOURSE_AMD_HOST_THREADS")) nt = static_cast<unsigned>(std::max(1, std::atoi(e)));
      pool.reset(new HostPool(nt - 1));
    }
    return *pool;
  }
  std::vector<Group> groups;
  uint32_t n_agents_total = 0;
  uint64_t agents_hash = 0;   // FNV-1a of the installed agent set (checkpoint compatibility)
  // device-resident instruction ingress (bk_device_ingress_enable): the queues, the id counters and the immutable halves of
  // the orders live on the device; the per-order host entries are refused, the readers mirror on demand
  bool device_ingress = false;
  uint32_t qcap = 0;               // events per book (market) and step
  DevBuf<uint4> dq, dorders;       // [n_markets][qcap] event records; [n_books][max_orders][2] immutable halves
  DevBuf<uint32_t> dqlen;          // [n_markets] queue lengths
  uint64_t ingest_epoch = 0;       // bumped by every submit / step: invalidates the readers' mirrors
  // HOST arrays through the device ingress (bk_submit_instructions_host): two slots of pinned + device staging, the
  // upload of ticket t + 1 on a copy stream of its own under the step kernel of ticket t, ids / status back on a third
  struct HostIngress {
    static constexpr int SLOTS = 2;
    size_t cap = 0;                 // elements per slot
    char* pin[SLOTS] = {nullptr, nullptr};
    char* dev[SLOTS] = {nullptr, nullptr};
    hipStream_t in = nullptr, out = nullptr;
    hipEvent_t e_h2d[SLOTS] = {}, e_ing[SLOTS] = {}, e_done[SLOTS] = {};
    uint64_t ticket_of[SLOTS] = {~0ull, ~0ull};  // the ticket whose arrays / results a slot holds
    uint64_t n_elem[SLOTS] = {0, 0};
    uint64_t next_ticket = 0;
    // pinned blocks a growth replaced: a view of their tickets' results (bk_submit_result_view) stays valid until two more
    // submits, so they are freed only then (they were freed at once and the views dangled)
    struct Retired {
      char* pin;
      uint64_t free_from_ticket;  // freed by the submit that takes this ticket (or by bk_env_destroy)
    };
    std::vector<Retired> retired;
    // one slot = [offsets (B + 1) u64 | order_id cap u64 | out_ids cap u64 | status 2 B u32 | action | vol | trader | price
    // cap u32 each | side cap u8], the same layout in pinned and in device memory
    size_t o_off = 0, o_oid = 0, o_out = 0, o_st = 0, o_act = 0, o_vol = 0, o_trd = 0, o_prc = 0, o_side = 0, bytes = 0;
    void layout(size_t B, size_t c) {
      cap = c;
      o_off = 0;
      o_oid = o_off + (B + 1) * 8;
      o_out = o_oid + c * 8;
      o_st = o_out + c * 8;
      o_act = o_st + B * 8;
      o_vol = o_act + c * 4;
      o_trd = o_vol + c * 4;
      o_prc = o_trd + c * 4;
      o_side = o_prc + c * 4;
      bytes = (o_side + c + 63) & ~size_t(63);
    }
  } hi;
  bool ev_seq_shuffle = false;  // measurement knobs of k_step_events' shuffle, from the environment at bk_env_create:
  int ev_shuffle_min = -1;      // draw by draw always / the queue length the wave-parallel form starts at (-1: the rule)
  std::atomic<bool> ev_mods_seen{false};  // a modification was submitted to this env (sticky): k_step_events<.., MODS = true> from then on
  uint32_t* mods_flag_host = nullptr;     // device ingress: k_ingest's hint word in mapped host memory, and its device address
  uint32_t* mods_flag_dev = nullptr;
  bool device_flow = false;   // bk_run has stepped this env with on-device agents: host-driven orders are refused
  uint64_t steps_done = 0, hist_base = 0;
  uint32_t trading = 1;
  size_t ev_capacity = 0;
  // profiling
  int profile = 0;       // 0 off, N: HIP-event-time the kernels of every Nth step
  uint64_t prof_tick = 0;  // steps seen while profiling
  bool prof_now = false;
  struct ProfEv {
    hipEvent_t a, b;
    int kind;
  };
  std::vector<ProfEv> prof_events;
  double prof_bracket_ms = 0;  // what two events recorded back to back on a stream read (calibrated, subtracted)
  std::vector<hipEvent_t> prof_pool;  // events created ahead of the launches that use them
  hipEvent_t prof_event() {
    hipEvent_t e = nullptr;
    if (!prof_pool.empty()) {
      e = prof_pool.back();
      prof_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
    }
    return e;
  }
  // per kernel kind: 0 the fused kernels, 1 the agents kernel of a split pipeline, 2 k_step_batch, 3 k_step_events
  static constexpr int PROF_KINDS = 4;
  double prof_ms[PROF_KINDS] = {0, 0, 0, 0};
  uint64_t prof_launches[PROF_KINDS] = {0, 0, 0, 0};

  DevArgs args() const {
    DevArgs a{};
    a.n_books = cfg.n_books;
    a.levels = cfg.levels;
    a.tick_size = cfg.tick_size;
    a.n_groups = static_cast<uint32_t>(groups.size());
    a.step_lo = static_cast<uint32_t>(cfg.step_size);
    a.step_hi = static_cast<uint32_t>(cfg.step_size >> 32);
    a.state_stride = stride;
    a.l2_width = W;
    a.trade_cap = warming ? 0u : cfg.trade_capacity;
    a.hist_cap = warming ? 0u : cfg.history_capacity;
    a.hist_slot0 = a.hist_cap ? static_cast<uint32_t>(steps_done % cfg.history_capacity) : 0u;
    a.step_prio = 0;
    a.n_agents_total = n_agents_total;
    a.log_cap = cfg.max_orders;
    a.state = state.p;
    a.l2_last = l2_last.p;
    a.hist = hist.p;
    a.trades = trades.p;
    a.order_log = order_log.p;
    a.ev_off = ev_off.p;
    a.ev = device_ingress ? dq.p : ev.p;
    a.ev_len = device_ingress ? dqlen.p : nullptr;
    a.ev_stride = qcap;
    a.batch = batch.p;
    a.batch_stride = batch_stride;
    a.book_begin = 0;
    a.book_end = cfg.n_books / M;
    a.assets = M;
    for (int i = 0; i < MAX_ASSETS; ++i) a.asset_tick[i] = asset_tick[i];
    auto dv = [](uint32_t d) {
      const HostUDiv h = make_udiv(d ? d : 1u);
      return UDiv{h.m, h.sh1, h.sh2, h.d};
    };
    a.tick_div = dv(cfg.tick_size);
    for (int i = 0; i < MAX_ASSETS; ++i) a.asset_div[i] = dv(asset_tick[i]);
    for (size_t g = 0; g < groups.size(); ++g) a.groups[g] = groups[g];
    return a;
  }
};

namespace {

int use_device(bk_env* env) {
  HIPCHK(hipSetDevice(env->cfg.device));
  return BK_OK;
}

struct ProfScope {  // HIP events around a launch on the env's stream
  bk_env* env;
  int kind;
  hipStream_t st;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(bk_env* e, int k, hipStream_t s = nullptr) : env(e), kind(k), st(s ? s : e->stream) {
    if (env->prof_now && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
      (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (a && b) {
      (void)hipEventRecord(b, st);
      env->prof_events.push_back({a, b, kind});
    }
  }
};

// A launch of the multi-stream pipelines, timed - when it is sampled - by two events recorded on its stream.
template <typename K, typename... Args>
void launch_timed(bk_env* env, int kind, K kernel, dim3 grid, dim3 block, uint32_t lds, hipStream_t st, Args... args) {
  hipEvent_t a = nullptr, b = nullptr;
  // Sampling whole STEPS (round 1) was biased once four parts overlapped: the sampled step's launches reach the GPU late
  // (six event creations on the launch path), overlap less and run 10 % faster than the rest - in the kernel trace
  // itself, scripts/kt_check.sh.  Events attached to the dispatch (hipExtLaunchKernelGGL) read 4-11 % long on these
  // 35-140 us kernels.  So: single launches are sampled, one in
  // (2 x profile + 1), which with 2 x parts launches per step walks through every part and both kernels; the events come
  // from a pool created when profiling is switched on, not from hipEventCreate on the launch path.
  const uint64_t every = env->profile == 1 ? 1u : 2u * static_cast<uint64_t>(env->profile) + 1u;  // odd: walks all parts
  const bool sample = env->profile > 0 && (env->prof_tick++ % every) == 0;
  if (sample && (a = env->prof_event()) && (b = env->prof_event())) {
    (void)hipEventRecord(a, st);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    (void)hipEventRecord(b, st);
    env->prof_events.push_back({a, b, kind});
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}

template <int R>
int launch_run(bk_env* env, const DevArgs& a, uint64_t first_step, uint32_t n_steps) {
  const uint32_t blocks = (env->cfg.n_books + 3) / 4;
  env->prof_now = env->profile > 0;
  ProfScope ps(env, 0);
  hipLaunchKernelGGL(k_run_random<R>, dim3(blocks), dim3(256), 0, env->stream, a, first_step, n_steps);
  HIPCHK(hipGetLastError());
  return BK_OK;
}
int wave_args(bk_env* env, WaveArgs* wva);
template <int R>
int launch_events(bk_env* env, const DevArgs& a, uint64_t step_index, uint32_t max_queue) {
  env->prof_now = env->profile > 0;
  ProfScope ps(env, 3);
  uint32_t perm_bytes = ((max_queue + 63u) & ~63u) * 2u + 128u;  // u16 permutation of the longest queue
  if (perm_bytes < ev_lds_bytes(R)) perm_bytes = ev_lds_bytes(R);  // ... the wave-parallel shuffle's and the keyed form's lists (step_events.hpp)
  // a queue longer than the pool runs the keyed form chunk by chunk (round 6): its work area sits BEHIND the permutation
  if (max_queue > 64u * R) perm_bytes = std::max(perm_bytes, ((max_queue + 63u) & ~63u) * 2u + ev_keyed_lds_bytes(R));
  // the shuffle borrows the decode's jump tables and per-book lane-state cache (BOURSE_AMD_EV_SEQ_SHUFFLE=1: the draw-by-draw
  // loop, for measurements)
  // (both knobs are read ONCE, when the env is created - bk_env_create -: a getenv per launch raced with another thread's
  // setenv; a test sets them before it creates its env)
  const bool seq_shuffle = env->ev_seq_shuffle;
  // (its fixed cost - the cache record, a block of draws, a resolution over all 64 R positions - pays from a queue length that
  // grows with the pool: docs/EXPERIMENTS.md; BOURSE_AMD_EV_WAVE_SHUFFLE_MIN overrides, for measurements)
  const int min_env = env->ev_shuffle_min;
  const uint32_t shuffle_min = min_env >= 0 ? static_cast<uint32_t>(min_env) : (12u * R > 32u ? 12u * R : 32u);  // (measured: 256 slots 24 events -3 %, 48 +5 %; 512 slots 48 -7 %, 96 +3 %)
  // the lane-state cache (1 280 B per book) and the jump tables exist only once a step CAN take the wave-parallel shuffle: no
  // queue of this launch reaches its threshold -> wcache stays null and the kernel draws one by one (every env's
  // first bk_step allocated 84 MB at 65 536 books whether or not it ever shuffled that way)
  WaveArgs wva{};
  if (!seq_shuffle && max_queue >= shuffle_min && max_queue >= 2u)
    if (int rc = wave_args(env, &wva)) return rc;
  const bool chunks = max_queue > 64u * R;  // a queue longer than the pool: the instantiation whose keyed form runs chunk by chunk
  const dim3 grid(env->cfg.n_books), block(64);
  // 512-slot pools: the kernel WITH the keyed modifications only once this env has seen one (step_events.hpp k_step_events: the
  // clean launch is 13 % faster without that code).  Sticky; k_ingest's hint may lag by a step - that step's modifications then
  // run event by event, with the same results.
  if (env->mods_flag_host && *static_cast<volatile uint32_t*>(env->mods_flag_host)) env->ev_mods_seen.store(true, std::memory_order_relaxed);
  if constexpr (R == 8) {
    if (!chunks && !env->ev_mods_seen.load(std::memory_order_relaxed)) {
      if (env->M == 1)
        hipLaunchKernelGGL((k_step_events<R, false, false, false>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
      else
        hipLaunchKernelGGL((k_step_events<R, true, false, false>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
      HIPCHK(hipGetLastError());
      return BK_OK;
    }
  }
  if (env->M == 1 && !chunks)
    hipLaunchKernelGGL((k_step_events<R, false, false>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
  else if (env->M == 1)
    hipLaunchKernelGGL((k_step_events<R, false, true>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
  else if (!chunks)
    hipLaunchKernelGGL((k_step_events<R, true, false>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
  else
    hipLaunchKernelGGL((k_step_events<R, true, true>), grid, block, perm_bytes, env->stream, a, wva, step_index, shuffle_min, perm_bytes);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

template <int R>
int launch_mixed(bk_env* env, const DevArgs& a, uint64_t first_step, uint32_t n_steps) {
  const uint32_t blocks = (env->cfg.n_books + 3) / 4;
  env->prof_now = env->profile > 0;
  ProfScope ps(env, 0);
  const MixedArgs ma = env->margs();
  env->wl_valid = false;  // (the wave-per-book lists no longer describe the pools)
  hipLaunchKernelGGL(k_run_mixed<R>, dim3(blocks), dim3(256), 0, env->stream, a, ma, first_step, n_steps);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

// jump tables (T^256, then T^(4 << b)) and the per-book lane-state cache of the wave-parallel decode, on first use
int wave_args(bk_env* env, WaveArgs* wva) {
  if (!env->jump_tabs.p) {
    std::vector<uint32_t> all;
    for (int t = 0; t < 7; ++t) {
      const std::vector<uint32_t> tab = xoroshiro_jump_table(t == 0 ? WV_BLOCK : static_cast<uint64_t>(WV_K) << (t - 1));
      all.insert(all.end(), tab.begin(), tab.end());
    }
    HIPCHK(env->jump_tabs.alloc(all.size() / 4));
    HIPCHK(env->wcache.alloc(static_cast<size_t>(env->cfg.n_books) * WC_STRIDE));
    // on the ENV's stream (a hipStreamNonBlocking stream is not ordered behind the null stream), and waited for: `all` is a local
    HIPCHK(hipMemcpyAsync(env->jump_tabs.p, all.data(), all.size() * 4, hipMemcpyHostToDevice, env->stream));
    HIPCHK(hipMemsetAsync(env->wcache.p, 0, static_cast<size_t>(env->cfg.n_books) * WC_STRIDE * 4, env->stream));
    HIPCHK(hipStreamSynchronize(env->stream));
  }
  wva->jt_block = env->jump_tabs.p;
  wva->jt_lane = env->jump_tabs.p + 512;
  wva->wcache = env->wcache.p;
  wva->lookahead = env->wave_lookahead;
  return BK_OK;
}

template <int R>
int launch_wave_fused(bk_env* env, const DevArgs& a, uint64_t first_step, uint32_t n_steps) {
  WaveArgs wva{};
  if (int rc = wave_args(env, &wva)) return rc;
  env->prof_now = env->profile > 0;
  ProfScope ps(env, 0);
  hipLaunchKernelGGL(k_run_wave<R>, dim3((env->cfg.n_books + 7) / 8), dim3(512), 0, env->stream, a, wva, first_step, n_steps);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

// The parts' streams: one process-wide set per device, every member PROBED to sit on a hardware queue of its own.
// HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues - a new stream joins the least-referenced queue
// once four exist - and two parts that share a queue run back to back: the pipeline loses its overlap (C3 220 -> 124 M
// book-steps/s with a second env that had created its own streams first, 216 -> 119 M under torch.distributed.run, where
// RCCL's streams come first).  Which queue a stream got cannot be asked, but it can be measured: two 300 us spin kernels
// on two streams take 300 us on different queues and 600 us on one.  Eight candidates are created, a mutually
// concurrent subset of up to four is kept (greedy), the rest destroyed; every env on the device uses that set for all
// its parts (the caller's stream only forks and joins), envs sharing it just interleave.  CU-masked streams, which do
// get a queue each, were measured and are not an option (98 M).  ~10 ms once per process and device.
const std::vector<hipStream_t>& part_streams(int device) {
  static std::mutex mu;
  static std::map<int, std::vector<hipStream_t>> pool;
  std::lock_guard<std::mutex> lk(mu);
  auto it = pool.find(device);
  if (it != pool.end()) return it->second;
  std::vector<hipStream_t> cand, chosen;
  // BOURSE_AMD_PART_STREAMS=N (1..8): take N fresh streams WITHOUT probing.  For launchers that start many ranks at once
  // (one process per GPU: each rank would otherwise run its 8 x 300 us probe kernels at start-up, all at the same time,
  // next to RCCL's communicator set-up), and for boxes where the host-clock probe is unreliable.  Default: probe.
  if (const char* e = std::getenv("BOURSE_AMD_PART_STREAMS")) {
    const int want = std::atoi(e);
    if (want >= 1 && want <= 8) {
      for (int k = 0; k < want; ++k) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        chosen.push_back(st);
      }
      (void)hipGetLastError();
      if (getenv("BOURSE_AMD_VERBOSE"))
        fprintf(stderr, "bourse_amd: device %d: %zu part streams taken unprobed (BOURSE_AMD_PART_STREAMS)\n", device, chosen.size());
      return pool.emplace(device, std::move(chosen)).first->second;
    }
  }
  for (int k = 0; k < 8; ++k) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, 1u);  // first use: the stream takes its hardware queue now
    cand.push_back(st);
  }
  (void)hipDeviceSynchronize();  // the probe times kernels: nothing of ours may still be running
  constexpr uint32_t SPIN_US = 300;
  auto concurrent = [&](hipStream_t x, hipStream_t y) {
    for (int attempt = 0; attempt < 2; ++attempt) {  // a slow pair is measured twice (someone else's work on the GPU)
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, x, SPIN_US * 100u);
      hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, y, SPIN_US * 100u);
      (void)hipStreamSynchronize(x);
      (void)hipStreamSynchronize(y);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < 1.6 * SPIN_US) return true;
    }
    return false;
  };
  for (hipStream_t c : cand) {
    bool ok = chosen.size() < 4;
    for (size_t j = 0; ok && j < chosen.size(); ++j) ok = concurrent(c, chosen[j]);
    if (ok)
      chosen.push_back(c);
    else
      (void)hipStreamDestroy(c);
  }
  (void)hipGetLastError();
  if (getenv("BOURSE_AMD_VERBOSE"))
    fprintf(stderr, "bourse_amd: device %d: %zu of %zu candidate streams on hardware queues of their own\n", device,
            chosen.size(), cand.size());
  // fewer than four: the parts of a split launch share queues and run back to back (C3 220 -> 124 M book-steps/s when
  // every part sat on one queue).  Results are unaffected; say so once instead of being silently slow.
  if (chosen.size() < 4 && !cand.empty())
    fprintf(stderr, "bourse_amd: warning: device %d: only %zu of %zu candidate streams run concurrently (busy GPU or serialised "
            "kernels?): multi-part launches will overlap less; BOURSE_AMD_PART_STREAMS=4 skips the probe\n", device,
            chosen.size(), cand.size());
  return pool.emplace(device, std::move(chosen)).first->second;
}

// split pipeline: per step and per part one lane-per-book launch (RNG-serial phases) + one wave-per-book launch
// MIXED: 0 RandomAgents groups (k_agents_fsm), 1 AgentSet members one wave per book (k_agents_mixed), 2 members one lane
// per book (k_agents_mixed_lanes)
template <int R, int MIXED = 0>
int launch_split(bk_env* env, const DevArgs& a0, uint64_t first_step, uint32_t n_steps) {
  const MixedArgs ma = env->margs();
  const uint32_t fsm_lds = fsm_lds_bytes(R);
  if (MIXED == 0 && !env->fsm_attr_set) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agents_fsm<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(fsm_lds)));
    env->fsm_attr_set = true;
  }
  if (MIXED == 2) {
    const size_t NB = env->cfg.n_books, cap = static_cast<size_t>(R) * 64, NU = NB / env->M;
    if (!env->ml_list.p) {
      HIPCHK(env->ml_list.alloc(MAX_MEMBERS * cap * NU));
      HIPCHK(env->ml_len.alloc(MAX_MEMBERS * NU));
      HIPCHK(env->ml_inl.alloc(2 * static_cast<size_t>(R) * NB));
      env->ml_valid = false;
    }
    if (!env->lds_attr_set) {  // allow > 64 KB of dynamic LDS (160 KB per workgroup on MI355X); the attribute is per device
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agents_mixed_lanes<R, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mixed_lanes_lds_bytes(R, true))));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agents_mixed_lanes<R, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mixed_lanes_lds_bytes(R, false))));
      env->lds_attr_set = true;
    }
    if (!env->ml_valid) {
      hipLaunchKernelGGL(k_mixed_lists_rebuild<R>, dim3((env->cfg.n_books + 3) / 4), dim3(256), 0, env->stream, a0, ma,
                         env->lists());
      HIPCHK(hipGetLastError());
      env->ml_valid = true;
    }
  } else if (MIXED == 1) {
    env->ml_valid = false;
  }
  if (MIXED == 1 || MIXED == 2) env->wl_valid = false;  // (the wave-per-book lists no longer describe the pools)
  const MixedLists ml = env->lists();
  const bool wave = (MIXED == 0 && env->use_wave()) || MIXED == 3;
  WaveArgs wva{};
  if (wave)
    if (int rc = wave_args(env, &wva)) return rc;
  // (diagnostic: BOURSE_AMD_MW_LDS_PAD=bytes of unused dynamic LDS added to every k_agents_mixed_wave workgroup - an occupancy
  // sensitivity probe: how much does the members' decode lose at FEWER waves per SIMD, i.e. what could more buy?)
  static const uint32_t mw_pad = [] {
    const char* e = std::getenv("BOURSE_AMD_MW_LDS_PAD");
    return e ? static_cast<uint32_t>(std::max(0, std::atoi(e))) & ~3u : 0u;
  }();
  // (experiment, docs/EXPERIMENTS.md: BOURSE_AMD_STEP_DECODE=1 runs a part's inner steps of the wave_split pipeline as ONE launch
  // each - k_step_decode = events of step s + decode of step s + 1)
  static const bool step_decode = [] {
    const char* e = std::getenv("BOURSE_AMD_STEP_DECODE");
    return e && std::atoi(e) != 0;
  }();
  const bool fuse_sd = step_decode && wave && MIXED == 0 && env->M == 1 && !env->warming;
  static const uint32_t wave_pad = [] {  // (the same probe for k_agents_wave: BOURSE_AMD_WAVE_LDS_PAD)
    const char* e = std::getenv("BOURSE_AMD_WAVE_LDS_PAD");
    return e ? static_cast<uint32_t>(std::max(0, std::atoi(e))) & ~3u : 0u;
  }();
  if (MIXED == 3) {
    if (!env->mw_attr_set) {  // > 64 KB of dynamic LDS at R = 8 (160 KB per workgroup on MI355X); per device
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agents_mixed_wave<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 static_cast<int>(mixed_wave_lds_bytes(R) + mw_pad)));
      env->mw_attr_set = true;
    }
    if (!env->wl_list.p) {
      HIPCHK(env->wl_list.alloc(static_cast<size_t>(env->cfg.n_books) * MAX_MEMBERS * R * 64));
      HIPCHK(env->wl_len.alloc(static_cast<size_t>(env->cfg.n_books) * MAX_MEMBERS));
      env->wl_valid = false;
    }
    if (!env->wl_valid) {  // another pipeline (or a restore / a fresh env) changed the pools: lists from the owner tags
      hipLaunchKernelGGL(k_wave_lists_rebuild<R>, dim3((env->cfg.n_books + 3) / 4), dim3(256), 0, env->stream, a0, ma,
                         WaveLists{env->wl_list.p, env->wl_len.p, static_cast<uint32_t>(R) * 64u});
      HIPCHK(hipGetLastError());
      env->wl_valid = true;
    }
    env->ml_valid = false;
  }
  const uint32_t M = env->M;
  const uint32_t B = env->cfg.n_books / M;  // units the parts are cut in: books, or markets of M books
  // small batches: one part on the caller's stream
  const int P = wave ? env->wave_split_parts() : env->parts();
  if (P > 1 && !env->ev_fork) {
    const std::vector<hipStream_t>& ps = part_streams(env->cfg.device);
    if (ps.empty()) return fail(BK_HIP_ERROR, "could not create the parts' streams");
    for (int i = 0; i < bk_env::MAX_PARTS; ++i) {
      env->part_stream[i] = ps[static_cast<size_t>(i) % ps.size()];  // more parts than queues: they share
      if (!env->ev_first[i]) HIPCHK(hipEventCreateWithFlags(&env->ev_first[i], hipEventDisableTiming));
      if (!env->ev_join[i]) HIPCHK(hipEventCreateWithFlags(&env->ev_join[i], hipEventDisableTiming));
    }
    // last: ev_fork doubles as "the parts' streams and events are set up" (a failure above leaves it unset, so the
    // next bk_run tries again instead of launching on null streams)
    HIPCHK(hipEventCreateWithFlags(&env->ev_fork, hipEventDisableTiming));
  }
  if (P > 1) {
    HIPCHK(hipEventRecord(env->ev_fork, env->stream));
    for (int i = 0; i < P; ++i) HIPCHK(hipStreamWaitEvent(env->part_stream[i], env->ev_fork, 0));
  }
  for (uint32_t s = 0; s < n_steps; ++s) {
    env->prof_now = false;  // (launch_timed samples single launches)
    for (int i = 0; i < P; ++i) {
      DevArgs a = a0;
      a.book_begin = static_cast<uint32_t>(static_cast<uint64_t>(B) * i / P) & ~3u;
      a.book_end = (i + 1 == P) ? B : (static_cast<uint32_t>(static_cast<uint64_t>(B) * (i + 1) / P) & ~3u);
      a.hist_slot0 = a.hist_cap ? static_cast<uint32_t>((first_step + s) % a.hist_cap) : 0u;
      a.step_prio = (wave && B >= wave_step_prio_books(R)) ? 1u : 0u;
      const uint32_t nb = a.book_end - a.book_begin;
      hipStream_t st = P > 1 ? env->part_stream[i] : env->stream;
      if (P > 1 && s == 0 && i > 0) {  // stagger the parts
        // by time: i x stagger_us.  The lane split's parts cycle through a ~180 us agents kernel and a ~140 us event
        // kernel; one agents kernel apart (the round-1 rule) puts part 2 at 360 us = almost in phase with part 0 again.
        // Measured at C3 (driver's 20-step regions): 60 us apart 186-189 M first region / 199-201 M later ones against
        // 182 / 192-195 M (BOURSE_AMD_STAGGER_US overrides; other pipelines keep the event-based stagger)
        // (round 3, 20-step regions, first / median of five: 0 us 226 / 229 M, 20 us 240 / 246, 35 us 238 / 244, 50 us 241 / 242,
        // 70 us 235 / 238; no difference over 200 steps)
        const uint32_t stagger = env->stagger_us != ~0u ? env->stagger_us : ((MIXED == 0 && !wave && P >= 3) ? 30u : 0u);
        if (stagger > 0)
          hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, static_cast<uint32_t>(i) * stagger * 100u);
        else                           // by one agents kernel each
          HIPCHK(hipStreamWaitEvent(st, env->ev_first[i - 1], 0));
      }
      const uint64_t step_no = first_step + s;
      if (MIXED == 3)
        launch_timed(env, 1, &k_agents_mixed_wave<R>, dim3((nb + MW_WPB - 1) / MW_WPB), dim3(64 * MW_WPB),
                     static_cast<uint32_t>(mixed_wave_lds_bytes(R)) + mw_pad, st, a, ma, wva,
                     WaveLists{env->wl_list.p, env->wl_len.p, static_cast<uint32_t>(R) * 64u});
      else if (MIXED == 2 && M > 1)
        launch_timed(env, 1, &k_agents_mixed_lanes<R, true>, dim3((nb + 63) / 64), dim3(64), mixed_lanes_lds_bytes(R, true), st, a,
                     ma, ml);
      else if (MIXED == 2)
        launch_timed(env, 1, &k_agents_mixed_lanes<R, false>, dim3((nb + 63) / 64), dim3(64), mixed_lanes_lds_bytes(R, false), st,
                     a, ma, ml);
      else if (MIXED == 1)
        launch_timed(env, 1, &k_agents_mixed<R>, dim3((nb + 3) / 4), dim3(256), 0u, st, a, ma);
      else if (wave && fuse_sd && s > 0)
        ;  // (the previous step's k_step_decode has decoded this step already)
      else if (wave)
        launch_timed(env, 1, &k_agents_wave<R>, dim3((nb + 3) / 4), dim3(256), wave_pad, st, a, wva);
      else
        launch_timed(env, 1, &k_agents_fsm<R>, dim3((nb + 63) / 64), dim3(64), fsm_lds, st, a);
      if (P > 1 && s == 0) HIPCHK(hipEventRecord(env->ev_first[i], st));
      // the lane-per-book members' update reads the touches from the latest level-2 record: keep it current
      const uint32_t write_last = (s + 1 == n_steps || a.hist_cap == 0 || MIXED >= 2) ? 1u : 0u;
      if (MIXED && M > 1)
        launch_timed(env, 2, &k_step_batch<R, true, true>, dim3(nb * M), dim3(64), 0u, st, a, step_no, write_last);
      else if (MIXED)
        launch_timed(env, 2, &k_step_batch<R, false, true>, dim3(nb), dim3(64), 0u, st, a, step_no, write_last);
      else if (M > 1)
        launch_timed(env, 2, &k_step_batch<R, true>, dim3(nb * M), dim3(64), 0u, st, a, step_no, write_last);
      else if (fuse_sd && s + 1 < n_steps) {
        DevArgs an = a;  // (the decode half belongs to the NEXT step; nothing in it reads the history slot)
        launch_timed(env, 2, &k_step_decode<R>, dim3((nb + 3) / 4), dim3(256), 0u, st, an, wva, step_no, write_last);
      } else
        launch_timed(env, 2, &k_step_batch<R, false>, dim3(nb), dim3(64), 0u, st, a, step_no, write_last);
    }
  }
  HIPCHK(hipGetLastError());
  if (P > 1) {
    for (int i = 0; i < P; ++i) {
      HIPCHK(hipEventRecord(env->ev_join[i], env->part_stream[i]));
      HIPCHK(hipStreamWaitEvent(env->stream, env->ev_join[i], 0));
    }
  }
  return BK_OK;
}

// make `later` wait for the work queued on `earlier` so far
int order_after(hipStream_t later, hipStream_t earlier) {
  hipEvent_t ev = nullptr;
  HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, earlier);
  if (e == hipSuccess) e = hipStreamWaitEvent(later, ev, 0);
  (void)hipEventDestroy(ev);  // released once the wait has been satisfied
  if (e != hipSuccess) return fail(BK_HIP_ERROR, std::string("stream ordering: ") + hipGetErrorString(e));
  return BK_OK;
}

// Host-driven orders on an env whose books are populated by on-device agents would restart the order ids at 0 (colliding
// with the agents' ids) and could take an agent's pool slot: one env runs ONE of the two flows.
int host_flow_ok(bk_env* env) {
  if (env->device_ingress)
    return fail(BK_INVALID_ARGUMENT, "this env takes its instructions from device memory (bk_device_ingress_enable): "
                                     "per-order host calls would collide with the ids assigned on the device");
  if (env->device_flow)
    return fail(BK_INVALID_ARGUMENT, "host-driven orders cannot be mixed with bk_run's on-device agents on the same env");
  return BK_OK;
}

int check_book(bk_env* env, uint32_t book) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (book >= env->cfg.n_books) return fail(BK_INVALID_ARGUMENT, "book index out of range");
  return BK_OK;
}

int refresh_log(bk_env* env, uint32_t book) {
  BookHost& bh = env->books[book];
  if (bh.log_fresh) return BK_OK;
  const uint64_t n = std::min<uint64_t>(bh.n_uploaded, env->cfg.max_orders);
  bh.log_cache.resize(n);
  if (n) {
    HIPCHK(hipStreamSynchronize(env->stream));
    HIPCHK(hipMemcpy(bh.log_cache.data(), env->order_log.p + static_cast<size_t>(book) * env->cfg.max_orders,
                     n * sizeof(DevOrderLog), hipMemcpyDeviceToHost));
  }
  bh.log_fresh = true;
  return BK_OK;
}

// Device-resident ingress: the readers' view of one book's orders.  The immutable halves written by k_ingest and the
// book's id counter are fetched into the same BookHost fields the host-driven path fills at bk_place_order, so every
// reader below works unchanged; valid until the next submit / step (ingest_epoch).
int mirror_orders(bk_env* env, uint32_t book) {
  if (!env->device_ingress) return BK_OK;
  BookHost& bh = env->books[book];
  if (bh.mirror_epoch == env->ingest_epoch) return BK_OK;
  HIPCHK(hipSetDevice(env->cfg.device));
  HIPCHK(hipStreamSynchronize(env->stream));
  uint32_t count = 0;
  HIPCHK(hipMemcpy(&count, env->state.p + static_cast<size_t>(book) * env->stride + H_NEXT_ID, 4, hipMemcpyDeviceToHost));
  const uint64_t logged = std::min<uint64_t>(count, env->cfg.max_orders);
  std::vector<uint4> rows(logged * 2);
  if (logged)
    HIPCHK(hipMemcpy(rows.data(), env->dorders.p + static_cast<size_t>(book) * env->cfg.max_orders * 2, logged * 32,
                     hipMemcpyDeviceToHost));
  bh.orders.assign(count, HostOrder{0, 0, 0, 0, 0});
  for (uint64_t id = 0; id < logged; ++id) {
    const uint4 a = rows[2 * id], b = rows[2 * id + 1];
    bh.orders[id] = HostOrder{static_cast<uint8_t>(a.w & 1u), a.x, a.z, a.y, (static_cast<uint64_t>(b.y) << 32) | b.x};
  }
  bh.n_uploaded = count;  // every order has had a log entry since k_ingest created it
  bh.log_fresh = false;
  bh.mirror_epoch = env->ingest_epoch;
  return BK_OK;
}

int prof_collect(bk_env* env) {
  HIPCHK(hipStreamSynchronize(env->stream));
  for (auto& pr : env->prof_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      env->prof_ms[pr.kind] += std::max(0.0, static_cast<double>(ms) - env->prof_bracket_ms);
      env->prof_launches[pr.kind] += 1;
    }
    env->prof_pool.push_back(pr.a);
    env->prof_pool.push_back(pr.b);
  }
  env->prof_events.clear();
  return BK_OK;
}

void fill_order(const bk_env* env, const BookHost& bh, uint64_t id, bk_order* o) {
  const HostOrder& h = bh.orders[id];
  std::memset(o, 0, sizeof(*o));
  o->side_is_bid = h.bid;
  o->start_vol = h.start_vol;
  o->trader_id = h.trader;
  o->order_id = id;
  if (id < bh.n_uploaded && id < bh.log_cache.size()) {
    const DevOrderLog& d = bh.log_cache[id];
    o->status = static_cast<uint8_t>(d.status);
    o->vol = d.vol;
    o->price = d.price;
    o->arr_time = (static_cast<uint64_t>(d.arr_hi) << 32) | d.arr_lo;
    o->end_time = (static_cast<uint64_t>(d.end_hi) << 32) | d.end_lo;
  } else {  // created, New event still queued (or beyond the log capacity)
    o->status = 0;
    o->vol = h.start_vol;
    o->price = h.price;
    o->arr_time = h.create_time;
    o->end_time = ~0ull;
  }
  (void)env;
}

}  // namespace

extern "C" {

const char* bk_last_error(void) { return g_err.c_str(); }

int bk_hip_versions(int* built, int* runtime) {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0;
  if (built) *built = HIP_VERSION;
  if (runtime) *runtime = v;
  return BK_OK;
}

int bk_device_count(int* out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (out) *out = n;
  return BK_OK;
}

int bk_env_create(const bk_config* cfg, bk_env** out) {
  if (!cfg || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (cfg->n_books == 0) return fail(BK_INVALID_ARGUMENT, "n_books must be >= 1");
  if (cfg->tick_size == 0) return fail(BK_INVALID_ARGUMENT, "tick_size must be > 0");  // orderbook.rs:159
  if (cfg->levels == 0 || cfg->levels > 64) return fail(BK_INVALID_ARGUMENT, "levels must be in 1..64");
  const uint32_t M = cfg->assets ? cfg->assets : 1u;
  if (M > MAX_ASSETS || cfg->n_books % M != 0)
    return fail(BK_INVALID_ARGUMENT, "assets must be <= 8 and divide n_books");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BK_NO_DEVICE, "no HIP device available: bourse_amd has no CPU execution path");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(BK_INVALID_ARGUMENT, "device ordinal out of range");

  std::unique_ptr<bk_env> env(new bk_env());
  env->cfg = *cfg;
  env->M = M;
  for (uint32_t i = 0; i < MAX_ASSETS; ++i) env->asset_tick[i] = cfg->tick_size;
  uint32_t pool = cfg->max_live_orders ? cfg->max_live_orders : 128;
  int R = 1;
  while (R * 64u < pool) R *= 2;
  if (R > 8) return fail(BK_INVALID_ARGUMENT, "max_live_orders must be <= 512");
  env->R = R;
  env->cfg.max_live_orders = R * 64;
  env->W = 5 + 4 * cfg->levels;
  env->stride = HDR_DW + R * POOL_FIELDS * 64;
  env->trading = cfg->trading ? 1 : 0;
  HIPCHK(hipSetDevice(cfg->device));

  const size_t B = cfg->n_books;
  HIPCHK(env->state.alloc(B * env->stride));
  HIPCHK(env->l2_last.alloc(B * env->W));
  HIPCHK(env->hist.alloc(static_cast<size_t>(cfg->history_capacity) * B * env->W));
  HIPCHK(env->trades.alloc(B * cfg->trade_capacity));
  HIPCHK(env->order_log.alloc(B * cfg->max_orders));
  HIPCHK(env->stats.alloc(2));  // [0] the record, [1] its reset template
  {
    DevStats init{};
    init.min_bid = 0xFFFFFFFFu;
    init.min_ask = 0xFFFFFFFFu;
    HIPCHK(hipMemcpy(env->stats.p + 1, &init, sizeof(init), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(env->stats.p, &init, sizeof(init), hipMemcpyHostToDevice));
  }
  env->batch_stride = 64 + 160 * R;
  HIPCHK(env->batch.alloc(B * env->batch_stride));
  HIPCHK(hipMemset(env->batch.p, 0, B * env->batch_stride * sizeof(uint32_t)));
  if (const char* pm = std::getenv("BOURSE_AMD_PIPELINE")) {
    if (std::strcmp(pm, "fused") == 0) env->pipeline = 1;
    if (std::strcmp(pm, "split") == 0) env->pipeline = 2;
    if (std::strcmp(pm, "wave_split") == 0) env->pipeline = 4;
    if (std::strcmp(pm, "wave") == 0) env->pipeline = 5;
  }
  if (const char* np = std::getenv("BOURSE_AMD_SPLIT_PARTS")) {
    const int v = std::atoi(np);
    if (v >= 1 && v <= bk_env::MAX_PARTS) env->n_parts = v;
  }
  if (const char* sq = std::getenv("BOURSE_AMD_EV_SEQ_SHUFFLE")) env->ev_seq_shuffle = *sq == '1';
  if (const char* sm = std::getenv("BOURSE_AMD_EV_WAVE_SHUFFLE_MIN")) env->ev_shuffle_min = std::atoi(sm);
  if (const char* su = std::getenv("BOURSE_AMD_STAGGER_US")) env->stagger_us = static_cast<uint32_t>(std::max(0, std::atoi(su)));
  if (const char* mp = std::getenv("BOURSE_AMD_MIN_PART")) {
    const int v = std::atoi(mp);
    if (v >= 64) env->min_part = static_cast<uint32_t>(v);
  }
  HIPCHK(env->ev_off.alloc(B + 1));
  HIPCHK(hipMemset(env->ev_off.p, 0, (B + 1) * sizeof(uint32_t)));

  // initial device state: empty books, per-book RNG streams, Env::new's empty-book snapshot
  std::vector<uint32_t> st(B * env->stride, 0u), l2(B * env->W, 0u);
  for (size_t b = 0; b < B; ++b) {
    uint32_t* h = st.data() + b * env->stride;
    uint64_t s0, s1;
    seed_from_u64(cfg->seed + cfg->book_offset + b / M, s0, s1);  // one stream per market (runner.rs:115)
    h[H_T_LO] = static_cast<uint32_t>(cfg->start_time);
    h[H_T_HI] = static_cast<uint32_t>(cfg->start_time >> 32);
    h[H_S0_LO] = static_cast<uint32_t>(s0);
    h[H_S0_HI] = static_cast<uint32_t>(s0 >> 32);
    h[H_S1_LO] = static_cast<uint32_t>(s1);
    h[H_S1_HI] = static_cast<uint32_t>(s1 >> 32);
    h[H_TRADING] = env->trading;
    uint32_t* l = l2.data() + b * env->W;
    l[1] = 0u;           // bid touch of an empty side (side.rs:194-196)
    l[2] = 0xFFFFFFFFu;  // ask touch of an empty side (side.rs:99-104)
  }
  HIPCHK(hipMemcpy(env->state.p, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(env->l2_last.p, l2.data(), l2.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  env->books.resize(B);
  *out = env.release();
  return BK_OK;
}

void bk_env_destroy(bk_env* env) {
  if (!env) return;
  (void)hipSetDevice(env->cfg.device);
  (void)hipStreamSynchronize(env->stream);
  for (auto& pr : env->prof_events) {
    (void)hipEventDestroy(pr.a);
    (void)hipEventDestroy(pr.b);
  }
  for (hipEvent_t e : env->prof_pool) (void)hipEventDestroy(e);
  if (env->ev_stage) (void)hipHostFree(env->ev_stage);
  if (env->off_stage) (void)hipHostFree(env->off_stage);
  if (env->hi.in) (void)hipStreamSynchronize(env->hi.in);
  if (env->hi.out) (void)hipStreamSynchronize(env->hi.out);
  if (env->mods_flag_host) (void)hipHostFree(env->mods_flag_host);
  for (auto& r : env->hi.retired) (void)hipHostFree(r.pin);
  env->hi.retired.clear();
  for (int i = 0; i < bk_env::HostIngress::SLOTS; ++i) {
    if (env->hi.pin[i]) (void)hipHostFree(env->hi.pin[i]);
    if (env->hi.dev[i]) (void)hipFree(env->hi.dev[i]);
    for (hipEvent_t e : {env->hi.e_h2d[i], env->hi.e_ing[i], env->hi.e_done[i]})
      if (e) (void)hipEventDestroy(e);
  }
  if (env->hi.in) (void)hipStreamDestroy(env->hi.in);
  if (env->hi.out) (void)hipStreamDestroy(env->hi.out);
  if (env->ev_fork) (void)hipEventDestroy(env->ev_fork);
  for (int i = 0; i < bk_env::MAX_PARTS; ++i) {
    if (env->part_stream[i]) (void)hipStreamSynchronize(env->part_stream[i]);  // shared (part_streams): not destroyed
    if (env->ev_first[i]) (void)hipEventDestroy(env->ev_first[i]);
    if (env->ev_join[i]) (void)hipEventDestroy(env->ev_join[i]);
  }
  delete env;
}

int bk_env_set_stream(bk_env* env, void* hip_stream) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  env->stream = static_cast<hipStream_t>(hip_stream);
  return BK_OK;
}

int bk_env_sync(bk_env* env) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  return BK_OK;
}

// ------------------------------------------------------------------ host-driven order flow
int bk_place_order(bk_env* env, uint32_t book, int bid, uint32_t vol, uint32_t trader_id, int has_price,
                   uint32_t price, uint64_t* out_order_id) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = host_flow_ok(env)) return rc;
  const uint32_t asset = book % env->M, tick = env->asset_tick[asset];
  if (has_price && price % tick != 0)  // create_order's tick check, orderbook.rs:367-382
    return fail(BK_PRICE_NOT_TICK_MULTIPLE,
                "Price " + std::to_string(price) + " was not a multiple of tick-size " + std::to_string(tick));
  BookHost& bh = env->books[book];
  BookHost& qh = env->books[book - asset];  // the queue is the market's (market_env.rs:58)
  const uint64_t id = bh.orders.size();  // current_order_id, orderbook.rs:327-329
  if (id >= 0xFFFFFFFFull) return fail(BK_CAPACITY, "order id space exhausted");
  const uint32_t p = has_price ? price : (bid ? 0xFFFFFFFFu : 0u);  // market sentinels, types.rs:168,221
  const uint64_t now = env->cfg.start_time + env->steps_done * env->cfg.step_size + bh.time_offset;
  bh.orders.push_back(HostOrder{static_cast<uint8_t>(bid ? 1 : 0), vol, p, trader_id, now});
  qh.queue.push_back(HostEvent{0u | (bid ? 1u << 8 : 0u) | (asset << 16), static_cast<uint32_t>(id), p, vol});
  if (out_order_id) *out_order_id = id;
  return BK_OK;
}

int bk_cancel_order(bk_env* env, uint32_t book, uint64_t order_id) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = host_flow_ok(env)) return rc;
  // an id that was never created panics only when the event is PROCESSED (orderbook.rs:642): checked in bk_step
  const uint32_t asset = book % env->M;
  env->books[book - asset].queue.push_back(
      HostEvent{1u | (asset << 16), static_cast<uint32_t>(std::min<uint64_t>(order_id, 0xFFFFFFFFull)), 0u, 0u});
  return BK_OK;
}

int bk_modify_order(bk_env* env, uint32_t book, uint64_t order_id, int has_price, uint32_t new_price, int has_vol,
                    uint32_t new_vol) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = host_flow_ok(env)) return rc;
  const uint32_t asset = book % env->M;
  const uint32_t w = 2u | (has_price ? 1u << 9 : 0u) | (has_vol ? 1u << 10 : 0u) | (asset << 16);
  env->ev_mods_seen.store(true, std::memory_order_relaxed);  // (launch_events: the k_step_events form with the keyed modifications)
  env->books[book - asset].queue.push_back(
      HostEvent{w, static_cast<uint32_t>(std::min<uint64_t>(order_id, 0xFFFFFFFFull)), new_price, new_vol});
  return BK_OK;
}

int bk_submit_instructions(bk_env* env, uint32_t book, size_t n, const uint32_t* action, const uint8_t* side,
                           const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                           const uint64_t* order_id, uint64_t* out_ids, size_t* n_done) {
  if (int rc = check_book(env, book)) return rc;
  if (n_done) *n_done = 0;
  if (n && (!action || !side || !vol || !trader_id || !price || !order_id))
    return fail(BK_INVALID_ARGUMENT, "null instruction array");
  for (size_t i = 0; i < n; ++i) {
    uint64_t id = ~0ull;  // usize::MAX for non-new instructions (step_sim_numpy.rs:255-268)
    if (action[i] == 1) {
      // (bit 0 is the side on EVERY entry - k_ingest reads `side & 1`; bits 1 / 2 belong to BK_ACTION_MODIFY: ADVICE r5)
      int rc = bk_place_order(env, book, (side[i] & 1) != 0, vol[i], trader_id[i], 1, price[i], &id);
      if (rc != BK_OK) return rc;  // earlier elements stay created and queued (:167-177)
    } else if (action[i] == 2) {
      int rc = bk_cancel_order(env, book, order_id[i]);
      if (rc != BK_OK) return rc;
    } else if (action[i] == BK_ACTION_MODIFY) {  // the extension both entries share (k_ingest); anything else: no-op (:266)
      int rc = bk_modify_order(env, book, order_id[i], (side[i] & 2) != 0, price[i], (side[i] & 4) != 0, vol[i]);
      if (rc != BK_OK) return rc;
    }
    if (out_ids) out_ids[i] = id;
    if (n_done) *n_done = i + 1;
  }
  return BK_OK;
}

// The same for EVERY book in one call (SURVEY §8b "batched SoA ops"): the instructions of book b are elements
// [book_offsets[b], book_offsets[b + 1]) of the arrays.  Semantics per book as bk_submit_instructions; stops at the first
// bad price of any book (*n_done = index of the offending element, earlier elements stay queued).
int bk_submit_instructions_csr(bk_env* env, const uint64_t* book_offsets, const uint32_t* action, const uint8_t* side,
                               const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                               const uint64_t* order_id, uint64_t* out_ids, size_t* n_done) {
  if (!env || !book_offsets) return fail(BK_INVALID_ARGUMENT, "null argument");
  const uint32_t B = env->cfg.n_books, M = env->M, NM = B / M;
  if (n_done) *n_done = 0;
  if (book_offsets[B] > book_offsets[0] && (!action || !side || !vol || !trader_id || !price || !order_id))
    return fail(BK_INVALID_ARGUMENT, "null instruction array");
  for (uint32_t b = 0; b < B; ++b)
    if (book_offsets[b + 1] < book_offsets[b]) return fail(BK_INVALID_ARGUMENT, "book_offsets must be non-decreasing");
  // The host half of Env (tick check, id assignment, queueing) is independent per book (per market: the queue is the
  // market's): large batches are spread over host threads, each owning a contiguous range of markets.
  auto work = [&](uint32_t m_lo, uint32_t m_hi, int* rc_out, size_t* done_out, std::string* err) {
    *rc_out = BK_OK;
    for (uint32_t b = m_lo * M; b < m_hi * M; ++b) {
      const uint64_t lo = book_offsets[b], hi = book_offsets[b + 1];
      if (hi == lo) continue;
      size_t done = 0;
      const int rc = bk_submit_instructions(env, b, hi - lo, action + lo, side + lo, vol + lo, trader_id + lo, price + lo,
                                            order_id + lo, out_ids ? out_ids + lo : nullptr, &done);
      if (rc != BK_OK) {
        *rc_out = rc;
        *done_out = lo + done;
        *err = bk_last_error();  // thread-local in the worker
        return;
      }
    }
    *done_out = book_offsets[static_cast<size_t>(m_hi) * M];
  };
  const uint64_t total = book_offsets[B];
  unsigned nt = 1;
  if (total >= 65536 && NM >= 64) nt = std::min<unsigned>(env->host_pool().threads(), NM);
  std::vector<int> rcs(nt, BK_OK);
  std::vector<size_t> dones(nt, 0);
  std::vector<std::string> errs(nt);
  auto task = [&](unsigned t) {
    work(static_cast<uint32_t>(static_cast<uint64_t>(NM) * t / nt),
         static_cast<uint32_t>(static_cast<uint64_t>(NM) * (t + 1) / nt), &rcs[t], &dones[t], &errs[t]);
  };
  if (nt == 1)
    task(0);
  else if (!env->host_pool().run(nt, task))
    return fail(BK_CAPACITY, "host pool: too many tasks");
  for (unsigned t = 0; t < nt; ++t) {  // the first failing range in book order decides (books after it may be queued too)
    if (rcs[t] != BK_OK) {
      if (n_done) *n_done = dones[t];
      return fail(rcs[t], errs[t]);
    }
  }
  if (n_done) *n_done = total;
  return BK_OK;
}

int bk_enable_trading(bk_env* env, int enabled) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  env->trading = enabled ? 1 : 0;
  const uint32_t B = env->cfg.n_books;
  hipLaunchKernelGGL(k_book_service, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, B, 1,
                     env->trading);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

int bk_step(bk_env* env) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (env->device_ingress) {  // the queues are on the device already: launch, then wait as this entry always has
    if (int rc = bk_step_async(env)) return rc;
    HIPCHK(hipStreamSynchronize(env->stream));
    return BK_OK;
  }
  if (int rc = host_flow_ok(env)) return rc;
  if (int rc = use_device(env)) return rc;
  const size_t B = env->cfg.n_books, M = env->M, NM = B / M;
  // CSR offsets of the queues (one row per market; a market of one book when assets == 1)
  std::vector<uint32_t> off(NM + 1, 0u);
  size_t total = 0;
  uint32_t max_queue = 0;
  for (size_t m = 0; m < NM; ++m) {
    const size_t q = env->books[m * M].queue.size();
    if (q > EV_LDS_CAP) return fail(BK_CAPACITY, "more than 8192 events queued for one book (market) in one step");
    off[m] = static_cast<uint32_t>(total);
    total += q;
    max_queue = std::max(max_queue, static_cast<uint32_t>(q));
  }
  off[NM] = static_cast<uint32_t>(total);
  HIPCHK(hipStreamSynchronize(env->stream));  // previous step may still read the event buffers
  if (total > env->ev_capacity || !env->off_stage) {
    const size_t cap = std::max<size_t>(total * 2, 1024);
    HIPCHK(env->ev.alloc(cap));
    if (env->ev_stage) HIPCHK(hipHostFree(env->ev_stage));
    env->ev_stage = nullptr;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&env->ev_stage), cap * sizeof(HostEvent), hipHostMallocDefault));
    if (!env->off_stage)
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&env->off_stage), (B + 1) * sizeof(uint32_t), hipHostMallocDefault));
    env->ev_capacity = cap;
  }
  // validate (an id that was never created: orderbook.rs:642) and flatten straight into pinned memory: one 16-byte
  // record per event, uploaded at link speed.  Markets are independent: large batches are spread over the host threads.
  unsigned nt = 1;
  if (total >= 32768 && NM >= 64) nt = std::min<unsigned>(env->host_pool().threads(), static_cast<unsigned>(NM));
  std::vector<size_t> bad_market(nt, NM);
  std::vector<uint32_t> bad_id(nt, 0u);
  auto task = [&](unsigned t) {
    const size_t m_lo = NM * t / nt, m_hi = NM * (t + 1) / nt;
    for (size_t m = m_lo; m < m_hi; ++m) {
      const std::vector<HostEvent>& q = env->books[m * M].queue;
      for (const HostEvent& e : q)
        if ((e.word & 0xFFu) != 0 && e.id >= env->books[m * M + ((e.word >> 16) & 0xFFu)].orders.size()) {
          bad_market[t] = m;
          bad_id[t] = e.id;
          return;
        }
      if (!q.empty()) std::memcpy(env->ev_stage + off[m], q.data(), q.size() * sizeof(HostEvent));
    }
  };
  if (nt == 1)
    task(0);
  else if (!env->host_pool().run(nt, task))
    return fail(BK_CAPACITY, "host pool: too many tasks");
  for (unsigned t = 0; t < nt; ++t)  // the first offender in market order; nothing has been uploaded or cleared
    if (bad_market[t] != NM) return fail(BK_UNKNOWN_ORDER_ID, "No order with id " + std::to_string(bad_id[t]) + " exists");
  std::memcpy(env->off_stage, off.data(), (NM + 1) * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(env->ev_off.p, env->off_stage, (NM + 1) * 4, hipMemcpyHostToDevice, env->stream));
  if (total)
    HIPCHK(hipMemcpyAsync(env->ev.p, env->ev_stage, total * sizeof(HostEvent), hipMemcpyHostToDevice, env->stream));
  const DevArgs a = env->args();
  int rc = BK_OK;
  switch (env->R) {
    case 1: rc = launch_events<1>(env, a, env->steps_done, max_queue); break;
    case 2: rc = launch_events<2>(env, a, env->steps_done, max_queue); break;
    case 4: rc = launch_events<4>(env, a, env->steps_done, max_queue); break;
    default: rc = launch_events<8>(env, a, env->steps_done, max_queue); break;
  }
  if (rc != BK_OK) return rc;
  env->steps_done += 1;
  for (size_t b = 0; b < B; ++b) {
    BookHost& bh = env->books[b];
    bh.queue.clear();
    bh.n_uploaded = bh.orders.size();
    bh.log_fresh = false;
  }
  HIPCHK(hipStreamSynchronize(env->stream));
  return BK_OK;
}

// ------------------------------------------------------------------ device-resident instruction ingress
// (rust/src/step_sim_numpy.rs:233-275 submit_instructions + env.rs:166-219, for every book, with no host in the loop)
int bk_device_ingress_enable(bk_env* env, uint32_t queue_capacity) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (env->device_ingress) {
    if (queue_capacity != env->qcap) return fail(BK_INVALID_ARGUMENT, "device ingress is already enabled with another queue capacity");
    return BK_OK;
  }
  if (queue_capacity == 0 || queue_capacity > EV_LDS_CAP)
    return fail(BK_INVALID_ARGUMENT, "queue_capacity must be in 1..8192 events per book (market) and step");
  if (env->device_flow || env->n_mixed || !env->groups.empty())
    return fail(BK_INVALID_ARGUMENT, "an env runs ONE order flow: on-device agents (bk_run) or submitted instructions");
  for (const BookHost& bh : env->books)
    if (!bh.queue.empty() || !bh.orders.empty())
      return fail(BK_INVALID_ARGUMENT, "enable the device ingress before any order is placed through the host entries");
  if (int rc = use_device(env)) return rc;
  const size_t NM = env->cfg.n_books / env->M;
  HIPCHK(env->dq.alloc(NM * queue_capacity));
  HIPCHK(env->dqlen.alloc(NM));
  HIPCHK(hipMemsetAsync(env->dqlen.p, 0, NM * 4, env->stream));
  HIPCHK(env->dorders.alloc(static_cast<size_t>(env->cfg.n_books) * env->cfg.max_orders * 2));
  // k_ingest's "a modification was submitted" hint: one word of mapped host memory (launch_events reads it without a copy)
  if (!env->mods_flag_host) {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
      *static_cast<uint32_t*>(hp) = 0u;
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
        env->mods_flag_host = static_cast<uint32_t*>(hp);
        env->mods_flag_dev = static_cast<uint32_t*>(dp);
      } else {
        (void)hipHostFree(hp);
      }
    }
    (void)hipGetLastError();
    if (!env->mods_flag_host) env->ev_mods_seen.store(true);  // (no hint available: always the full kernel)
  }
  env->qcap = queue_capacity;
  env->device_ingress = true;
  return BK_OK;
}

int bk_submit_instructions_device(bk_env* env, const uint64_t* book_offsets_dev, const uint32_t* action_dev, const uint8_t* side_dev,
                                  const uint32_t* vol_dev, const uint32_t* trader_dev, const uint32_t* price_dev,
                                  const uint64_t* order_id_dev, uint64_t* out_ids_dev, uint32_t* status_dev) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (!env->device_ingress) return fail(BK_INVALID_ARGUMENT, "call bk_device_ingress_enable first");
  if (!book_offsets_dev || !action_dev || !side_dev || !vol_dev || !trader_dev || !price_dev || !order_id_dev)
    return fail(BK_INVALID_ARGUMENT, "null instruction array");
  if (int rc = use_device(env)) return rc;
  IngestArgs g{};
  g.off = reinterpret_cast<const unsigned long long*>(book_offsets_dev);
  g.action = action_dev;
  g.side = side_dev;
  g.vol = vol_dev;
  g.trader = trader_dev;
  g.price = price_dev;
  g.order_id = reinterpret_cast<const unsigned long long*>(order_id_dev);
  g.out_ids = reinterpret_cast<unsigned long long*>(out_ids_dev);
  g.status = status_dev;
  g.q = env->dq.p;
  g.qlen = env->dqlen.p;
  g.qcap = env->qcap;
  g.dorders = env->dorders.p;
  g.mods_flag = env->mods_flag_dev;
  const DevArgs a = env->args();
  hipLaunchKernelGGL(k_ingest, dim3(env->cfg.n_books / env->M), dim3(64), 0, env->stream, a, g);
  HIPCHK(hipGetLastError());
  env->ingest_epoch += 1;
  return BK_OK;
}

// ==================================================================================
// HOST arrays through the device ingress (round 5; VERDICT r4 "missing" #2).  A `BaseNumpyAgent`-style caller
// (src/bourse/step_sim/agents/base_agent.py:67-116, runner.py:103-112) hands over HOST numpy arrays; rounds 1-4 walked them
// through the host half of Env (tick check, ids and queues on CPU threads, 16 B per event uploaded: 3.2 M book-steps/s at
// 8 192 books x 48 instructions).  Here the SAME six arrays (rust/src/step_sim_numpy.rs:233-275) go pinned staging ->
// one async upload per array on a copy stream -> k_ingest on the env's stream -> ids / per-book status back on a second copy
// stream.  Two slots: the upload of ticket t + 1 runs under the step kernel of ticket t; a ticket's results stay
// readable until two more submits.  Same per-book semantics as every other submit entry (k_ingest: dense ids in element
// order, a book's batch stops at its first bad price with the earlier elements applied, other books unaffected).
// ==================================================================================
namespace {
int hi_ensure(bk_env* env, size_t n_elem) {
  bk_env::HostIngress& h = env->hi;
  if (!h.in) {
    HIPCHK(hipStreamCreateWithFlags(&h.in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h.out, hipStreamNonBlocking));
    for (int i = 0; i < bk_env::HostIngress::SLOTS; ++i) {
      HIPCHK(hipEventCreateWithFlags(&h.e_h2d[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&h.e_ing[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&h.e_done[i], hipEventDisableTiming));
    }
  }
  if (n_elem <= h.cap && h.pin[0] && h.pin[1]) return BK_OK;
  // grow: everything in flight must have landed first; the results the slots hold (ids + status of the last two tickets - a
  // caller that fetches one submit late has not read them yet) move into the new staging
  HIPCHK(hipStreamSynchronize(h.in));
  HIPCHK(hipStreamSynchronize(env->stream));
  HIPCHK(hipStreamSynchronize(h.out));
  const size_t cap = std::max<size_t>((n_elem + n_elem / 4 + 1023) & ~size_t(1023), 1024);
  // BOTH new slots are allocated before anything of the old staging is given up: a failed allocation leaves the env as it was
  bk_env::HostIngress lay = h;
  lay.layout(env->cfg.n_books, cap);
  char *npin[bk_env::HostIngress::SLOTS] = {}, *ndev[bk_env::HostIngress::SLOTS] = {};
  for (int i = 0; i < bk_env::HostIngress::SLOTS; ++i) {
    if (hipHostMalloc(reinterpret_cast<void**>(&npin[i]), lay.bytes, hipHostMallocDefault) != hipSuccess) npin[i] = nullptr;
    if (npin[i] && hipMalloc(reinterpret_cast<void**>(&ndev[i]), lay.bytes) != hipSuccess) ndev[i] = nullptr;
    if (!npin[i] || !ndev[i]) {
      (void)hipGetLastError();
      for (int j = 0; j <= i; ++j) {
        if (npin[j]) (void)hipHostFree(npin[j]);
        if (ndev[j]) (void)hipFree(ndev[j]);
      }
      return fail(BK_CAPACITY, "bk_submit_instructions_host: out of pinned / device memory for the staging");
    }
  }
  const bk_env::HostIngress old = h;
  h.layout(env->cfg.n_books, cap);
  for (int i = 0; i < bk_env::HostIngress::SLOTS; ++i) {
    h.pin[i] = npin[i];
    h.dev[i] = ndev[i];
    if (old.pin[i] && old.ticket_of[i] != ~0ull) {
      std::memcpy(h.pin[i] + h.o_out, old.pin[i] + old.o_out, old.n_elem[i] * 8);
      std::memcpy(h.pin[i] + h.o_st, old.pin[i] + old.o_st, static_cast<size_t>(env->cfg.n_books) * 8);
    }
    // a view into the old pinned block (of a ticket < next_ticket) is valid until two more submits: the block goes when the
    // submit of ticket next_ticket + 1 starts.  The device block has no outside references (every stream was drained above).
    if (old.pin[i]) h.retired.push_back({old.pin[i], h.next_ticket + 1});
    if (old.dev[i]) HIPCHK(hipFree(old.dev[i]));
  }
  return BK_OK;
}
void hi_release_retired(bk_env* env, bool all) {
  auto& r = env->hi.retired;
  for (size_t i = 0; i < r.size();) {
    if (all || env->hi.next_ticket >= r[i].free_from_ticket) {
      (void)hipHostFree(r[i].pin);
      r[i] = r.back();
      r.pop_back();
    } else {
      ++i;
    }
  }
}

// The staging copies: the calling thread alone below 1 MB, else FOUR threads of the env's pool claiming 256 KB pieces one at a
// time.  Measured on three boxes at 9.8 MB per step (scripts/host_driven_rate.py, `tickets`): one thread 0.30 - 0.40 ms (the copy
// then bounds the step: 15 M book-steps/s), sixteen threads 0.19 ms on a quiet box (24.6 M) but 0.9 ms on a busy one (a parallel
// copy is as fast as its slowest worker gets scheduled, and sixteen runnable threads meet the container's CPU quota), four
// threads with pieces claimed dynamically: see docs/EXPERIMENTS.md.
struct HiSeg {
  void* dst;
  const void* src;
  size_t bytes;
};
void hi_copy(bk_env* env, const HiSeg* segs, int n_segs) {
  constexpr size_t PIECE = 256u << 10, PARALLEL_FROM = 1u << 20;
  size_t total = 0;
  for (int i = 0; i < n_segs; ++i)
    if (segs[i].dst != segs[i].src) total += segs[i].bytes;  // (dst == src: the caller filled the staging array in place)
  if (total < PARALLEL_FROM) {
    for (int i = 0; i < n_segs; ++i)
      if (segs[i].dst != segs[i].src && segs[i].bytes) std::memcpy(segs[i].dst, segs[i].src, segs[i].bytes);
    return;
  }
  struct Piece {
    char* d;
    const char* s;
    size_t n;
  };
  std::vector<Piece> pieces;
  for (int i = 0; i < n_segs; ++i) {
    const HiSeg& g = segs[i];
    if (g.dst == g.src) continue;
    for (size_t o = 0; o < g.bytes; o += PIECE)
      pieces.push_back({static_cast<char*>(g.dst) + o, static_cast<const char*>(g.src) + o, std::min(PIECE, g.bytes - o)});
  }
  std::atomic<size_t> next{0};
  auto worker = [&](unsigned) {
    for (size_t k = next.fetch_add(1); k < pieces.size(); k = next.fetch_add(1)) std::memcpy(pieces[k].d, pieces[k].s, pieces[k].n);
  };
  const unsigned nt = std::min<unsigned>(4u, env->host_pool().threads());
  if (nt <= 1 || !env->host_pool().run(nt, worker)) worker(0);
}
}  // namespace

int bk_ingress_staging(bk_env* env, uint64_t min_elements, bk_ingress_arrays* out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (!env->device_ingress) return fail(BK_INVALID_ARGUMENT, "call bk_device_ingress_enable first");
  if (int rc = use_device(env)) return rc;
  if (int rc = hi_ensure(env, min_elements)) return rc;
  bk_env::HostIngress& h = env->hi;
  const int s = static_cast<int>(h.next_ticket % bk_env::HostIngress::SLOTS);
  // the slot's previous ticket (two submits ago) must have been consumed before the caller writes into its arrays
  if (h.ticket_of[s] != ~0ull) HIPCHK(hipEventSynchronize(h.e_done[s]));
  char* p = h.pin[s];
  out->capacity = h.cap;
  out->book_offsets = reinterpret_cast<uint64_t*>(p + h.o_off);
  out->action = reinterpret_cast<uint32_t*>(p + h.o_act);
  out->side = reinterpret_cast<uint8_t*>(p + h.o_side);
  out->vol = reinterpret_cast<uint32_t*>(p + h.o_vol);
  out->trader_id = reinterpret_cast<uint32_t*>(p + h.o_trd);
  out->price = reinterpret_cast<uint32_t*>(p + h.o_prc);
  out->order_id = reinterpret_cast<uint64_t*>(p + h.o_oid);
  return BK_OK;
}

int bk_submit_instructions_host(bk_env* env, const uint64_t* book_offsets, const uint32_t* action, const uint8_t* side,
                                const uint32_t* vol, const uint32_t* trader_id, const uint32_t* price,
                                const uint64_t* order_id, uint64_t* out_ticket) {
  if (!env || !book_offsets) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (!env->device_ingress) return fail(BK_INVALID_ARGUMENT, "call bk_device_ingress_enable first");
  const size_t B = env->cfg.n_books;
  const uint64_t n = book_offsets[B];
  if (book_offsets[0] != 0) return fail(BK_INVALID_ARGUMENT, "book_offsets[0] must be 0");
  for (size_t b = 0; b < B; ++b)
    if (book_offsets[b + 1] < book_offsets[b]) return fail(BK_INVALID_ARGUMENT, "book_offsets must be non-decreasing");
  if (n && (!action || !side || !vol || !trader_id || !price || !order_id))
    return fail(BK_INVALID_ARGUMENT, "null instruction array");
  if (int rc = use_device(env)) return rc;
  if (int rc = hi_ensure(env, n)) return rc;
  hi_release_retired(env, false);  // (pinned blocks replaced by a growth two submits ago: their tickets' views have expired)
  bk_env::HostIngress& h = env->hi;
  const uint64_t ticket = h.next_ticket;
  const int s = static_cast<int>(ticket % bk_env::HostIngress::SLOTS);
  if (h.ticket_of[s] != ~0ull) HIPCHK(hipEventSynchronize(h.e_done[s]));  // slot free: its upload, ingest and download are over
  char *pp = h.pin[s], *dd = h.dev[s];
  const HiSeg segs[7] = {{pp + h.o_off, book_offsets, (B + 1) * 8}, {pp + h.o_oid, order_id, n * 8}, {pp + h.o_act, action, n * 4},
                         {pp + h.o_vol, vol, n * 4},     {pp + h.o_trd, trader_id, n * 4}, {pp + h.o_prc, price, n * 4},
                         {pp + h.o_side, side, n}};
  auto up = [&](size_t off, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(dd + off, pp + off, bytes, hipMemcpyHostToDevice, h.in) : hipSuccess;
  };
  // Large batches in three groups of ~a third of the bytes, each uploaded as soon as it is staged: the link works on group g
  // while the host copies group g + 1 (a caller that fetches the ids before every step - the reference's call shape - waits
  // for copy + upload + k_ingest + download in sequence: 0.25 + 0.2 ms of it at 8 192 books x 48 instructions were these two).
  const char* one_pass = getenv("BOURSE_AMD_HI_ONE_PASS");  // (=1: stage everything, then upload - for measurements)
  const bool grouped = n * 27 >= (3u << 20) && !(one_pass && *one_pass == '1');
  if (!grouped) hi_copy(env, segs, 7);
  if (grouped) hi_copy(env, segs + 0, 2);
  HIPCHK(up(h.o_off, (B + 1) * 8));
  HIPCHK(up(h.o_oid, n * 8));
  if (grouped) hi_copy(env, segs + 2, 2);
  HIPCHK(up(h.o_act, n * 4));
  HIPCHK(up(h.o_vol, n * 4));
  if (grouped) hi_copy(env, segs + 4, 3);
  HIPCHK(up(h.o_trd, n * 4));
  HIPCHK(up(h.o_prc, n * 4));
  HIPCHK(up(h.o_side, n));
  HIPCHK(hipEventRecord(h.e_h2d[s], h.in));
  HIPCHK(hipStreamWaitEvent(env->stream, h.e_h2d[s], 0));
  if (n) {
    // (k_ingest leaves out_ids alone from a book's failing element on: those read u64::MAX here, not a stale ticket's ids)
    HIPCHK(hipMemsetAsync(dd + h.o_out, 0xFF, n * 8, env->stream));
    if (int rc = bk_submit_instructions_device(
            env, reinterpret_cast<const uint64_t*>(dd + h.o_off), reinterpret_cast<const uint32_t*>(dd + h.o_act),
            reinterpret_cast<const uint8_t*>(dd + h.o_side), reinterpret_cast<const uint32_t*>(dd + h.o_vol),
            reinterpret_cast<const uint32_t*>(dd + h.o_trd), reinterpret_cast<const uint32_t*>(dd + h.o_prc),
            reinterpret_cast<const uint64_t*>(dd + h.o_oid), reinterpret_cast<uint64_t*>(dd + h.o_out),
            reinterpret_cast<uint32_t*>(dd + h.o_st)))
      return rc;
  } else {
    HIPCHK(hipMemsetAsync(dd + h.o_st, 0, B * 8, env->stream));  // no element for any book: every status {OK, 0}
  }
  HIPCHK(hipEventRecord(h.e_ing[s], env->stream));
  HIPCHK(hipStreamWaitEvent(h.out, h.e_ing[s], 0));
  if (n) HIPCHK(hipMemcpyAsync(pp + h.o_out, dd + h.o_out, n * 8, hipMemcpyDeviceToHost, h.out));
  HIPCHK(hipMemcpyAsync(pp + h.o_st, dd + h.o_st, B * 8, hipMemcpyDeviceToHost, h.out));
  HIPCHK(hipEventRecord(h.e_done[s], h.out));
  h.ticket_of[s] = ticket;
  h.n_elem[s] = n;
  h.next_ticket = ticket + 1;
  if (out_ticket) *out_ticket = ticket;
  return BK_OK;
}

int bk_submit_result_view(bk_env* env, uint64_t ticket, const uint64_t** out_ids, const uint32_t** status,
                          uint32_t* first_failed_book) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  bk_env::HostIngress& h = env->hi;
  const int s = static_cast<int>(ticket % bk_env::HostIngress::SLOTS);
  if (ticket >= h.next_ticket || h.ticket_of[s] != ticket)
    return fail(BK_INVALID_ARGUMENT, "unknown or expired ticket (a ticket's results stay readable until two more submits)");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipEventSynchronize(h.e_done[s]));
  const size_t B = env->cfg.n_books;
  const uint32_t* st = reinterpret_cast<const uint32_t*>(h.pin[s] + h.o_st);
  if (out_ids) *out_ids = reinterpret_cast<const uint64_t*>(h.pin[s] + h.o_out);
  if (status) *status = st;
  if (first_failed_book) {
    *first_failed_book = 0xFFFFFFFFu;
    for (size_t b = 0; b < B; ++b)
      if (st[2 * b] != BK_OK) {
        *first_failed_book = static_cast<uint32_t>(b);
        break;
      }
  }
  return BK_OK;
}

int bk_submit_result(bk_env* env, uint64_t ticket, uint64_t* out_ids, uint32_t* status, uint32_t* first_failed_book) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  bk_env::HostIngress& h = env->hi;
  const int s = static_cast<int>(ticket % bk_env::HostIngress::SLOTS);
  if (ticket >= h.next_ticket || h.ticket_of[s] != ticket)
    return fail(BK_INVALID_ARGUMENT, "unknown or expired ticket (a ticket's results stay readable until two more submits)");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipEventSynchronize(h.e_done[s]));
  const size_t B = env->cfg.n_books;
  const uint32_t* st = reinterpret_cast<const uint32_t*>(h.pin[s] + h.o_st);
  if (out_ids) {
    const HiSeg seg{out_ids, h.pin[s] + h.o_out, h.n_elem[s] * 8};
    hi_copy(env, &seg, 1);
  }
  if (status) std::memcpy(status, st, B * 8);
  if (first_failed_book) {
    *first_failed_book = 0xFFFFFFFFu;
    for (size_t b = 0; b < B; ++b)
      if (st[2 * b] != BK_OK) {
        *first_failed_book = static_cast<uint32_t>(b);
        break;
      }
  }
  return BK_OK;
}

// Env::step over the device-resident queues, asynchronous on the env's stream (bk_step = this + a wait)
int bk_step_async(bk_env* env) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (!env->device_ingress) return fail(BK_INVALID_ARGUMENT, "bk_step_async steps the device-resident queues: call bk_device_ingress_enable first");
  if (int rc = use_device(env)) return rc;
  const DevArgs a = env->args();
  int rc = BK_OK;
  // (the shuffle permutation's LDS is sized for a full queue: the host does not know the queues' lengths - nothing of this
  // step has been on the host)
  switch (env->R) {
    case 1: rc = launch_events<1>(env, a, env->steps_done, env->qcap); break;
    case 2: rc = launch_events<2>(env, a, env->steps_done, env->qcap); break;
    case 4: rc = launch_events<4>(env, a, env->steps_done, env->qcap); break;
    default: rc = launch_events<8>(env, a, env->steps_done, env->qcap); break;
  }
  if (rc != BK_OK) return rc;
  HIPCHK(hipMemsetAsync(env->dqlen.p, 0, static_cast<size_t>(env->cfg.n_books / env->M) * 4, env->stream));
  env->steps_done += 1;
  env->ingest_epoch += 1;
  return BK_OK;
}

int bk_order_status(bk_env* env, uint32_t book, uint64_t order_id, uint8_t* out_status) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = mirror_orders(env, book)) return rc;
  BookHost& bh = env->books[book];
  if (order_id >= bh.orders.size())
    return fail(BK_UNKNOWN_ORDER_ID, "No order with id " + std::to_string(order_id) + " exists");
  if (env->cfg.max_orders == 0) return fail(BK_INVALID_ARGUMENT, "order log disabled (max_orders == 0)");
  if (order_id >= env->cfg.max_orders) return fail(BK_CAPACITY, "order id beyond the order-log capacity");
  if (int rc = use_device(env)) return rc;
  if (int rc = refresh_log(env, book)) return rc;
  bk_order o;
  fill_order(env, bh, order_id, &o);
  if (out_status) *out_status = o.status;
  return BK_OK;
}

int bk_order_count(bk_env* env, uint32_t book, uint64_t* out) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = mirror_orders(env, book)) return rc;
  if (out) *out = env->books[book].orders.size();
  return BK_OK;
}

int bk_get_orders(bk_env* env, uint32_t book, uint64_t first, uint64_t n, bk_order* out) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = mirror_orders(env, book)) return rc;
  BookHost& bh = env->books[book];
  if (first > bh.orders.size() || n > bh.orders.size() - first) return fail(BK_INVALID_ARGUMENT, "order range out of bounds");
  if (n && !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (bh.orders.size() > env->cfg.max_orders && env->cfg.max_orders > 0 && bh.n_uploaded > env->cfg.max_orders)
    return fail(BK_CAPACITY, "order log capacity exceeded");
  if (int rc = use_device(env)) return rc;
  if (env->cfg.max_orders)
    if (int rc = refresh_log(env, book)) return rc;
  for (uint64_t i = 0; i < n; ++i) fill_order(env, bh, first + i, out + i);
  return BK_OK;
}

// ------------------------------------------------------------------ on-device order flow
int bk_set_random_market_agents(bk_env* env, uint32_t n_groups, const bk_random_agents* groups,
                                const uint32_t* assets) {
  if (!env || (!groups && n_groups)) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (n_groups > MAX_GROUPS) return fail(BK_INVALID_ARGUMENT, "at most 8 agent groups");
  std::vector<Group> gs;
  uint64_t total = 0;
  for (uint32_t g = 0; g < n_groups; ++g) {
    const bk_random_agents& r = groups[g];
    if (r.tick_lo >= r.tick_hi || r.vol_lo >= r.vol_hi)
      return fail(BK_INVALID_ARGUMENT, "empty tick/vol range");  // gen_range asserts low < high
    // every sampled price tick * tick_size must pass create_order's tick check (else `.unwrap()` panics,
    // random_agent.rs:103-110)
    const uint32_t asset = assets ? assets[g] : 0u;
    if (asset >= env->M) return fail(BK_INVALID_ARGUMENT, "group asset index out of range");
    if (r.tick_size % env->asset_tick[asset] != 0)
      return fail(BK_PRICE_NOT_TICK_MULTIPLE, "agent tick_size must be a multiple of the env tick_size");
    if (static_cast<uint64_t>(r.tick_hi - 1) * r.tick_size >= 0xFFFFFFFFull || r.tick_lo == 0)
      return fail(BK_INVALID_ARGUMENT, "limit prices must lie in (0, u32::MAX)");
    Group G{};
    G.n = r.n_agents;
    G.thr = activity_threshold(r.activity_rate);
    G.tick_lo = r.tick_lo;
    G.tick_rng = r.tick_hi - r.tick_lo;
    G.tick_zone = sample_zone(G.tick_rng);
    G.vol_lo = r.vol_lo;
    G.vol_rng = r.vol_hi - r.vol_lo;
    G.vol_zone = sample_zone(G.vol_rng);
    G.tick_size = r.tick_size;
    G.asset = asset;
    total += r.n_agents;
    gs.push_back(G);
  }
  if (total > env->cfg.max_live_orders)
    return fail(BK_CAPACITY, "sum of n_agents exceeds max_live_orders (one pool slot per agent)");
  env->groups = gs;
  env->n_agents_total = static_cast<uint32_t>(total);
  env->n_mixed = 0;
  env->agents_hash = gs.empty() ? 0 : fnv1a(gs.data(), gs.size() * sizeof(Group));
  return BK_OK;
}

int bk_set_random_agents(bk_env* env, uint32_t n_groups, const bk_random_agents* groups) {
  return bk_set_random_market_agents(env, n_groups, groups, nullptr);  // every group on asset 0
}

int bk_set_tick_sizes(bk_env* env, uint32_t n, const uint32_t* tick_sizes) {
  if (!env || !tick_sizes) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (n != env->M) return fail(BK_INVALID_ARGUMENT, "one tick size per asset");
  if (env->steps_done || !env->groups.empty()) return fail(BK_INVALID_ARGUMENT, "set the tick sizes before anything else");
  for (const BookHost& bh : env->books)
    if (!bh.orders.empty()) return fail(BK_INVALID_ARGUMENT, "set the tick sizes before anything else");
  for (uint32_t i = 0; i < n; ++i)
    if (tick_sizes[i] == 0) return fail(BK_INVALID_ARGUMENT, "tick_size must be > 0");  // orderbook.rs:159
  for (uint32_t i = 0; i < n; ++i) env->asset_tick[i] = tick_sizes[i];
  return BK_OK;
}

static int set_agents_impl(bk_env* env, uint32_t n_members, const bk_agent_desc* members, const uint32_t* assets);

int bk_set_agents(bk_env* env, uint32_t n_members, const bk_agent_desc* members) {
  if (!env || (!members && n_members)) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (env->M > 1) return fail(BK_INVALID_ARGUMENT, "markets (assets > 1) take bk_set_market_agents");
  return set_agents_impl(env, n_members, members, nullptr);
}

// A MarketAgentSet of RandomMarketAgents / NoiseMarketAgent / MomentumMarketAgent members (random_agent.rs:164-247,
// noise_agent.rs:226-340, momentum_agent.rs:282-397): member i trades asset assets[i] of every market.
int bk_set_market_agents(bk_env* env, uint32_t n_members, const bk_agent_desc* members, const uint32_t* assets) {
  if (!env || (!members && n_members) || (!assets && n_members)) return fail(BK_INVALID_ARGUMENT, "null argument");
  for (uint32_t i = 0; i < n_members; ++i)
    if (assets[i] >= env->M) return fail(BK_INVALID_ARGUMENT, "member asset index out of range");
  return set_agents_impl(env, n_members, members, assets);
}

static int set_agents_impl(bk_env* env, uint32_t n_members, const bk_agent_desc* members, const uint32_t* assets) {
  bool all_random = true;
  for (uint32_t i = 0; i < n_members; ++i) all_random = all_random && members[i].type == BK_AGENT_RANDOM;
  if (all_random) {
    std::vector<bk_random_agents> g(n_members);
    for (uint32_t i = 0; i < n_members; ++i)
      g[i] = bk_random_agents{members[i].n_agents, members[i].tick_lo, members[i].tick_hi, members[i].vol_lo,
                              members[i].vol_hi, members[i].tick_size, members[i].activity_rate};
    return bk_set_random_market_agents(env, n_members, g.data(), assets);
  }
  if (n_members > MAX_MEMBERS) return fail(BK_INVALID_ARGUMENT, "at most 4 members in a set with Noise/Momentum agents");
  if (int rc = use_device(env)) return rc;
  std::vector<MixedDesc> ds(n_members);
  uint32_t fixed_a[MAX_ASSETS] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < n_members; ++i) {
    const uint32_t as = assets ? assets[i] : 0u;
    uint32_t& fixed = fixed_a[as];  // fixed RandomAgents slots are counted per book
    const bk_agent_desc& m = members[i];
    MixedDesc& D = ds[i];
    std::memset(&D, 0, sizeof(D));
    D.type = m.type;
    D.n = m.n_agents;
    if (m.tick_size == 0 || m.tick_size % env->asset_tick[as] != 0)
      return fail(BK_PRICE_NOT_TICK_MULTIPLE, "member tick_size must be a non-zero multiple of the env tick_size");
    if (m.type == BK_AGENT_RANDOM) {
      if (m.tick_lo >= m.tick_hi || m.vol_lo >= m.vol_hi || m.tick_lo == 0 ||
          static_cast<uint64_t>(m.tick_hi - 1) * m.tick_size >= 0xFFFFFFFFull)
        return fail(BK_INVALID_ARGUMENT, "bad RandomAgents ranges");
      D.thr = activity_threshold(m.activity_rate);
      D.tick_lo = m.tick_lo;
      D.tick_rng = m.tick_hi - m.tick_lo;
      D.tick_zone = sample_zone(D.tick_rng);
      D.vol_lo = m.vol_lo;
      D.vol_rng = m.vol_hi - m.vol_lo;
      D.vol_zone = sample_zone(D.vol_rng);
      D.tick_size = m.tick_size;
      D.slot_base = fixed;
      fixed += m.n_agents;
    } else if (m.type == BK_AGENT_NOISE || m.type == BK_AGENT_MOMENTUM) {
      if (m.n_agents > 0xFFFFu) return fail(BK_INVALID_ARGUMENT, "n_agents is a u16 in the reference");
      if (!(m.price_dist_sigma >= 0.0) || !std::isfinite(m.price_dist_sigma) || !std::isfinite(m.price_dist_mu))
        return fail(BK_INVALID_ARGUMENT, "LogNormal::new(mu, sigma) needs finite mu and sigma >= 0");  // .unwrap()
      D.thr_limit = activity_threshold(m.p_limit);
      D.thr_market = activity_threshold(m.p_market);
      D.keep_thr = keep_threshold(m.p_cancel);
      D.trade_vol = m.trade_vol;
      D.mu = m.price_dist_mu;
      D.sigma = m.price_dist_sigma;
      D.decay = m.decay;
      D.demand = m.demand;
      D.scale = m.scale;
      D.order_ratio = m.order_ratio;
      D.n_f = static_cast<double>(m.n_agents);
      D.tick_f = static_cast<double>(m.tick_size);
    } else {
      return fail(BK_INVALID_ARGUMENT, "unknown agent type");
    }
  }
  uint32_t fixed = fixed_a[0];
  for (uint32_t as = 0; as < env->M; ++as)
    if (fixed_a[as] >= env->cfg.max_live_orders)
      return fail(BK_CAPACITY, "RandomAgents members leave no pool slots for the other members' orders");
  HIPCHK(env->mixed_descs.alloc(n_members));
  HIPCHK(hipMemcpy(env->mixed_descs.p, ds.data(), ds.size() * sizeof(MixedDesc), hipMemcpyHostToDevice));
  env->n_mixed = n_members;
  env->n_fixed = fixed;
  env->wl_valid = false;
  for (uint32_t i = 0; i < MAX_MEMBERS; ++i) env->member_asset[i] = (assets && i < n_members) ? assets[i] : 0u;
  for (uint32_t as = 0; as < MAX_ASSETS; ++as) env->n_fixed_a[as] = fixed_a[as];
  env->ml_valid = false;
  env->groups.clear();
  env->n_agents_total = 0;
  env->agents_hash = fnv1a(env->member_asset, sizeof(env->member_asset), fnv1a(ds.data(), ds.size() * sizeof(MixedDesc)));
  return BK_OK;
}

// books one residency round of the fused wave kernel holds (the auto rule's `wave` limit): asked of the runtime once
static void query_fused_resident(bk_env* env) {
  if (env->fused_resident || !env->wave_ok()) return;
  if (hipSetDevice(env->cfg.device) != hipSuccess) return;
  int blocks = 0, cus = 0;
  hipError_t e = hipSuccess;
  switch (env->R) {
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_run_wave<1>, 512, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_run_wave<2>, 512, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_run_wave<4>, 512, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_run_wave<8>, 512, 0); break;
  }
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, env->cfg.device);
  if (e == hipSuccess && blocks > 0 && cus > 0) env->fused_resident = static_cast<uint32_t>(blocks) * 8u * static_cast<uint32_t>(cus);
  (void)hipGetLastError();
  if (getenv("BOURSE_AMD_VERBOSE"))
    fprintf(stderr, "bourse_amd: k_run_wave<%d>: %d workgroups per CU x %d CUs -> one round holds %u books\n", env->R, blocks, cus,
            env->fused_resident);
}

int bk_run(bk_env* env, uint64_t n_steps) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (n_steps == 0) return BK_OK;
  query_fused_resident(env);
  if (n_steps > 0xFFFFFFFFull) return fail(BK_INVALID_ARGUMENT, "n_steps too large for one launch");
  if (int rc = use_device(env)) return rc;
  if (env->device_ingress) return fail(BK_INVALID_ARGUMENT, "bk_run cannot be mixed with submitted instructions on the same env");
  for (const BookHost& bh : env->books)
    if (!bh.queue.empty() || !bh.orders.empty())
      return fail(BK_INVALID_ARGUMENT, "bk_run cannot be mixed with host-driven orders on the same env");
  DevArgs a = env->args();
  if (a.n_groups == 0) {  // no agents: plain steps
    a.groups[0] = Group{};
    a.n_agents_total = 0;
  }
  int rc = BK_OK;
  const uint32_t ns = static_cast<uint32_t>(n_steps);
  if (env->n_mixed || a.n_groups) env->device_flow = true;
  // Which kernels (bk_env::plan - the rule itself is documented there and in DESIGN.md 2.1):
  //  * the fused kernels keep a book in registers across all steps of the launch but run the RNG-serial phases on the
  //    scalar unit of ONE wave per book; the split forms move them to one lane per book (>= 64 books per wave to pay off)
  //    or decode the stream wave-parallel, in front of the lean event kernel, the batch cut in parts that overlap;
  //  * markets always take a split form: the market's RNG-serial phase is one lane, its books are M waves;
  //  * the lane-per-book members' update keeps a filled order's pool slot reserved until its member's next update, so a
  //    pool the wave-per-book kernels just fit can overflow there (flagged): it is only ever taken on request (mode 2) or
  //    for markets, where the other kernels do not exist - nothing of the library's choosing is left to guard.
#define BK_BY_R(CALL_1, CALL_2, CALL_4, CALL_8) \
  switch (env->R) {                             \
    case 1: rc = CALL_1; break;                 \
    case 2: rc = CALL_2; break;                 \
    case 4: rc = CALL_4; break;                 \
    default: rc = CALL_8; break;                \
  }
#define BK_SPLIT(MODE) \
  BK_BY_R((launch_split<1, MODE>(env, a, env->steps_done, ns)), (launch_split<2, MODE>(env, a, env->steps_done, ns)), \
          (launch_split<4, MODE>(env, a, env->steps_done, ns)), (launch_split<8, MODE>(env, a, env->steps_done, ns)))
  switch (env->plan().kind) {
    case bk_env::PL_MIXED_WAVE: BK_SPLIT(3) break;
    case bk_env::PL_MIXED_LANES: BK_SPLIT(2) break;
    case bk_env::PL_MIXED_WPB: BK_SPLIT(1) break;
    case bk_env::PL_MIXED_FUSED:
      env->ml_valid = false;
      BK_BY_R(launch_mixed<1>(env, a, env->steps_done, ns), launch_mixed<2>(env, a, env->steps_done, ns),
              launch_mixed<4>(env, a, env->steps_done, ns), launch_mixed<8>(env, a, env->steps_done, ns))
      break;
    case bk_env::PL_FUSED_WAVE:
      BK_BY_R(launch_wave_fused<1>(env, a, env->steps_done, ns), launch_wave_fused<2>(env, a, env->steps_done, ns),
              launch_wave_fused<4>(env, a, env->steps_done, ns), launch_wave_fused<8>(env, a, env->steps_done, ns))
      break;
    case bk_env::PL_SPLIT_WAVE:
    case bk_env::PL_SPLIT_LANES: BK_SPLIT(0) break;  // (launch_split<R, 0> takes k_agents_wave when env->use_wave())
    case bk_env::PL_FUSED_RANDOM:
      BK_BY_R(launch_run<1>(env, a, env->steps_done, ns), launch_run<2>(env, a, env->steps_done, ns),
              launch_run<4>(env, a, env->steps_done, ns), launch_run<8>(env, a, env->steps_done, ns))
      break;
  }
#undef BK_SPLIT
#undef BK_BY_R
  if (rc != BK_OK) return rc;
  env->steps_done += n_steps;
  return BK_OK;
}

// Warm-up without side effects: n_steps of THIS env's own kernels (same agents, same pipeline, same streams) on its own
// books, then everything put back - state blocks, level-2 records, step counter; the scratch steps write no history slot
// and no trade record.  Why it exists: an MI355X drops its clocks within milliseconds of idling and needs ~15 ms of load
// to come back, and the first launch of a pipeline also pays one-off set-up (the parts' streams, dynamic-LDS
// attributes, the decode's lane-state cache); a latency-sensitive caller - a short bk_run after host-side work - calls
// this first.  Asynchronous on the env's stream like bk_run (two device-to-device copies around the steps).
int bk_warm(bk_env* env, uint64_t n_steps) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (n_steps == 0) return BK_OK;
  if (int rc = use_device(env)) return rc;
  const size_t sb = static_cast<size_t>(env->cfg.n_books) * env->stride, lb = static_cast<size_t>(env->cfg.n_books) * env->W;
  if (!env->warm_snap.p) HIPCHK(env->warm_snap.alloc(sb + lb));
  HIPCHK(hipMemcpyAsync(env->warm_snap.p, env->state.p, sb * 4, hipMemcpyDeviceToDevice, env->stream));
  HIPCHK(hipMemcpyAsync(env->warm_snap.p + sb, env->l2_last.p, lb * 4, hipMemcpyDeviceToDevice, env->stream));
  const uint64_t steps0 = env->steps_done;
  const bool flow0 = env->device_flow;
  env->warming = true;
  const int rc = bk_run(env, n_steps);
  env->warming = false;
  env->steps_done = steps0;
  env->device_flow = flow0;
  env->wl_valid = false;  // the wave-per-book lists described the scratch steps' pools
  env->ml_valid = false;  // the members' lists described the scratch steps' pool
  // put the books back WHATEVER bk_run returned: a launch that failed half-way (an event record / wait between two kernel
  // launches) has already stepped some parts, and the step counter above is rolled back either way.  bk_run's own error
  // (and its message) wins over one of the restore copies.
  std::string run_err = rc != BK_OK ? g_err : std::string();
  if (rc != BK_OK)  // (a failed multi-part launch never reached its join: the parts' streams may still be stepping)
    for (int i = 0; i < bk_env::MAX_PARTS; ++i)
      if (env->part_stream[i]) (void)hipStreamSynchronize(env->part_stream[i]);
  const hipError_t e1 = hipMemcpyAsync(env->state.p, env->warm_snap.p, sb * 4, hipMemcpyDeviceToDevice, env->stream);
  const hipError_t e2 = hipMemcpyAsync(env->l2_last.p, env->warm_snap.p + sb, lb * 4, hipMemcpyDeviceToDevice, env->stream);
  if (rc != BK_OK) return fail(rc, run_err);
  HIPCHK(e1);
  HIPCHK(e2);
  return BK_OK;
}

// ------------------------------------------------------------------ readers
uint32_t bk_l2_width(const bk_env* env) { return env ? env->W : 0; }

int bk_level2(bk_env* env, uint32_t first_book, uint32_t n_books, uint32_t* out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (static_cast<uint64_t>(first_book) + n_books > env->cfg.n_books)
    return fail(BK_INVALID_ARGUMENT, "book range out of bounds");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  HIPCHK(hipMemcpy(out, env->l2_last.p + static_cast<size_t>(first_book) * env->W,
                   static_cast<size_t>(n_books) * env->W * 4, hipMemcpyDeviceToHost));
  return BK_OK;
}

// first retained step of the history ring: the last history_capacity steps, or since the last bk_clear_history()
static uint64_t hist_first(const bk_env* env) {
  const uint64_t cap = env->cfg.history_capacity;
  const uint64_t lo = env->steps_done > cap ? env->steps_done - cap : 0;
  return std::max(lo, env->hist_base);
}

int bk_history_len(bk_env* env, uint64_t* first_step, uint64_t* n_steps) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  const uint64_t f = hist_first(env);
  if (first_step) *first_step = f;
  if (n_steps) *n_steps = env->cfg.history_capacity ? env->steps_done - f : 0;
  return BK_OK;
}

// copy steps [first_step, first_step + n_steps) of the ring to `out` (handles the wrap); async if cs != nullptr
static int hist_copy(bk_env* env, uint64_t first_step, uint64_t n_steps, uint32_t first_book, uint32_t n_books,
                     uint32_t* out, hipStream_t cs, bool async) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (!env->cfg.history_capacity || first_step < hist_first(env) || first_step > env->steps_done ||
      n_steps > env->steps_done - first_step)
    return fail(BK_INVALID_ARGUMENT, "step range not retained (history is a ring of history_capacity steps)");
  if (static_cast<uint64_t>(first_book) + n_books > env->cfg.n_books)
    return fail(BK_INVALID_ARGUMENT, "book range out of bounds");
  if (int rc = use_device(env)) return rc;
  const size_t W = env->W, B = env->cfg.n_books, cap = env->cfg.history_capacity;
  const size_t row = static_cast<size_t>(n_books) * W * 4;
  uint64_t done = 0;
  while (done < n_steps) {
    const uint64_t slot = (first_step + done) % cap;
    const uint64_t n = std::min<uint64_t>(n_steps - done, cap - slot);
    const uint32_t* src = env->hist.p + (slot * B + first_book) * W;
    uint32_t* dst = out + done * static_cast<size_t>(n_books) * W;
    if (async && n_books == B)  // whole rows are contiguous: one linear copy
      HIPCHK(hipMemcpyAsync(dst, src, row * n, hipMemcpyDeviceToHost, cs));
    else if (async)
      HIPCHK(hipMemcpy2DAsync(dst, row, src, B * W * 4, row, n, hipMemcpyDeviceToHost, cs));
    else
      HIPCHK(hipMemcpy2D(dst, row, src, B * W * 4, row, n, hipMemcpyDeviceToHost));
    done += n;
  }
  return BK_OK;
}

int bk_history(bk_env* env, uint64_t first_step, uint64_t n_steps, uint32_t first_book, uint32_t n_books,
               uint32_t* out) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  return hist_copy(env, first_step, n_steps, first_book, n_books, out, nullptr, false);
}

// Egress overlapped with stepping: the copy is ordered after the work queued on the env's stream so far and runs on
// `copy_stream`; the caller keeps stepping and must not let the ring wrap onto the steps being copied
// (history_capacity >= 2 x chunk), then waits with bk_stream_sync(copy_stream).  `out` should be pinned
// (bk_pinned_alloc) for the copy to be truly asynchronous.
int bk_history_copy_async(bk_env* env, uint64_t first_step, uint64_t n_steps, uint32_t first_book, uint32_t n_books,
                          uint32_t* out, void* copy_stream) {
  if (!env || !copy_stream) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (int rc = use_device(env)) return rc;
  hipStream_t cs = static_cast<hipStream_t>(copy_stream);
  if (int rc = order_after(cs, env->stream)) return rc;
  return hist_copy(env, first_step, n_steps, first_book, n_books, out, cs, true);
}

int bk_stream_create(void** out) {
  if (!out) return fail(BK_INVALID_ARGUMENT, "null argument");
  hipStream_t s;
  HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = s;
  return BK_OK;
}
int bk_stream_sync(void* stream) {
  HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return BK_OK;
}
int bk_stream_destroy(void* stream) {
  HIPCHK(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return BK_OK;
}
int bk_pinned_alloc(uint64_t nbytes, void** out) {
  if (!out) return fail(BK_INVALID_ARGUMENT, "null argument");
  HIPCHK(hipHostMalloc(out, nbytes, hipHostMallocDefault));
  return BK_OK;
}
int bk_pinned_free(void* p) {
  HIPCHK(hipHostFree(p));
  return BK_OK;
}

int bk_clear_history(bk_env* env) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  env->hist_base = env->steps_done;
  return BK_OK;
}

static int read_hdr(bk_env* env, uint32_t book, int first_dw, int n_dw, uint32_t* out) {
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  HIPCHK(hipMemcpy(out, env->state.p + static_cast<size_t>(book) * env->stride + first_dw, n_dw * 4,
                   hipMemcpyDeviceToHost));
  return BK_OK;
}

int bk_trade_count(bk_env* env, uint32_t book, uint64_t* total, uint64_t* first_retained) {
  if (int rc = check_book(env, book)) return rc;
  uint32_t h[HDR_DW];
  if (int rc = read_hdr(env, book, 0, HDR_DW, h)) return rc;
  if (total) *total = (static_cast<uint64_t>(h[H_TRADES_HI]) << 32) | h[H_TRADES_LO];
  if (first_retained) *first_retained = (static_cast<uint64_t>(h[H_TRADE_BASE_HI]) << 32) | h[H_TRADE_BASE_LO];
  return BK_OK;
}

// one header counter of every book -> host (stream-ordered gather into a contiguous device array, one copy)
int gather_header(bk_env* env, uint32_t word, uint32_t n_words, uint64_t* totals) {
  const uint32_t NB = env->cfg.n_books;
  if (!env->gather_buf.p) HIPCHK(env->gather_buf.alloc(NB));
  hipLaunchKernelGGL(k_gather_header, dim3((NB + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, word, n_words,
                     NB, reinterpret_cast<unsigned long long*>(env->gather_buf.p));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(totals, env->gather_buf.p, static_cast<size_t>(NB) * 8, hipMemcpyDeviceToHost, env->stream));
  HIPCHK(hipStreamSynchronize(env->stream));
  return BK_OK;
}

int bk_trade_counts(bk_env* env, uint64_t* totals) {
  if (!env || !totals) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (int rc = use_device(env)) return rc;
  return gather_header(env, H_TRADES_LO, 2, totals);
}

int bk_get_trades(bk_env* env, uint32_t book, uint64_t first, uint64_t n, bk_trade* out) {
  if (int rc = check_book(env, book)) return rc;
  uint64_t total = 0, base = 0;
  if (int rc = bk_trade_count(env, book, &total, &base)) return rc;
  if (first < base || first > total || n > total - first) return fail(BK_INVALID_ARGUMENT, "trade range not retained");
  if (n && !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (first + n - base > env->cfg.trade_capacity)
    return fail(BK_CAPACITY, "trade records beyond trade_capacity were dropped");
  if (n == 0) return BK_OK;
  std::vector<DevTrade> tmp(n);
  HIPCHK(hipMemcpy(tmp.data(), env->trades.p + static_cast<size_t>(book) * env->cfg.trade_capacity + (first - base),
                   n * sizeof(DevTrade), hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < n; ++i) {
    const DevTrade& d = tmp[i];
    bk_trade& t = out[i];
    t.t = (static_cast<uint64_t>(d.t_hi) << 32) | d.t_lo;
    t.side_is_bid = d.side_is_bid;
    t.price = d.price;
    t.vol = d.vol;
    t.reserved = 0;
    t.active_order_id = d.active;
    t.passive_order_id = d.passive;
  }
  return BK_OK;
}

// Trade egress at scale (Env::get_trades for every book at once): gather every book's retained
// records into ONE dense device buffer in the bk_trade layout with CSR offsets (book b owns [off[b], off[b+1])), and mark
// them consumed (as bk_clear_trades).  Returns the number of records; fetch them with bk_trades_compact_copy_async.
int bk_trades_compact(bk_env* env, uint64_t* out_total) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  const uint32_t B = env->cfg.n_books;
  if (!env->tr_off.p) HIPCHK(env->tr_off.alloc(static_cast<size_t>(B) + 1));
  hipLaunchKernelGGL(k_trade_scan, dim3(1), dim3(1024), 0, env->stream, env->state.p, env->stride, B,
                     env->cfg.trade_capacity, env->tr_off.p);
  HIPCHK(hipGetLastError());
  unsigned long long total = 0;
  HIPCHK(hipMemcpyAsync(&total, env->tr_off.p + B, 8, hipMemcpyDeviceToHost, env->stream));
  HIPCHK(hipStreamSynchronize(env->stream));
  if (total > env->tr_dense.n) HIPCHK(env->tr_dense.alloc(std::max<size_t>(total + total / 4, 1024)));
  hipLaunchKernelGGL(k_trade_gather, dim3((B + 3) / 4), dim3(256), 0, env->stream, env->state.p, env->stride, B,
                     env->cfg.trade_capacity, env->trades.p, env->tr_off.p, env->tr_dense.p);
  HIPCHK(hipGetLastError());
  env->tr_total = total;
  if (out_total) *out_total = total;
  return BK_OK;
}

// Copy the stream made by the last bk_trades_compact: `records` (tr_total x bk_trade) and `offsets` (n_books + 1) to
// host memory on `copy_stream` (ordered after the compaction; NULL = the env's stream, synchronous).  Pinned
// destinations (bk_pinned_alloc) make the copy overlap the next bk_run.
int bk_trades_compact_copy_async(bk_env* env, bk_trade* records, uint64_t* offsets, void* copy_stream) {
  if (!env || !offsets || (env->tr_total && !records)) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (!env->tr_off.p) return fail(BK_INVALID_ARGUMENT, "call bk_trades_compact first");
  if (int rc = use_device(env)) return rc;
  hipStream_t cs = copy_stream ? static_cast<hipStream_t>(copy_stream) : env->stream;
  if (copy_stream)
    if (int rc = order_after(cs, env->stream)) return rc;
  HIPCHK(hipMemcpyAsync(offsets, env->tr_off.p, (static_cast<size_t>(env->cfg.n_books) + 1) * 8, hipMemcpyDeviceToHost, cs));
  if (env->tr_total)
    HIPCHK(hipMemcpyAsync(records, env->tr_dense.p, env->tr_total * sizeof(bk_trade), hipMemcpyDeviceToHost, cs));
  if (!copy_stream) HIPCHK(hipStreamSynchronize(cs));
  return BK_OK;
}

int bk_clear_trades(bk_env* env) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  const uint32_t B = env->cfg.n_books;
  hipLaunchKernelGGL(k_book_service, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, B, 0,
                     0u);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

// OrderEntry.key of orders [first, first + n) (orderbook.rs:34-39): the price and time the order's priority key was
// last set with.  key = (side, price_key(key_price), key_time); price_key = u32::MAX - price for bids (side.rs:300-302).
int bk_get_order_keys(bk_env* env, uint32_t book, uint64_t first, uint64_t n, uint32_t* key_price,
                      uint64_t* key_time) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = mirror_orders(env, book)) return rc;
  BookHost& bh = env->books[book];
  if (first > bh.orders.size() || n > bh.orders.size() - first) return fail(BK_INVALID_ARGUMENT, "order range out of bounds");
  if (env->cfg.max_orders == 0) return fail(BK_INVALID_ARGUMENT, "order log disabled (max_orders == 0)");
  if (bh.n_uploaded > env->cfg.max_orders) return fail(BK_CAPACITY, "order log capacity exceeded");
  if (int rc = use_device(env)) return rc;
  if (int rc = refresh_log(env, book)) return rc;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t id = first + i;
    if (id < bh.n_uploaded && id < bh.log_cache.size()) {
      const DevOrderLog& d = bh.log_cache[id];
      if (key_price) key_price[i] = d.key_price;
      if (key_time) key_time[i] = (static_cast<uint64_t>(d.key_hi) << 32) | d.key_lo;
    } else {  // created, New event still queued: provisional key (orderbook.rs:388-391)
      if (key_price) key_price[i] = bh.orders[id].price;
      if (key_time) key_time[i] = 0;
    }
  }
  return BK_OK;
}

// Replace one book's state with a snapshot (OrderBook::load_json -> TryFrom<OrderBookState>, orderbook.rs:827-918):
// clock, per-step trade volume, every order ever created (with its key) and every trade.  Active orders go back into
// the pool in key order, the level-2 record is rebuilt from them.  The book must have no queued events.
int bk_load_book(bk_env* env, uint32_t book, uint64_t t, uint32_t trade_vol, uint64_t n_orders, const bk_order* orders,
                 const uint32_t* key_price, const uint64_t* key_time, uint64_t n_trades, const bk_trade* trades) {
  if (int rc = check_book(env, book)) return rc;
  if ((n_orders && (!orders || !key_price || !key_time)) || (n_trades && !trades))
    return fail(BK_INVALID_ARGUMENT, "null argument");
  BookHost& bh = env->books[book];
  if (!env->books[book - book % env->M].queue.empty())
    return fail(BK_INVALID_ARGUMENT, "events are queued for this book (market)");
  if (n_orders > env->cfg.max_orders) return fail(BK_CAPACITY, "snapshot holds more orders than max_orders");
  if (n_trades > env->cfg.trade_capacity) return fail(BK_CAPACITY, "snapshot holds more trades than trade_capacity");
  if (n_orders >= 0xFFFFFFFFull) return fail(BK_CAPACITY, "order id space exhausted");
  std::vector<uint64_t> act;  // Active orders, to be ranked by key time (price-time priority within a price)
  for (uint64_t i = 0; i < n_orders; ++i) {
    if (orders[i].order_id != i) return fail(BK_INVALID_ARGUMENT, "orders must be listed by id");
    if (orders[i].status > 4) return fail(BK_INVALID_ARGUMENT, "bad order status");
    if (orders[i].status == 1) act.push_back(i);
  }
  const uint32_t pool = static_cast<uint32_t>(env->R) * 64u;
  if (act.size() > pool) return fail(BK_CAPACITY, "snapshot holds more Active orders than max_live_orders");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  std::stable_sort(act.begin(), act.end(), [&](uint64_t a, uint64_t b) { return key_time[a] < key_time[b]; });

  const uint32_t L = env->cfg.levels, W = env->W, tick = env->asset_tick[book % env->M];
  std::vector<uint32_t> st(env->stride, 0u), l2(W, 0u);
  uint32_t* hdr = st.data();
  HIPCHK(hipMemcpy(hdr, env->state.p + static_cast<size_t>(book) * env->stride, HDR_DW * 4, hipMemcpyDeviceToHost));
  hdr[H_T_LO] = static_cast<uint32_t>(t);
  hdr[H_T_HI] = static_cast<uint32_t>(t >> 32);
  hdr[H_NEXT_ID] = static_cast<uint32_t>(n_orders);
  hdr[H_SEQ] = static_cast<uint32_t>(act.size());
  hdr[H_TRADES_LO] = static_cast<uint32_t>(n_trades);
  hdr[H_TRADES_HI] = static_cast<uint32_t>(n_trades >> 32);
  hdr[H_TRADE_BASE_LO] = hdr[H_TRADE_BASE_HI] = 0;
  hdr[H_FLAGS] = 0;
  hdr[H_TRADE_VOL] = trade_vol;
  hdr[H_LAST_NTRADES] = hdr[H_LAST_NEVENTS] = 0;
  for (int r = 0; r < 8; ++r) hdr[H_LIVE0 + 2 * r] = hdr[H_LIVE0 + 2 * r + 1] = 0;
  uint32_t bid_best = 0, ask_best = 0xFFFFFFFFu, bid_vol = 0, ask_vol = 0;
  for (size_t k = 0; k < act.size(); ++k) {
    const bk_order& o = orders[act[k]];
    const uint32_t r = static_cast<uint32_t>(k) / 64u, lane = static_cast<uint32_t>(k) % 64u;
    uint32_t* p = st.data() + HDR_DW + r * POOL_FIELDS * 64;
    p[0 * 64 + lane] = o.price;
    p[1 * 64 + lane] = o.vol;
    p[2 * 64 + lane] = static_cast<uint32_t>(o.order_id);
    p[3 * 64 + lane] = static_cast<uint32_t>(k);  // seq: rank in key-time order
    p[4 * 64 + lane] = 1u | (o.side_is_bid ? 2u : 0u);
    hdr[H_LIVE0 + 2 * r + (lane >> 5)] |= 1u << (lane & 31);
    if (o.side_is_bid) {
      bid_best = std::max(bid_best, o.price);
      bid_vol += o.vol;
    } else {
      ask_best = std::min(ask_best, o.price);
      ask_vol += o.vol;
    }
  }
  // level-2 record of the loaded book (orderbook.rs:229-264): [trade_vol, bid, ask, ask_vol, bid_vol, levels...]
  l2[0] = trade_vol;
  l2[1] = bid_best;
  l2[2] = ask_best;
  l2[3] = ask_vol;
  l2[4] = bid_vol;
  for (uint64_t i : act) {
    const bk_order& o = orders[i];
    const uint32_t d = o.side_is_bid ? bid_best - o.price : o.price - ask_best;
    if (d % tick == 0 && d / tick < L) {
      l2[5 + 4 * (d / tick) + (o.side_is_bid ? 0 : 2)] += o.vol;
      l2[5 + 4 * (d / tick) + (o.side_is_bid ? 1 : 3)] += 1;
    }
  }
  std::vector<DevOrderLog> log(n_orders);
  bh.orders.clear();
  for (uint64_t i = 0; i < n_orders; ++i) {
    const bk_order& o = orders[i];
    bh.orders.push_back(HostOrder{static_cast<uint8_t>(o.side_is_bid ? 1 : 0), o.start_vol, o.price, o.trader_id,
                                  o.arr_time});
    DevOrderLog& d = log[i];
    std::memset(&d, 0, sizeof(d));
    d.status = o.status;
    d.vol = o.vol;
    d.price = o.price;
    d.key_price = key_price[i];
    d.arr_lo = static_cast<uint32_t>(o.arr_time);
    d.arr_hi = static_cast<uint32_t>(o.arr_time >> 32);
    d.end_lo = static_cast<uint32_t>(o.end_time);
    d.end_hi = static_cast<uint32_t>(o.end_time >> 32);
    d.key_lo = static_cast<uint32_t>(key_time[i]);
    d.key_hi = static_cast<uint32_t>(key_time[i] >> 32);
  }
  std::vector<DevTrade> tr(n_trades);
  for (uint64_t i = 0; i < n_trades; ++i) {
    const bk_trade& x = trades[i];
    tr[i] = DevTrade{static_cast<uint32_t>(x.t), static_cast<uint32_t>(x.t >> 32), x.price, x.vol,
                     static_cast<uint32_t>(x.active_order_id), static_cast<uint32_t>(x.passive_order_id),
                     x.side_is_bid ? 1u : 0u, 0u};
  }
  HIPCHK(hipMemcpy(env->state.p + static_cast<size_t>(book) * env->stride, st.data(), st.size() * 4,
                   hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(env->l2_last.p + static_cast<size_t>(book) * W, l2.data(), W * 4, hipMemcpyHostToDevice));
  if (n_orders)
    HIPCHK(hipMemcpy(env->order_log.p + static_cast<size_t>(book) * env->cfg.max_orders, log.data(),
                     n_orders * sizeof(DevOrderLog), hipMemcpyHostToDevice));
  if (n_trades)
    HIPCHK(hipMemcpy(env->trades.p + static_cast<size_t>(book) * env->cfg.trade_capacity, tr.data(),
                     n_trades * sizeof(DevTrade), hipMemcpyHostToDevice));
  env->ml_valid = false;
  env->wl_valid = false;
  bh.n_uploaded = n_orders;
  bh.log_fresh = false;
  bh.time_offset = t - (env->cfg.start_time + env->steps_done * env->cfg.step_size);
  return BK_OK;
}

// OrderBook::set_time (crates/order_book/src/orderbook.rs:183-185) for one book (host-driven path)
int bk_set_time(bk_env* env, uint32_t book, uint64_t t) {
  if (int rc = check_book(env, book)) return rc;
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  const uint32_t w[2] = {static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32)};
  HIPCHK(hipMemcpy(env->state.p + static_cast<size_t>(book) * env->stride + H_T_LO, w, 8, hipMemcpyHostToDevice));
  env->books[book].time_offset = t - (env->cfg.start_time + env->steps_done * env->cfg.step_size);
  return BK_OK;
}

int bk_time(bk_env* env, uint32_t book, uint64_t* out) {
  if (int rc = check_book(env, book)) return rc;
  uint32_t h[2];
  if (int rc = read_hdr(env, book, H_T_LO, 2, h)) return rc;
  if (out) *out = (static_cast<uint64_t>(h[1]) << 32) | h[0];
  return BK_OK;
}

int bk_trade_vol(bk_env* env, uint32_t book, uint32_t* out) {
  if (int rc = check_book(env, book)) return rc;
  if (!out) return fail(BK_INVALID_ARGUMENT, "null argument");
  return read_hdr(env, book, H_TRADE_VOL, 1, out);
}

int bk_steps_done(bk_env* env, uint64_t* out) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (out) *out = env->steps_done;
  return BK_OK;
}

int bk_book_flags(bk_env* env, uint32_t* out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (int rc = use_device(env)) return rc;
  std::vector<uint64_t> v(env->cfg.n_books);
  if (int rc = gather_header(env, H_FLAGS, 1, v.data())) return rc;
  for (size_t b = 0; b < v.size(); ++b) out[b] = static_cast<uint32_t>(v[b]);
  return BK_OK;
}

// Clear sticky flag bits of every book (the caller has seen and handled them: e.g. after draining the trade records that
// overflowed, or after reporting a capacity error once)
int bk_clear_flags(bk_env* env, uint32_t mask) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  const uint32_t B = env->cfg.n_books;
  hipLaunchKernelGGL(k_book_service, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, B, 2, mask);
  HIPCHK(hipGetLastError());
  return BK_OK;
}

// What a strict caller polls after every step: the OR of all books' sticky flags and the largest number of trade
// records any book retains (towards trade_capacity) - one small reduction + an 8-byte copy instead of n_books words.
int bk_flags_summary(bk_env* env, uint32_t* flags_or, uint64_t* max_retained_trades) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  const uint32_t B = env->cfg.n_books;
  if (!env->gather_buf.p) HIPCHK(env->gather_buf.alloc(B));
  uint32_t* out = reinterpret_cast<uint32_t*>(env->gather_buf.p);
  HIPCHK(hipMemsetAsync(out, 0, 8, env->stream));
  hipLaunchKernelGGL(k_flags_summary, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, B, out);
  HIPCHK(hipGetLastError());
  uint32_t res[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(res, out, 8, hipMemcpyDeviceToHost, env->stream));
  HIPCHK(hipStreamSynchronize(env->stream));
  if (flags_or) *flags_or = res[0];
  if (max_retained_trades) *max_retained_trades = res[1];
  return BK_OK;
}

int bk_rng_state(bk_env* env, uint32_t book, uint64_t out_state[2]) {
  if (int rc = check_book(env, book)) return rc;
  if (!out_state) return fail(BK_INVALID_ARGUMENT, "null argument");
  uint32_t h[4];
  if (int rc = read_hdr(env, book, H_S0_LO, 4, h)) return rc;
  out_state[0] = (static_cast<uint64_t>(h[1]) << 32) | h[0];
  out_state[1] = (static_cast<uint64_t>(h[3]) << 32) | h[2];
  return BK_OK;
}

int bk_live_orders(bk_env* env, uint32_t book, uint32_t cap, bk_order* out, uint32_t* n_out) {
  if (int rc = check_book(env, book)) return rc;
  if (cap && !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  std::vector<uint32_t> st(env->stride);
  if (int rc = read_hdr(env, book, 0, static_cast<int>(env->stride), st.data())) return rc;
  struct Live {
    uint32_t price, vol, id, seq, bid;
  };
  std::vector<Live> v;
  for (int r = 0; r < env->R; ++r) {
    const uint32_t* p = st.data() + HDR_DW + r * POOL_FIELDS * 64;
    for (int l = 0; l < 64; ++l)
      if (p[4 * 64 + l] & 1u) v.push_back(Live{p[l], p[64 + l], p[128 + l], p[192 + l], (p[4 * 64 + l] >> 1) & 1u});
  }
  std::sort(v.begin(), v.end(), [](const Live& x, const Live& y) {
    if (x.bid != y.bid) return x.bid > y.bid;  // bids first
    if (x.price != y.price) return x.bid ? x.price > y.price : x.price < y.price;
    return x.seq < y.seq;
  });
  if (n_out) *n_out = static_cast<uint32_t>(v.size());
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    std::memset(&out[i], 0, sizeof(bk_order));
    out[i].side_is_bid = static_cast<uint8_t>(v[i].bid);
    out[i].status = 1;
    out[i].vol = v[i].vol;
    out[i].price = v[i].price;
    out[i].order_id = v[i].id;
    out[i].end_time = ~0ull;
  }
  return BK_OK;
}

// ------------------------------------------------------------------ stats
int bk_stats_compute(bk_env* env, bk_stats* out_host) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  // reset the record from the template kept next to it: a device-to-device copy in stream order, so that queueing the
  // reduction never waits for the stepping kernels in front of it (the per-launch all-gather of a sharded run)
  HIPCHK(hipMemcpyAsync(env->stats.p, env->stats.p + 1, sizeof(DevStats), hipMemcpyDeviceToDevice, env->stream));
  const uint32_t B = env->cfg.n_books;
  const uint32_t blocks = std::min<uint32_t>((B + 255) / 256, 1024);
  hipLaunchKernelGGL(k_stats, dim3(blocks), dim3(256), 0, env->stream, env->state.p, env->stride, env->l2_last.p,
                     env->W, B, env->stats.p);
  HIPCHK(hipGetLastError());
  if (out_host) {
    HIPCHK(hipStreamSynchronize(env->stream));
    HIPCHK(hipMemcpy(out_host, env->stats.p, sizeof(bk_stats), hipMemcpyDeviceToHost));
  }
  return BK_OK;
}

int bk_stats_device_ptr(bk_env* env, void** out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  *out = env->stats.p;
  return BK_OK;
}

// the latest level-2 records on the device: u32[n_books][bk_l2_width()], for on-device consumers (e.g. the optional
// per-book L1 all-gather across GPUs); valid until the env is destroyed, contents as of the work queued so far
int bk_level2_device_ptr(bk_env* env, void** out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  *out = env->l2_last.p;
  return BK_OK;
}

// ------------------------------------------------------------------ measurement
int bk_profile_enable(bk_env* env, int on) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  env->profile = on < 0 ? 0 : on;
  env->prof_tick = 0;
  env->prof_now = false;
  if (env->profile > 0 && use_device(env) == BK_OK) {
    while (env->prof_pool.size() < 256) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) break;
      env->prof_pool.push_back(e);
    }
    // Two events recorded on a stream bracket the launch gap as well as the kernel (the first fires when the previous
    // kernel of the stream ends): ~4 us, 3 % of a 135 us launch and 12 % of a 35 us one against the kernel trace
    // (scripts/kt_check.sh).  The reading of two events recorded back to back, calibrated once per env, is subtracted
    // from every sample.
    if (env->prof_bracket_ms == 0 && env->prof_pool.size() >= 2) {
      HIPCHK(hipStreamSynchronize(env->stream));
      std::vector<float> v;
      for (int i = 0; i < 9; ++i) {
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, env->stream, 1u);  // something for the first event to follow
        (void)hipEventRecord(env->prof_pool[0], env->stream);
        (void)hipEventRecord(env->prof_pool[1], env->stream);
        (void)hipStreamSynchronize(env->stream);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, env->prof_pool[0], env->prof_pool[1]) == hipSuccess) v.push_back(ms);
      }
      if (!v.empty()) {
        std::sort(v.begin(), v.end());
        env->prof_bracket_ms = v[v.size() / 2];
      }
      if (getenv("BOURSE_AMD_VERBOSE")) fprintf(stderr, "bourse_amd: two events back to back read %.1f us\n", env->prof_bracket_ms * 1e3);
    }
  }
  return BK_OK;
}

int bk_profile_read(bk_env* env, double* total_ms, uint64_t* n_launches, int reset) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (int rc = use_device(env)) return rc;
  if (int rc = prof_collect(env)) return rc;
  double ms = 0;
  uint64_t n = 0;
  for (int k = 0; k < bk_env::PROF_KINDS; ++k) {
    ms += env->prof_ms[k];
    n += env->prof_launches[k];
    if (reset) {
      env->prof_ms[k] = 0.0;
      env->prof_launches[k] = 0;
    }
  }
  if (total_ms) *total_ms = ms;
  if (n_launches) *n_launches = n;
  return BK_OK;
}

int bk_profile_read_kind(bk_env* env, int kind, double* total_ms, uint64_t* n_launches) {
  if (!env || kind < 0 || kind >= bk_env::PROF_KINDS) return fail(BK_INVALID_ARGUMENT, "bad argument");
  if (int rc = use_device(env)) return rc;
  if (int rc = prof_collect(env)) return rc;
  if (total_ms) *total_ms = env->prof_ms[kind];
  if (n_launches) *n_launches = env->prof_launches[kind];
  return BK_OK;
}

int bk_get_pipeline(bk_env* env, int* split, int* n_parts) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  query_fused_resident(env);
  const bk_env::Plan pl = env->plan();  // the same function bk_run launches from
  int code = 0;
  switch (pl.kind) {
    case bk_env::PL_FUSED_RANDOM:
    case bk_env::PL_MIXED_FUSED: code = 0; break;
    case bk_env::PL_SPLIT_LANES:
    case bk_env::PL_MIXED_LANES:
    case bk_env::PL_MIXED_WPB: code = 1; break;
    case bk_env::PL_SPLIT_WAVE:
    case bk_env::PL_MIXED_WAVE: code = 2; break;
    case bk_env::PL_FUSED_WAVE: code = 3; break;
  }
  if (split) *split = code;
  if (n_parts) *n_parts = pl.parts;
  return BK_OK;
}

int bk_set_pipeline(bk_env* env, int mode) {
  if (!env || mode < 0 || mode > 5)
    return fail(BK_INVALID_ARGUMENT,
                "pipeline mode must be 0 (auto), 1 (fused), 2 (split), 3 (split, wave-per-book AgentSet members), 4 (wave_split: "
                "wave-parallel RNG decode kernel + event kernel) or 5 (wave: both fused in one persistent kernel)");
  env->pipeline = mode;
  return BK_OK;
}

// DEPRECATED (kept so that clients built against rounds 1-3 still link): those rounds rolled a launch of the library's
// own pipeline choice back when it overflowed a pool the other kernels fit; no such launch exists any more.  Always 0.
int bk_pipeline_fallbacks(bk_env* env, uint64_t* out) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  *out = 0;
  return BK_OK;
}

// k_agents_wave knobs: look-ahead of the vector path (1..64 draws; small values force the scalar slow path - tests),
// parts the wave pipeline cuts the batch in (0 = as the lane split)
int bk_set_wave_options(bk_env* env, uint32_t lookahead, int parts) {
  if (!env || lookahead < 1 || lookahead > 64 || parts < 0 || parts > bk_env::MAX_PARTS)
    return fail(BK_INVALID_ARGUMENT, "lookahead must be in 1..64 and parts in 0..8");
  env->wave_lookahead = lookahead;
  env->wave_parts = parts;
  return BK_OK;
}

// split pipeline geometry: the batch is cut in min(n_parts, units / min_part) contiguous parts on separate streams
int bk_set_split_parts(bk_env* env, int n_parts, uint32_t min_part) {
  if (!env || n_parts < 1 || n_parts > bk_env::MAX_PARTS || min_part < 64)
    return fail(BK_INVALID_ARGUMENT, "n_parts must be in 1..8 and min_part >= 64");
  env->n_parts = n_parts;
  env->min_part = min_part;
  return BK_OK;
}

int bk_get_split_parts(bk_env* env, int* n_parts, uint32_t* min_part) {
  if (!env) return fail(BK_INVALID_ARGUMENT, "null env");
  if (n_parts) *n_parts = env->n_parts;
  if (min_part) *min_part = env->min_part;
  return BK_OK;
}

// orders created so far per book by the on-device agents (OrderBook::current_order_id, orderbook.rs:327-329)
int bk_order_counts(bk_env* env, uint64_t* totals) {
  if (!env || !totals) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (int rc = use_device(env)) return rc;
  return gather_header(env, H_NEXT_ID, 1, totals);
}

int bk_event_steps_keyed(bk_env* env, uint64_t* counts) {
  if (!env || !counts) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (int rc = use_device(env)) return rc;
  return gather_header(env, H_EV_KEYED, 1, counts);
}

// ---------------------------------------------------------------- checkpoint / resume (on-device order flow)
// The reference cannot resume a running simulation (Env, agents and RNG are not serialisable);
// here the whole simulation state IS the per-book device block (pool, clock, counters, RNG), so a checkpoint
// is one device-to-host copy.  Host-driven envs (order log + host order table) are not covered.
// header: 8 x u64 = {magic "BKCKPT02", steps_done, n_books << 32 | stride (pool size), levels width W << 32 | assets,
// number of Noise/Momentum members << 32 | RandomAgents groups, hash of the installed agent set, trading flag, 0}
constexpr uint64_t CKPT_MAGIC = 0x3230545043434B42ull;  // "BKCKPT02" little-endian
constexpr size_t CKPT_HDR = 8;                            // u64 words
static void ckpt_header(const bk_env* env, uint64_t* h) {
  h[0] = CKPT_MAGIC;
  h[1] = env->steps_done;
  h[2] = (static_cast<uint64_t>(env->cfg.n_books) << 32) | env->stride;
  h[3] = (static_cast<uint64_t>(env->W) << 32) | env->M;
  h[4] = (static_cast<uint64_t>(env->n_mixed) << 32) | static_cast<uint32_t>(env->groups.size());
  h[5] = env->agents_hash;
  h[6] = env->trading;
  h[7] = 0;
}
uint64_t bk_checkpoint_bytes(const bk_env* env) {
  // header + per-book state blocks + the latest level-2 records (Env::level_2_data; the lane-per-book members' update
  // reads the touches from there)
  return env ? CKPT_HDR * 8 + static_cast<uint64_t>(env->cfg.n_books) * (env->stride + env->W) * 4 : 0;
}

int bk_checkpoint_save(bk_env* env, void* out, uint64_t nbytes) {
  if (!env || !out) return fail(BK_INVALID_ARGUMENT, "null argument");
  if (nbytes < bk_checkpoint_bytes(env)) return fail(BK_INVALID_ARGUMENT, "checkpoint buffer too small");
  if (env->device_ingress) return fail(BK_INVALID_ARGUMENT, "checkpointing a host-driven env is not supported");
  for (const BookHost& bh : env->books)
    if (!bh.orders.empty()) return fail(BK_INVALID_ARGUMENT, "checkpointing a host-driven env is not supported");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  uint64_t* h = static_cast<uint64_t*>(out);
  ckpt_header(env, h);
  const size_t sb = static_cast<size_t>(env->cfg.n_books) * env->stride * 4;
  HIPCHK(hipMemcpy(h + CKPT_HDR, env->state.p, sb, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(reinterpret_cast<char*>(h + CKPT_HDR) + sb, env->l2_last.p,
                   static_cast<size_t>(env->cfg.n_books) * env->W * 4, hipMemcpyDeviceToHost));
  return BK_OK;
}

int bk_checkpoint_load(bk_env* env, const void* in, uint64_t nbytes) {
  if (!env || !in) return fail(BK_INVALID_ARGUMENT, "null argument");
  const uint64_t* h = static_cast<const uint64_t*>(in);
  if (nbytes < CKPT_HDR * 8 || h[0] != CKPT_MAGIC)
    return fail(BK_INVALID_ARGUMENT, "not a bourse_amd checkpoint of this version (bad magic)");
  uint64_t want[CKPT_HDR];
  ckpt_header(env, want);
  if (nbytes < bk_checkpoint_bytes(env) || h[2] != want[2] || h[3] != want[3])
    return fail(BK_INVALID_ARGUMENT, "checkpoint does not match this env (n_books / pool size / levels / assets)");
  if (h[4] != want[4] || h[5] != want[5])
    return fail(BK_INVALID_ARGUMENT, "checkpoint was taken with a different agent set: install the same agents first");
  if (int rc = use_device(env)) return rc;
  HIPCHK(hipStreamSynchronize(env->stream));
  const size_t sb = static_cast<size_t>(env->cfg.n_books) * env->stride * 4;
  HIPCHK(hipMemcpy(env->state.p, h + CKPT_HDR, sb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(env->l2_last.p, reinterpret_cast<const char*>(h + CKPT_HDR) + sb,
                   static_cast<size_t>(env->cfg.n_books) * env->W * 4, hipMemcpyHostToDevice));
  env->ml_valid = false;
  env->wl_valid = false;
  env->steps_done = h[1];
  env->hist_base = h[1];  // retained history/trade records restart at the restored step
  env->trading = h[6] ? 1u : 0u;  // the host mirror of the books' trading flag (H_TRADING travels in the state blocks)
  // an image is only ever taken from an on-device-order-flow env: the restored books hold the agents' orders and ids, so
  // host-driven orders (whose ids would restart at 0) are refused from here on, exactly as after a bk_run
  if (env->n_mixed || !env->groups.empty() || h[1] > 0) env->device_flow = true;
  const uint32_t B = env->cfg.n_books;
  hipLaunchKernelGGL(k_book_service, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->state.p, env->stride, B, 0,
                     0u);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(env->stream));
  return BK_OK;
}

uint64_t bk_state_bytes_per_book(const bk_env* env) { return env ? static_cast<uint64_t>(env->stride) * 4 : 0; }

#if BOURSE_AMD_STAMPS
// diagnostic build only (book_device.hpp BK_STAMP): per-book phase accumulators, BK_STAMP_WORDS u32 per book.
// bk_debug_stamps(n_books, out): the first call allocates + zeroes them and returns nothing; later calls copy them out
// (out: n_books * BK_STAMP_WORDS) and zero.
int bk_debug_stamps(uint32_t n_books, unsigned int* out) {
  static unsigned int* dev = nullptr;
  static uint32_t cap = 0;
  HIPCHK(hipDeviceSynchronize());
  if (!dev || cap < n_books) {
    if (dev) (void)hipFree(dev);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dev), static_cast<size_t>(n_books) * BK_STAMP_WORDS * 4));
    cap = n_books;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(bkd::g_stamp_ptr), &dev, sizeof(dev)));
  } else if (out) {
    HIPCHK(hipMemcpy(out, dev, static_cast<size_t>(n_books) * BK_STAMP_WORDS * 4, hipMemcpyDeviceToHost));
  }
  HIPCHK(hipMemset(dev, 0, static_cast<size_t>(cap) * BK_STAMP_WORDS * 4));
  return BK_OK;
}
#endif

// DPP reduction self-test (tests only): in[n_waves*64] -> out[n_waves*4] = {min, max, sum, sel-sum}
int bk_selftest_reduce(const uint32_t* in_host, uint32_t n_waves, uint32_t* out_host) {
  if (!in_host || !out_host) return fail(BK_INVALID_ARGUMENT, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BK_NO_DEVICE, "no HIP device available");
  DevBuf<uint32_t> in, out;
  HIPCHK(in.alloc(static_cast<size_t>(n_waves) * 64));
  HIPCHK(out.alloc(static_cast<size_t>(n_waves) * 4));
  HIPCHK(hipMemcpy(in.p, in_host, static_cast<size_t>(n_waves) * 64 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_reduce, dim3(n_waves), dim3(64), 0, nullptr, in.p, out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, out.p, static_cast<size_t>(n_waves) * 4 * 4, hipMemcpyDeviceToHost));
  return BK_OK;
}

// pm_math.hpp on the device (tests only): out[i] = f(in[i]), op 0 pm::exp, 1 pm::log, 2 pm::tanh - the routines the
// Noise/Momentum members' log-normal offsets and momentum signal go through (agents/common.rs:104,137;
// momentum_agent.rs:156); tests compare them with the host build of the same header bit for bit and with libm in ulps
int bk_selftest_math(int op, const double* in_host, uint64_t n, double* out_host) {
  if (!in_host || !out_host || op < 0 || op > 2) return fail(BK_INVALID_ARGUMENT, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BK_NO_DEVICE, "no HIP device available");
  DevBuf<double> in, out;
  HIPCHK(in.alloc(n));
  HIPCHK(out.alloc(n));
  HIPCHK(hipMemcpy(in.p, in_host, n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_math, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, nullptr, op, in.p, out.p, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, out.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return BK_OK;
}

}  // extern "C"
