/* Plain-C client of the drop-in boundary (include/bourse_amd.h): what a cgo / Rust-FFI / JNI host would do.
 * Build: gcc -std=c11 -Iinclude tests/c/abi_smoke.c -Lbourse_amd/csrc -lbourse_amd -Wl,-rpath,$PWD/bourse_amd/csrc
 * Exit code 0 = all checks passed on a GPU; 77 = no HIP device (the library has no CPU path and says so). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bourse_amd.h"

#define CHECK(expr)                                                                 \
  do {                                                                              \
    int rc_ = (expr);                                                               \
    if (rc_ != BK_OK) {                                                             \
      fprintf(stderr, "%s -> %d (%s)\n", #expr, rc_, bk_last_error());              \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
#define EXPECT(cond)                                                                \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      fprintf(stderr, "line %d: expectation failed: %s\n", __LINE__, #cond);        \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main(void) {
  bk_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_books = 2;
  cfg.levels = 10;
  cfg.tick_size = 1;
  cfg.trading = 1;
  cfg.step_size = 1000;
  cfg.seed = 101;
  cfg.max_live_orders = 64;
  cfg.max_orders = 64;
  cfg.trade_capacity = 64;
  cfg.history_capacity = 8;
  bk_env* env = NULL;
  int rc = bk_env_create(&cfg, &env);
  if (rc == BK_NO_DEVICE) {
    printf("no HIP device: %s\n", bk_last_error());
    return 77;
  }
  CHECK(rc);

  /* ref crates/step_sim/src/env.rs:312-368 (three steps) on book 1; book 0 stays empty */
  uint64_t id = 99;
  CHECK(bk_place_order(env, 1, 1, 10, 101, 1, 10, &id));
  EXPECT(id == 0);
  CHECK(bk_place_order(env, 1, 0, 20, 101, 1, 20, &id));
  EXPECT(id == 1);
  CHECK(bk_step(env));
  uint32_t w = bk_l2_width(env);
  EXPECT(w == 45);
  uint32_t l2[2 * 45];
  CHECK(bk_level2(env, 0, 2, l2));
  EXPECT(l2[1] == 0 && l2[2] == 0xFFFFFFFFu);                 /* empty book: bid 0, ask u32::MAX */
  EXPECT(l2[45 + 1] == 10 && l2[45 + 2] == 20);               /* bid_ask == (10, 20) */
  CHECK(bk_place_order(env, 1, 1, 10, 101, 1, 11, &id));
  CHECK(bk_place_order(env, 1, 0, 20, 101, 1, 21, &id));
  CHECK(bk_step(env));
  CHECK(bk_place_order(env, 1, 1, 30, 101, 0, 0, &id));       /* market buy 30 */
  EXPECT(id == 4);
  CHECK(bk_step(env));
  CHECK(bk_level2(env, 1, 1, l2));
  EXPECT(l2[0] == 30 && l2[1] == 11 && l2[2] == 21 && l2[3] == 10 && l2[4] == 20);
  uint64_t n_tr = 0, base = 0;
  CHECK(bk_trade_count(env, 1, &n_tr, &base));
  EXPECT(n_tr == 2 && base == 0);
  bk_trade tr[2];
  CHECK(bk_get_trades(env, 1, 0, 2, tr));
  EXPECT(tr[0].price == 20 && tr[0].vol == 20 && tr[0].active_order_id == 4 && tr[0].passive_order_id == 1);
  EXPECT(tr[1].price == 21 && tr[1].vol == 10 && tr[1].passive_order_id == 3 && tr[0].side_is_bid == 0);
  uint8_t status = 0;
  CHECK(bk_order_status(env, 1, 1, &status));
  EXPECT(status == 2); /* Filled */
  uint64_t t = 0;
  CHECK(bk_time(env, 1, &t));
  EXPECT(t == 3000);
  /* error behaviour: tick check (orderbook.rs:367-382) and unknown ids */
  bk_env_destroy(env);
  cfg.tick_size = 2;
  cfg.n_books = 1;
  CHECK(bk_env_create(&cfg, &env));
  EXPECT(bk_place_order(env, 0, 1, 10, 0, 1, 11, &id) == BK_PRICE_NOT_TICK_MULTIPLE);
  EXPECT(strstr(bk_last_error(), "Price 11 was not a multiple of tick-size 2") != NULL);
  CHECK(bk_cancel_order(env, 0, 5));
  EXPECT(bk_step(env) == BK_UNKNOWN_ORDER_ID);
  bk_env_destroy(env);

  /* on-device agents: sim_runner with two RandomAgents groups on 256 books, twice from the same seed */
  uint32_t sum[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    memset(&cfg, 0, sizeof cfg);
    cfg.n_books = 256;
    cfg.levels = 16;
    cfg.tick_size = 2;
    cfg.trading = 1;
    cfg.step_size = 100000;
    cfg.seed = 101;
    cfg.max_live_orders = 64;
    cfg.trade_capacity = 4096;
    cfg.history_capacity = 20;
    CHECK(bk_env_create(&cfg, &env));
    bk_random_agents g[2] = {{32, 40, 56, 10, 20, 2, 0.8f}, {32, 40, 56, 50, 70, 2, 0.2f}};
    CHECK(bk_set_random_agents(env, 2, g));
    CHECK(bk_set_pipeline(env, rep == 0 ? 1 : 2)); /* fused, then split: identical results */
    if (rep == 1) CHECK(bk_warm(env, 7));          /* scratch steps: state, records and the step counter are put back */
    CHECK(bk_run(env, 20));
    CHECK(bk_env_sync(env));
    bk_stats st;
    CHECK(bk_stats_compute(env, &st));
    EXPECT(st.n_books == 256 && st.sum_trades > 0 && st.sum_events > 0);
    uint32_t* h = (uint32_t*)malloc((size_t)20 * 256 * bk_l2_width(env) * 4);
    CHECK(bk_history(env, 0, 20, 0, 256, h));
    for (size_t i = 0; i < (size_t)20 * 256 * bk_l2_width(env); ++i) sum[rep] = sum[rep] * 31u + h[i];
    free(h);
    uint32_t flags[256], any = 1;
    uint64_t retained = 0, steps = 0;
    CHECK(bk_book_flags(env, flags));
    for (int b = 0; b < 256; ++b) EXPECT(flags[b] == 0);
    CHECK(bk_flags_summary(env, &any, &retained));
    EXPECT(any == 0 && retained > 0 && retained <= 4096);
    CHECK(bk_clear_flags(env, 0xFFFFFFFFu));
    CHECK(bk_steps_done(env, &steps));
    EXPECT(steps == 20);
    int np = 0;
    uint32_t mp = 0;
    CHECK(bk_get_split_parts(env, &np, &mp));
    EXPECT(np == 4 && mp == 4096);
    bk_env_destroy(env);
  }
  EXPECT(sum[0] == sum[1]);

  /* host arrays through the device ingress (StepEnvNumpy.submit_instructions for many books, rust/src/step_sim_numpy.rs:233-275):
   * 64 books x 3 instructions, a bad price planted in book 5, a BK_ACTION_MODIFY and an action the reference ignores in book 7;
   * the second batch is written in place into the library's pinned staging, its results read as views */
  memset(&cfg, 0, sizeof cfg);
  cfg.n_books = 64;
  cfg.levels = 10;
  cfg.tick_size = 2;
  cfg.trading = 1;
  cfg.step_size = 1000;
  cfg.seed = 7;
  cfg.max_live_orders = 64;
  cfg.max_orders = 32;
  cfg.trade_capacity = 64;
  cfg.history_capacity = 4;
  CHECK(bk_env_create(&cfg, &env));
  EXPECT(bk_submit_instructions_host(env, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == BK_INVALID_ARGUMENT);
  CHECK(bk_device_ingress_enable(env, 16));
  {
    enum { NB = 64, PER = 3, N = NB * PER };
    uint64_t off[NB + 1], oid[N], ids[N], ticket = 99, t2 = 99;
    uint32_t act[N], vol[N], trd[N], prc[N], st[2 * NB], bad = 0;
    uint8_t side[N];
    for (int b = 0; b <= NB; ++b) off[b] = (uint64_t)b * PER;
    for (int i = 0; i < N; ++i) {
      act[i] = 1; side[i] = (uint8_t)(i % PER == 0); vol[i] = 10; trd[i] = (uint32_t)(i % PER); oid[i] = 0;
      prc[i] = i % PER == 0 ? 100 : 110 + 2 * (uint32_t)(i % PER);   /* bid 100, asks 112 and 114 */
    }
    prc[5 * PER + 1] = 113;                /* book 5: its second element is off-tick -> the book stops there, element 0 stays queued */
    CHECK(bk_submit_instructions_host(env, off, act, side, vol, trd, prc, oid, &ticket));
    EXPECT(ticket == 0);
    CHECK(bk_submit_result(env, ticket, ids, st, &bad));
    EXPECT(bad == 5 && st[2 * 5] == BK_PRICE_NOT_TICK_MULTIPLE && st[2 * 5 + 1] == 1 && st[0] == BK_OK && st[1] == PER);
    EXPECT(ids[0] == 0 && ids[1] == 1 && ids[2] == 2 && ids[5 * PER] == 0 && ids[5 * PER + 1] == UINT64_MAX && ids[5 * PER + 2] == UINT64_MAX);
    CHECK(bk_step(env));
    uint32_t lw = bk_l2_width(env), *rec = (uint32_t*)malloc((size_t)NB * lw * 4);
    CHECK(bk_level2(env, 0, NB, rec));
    EXPECT(rec[1] == 100 && rec[2] == 112 && rec[5 * lw + 1] == 100 && rec[5 * lw + 2] == UINT32_MAX);  /* book 5 holds its bid only */
    bk_ingress_arrays stg;
    CHECK(bk_ingress_staging(env, N, &stg));
    EXPECT(stg.capacity >= N);
    for (int b = 0; b <= NB; ++b) stg.book_offsets[b] = (uint64_t)b * 2;
    for (int b = 0; b < NB; ++b) {         /* every book: cancel its ask 112 (id 1), lift its bid's volume to 25 (a replace) */
      stg.action[2 * b] = 2; stg.order_id[2 * b] = 1; stg.side[2 * b] = 0; stg.vol[2 * b] = 0; stg.trader_id[2 * b] = 0; stg.price[2 * b] = 0;
      stg.action[2 * b + 1] = BK_ACTION_MODIFY; stg.order_id[2 * b + 1] = 0; stg.side[2 * b + 1] = 4; stg.vol[2 * b + 1] = 25;
      stg.trader_id[2 * b + 1] = 0; stg.price[2 * b + 1] = 0;
    }
    stg.action[2 * 7] = 3;                 /* book 7: action 3 is nothing the reference knows: a no-op, its ask 112 stays */
    CHECK(bk_submit_instructions_host(env, stg.book_offsets, stg.action, stg.side, stg.vol, stg.trader_id, stg.price, stg.order_id, &t2));
    EXPECT(t2 == 1);
    CHECK(bk_step_async(env));
    const uint64_t* vids = NULL;
    const uint32_t* vst = NULL;
    CHECK(bk_submit_result_view(env, t2, &vids, &vst, &bad));
    EXPECT(bad == UINT32_MAX && vids != NULL && vids[0] == UINT64_MAX && vst[1] == 2);
    EXPECT(bk_submit_result(env, 7, ids, st, &bad) == BK_INVALID_ARGUMENT);   /* not a ticket */
    CHECK(bk_env_sync(env));
    CHECK(bk_level2(env, 0, NB, rec));
    EXPECT(rec[1] == 100 && rec[2] == 114 && rec[4] == 25 && rec[7 * lw + 2] == 112 && rec[5 * lw + 2] == UINT32_MAX);
    uint32_t fl[NB];
    CHECK(bk_book_flags(env, fl));
    EXPECT(fl[0] == 0 && fl[5] == BK_FLAG_UNKNOWN_ORDER);   /* book 5 never created id 1: dropped and flagged, as on the device entry */
    free(rec);
  }
  bk_env_destroy(env);
  printf("abi_smoke: ok (history checksum %u)\n", sum[0]);
  return 0;
}
