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
  printf("abi_smoke: ok (history checksum %u)\n", sum[0]);
  return 0;
}
