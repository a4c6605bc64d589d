// CPU unit test of bkd::HostPool (bourse_amd/csrc/host_pool.hpp): every task runs exactly once, run() returns only when
// all are done, the pool is reusable back to back, and the caller takes part.  Built with -fsanitize=thread by
// tests/test_host_pool.py.
#include <atomic>
#include <cstdio>
#include <numeric>
#include <vector>

#include "../../bourse_amd/csrc/host_pool.hpp"

int main() {
  for (unsigned workers : {0u, 1u, 3u, 7u}) {
    bkd::HostPool pool(workers);
    if (pool.threads() != workers + 1) return 1;
    for (int round = 0; round < 200; ++round) {
      const unsigned n = 1 + (round * 7) % 23;
      std::vector<int> hits(n, 0);       // each element written by exactly one task
      std::atomic<long> sum{0};
      pool.run(n, [&](unsigned t) {
        hits[t] += 1;
        long s = 0;
        for (int i = 0; i < 1000 * ((int)t % 3 + 1); ++i) s += i;  // uneven task lengths
        sum += s > 0 ? 1 : 1;
      });
      for (int h : hits)
        if (h != 1) { std::printf("task ran %d times (workers %u round %d)\n", h, workers, round); return 2; }
      if (sum.load() != (long)n) return 3;
    }
    pool.run(0, [&](unsigned) {});  // no tasks: returns at once
    // task counts that GROW from one run() to the next with empty tasks: a worker still leaving run k must not claim
    // (or double-run) a task of run k + 1 under the old bound
    for (int round = 0; round < 20000; ++round) {
      const unsigned n = (round & 1) ? 64u : 1u;
      std::vector<std::atomic<int>> hits(n);
      for (auto& h : hits) h = 0;
      pool.run(n, [&](unsigned t) { hits[t].fetch_add(1); });
      for (unsigned t = 0; t < n; ++t)
        if (hits[t].load() != 1) { std::printf("stress: task %u ran %d times\n", t, hits[t].load()); return 4; }
    }
  }
  std::puts("host_pool ok");
  return 0;
}
