// host_pool.hpp - persistent host worker threads of a bk_env (plain C++17, no HIP): used by bourse_amd.hip, unit- and
// thread-sanitizer-tested on the CPU by tests/test_host_pool.py.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace bkd {

// Host threads for the per-book half of Env that stays on the host (tick check, id assignment, queueing, flattening
// the queues for upload): books are independent, each task owns a contiguous range of markets.  Workers persist for the
// life of the env (a std::thread per call costs more than the work of a small step).
class HostPool {
 public:
  explicit HostPool(unsigned n_workers) {
    for (unsigned i = 0; i < n_workers; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  unsigned threads() const { return static_cast<unsigned>(workers_.size()) + 1; }  // + the calling thread
  // fn(task) for task in [0, n_tasks); returns true when every task has finished, false (nothing run) when n_tasks does
  // not fit the 16-bit task counter of the control word.  One run() at a time.
  static constexpr unsigned MAX_TASKS = 0xFFFFu;
  bool run(unsigned n_tasks, const std::function<void(unsigned)>& fn) {
    if (n_tasks == 0) return true;
    if (n_tasks > MAX_TASKS) return false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_.store(&fn, std::memory_order_release);
      pending_ = n_tasks;
      ++generation_;
      // generation, bound and next index are ONE word: a worker that is still leaving the previous run() can only claim
      // a task by a compare-exchange on the value it read, so it can neither run a task of this generation twice nor
      // apply the old bound to the new counter (the hazard of separate n_tasks / next words)
      ctl_.store((generation_ << 32) | (static_cast<uint64_t>(n_tasks) << 16), std::memory_order_release);
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
    return true;
  }

 private:
  void work() {
    for (;;) {
      uint64_t cur = ctl_.load(std::memory_order_acquire);
      const unsigned n = static_cast<unsigned>((cur >> 16) & 0xFFFFu), t = static_cast<unsigned>(cur & 0xFFFFu);
      if (t >= n) return;
      if (!ctl_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
      // run() of the generation just claimed cannot return before this task is counted: fn_ is that generation's
      (*fn_.load(std::memory_order_acquire))(t);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::atomic<const std::function<void(unsigned)>*> fn_{nullptr};
  std::atomic<uint64_t> ctl_{0};  // [63:32] generation, [31:16] n_tasks, [15:0] next task index
  unsigned pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};


}  // namespace bkd
