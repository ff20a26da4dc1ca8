// worker_pool.h — helper threads of the pipelined host-buffer path (tdr_capi.cpp: run_host_pipelined).  Plain C++17,
// no HIP: tests/test_worker_pool_cpu.py builds it with -fsanitize=thread.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace tdr_amd {

// Helper threads of the pipelined host-buffer path, kept between calls: a pipelined call uses up to 17 of them
// (downloader, staging and copy-out workers), and creating them afresh cost ~0.4 ms per call — a fifth of a
// 1 M-system matrix call.  The pool grows on demand (a task never waits for a busy worker: the pipeline's tasks
// block on each other), threads sleep on a condition variable between calls and are joined with the context.
class WorkerPool {
 public:
  struct Task {
    std::function<void()> fn;
    bool done = false;
  };
  using Ticket = std::shared_ptr<Task>;
  WorkerPool() = default;
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stopping_ = true;
    }
    cv_work_.notify_all();
    for (auto& t : threads_)
      if (t.joinable()) t.join();
  }
  Ticket submit(std::function<void()> fn) {
    Ticket t = std::make_shared<Task>();
    t->fn = std::move(fn);
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (idle_ <= (int)queue_.size()) threads_.emplace_back([this] { run(); });  // may throw: nothing queued yet
      queue_.push_back(t);
    }
    cv_work_.notify_one();
    return t;
  }
  void wait(const Ticket& t) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return t->done; });
  }

 private:
  void run() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      ++idle_;
      cv_work_.wait(lk, [&] { return stopping_ || !queue_.empty(); });
      --idle_;
      if (queue_.empty()) return;  // stopping
      Ticket t = std::move(queue_.front());
      queue_.pop_front();
      lk.unlock();
      try {
        t->fn();
      } catch (...) {  // the pipeline's tasks report through their own state; never let one take the process down
      }
      lk.lock();
      t->done = true;
      cv_done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::deque<Ticket> queue_;
  std::vector<std::thread> threads_;
  int idle_ = 0;
  bool stopping_ = false;
};

// waits for every task of a group on scope exit (after telling them to stop), so that an exception or an early
// return between two submissions can never leave a task running on the caller's stack frame
//
// A task that throws (std::system_error from a lock, bad_alloc from a trace line) must not strand the others: the
// pool itself swallows what a task lets out, so the group wraps every task — the exception is recorded in `failed`
// and `stop` is called at once, which releases everybody who waits on the group's shared state; the caller checks
// failed after wait_all().  (Round 2 left this to the pool's catch-all: the main thread then waited for a chunk that
// would never be staged — ADVICE r2.)
struct TaskGroup {
  WorkerPool& pool;
  std::vector<WorkerPool::Ticket> tickets;
  std::function<void()> stop;
  std::atomic<bool> failed{false};
  explicit TaskGroup(WorkerPool& p) : pool(p) {}
  void run(std::function<void()> fn) {
    tickets.push_back(pool.submit([this, fn = std::move(fn)] {
      try {
        fn();
      } catch (...) {
        failed.store(true);
        try {
          if (stop) stop();
        } catch (...) {
        }
      }
    }));
  }
  void wait_all() {
    for (auto& t : tickets) pool.wait(t);
  }
  ~TaskGroup() {
    if (stop) stop();
    wait_all();
  }
};

}  // namespace tdr_amd
