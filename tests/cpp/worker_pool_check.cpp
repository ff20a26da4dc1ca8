// Exercises tdr_amd::WorkerPool / TaskGroup the way run_host_pipelined uses them (tasks that block on each other
// through a shared mutex / condition variable, pool reused across "calls", a throwing task, destruction with idle
// threads).  Built with -fsanitize=thread by tests/test_worker_pool_cpu.py; prints "ok" on success.
#include <atomic>
#include <cstdio>
#include <stdexcept>

#include "worker_pool.h"

using tdr_amd::TaskGroup;
using tdr_amd::WorkerPool;

static int one_call(WorkerPool& pool, int workers, int chunks) {
  // a miniature of the pipeline: `workers` stagers fill chunk k, a consumer takes chunks in order; every task
  // blocks on the others, so none of them may ever wait for a free pool thread
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> staged((size_t)chunks, 0);
  int consumed = 0;
  bool abort_all = false;
  long sum = 0;
  {
    TaskGroup group(pool);
    group.stop = [&] {
      std::lock_guard<std::mutex> lk(mu);
      abort_all = true;
      cv.notify_all();
    };
    group.run([&] {  // consumer
      for (int k = 0; k < chunks; ++k) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return staged[(size_t)k] == workers || abort_all; });
        if (abort_all) return;
        sum += k;
        consumed = k + 1;
        cv.notify_all();
      }
    });
    for (int t = 0; t < workers; ++t)
      group.run([&, t] {
        for (int k = 0; k < chunks; ++k) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return consumed + 4 > k || abort_all; });  // at most 4 chunks ahead
          if (abort_all) return;
          if (++staged[(size_t)k] == workers) cv.notify_all();
        }
      });
    pool.wait(group.tickets[0]);
    group.wait_all();
    group.stop = nullptr;
  }
  return sum == (long)chunks * (chunks - 1) / 2 ? 0 : 1;
}

int main() {
  int bad = 0;
  {
    WorkerPool pool;
    for (int call = 0; call < 50; ++call) bad += one_call(pool, 1 + call % 8, 1 + call % 23);
    // a task that throws must not take the process down, and its ticket must still complete
    {
      TaskGroup group(pool);
      group.run([] { throw std::runtime_error("boom"); });
      group.wait_all();
      bad += group.failed.load() ? 0 : 1;  // ... and the group knows
    }
    // a throwing task releases the tasks that wait for it (the pipeline's stagers / copy-out workers wait for
    // counters only another task advances): the group calls `stop` as soon as the exception arrives
    {
      std::mutex mu;
      std::condition_variable cv;
      bool stop = false, produced = false;
      std::atomic<int> released{0};
      TaskGroup group(pool);
      group.stop = [&] {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
        cv.notify_all();
      };
      for (int t = 0; t < 3; ++t)
        group.run([&] {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return produced || stop; });  // `produced` never comes: its producer throws
          ++released;
        });
      group.run([] { throw std::runtime_error("producer died"); });
      group.wait_all();  // would hang without the stop-on-exception rule
      bad += (released.load() == 3 && group.failed.load()) ? 0 : 1;
      group.stop = nullptr;
    }
    // early exit: the group's destructor stops and waits for tasks that would otherwise block forever
    {
      std::mutex mu;
      std::condition_variable cv;
      bool stop = false;
      std::atomic<int> exited{0};
      {
        TaskGroup group(pool);
        group.stop = [&] {
          std::lock_guard<std::mutex> lk(mu);
          stop = true;
          cv.notify_all();
        };
        for (int t = 0; t < 5; ++t)
          group.run([&] {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop; });
            ++exited;
          });
      }
      bad += exited.load() == 5 ? 0 : 1;
    }
    bad += one_call(pool, 8, 40);
  }  // pool destroyed with idle threads
  {
    WorkerPool unused;  // never used: no threads
  }
  std::puts(bad == 0 ? "ok" : "FAILED");
  return bad;
}
