// Build shim (test infrastructure) for thread::ThreadPool::ParallelFor: runs the whole range on the
// calling thread (the rows are independent). Not product code.
#pragma once
#include <cstdint>
#include <functional>
namespace tensorflow {
namespace thread {
class ThreadPool {
 public:
  void ParallelFor(int64_t total, int64_t /*cost_per_unit*/, const std::function<void(int64_t, int64_t)>& fn) {
    if (total > 0) fn(0, total);
  }
};
}  // namespace thread
struct CpuWorkerThreads {
  thread::ThreadPool* workers;
};
class DeviceBase {
 public:
  const CpuWorkerThreads* tensorflow_cpu_worker_threads() const { return &threads_; }
 private:
  thread::ThreadPool pool_;
  CpuWorkerThreads threads_{&pool_};
};
}  // namespace tensorflow
