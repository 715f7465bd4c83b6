// Shared host-side helpers for the gfx950 C-ABI library (libtfc_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

namespace tfc {

std::string& last_error();
int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define TFC_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess)                                                         \
      return ::tfc::fail("HIP error %s at %s:%d (%s)", hipGetErrorString(e__),     \
                         __FILE__, __LINE__, #expr);                               \
  } while (0)

// Stream-ordered device buffer.  Freed on the stream it was allocated on.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipStream_t st = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), st(o.st) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; st = o.st; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  hipError_t alloc(size_t n, hipStream_t s) {
    release();
    keep_pool_memory();
    st = s;
    bytes = n;
    if (n == 0) n = 16;
    return hipMallocAsync(&p, n, s);
  }
  // The default stream-ordered pool hands freed memory back to the driver at the next
  // synchronisation unless its release threshold is raised; every encode/decode call would
  // then pay a fresh driver allocation for its slabs.  Done once per device.
  static void keep_pool_memory() {
    static thread_local int done_for = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev == done_for) return;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
      unsigned long long keep = ~0ull;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      // Never let the pool hand a block that was freed on stream A to stream B by making B wait for
      // A's pending work ("internal dependencies"): with independent coding steps in flight on different
      // streams that silently chains them one behind the other.
      int off = 0;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowInternalDependencies, &off);
    }
    done_for = dev;
  }
  void release() {
    if (p) (void)hipFreeAsync(p, st);
    p = nullptr;
    bytes = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Non-owning view of a piece of a DevBuf (several small buffers of a handle share one allocation:
// every stream-ordered allocation is a driver call on the host path of a coding step).
struct DevView {
  void* p = nullptr;
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Optional per-kernel timing (tfc_profile_enable): HIP events recorded on the
// launch stream around the named kernel; read back by tfc_profile_query.
struct KernelTimer {
  const char* name;
  hipStream_t st;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  KernelTimer(const char* name, hipStream_t st);
  ~KernelTimer();
};
bool profiling_enabled();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace tfc
