// Shared host-side helpers for the gfx950 C-ABI library (libtfc_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
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

// Stream-ordered device buffer from the library's PRIVATE memory pool (one per device): the pool keeps
// freed blocks (release threshold = max) and never chains streams through "internal dependencies" —
// attributes that stay local to this library instead of changing the device's default pool under every
// other user of it.  Freed on the stream that used it last (`st`; handles retarget it on every call:
// a buffer allocated under stream A and last read by a kernel on stream B must not be returned to the
// pool in A's order).
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
    st = s;
    bytes = n;
    if (n == 0) n = 16;
    hipMemPool_t mp = pool();
    return mp ? hipMallocFromPoolAsync(&p, n, mp, s) : hipMallocAsync(&p, n, s);
  }
  // The library's pool on the current device, created on first use.  A stream-ordered pool hands freed
  // memory back to the driver at the next synchronisation unless its release threshold is raised (every
  // encode / decode call would then pay a fresh driver allocation for its slabs), and by default lets a
  // block freed on stream A go to stream B by making B wait for A's pending work — with independent
  // coding steps in flight on different streams that silently chains them one behind the other.
  static hipMemPool_t pool() {
    constexpr int kMaxDevices = 64;
    static hipMemPool_t pools[kMaxDevices] = {};
    static bool tried[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    static std::mutex mu;             // handles are created from several host threads
    std::lock_guard<std::mutex> lock(mu);
    if (!tried[dev]) {
      tried[dev] = true;
      hipMemPoolProps props = {};
      props.allocType = hipMemAllocationTypePinned;
      props.handleTypes = hipMemHandleTypeNone;
      props.location.type = hipMemLocationTypeDevice;
      props.location.id = dev;
      hipMemPool_t mp = nullptr;
      if (hipMemPoolCreate(&mp, &props) == hipSuccess && mp) {
        unsigned long long keep = ~0ull;
        int off = 0;
        (void)hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep);
        (void)hipMemPoolSetAttribute(mp, hipMemPoolReuseAllowInternalDependencies, &off);
        pools[dev] = mp;
      } else {
        // no private pool on this runtime: the device's default pool with the same two attributes
        (void)hipGetLastError();
        hipMemPool_t dflt = nullptr;
        if (hipDeviceGetDefaultMemPool(&dflt, dev) == hipSuccess && dflt) {
          unsigned long long keep = ~0ull;
          int off = 0;
          (void)hipMemPoolSetAttribute(dflt, hipMemPoolAttrReleaseThreshold, &keep);
          (void)hipMemPoolSetAttribute(dflt, hipMemPoolReuseAllowInternalDependencies, &off);
        }
      }
    }
    return pools[dev];
  }
  void release() {
    if (p) (void)hipFreeAsync(p, st);
    p = nullptr;
    bytes = 0;
  }
  // the stream whose order a later release() must follow
  void touch(hipStream_t s) { if (p) st = s; }
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
