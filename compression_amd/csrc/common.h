// Shared host-side helpers for the gfx950 C-ABI library (libtfc_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace tfc {

std::string& last_error();
int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// Diagnostics: TFC_SLOW_CALL_MS=<ms> in the environment reports (stderr) every HIP runtime call of the library,
// buffer allocation and kernel-launch scope that holds the calling host thread longer than that — how a call that
// was meant to only enqueue work is found blocking (e.g. a copy from pageable host memory behind a long kernel).
double slow_call_threshold_ms();
struct SlowCall {
  const char* what;
  const char* file;
  int line;
  double t0;
  unsigned long long detail = 0;       // e.g. the byte count of an allocation
  SlowCall(const char* w, const char* f, int l);
  ~SlowCall();
};

#define TFC_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t e__;                                                                \
    {                                                                              \
      ::tfc::SlowCall slow__(#expr, __FILE__, __LINE__);                           \
      e__ = (expr);                                                                \
    }                                                                              \
    if (e__ != hipSuccess)                                                         \
      return ::tfc::fail("HIP error %s at %s:%d (%s)", hipGetErrorString(e__),     \
                         __FILE__, __LINE__, #expr);                               \
  } while (0)

// Stream-ordered device buffer.  Memory comes from the library's PRIVATE HIP memory pool (one per device, so
// that its attributes stay local to this library instead of changing the device's default pool under every other
// user of it) and, once released, is kept in the library's own per-stream free lists: a released block goes to
// the free list together with the stream that used it last (`st`; handles retarget it on every call — a buffer
// allocated under stream A and last read by a kernel on stream B must not be reused in A's order) and an event
// recorded on that stream, and is handed out again to an allocation made for that same stream (safe by the stream's
// own order) or, once the event has completed, for any stream.
//
// Why not hipFreeAsync: measured on ROCm 7.2 / MI355X (TFC_SLOW_CALL_MS, tools/pipeline_host_log.py), a
// hipFreeAsync issued behind a kernel that is running on the same stream holds the calling thread until that
// kernel has finished — the packed-weight buffer of every convolution and the encoder's 24-byte counter block
// each cost the enqueuing thread the whole kernel (15-40 ms per model stage), so that a thread feeding two
// streams could never run ahead of either.  Blocks beyond TFC_CACHE_LIMIT_MB (default 65536) of cached memory
// are returned with hipFreeAsync.
struct BlockCache {
  struct Block {
    void* p;
    hipStream_t st;        // the stream that used it last
    hipEvent_t ev;         // recorded on `st` at release: complete = free for any stream
  };
  struct Key {
    int dev;
    size_t cap;
    bool operator<(const Key& o) const { return dev != o.dev ? dev < o.dev : cap < o.cap; }
  };
  std::mutex mu;                                            // handles are released from several host threads
  std::map<Key, std::vector<Block>> free;
  std::vector<hipEvent_t> events;                           // recycled
  size_t cached = 0;
  size_t limit;
  BlockCache() {
    const char* e = std::getenv("TFC_CACHE_LIMIT_MB");
    limit = static_cast<size_t>(e ? std::atoll(e) : 65536) << 20;
  }
  static BlockCache& get() {
    static BlockCache* c = new BlockCache;                  // leaked: released blocks may arrive during exit
    return *c;
  }
  // 8 size classes per octave (at most 12.5 % over the request), 512 bytes at least
  static size_t size_class(size_t n) {
    if (n <= 512) return 512;
    const int lg = 63 - __builtin_clzll(static_cast<unsigned long long>(n));
    const size_t step = static_cast<size_t>(1) << (lg - 3);
    return (n + step - 1) & ~(step - 1);
  }
  // A cached block of this size class: the most recently released one that was last used on `st` itself (the
  // stream's order makes the reuse safe), else one whose last use has completed.
  void* take(int dev, hipStream_t st, size_t cap) {
    std::lock_guard<std::mutex> lock(mu);
    auto f = free.find(Key{dev, cap});
    if (f == free.end()) return nullptr;
    auto& v = f->second;
    const size_t scan = std::min<size_t>(v.size(), 32);
    size_t hit = v.size();
    for (size_t k = 0; k < scan && hit == v.size(); ++k)
      if (v[v.size() - 1 - k].st == st) hit = v.size() - 1 - k;
    for (size_t k = 0; k < scan && hit == v.size(); ++k)
      if (hipEventQuery(v[k].ev) == hipSuccess) hit = k;    // oldest first: the likeliest to be complete
    if (hit == v.size()) {
      (void)hipGetLastError();                              // hipErrorNotReady of the queries
      return nullptr;
    }
    void* p = v[hit].p;
    events.push_back(v[hit].ev);
    v.erase(v.begin() + static_cast<long>(hit));
    cached -= cap;
    return p;
  }
  // Returns every cached block whose last use has completed to the driver (hipFreeAsync on its last-use stream — behind
  // completed work that call does not hold the thread); -> bytes released.
  size_t trim() {
    std::lock_guard<std::mutex> lock(mu);
    size_t released = 0;
    for (auto& kv : free) {
      auto& v = kv.second;
      for (size_t k = v.size(); k-- > 0;) {
        if (hipEventQuery(v[k].ev) != hipSuccess) continue;
        (void)hipFreeAsync(v[k].p, v[k].st);
        events.push_back(v[k].ev);
        released += kv.first.cap;
        v.erase(v.begin() + static_cast<long>(k));
      }
    }
    (void)hipGetLastError();
    cached -= released;
    return released;
  }
  // false: over the limit, the caller frees the block
  bool give(int dev, hipStream_t st, size_t cap, void* p) {
    std::lock_guard<std::mutex> lock(mu);
    if (cached + cap > limit) return false;
    hipEvent_t ev = nullptr;
    if (!events.empty()) {
      ev = events.back();
      events.pop_back();
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      return false;
    }
    if (hipEventRecord(ev, st) != hipSuccess) {
      events.push_back(ev);
      return false;
    }
    free[Key{dev, cap}].push_back(Block{p, st, ev});
    cached += cap;
    return true;
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;        // as requested
  size_t cap = 0;          // the block's size class
  int dev = 0;
  hipStream_t st = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap), dev(o.dev), st(o.st) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p; bytes = o.bytes; cap = o.cap; dev = o.dev; st = o.st;
      o.p = nullptr; o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  hipError_t alloc(size_t n, hipStream_t s) {
    release();
    st = s;
    bytes = n;
    cap = BlockCache::size_class(n);
    hipMemPool_t mp = pool();
    dev = 0;
    (void)hipGetDevice(&dev);
    p = BlockCache::get().take(dev, s, cap);
    if (p) return hipSuccess;
    SlowCall slow("DevBuf::alloc", __FILE__, __LINE__);
    slow.detail = cap;
    hipError_t rc = mp ? hipMallocFromPoolAsync(&p, cap, mp, s) : hipMallocAsync(&p, cap, s);
    if (rc != hipSuccess) {
      // out of memory with idle blocks of other sizes in the library's own cache: hand those back and ask once more
      (void)hipGetLastError();
      p = nullptr;
      if (BlockCache::get().trim() > 0)
        rc = mp ? hipMallocFromPoolAsync(&p, cap, mp, s) : hipMallocAsync(&p, cap, s);
      if (rc != hipSuccess) p = nullptr;
    }
    return rc;
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
    if (p && !BlockCache::get().give(dev, st, cap, p)) {
      SlowCall slow("DevBuf::release", __FILE__, __LINE__);
      slow.detail = cap;
      (void)hipFreeAsync(p, st);
    }
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
  SlowCall slow;                       // the launch scope as seen by the host thread
  KernelTimer(const char* name, hipStream_t st);
  ~KernelTimer();
};
bool profiling_enabled();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace tfc
