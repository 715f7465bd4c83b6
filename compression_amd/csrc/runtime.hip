// Device-runtime helpers of the C ABI that are not kernels: the library's cache of released device memory.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/tfc_hip.h"
#include "common.h"

using tfc::fail;

extern "C" int tfc_cache_bytes(long long* bytes) {
  if (!bytes) return tfc::fail("tfc_cache_bytes: null argument");
  tfc::BlockCache& c = tfc::BlockCache::get();
  std::lock_guard<std::mutex> lock(c.mu);
  *bytes = static_cast<long long>(c.cached);
  return 0;
}

extern "C" int tfc_cache_trim(long long* released) {
  const size_t n = tfc::BlockCache::get().trim();
  if (released) *released = static_cast<long long>(n);
  return 0;
}
