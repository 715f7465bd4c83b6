// Device-runtime helpers of the C ABI that are not kernels: compute-unit partitioning for model
// pipelines (include/tfc_hip.h, "HIP streams restricted to a subset of the compute units").
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>
#include <vector>

#include "../../include/tfc_hip.h"
#include "common.h"

using tfc::fail;

extern "C" int tfc_device_compute_units(int* cus) {
  int dev = 0;
  TFC_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TFC_HIP(hipGetDeviceProperties(&prop, dev));
  *cus = prop.multiProcessorCount;
  return 0;
}

// Streams are never handed back to the runtime: coder handles release their buffers in the order of the
// stream that used them last (common.h, DevBuf), and a handle may outlive the pipeline object that made
// the stream — hipFreeAsync on a destroyed stream is a crash.  tfc_stream_destroy parks the stream, the
// next tfc_stream_create_cu_mask with the same mask takes it again (which also spares the hardware
// queue a re-creation).  A process uses a handful of masks.
namespace {
struct ParkedStream {
  int device;
  std::vector<uint32_t> mask;
  hipStream_t stream;
  bool in_use;
};
std::mutex g_streams_mu;
std::vector<ParkedStream>& streams() {
  static std::vector<ParkedStream>* v = new std::vector<ParkedStream>();     // intentionally leaked
  return *v;
}
}  // namespace

extern "C" int tfc_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
  *stream = nullptr;
  if (!mask || words <= 0) return fail("empty CU mask");
  bool any = false;
  for (int i = 0; i < words; ++i) any |= mask[i] != 0u;
  if (!any) return fail("CU mask selects no compute unit");
  int dev = 0;
  TFC_HIP(hipGetDevice(&dev));
  const std::vector<uint32_t> want(mask, mask + words);
  std::lock_guard<std::mutex> lock(g_streams_mu);
  for (ParkedStream& p : streams()) {
    if (!p.in_use && p.device == dev && p.mask == want) {
      p.in_use = true;
      *stream = p.stream;
      return 0;
    }
  }
  hipStream_t st = nullptr;
  TFC_HIP(hipExtStreamCreateWithCUMask(&st, static_cast<uint32_t>(words), mask));
  streams().push_back(ParkedStream{dev, want, st, true});
  *stream = st;
  return 0;
}

extern "C" int tfc_stream_destroy(void* stream) {
  if (!stream) return 0;
  TFC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  std::lock_guard<std::mutex> lock(g_streams_mu);
  for (ParkedStream& p : streams())
    if (p.stream == static_cast<hipStream_t>(stream)) p.in_use = false;
  return 0;
}

// Bytes the library keeps cached for reuse (DevBuf's free lists, csrc/common.h), and a way to hand the idle ones back.
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
