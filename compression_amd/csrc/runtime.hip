// Device-runtime helpers of the C ABI that are not kernels: compute-unit partitioning for model
// pipelines (include/tfc_hip.h, "HIP streams restricted to a subset of the compute units").
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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

extern "C" int tfc_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
  *stream = nullptr;
  if (!mask || words <= 0) return fail("empty CU mask");
  bool any = false;
  for (int i = 0; i < words; ++i) any |= mask[i] != 0u;
  if (!any) return fail("CU mask selects no compute unit");
  hipStream_t st = nullptr;
  TFC_HIP(hipExtStreamCreateWithCUMask(&st, static_cast<uint32_t>(words), mask));
  *stream = st;
  return 0;
}

extern "C" int tfc_stream_destroy(void* stream) {
  if (stream) TFC_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return 0;
}
