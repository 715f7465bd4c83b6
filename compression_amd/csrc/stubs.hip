// Entry points whose kernels are not written yet return an explicit failure.
#include "../../include/tfc_hip.h"
#include "common.h"
extern "C" int tfc_gdn_backward(const void*, const void*, void*, int, int64_t, int64_t, const float*,
                                const float*, int, int, int, int, float*, float*, void*) {
  return tfc::fail("tfc_gdn_backward: kernel not built into this library yet");
}
