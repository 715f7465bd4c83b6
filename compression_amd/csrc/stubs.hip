// Entry points whose kernels are not written yet return an explicit failure.
#include "../../include/tfc_hip.h"
#include "common.h"
extern "C" int tfc_conv2d_down(const void*, const void*, const float*, void*, int, int64_t, int64_t,
                               int64_t, int64_t, int64_t, int, int, int, void*) {
  return tfc::fail("tfc_conv2d_down: kernel not built into this library yet");
}
extern "C" int tfc_conv2d_up(const void*, const void*, const float*, void*, int, int64_t, int64_t,
                             int64_t, int64_t, int64_t, int, int, int, void*) {
  return tfc::fail("tfc_conv2d_up: kernel not built into this library yet");
}
extern "C" int tfc_gdn_backward(const void*, const void*, void*, int, int64_t, int64_t, const float*,
                                const float*, int, int, int, int, float*, float*, void*) {
  return tfc::fail("tfc_gdn_backward: kernel not built into this library yet");
}
