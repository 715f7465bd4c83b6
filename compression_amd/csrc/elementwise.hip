// Elementwise passes of the model pipelines around the transforms and the coder, one kernel each instead of the
// three to five torch kernels they were (every one a full pass over the tensor):
//   * the image into the analysis transform: tf.cast(x, dtype) / 255 (models/bls2017.py:164-170, bmshj2018.py:219-224);
//   * the reconstruction out of the synthesis transform: saturate_cast(round(x_hat * 255), uint8)
//     (bls2017.py:186-190, bmshj2018.py:262-264), the product rounded to the compute dtype as the reference's is;
//   * the table indexes of an indexed entropy model: cast(min(max(indexes, 0), L - 1), int32)
//     (continuous_indexed.py:272-296 `_normalize_indexes` + the int32 cast of `_flatten_indexes`, one index range).
// HBM-bound: bytes in + bytes out.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/tfc_hip.h"
#include "common.h"

namespace tfc {

__device__ inline float bf16_bits_to_float(unsigned short b) { return __builtin_bit_cast(float, static_cast<unsigned int>(b) << 16); }
__device__ inline unsigned short float_to_bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));       // round to nearest even
}

template <bool BF>
__global__ void image_to_unit_kernel(const uint8_t* x, void* y, long long n) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 4;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i + k >= n) break;
    const float v = static_cast<float>(x[i + k]) / 255.0f;
    if (BF) static_cast<unsigned short*>(y)[i + k] = float_to_bf16_bits(v);
    else static_cast<float*>(y)[i + k] = v;
  }
}

template <bool BF>
__global__ void unit_to_image_kernel(const void* x, uint8_t* y, long long n) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 4;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i + k >= n) break;
    float v;
    if (BF) v = bf16_bits_to_float(float_to_bf16_bits(bf16_bits_to_float(static_cast<const unsigned short*>(x)[i + k]) * 255.0f));
    else v = static_cast<const float*>(x)[i + k] * 255.0f;
    v = fminf(fmaxf(rintf(v), 0.f), 255.f);
    y[i + k] = static_cast<uint8_t>(v);
  }
}

template <bool BF>
__global__ void index_prepare_kernel(const void* idx, int32_t* out, long long n, float hi) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 4;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i + k >= n) break;
    const float v = BF ? bf16_bits_to_float(static_cast<const unsigned short*>(idx)[i + k]) : static_cast<const float*>(idx)[i + k];
    // maximum(v, 0) then minimum(., hi) as the bound ops do (a NaN index stays NaN there and is undefined as an int;
    // here it becomes 0), cast toward zero
    out[i + k] = static_cast<int32_t>(fminf(fmaxf(v, 0.f), hi));
  }
}

}  // namespace tfc

extern "C" int tfc_image_to_unit(const void* x, void* y, int dtype, int64_t n, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_image_to_unit: dtype must be 0 (float32) or 1 (bfloat16)");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 1023) / 1024);
  tfc::KernelTimer timer("elementwise", st);
  if (dtype == 1) hipLaunchKernelGGL(tfc::image_to_unit_kernel<true>, dim3(blocks), dim3(256), 0, st, static_cast<const uint8_t*>(x), y, static_cast<long long>(n));
  else hipLaunchKernelGGL(tfc::image_to_unit_kernel<false>, dim3(blocks), dim3(256), 0, st, static_cast<const uint8_t*>(x), y, static_cast<long long>(n));
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_unit_to_image(const void* x, int dtype, void* y, int64_t n, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_unit_to_image: dtype must be 0 (float32) or 1 (bfloat16)");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 1023) / 1024);
  tfc::KernelTimer timer("elementwise", st);
  if (dtype == 1) hipLaunchKernelGGL(tfc::unit_to_image_kernel<true>, dim3(blocks), dim3(256), 0, st, x, static_cast<uint8_t*>(y), static_cast<long long>(n));
  else hipLaunchKernelGGL(tfc::unit_to_image_kernel<false>, dim3(blocks), dim3(256), 0, st, x, static_cast<uint8_t*>(y), static_cast<long long>(n));
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_index_prepare(const void* indexes, int dtype, int32_t* out, int64_t n, int num_tables, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_index_prepare: dtype must be 0 (float32) or 1 (bfloat16)");
  if (num_tables < 1) return tfc::fail("tfc_index_prepare: num_tables must be positive");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 1023) / 1024);
  tfc::KernelTimer timer("elementwise", st);
  const float hi = static_cast<float>(num_tables - 1);
  if (dtype == 1) hipLaunchKernelGGL(tfc::index_prepare_kernel<true>, dim3(blocks), dim3(256), 0, st, indexes, out, static_cast<long long>(n), hi);
  else hipLaunchKernelGGL(tfc::index_prepare_kernel<false>, dim3(blocks), dim3(256), 0, st, indexes, out, static_cast<long long>(n), hi);
  TFC_HIP(hipGetLastError());
  return 0;
}
