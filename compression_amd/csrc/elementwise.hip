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

// A thread takes 16 consecutive elements (8 for the indexes): one 16-byte load on the narrow side, two 16-byte stores or
// loads on the wide one — a wave instruction is 1 KB contiguous, whole lines.  (Round 6.  Before: four elements per thread,
// one at a time — 1-, 2- and 4-byte accesses, 64 partial lines per store instruction: 1.5 TB/s of their bytes; the
// arithmetic per element is unchanged.)  The last, partial group of a tensor and tensors whose base is not 16-byte aligned
// go element by element.
typedef __attribute__((ext_vector_type(4))) unsigned int ew_u32x4;

template <bool BF>
__device__ inline void image_to_unit_one(const uint8_t* x, void* y, long long i) {
  const float v = static_cast<float>(x[i]) / 255.0f;
  if (BF) static_cast<unsigned short*>(y)[i] = float_to_bf16_bits(v);
  else static_cast<float*>(y)[i] = v;
}

template <bool BF>
__global__ void image_to_unit_kernel(const uint8_t* x, void* y, long long n, int vec) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 16;
  if (i >= n) return;
  if (vec && i + 16 <= n) {
    const ew_u32x4 in = *reinterpret_cast<const ew_u32x4*>(x + i);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = static_cast<float>((in[k >> 2] >> (8 * (k & 3))) & 0xFFu) / 255.0f;
    if (BF) {
      ew_u32x4 o[2];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        o[k >> 2][k & 3] = static_cast<unsigned int>(float_to_bf16_bits(v[2 * k])) |
                           (static_cast<unsigned int>(float_to_bf16_bits(v[2 * k + 1])) << 16);
      ew_u32x4* const dst = reinterpret_cast<ew_u32x4*>(static_cast<unsigned short*>(y) + i);
      dst[0] = o[0];
      dst[1] = o[1];
    } else {
      ew_u32x4* const dst = reinterpret_cast<ew_u32x4*>(static_cast<float*>(y) + i);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[q] = ew_u32x4{__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3])};
    }
    return;
  }
  for (int k = 0; k < 16 && i + k < n; ++k) image_to_unit_one<BF>(x, y, i + k);
}

template <bool BF>
__device__ inline unsigned int unit_to_image_value(float xv) {
  float v = BF ? bf16_bits_to_float(float_to_bf16_bits(xv * 255.0f)) : xv * 255.0f;
  v = fminf(fmaxf(rintf(v), 0.f), 255.f);
  return static_cast<unsigned int>(static_cast<uint8_t>(v));
}

template <bool BF>
__global__ void unit_to_image_kernel(const void* x, uint8_t* y, long long n, int vec) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 16;
  if (i >= n) return;
  if (vec && i + 16 <= n) {
    float v[16];
    if (BF) {
      const ew_u32x4* const src = reinterpret_cast<const ew_u32x4*>(static_cast<const unsigned short*>(x) + i);
      const ew_u32x4 in[2] = {src[0], src[1]};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned int w = in[k >> 2][k & 3];
        v[2 * k] = __uint_as_float(w << 16);
        v[2 * k + 1] = __uint_as_float(w & 0xFFFF0000u);
      }
    } else {
      const ew_u32x4* const src = reinterpret_cast<const ew_u32x4*>(static_cast<const float*>(x) + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const ew_u32x4 w = src[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = __uint_as_float(w[e]);
      }
    }
    ew_u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k >> 2] |= unit_to_image_value<BF>(v[k]) << (8 * (k & 3));
    *reinterpret_cast<ew_u32x4*>(y + i) = o;
    return;
  }
  for (int k = 0; k < 16 && i + k < n; ++k) {
    const float xv = BF ? bf16_bits_to_float(static_cast<const unsigned short*>(x)[i + k]) : static_cast<const float*>(x)[i + k];
    y[i + k] = static_cast<uint8_t>(unit_to_image_value<BF>(xv));
  }
}

// maximum(v, 0) then minimum(., hi) as the bound ops do (a NaN index stays NaN there and is undefined as an int; here it
// becomes 0), cast toward zero
__device__ inline int32_t index_prepare_value(float v, float hi) { return static_cast<int32_t>(fminf(fmaxf(v, 0.f), hi)); }

template <bool BF>
__global__ void index_prepare_kernel(const void* idx, int32_t* out, long long n, float hi, int vec) {
  const long long i = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 8;
  if (i >= n) return;
  if (vec && i + 8 <= n) {
    float v[8];
    if (BF) {
      const ew_u32x4 in = *reinterpret_cast<const ew_u32x4*>(static_cast<const unsigned short*>(idx) + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = __uint_as_float(in[k] << 16);
        v[2 * k + 1] = __uint_as_float(in[k] & 0xFFFF0000u);
      }
    } else {
      const ew_u32x4* const src = reinterpret_cast<const ew_u32x4*>(static_cast<const float*>(idx) + i);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const ew_u32x4 w = src[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = __uint_as_float(w[e]);
      }
    }
    ew_u32x4* const dst = reinterpret_cast<ew_u32x4*>(out + i);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dst[q] = ew_u32x4{static_cast<unsigned int>(index_prepare_value(v[4 * q], hi)), static_cast<unsigned int>(index_prepare_value(v[4 * q + 1], hi)),
                        static_cast<unsigned int>(index_prepare_value(v[4 * q + 2], hi)), static_cast<unsigned int>(index_prepare_value(v[4 * q + 3], hi))};
    return;
  }
  for (int k = 0; k < 8 && i + k < n; ++k) {
    const float v = BF ? bf16_bits_to_float(static_cast<const unsigned short*>(idx)[i + k]) : static_cast<const float*>(idx)[i + k];
    out[i + k] = index_prepare_value(v, hi);
  }
}

// Spatial padding of an NHWC tensor in one pass (the pre-pad of SignalConv2D's `same_reflect` / pre-padded `same_zeros`
// modes, signal_conv.py:880-893 `tf.pad(outputs, padding, self._pad_mode)`): a thread copies one UNIT of a pixel's
// channel vector (16 bytes when the vector is a multiple of that, else one element) from the source pixel the mode
// names — REFLECT mirrors without repeating the edge sample (tf.pad "REFLECT"), CONSTANT writes zeros.
template <typename Unit>
__global__ void pad2d_kernel(const Unit* x, Unit* y, int h, int w, long long units, int top, int left, int oh, int ow,
                             int reflect, long long total) {
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (idx >= total) return;
  const long long u = idx % units;
  long long rest = idx / units;
  const int j = static_cast<int>(rest % ow); rest /= ow;
  const int i = static_cast<int>(rest % oh);
  const long long n = rest / oh;
  int si = i - top, sj = j - left;
  bool inside = si >= 0 && si < h && sj >= 0 && sj < w;
  if (reflect) {
    si = si < 0 ? -si : si;
    si = si >= h ? 2 * (h - 1) - si : si;
    sj = sj < 0 ? -sj : sj;
    sj = sj >= w ? 2 * (w - 1) - sj : sj;
    inside = true;
  }
  Unit v{};
  if (inside) v = x[((n * h + si) * w + sj) * units + u];
  y[idx] = v;
}

}  // namespace tfc

extern "C" int tfc_pad2d(const void* x, void* y, int elem_bytes, int64_t n, int64_t h, int64_t w, int64_t c, int top,
                         int bottom, int left, int right, int reflect, void* stream) {
  if (elem_bytes != 2 && elem_bytes != 4) return tfc::fail("tfc_pad2d: elements of 2 or 4 bytes");
  if (top < 0 || bottom < 0 || left < 0 || right < 0) return tfc::fail("tfc_pad2d: paddings must be non-negative");
  if (reflect && (top >= h || bottom >= h || left >= w || right >= w))
    return tfc::fail("tfc_pad2d: reflect padding must be smaller than the dimension (tf.pad REFLECT)");
  if (n <= 0 || c <= 0) return 0;
  const int64_t oh = h + top + bottom, ow = w + left + right;
  if (oh <= 0 || ow <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  tfc::KernelTimer timer("elementwise", st);
  const long long row_bytes = static_cast<long long>(c) * elem_bytes;
  const bool aligned = row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  const long long units = aligned ? row_bytes / 16 : c;
  const long long total = n * oh * ow * units;
  if (total >= (1ll << 31) * 256) return tfc::fail("tfc_pad2d: tensor too large for one launch");
  const unsigned blocks = static_cast<unsigned>((total + 255) / 256);
#define TFC_PAD_LAUNCH(Unit)                                                                                             \
  hipLaunchKernelGGL(tfc::pad2d_kernel<Unit>, dim3(blocks), dim3(256), 0, st, static_cast<const Unit*>(x),               \
                     static_cast<Unit*>(y), static_cast<int>(h), static_cast<int>(w), units, top, left,                  \
                     static_cast<int>(oh), static_cast<int>(ow), reflect, total)
  if (aligned) TFC_PAD_LAUNCH(uint4);
  else if (elem_bytes == 4) TFC_PAD_LAUNCH(uint32_t);
  else TFC_PAD_LAUNCH(uint16_t);
#undef TFC_PAD_LAUNCH
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_image_to_unit(const void* x, void* y, int dtype, int64_t n, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_image_to_unit: dtype must be 0 (float32) or 1 (bfloat16)");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 4095) / 4096);
  const int vec = reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  tfc::KernelTimer timer("elementwise", st);
  if (dtype == 1) hipLaunchKernelGGL(tfc::image_to_unit_kernel<true>, dim3(blocks), dim3(256), 0, st, static_cast<const uint8_t*>(x), y, static_cast<long long>(n), vec);
  else hipLaunchKernelGGL(tfc::image_to_unit_kernel<false>, dim3(blocks), dim3(256), 0, st, static_cast<const uint8_t*>(x), y, static_cast<long long>(n), vec);
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_unit_to_image(const void* x, int dtype, void* y, int64_t n, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_unit_to_image: dtype must be 0 (float32) or 1 (bfloat16)");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 4095) / 4096);
  const int vec = reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  tfc::KernelTimer timer("elementwise", st);
  if (dtype == 1) hipLaunchKernelGGL(tfc::unit_to_image_kernel<true>, dim3(blocks), dim3(256), 0, st, x, static_cast<uint8_t*>(y), static_cast<long long>(n), vec);
  else hipLaunchKernelGGL(tfc::unit_to_image_kernel<false>, dim3(blocks), dim3(256), 0, st, x, static_cast<uint8_t*>(y), static_cast<long long>(n), vec);
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_index_prepare(const void* indexes, int dtype, int32_t* out, int64_t n, int num_tables, void* stream) {
  if (dtype != 0 && dtype != 1) return tfc::fail("tfc_index_prepare: dtype must be 0 (float32) or 1 (bfloat16)");
  if (num_tables < 1) return tfc::fail("tfc_index_prepare: num_tables must be positive");
  if (n <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + 2047) / 2048);
  const int vec = reinterpret_cast<uintptr_t>(indexes) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  tfc::KernelTimer timer("elementwise", st);
  const float hi = static_cast<float>(num_tables - 1);
  if (dtype == 1) hipLaunchKernelGGL(tfc::index_prepare_kernel<true>, dim3(blocks), dim3(256), 0, st, indexes, out, static_cast<long long>(n), hi, vec);
  else hipLaunchKernelGGL(tfc::index_prepare_kernel<false>, dim3(blocks), dim3(256), 0, st, indexes, out, static_cast<long long>(n), hi, vec);
  TFC_HIP(hipGetLastError());
  return 0;
}
