// PmfToQuantizedCdf on gfx950: one wavefront per PMF row.
//
// Follows cc/kernels/pmf_to_cdf_kernels.cc:159-208: quantise every mass to
// max(1, rint(p * 2^precision)), then move the sum to exactly 2^precision by
// repeatedly decrementing the item whose next decrement costs least
// (p * (log2 v - log2(v-1))) or incrementing the one whose next increment gains
// most (p * (log2(v+1) - log2 v)), then prefix-sum.
//
// The reference keeps the candidates in a sorted vector and re-inserts the
// touched item BEHIND every item with an equal key (find_if + rotate), i.e. a
// FIFO among ties; its initial order among ties is whatever std::sort
// produces.  Ties are the normal case (symmetric tables: pmf[i] == pmf[n-1-i])
// and decide which symbol of a tied pair is adjusted, so the initial order is
// reproduced too: lane 0 runs libstdc++'s introsort (sort_order.h, the library
// the reference's Linux builds use) over (key, symbol) in LDS — serial, but a
// table is built once per model — and the resulting rank is the item's first
// ticket; a touched item takes the next ticket.  Selection is then a
// wave-wide arg-min over (key, ticket), which visits items in exactly the
// order of the reference's vector.
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "../../include/tfc_hip.h"
#include "common.h"
#include "sort_order.h"

namespace tfc {

struct Best {
  double key;
  unsigned int ticket;
  int idx;
};

template <bool SHRINK>
__device__ inline bool better(const Best& a, const Best& b) {
  // SHRINK: smallest key first; GROW: largest key first; FIFO among equals.
  if (a.key != b.key) return SHRINK ? (a.key < b.key) : (a.key > b.key);
  return a.ticket < b.ticket;
}

template <bool SHRINK>
__device__ inline double key_of(double mass, int v) {
  if (SHRINK) {
    if (v <= 1) return INFINITY;
    return mass * (log2(static_cast<double>(v)) - log2(static_cast<double>(v - 1)));
  }
  if (v < 1) return -INFINITY;
  return mass * (log2(static_cast<double>(v + 1)) - log2(static_cast<double>(v)));
}

template <bool SHRINK>
__device__ void rebalance(const float* pmf, int n, int steps, int* v, double* key,
                          unsigned int* ticket, int lane) {
  for (int i = lane; i < n; i += 64) {
    key[i] = key_of<SHRINK>(static_cast<double>(pmf[i]), v[i]);
    ticket[i] = static_cast<unsigned int>(i);
  }
  __syncthreads();
  // initial order: position p of the sorted sequence holds symbol ticket[p]
  if (lane == 0) {
    using Before = typename std::conditional<SHRINK, KeyAscending, KeyDescending>::type;
    SortOrder<Before>{key, ticket, Before{}}.sort(n);
  }
  __syncthreads();
  // invert into rank-by-symbol (through the key array, which is rebuilt afterwards)
  unsigned int* rank = reinterpret_cast<unsigned int*>(key);
  for (int pos = lane; pos < n; pos += 64) rank[ticket[pos]] = static_cast<unsigned int>(pos);
  __syncthreads();
  for (int i = lane; i < n; i += 64) ticket[i] = rank[i];
  __syncthreads();
  for (int i = lane; i < n; i += 64) key[i] = key_of<SHRINK>(static_cast<double>(pmf[i]), v[i]);
  __syncthreads();
  unsigned int next_ticket = static_cast<unsigned int>(n);
  for (int it = 0; it < steps; ++it) {
    Best b;
    b.key = SHRINK ? INFINITY : -INFINITY;
    b.ticket = 0xFFFFFFFFu;
    b.idx = -1;
    for (int i = lane; i < n; i += 64) {
      Best c{key[i], ticket[i], i};
      if (b.idx < 0 || better<SHRINK>(c, b)) b = c;
    }
    for (int off = 32; off > 0; off >>= 1) {
      Best o;
      o.key = __shfl_xor(b.key, off, 64);
      o.ticket = __shfl_xor(b.ticket, off, 64);
      o.idx = __shfl_xor(b.idx, off, 64);
      if (o.idx >= 0 && (b.idx < 0 || better<SHRINK>(o, b))) b = o;
    }
    // every lane now holds the same winner
    if (lane == 0 && b.idx >= 0) {
      const int i = b.idx;
      v[i] += SHRINK ? -1 : 1;
      key[i] = key_of<SHRINK>(static_cast<double>(pmf[i]), v[i]);
      ticket[i] = next_ticket;
    }
    ++next_ticket;
    __syncthreads();  // one wave per block: orders lane 0's LDS update before the next sweep
  }
}

__global__ void __launch_bounds__(64) pmf_to_cdf_kernel(const float* pmf, int64_t rows, int n,
                                                        int precision, int32_t* cdf) {
  extern __shared__ unsigned char smem[];
  double* key = reinterpret_cast<double*>(smem);
  int* v = reinterpret_cast<int*>(key + n);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(v + n);
  const int lane = threadIdx.x;
  const int64_t r = blockIdx.x;
  const float* p = pmf + r * n;
  int32_t* out = cdf + r * (n + 1);
  const int total = 1 << precision;

  int sum = 0;
  for (int i = lane; i < n; i += 64) {
    // float * int -> float product, rint in float, like the reference's
    // std::rint(mass * normalizer).
    int q = static_cast<int>(rintf(p[i] * static_cast<float>(total)));
    q = max(q, 1);
    v[i] = q;
    sum += q;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  __syncthreads();
  if (sum > total) {
    rebalance<true>(p, n, sum - total, v, key, ticket, lane);
  } else if (sum < total) {
    rebalance<false>(p, n, total - sum, v, key, ticket, lane);
  }
  __syncthreads();
  // inclusive prefix sum, 64 items per sweep
  if (lane == 0) out[0] = 0;
  int carry = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    int x = i < n ? v[i] : 0;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (i < n) out[i + 1] = carry + x;
    carry += __shfl(x, 63, 64);
  }
}

// ---- a whole model's tables in one launch (continuous_base.py:217-296 `_build_tables`) ---------------------------
// Row r of `pmf` [rows, stride] holds lengths[r] probabilities (the prior sampled on the row's integer support); the
// reference appends the mass the support does not cover,
//     overflow = max(1 - sum(pmf[:length]), 0)                                        (continuous_base.py:277-279)
// runs PmfToQuantizedCdf on the length + 1 values and writes the row as [-precision, cdf...] into a ragged 1-D table.
// Here a wave does all of that for its row: the sum in float32 in a FIXED order (lane l adds elements l, l + 64, ... in
// turn, then the 64 partial sums are combined by the xor butterfly 32, 16, ..., 1 — what tests/test_tables_gpu.py
// restates in numpy), the quantisation of pmf_to_cdf_kernel above on an LDS copy of the row, and the header + cdf at
// out[offsets[r]] — offsets are the caller's prefix sums of length + 3.
// `overflow` (or null): the caller's overflow mass per row — a prior in another dtype than float32 sums in ITS arithmetic
// before the cast (continuous_base.py:277-279: reduce_sum in prior.dtype, then tf.cast(..., float32)).
__global__ void __launch_bounds__(64) pmf_to_cdf_ragged_kernel(const float* pmf, int64_t stride, const int32_t* lengths,
                                                               const int64_t* offsets, int precision, const float* overflow,
                                                               int32_t* out) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t r = blockIdx.x;
  const int len = lengths[r];
  const int n = len + 1;
  double* key = reinterpret_cast<double*>(smem);
  int* v = reinterpret_cast<int*>(key + n);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(v + n);
  float* p = reinterpret_cast<float*>(ticket + n);
  const float* src = pmf + r * stride;
  float part = 0.f;
  for (int i = lane; i < len; i += 64) {
    const float x = src[i];
    p[i] = x;
    part += x;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) p[len] = overflow ? overflow[r] : fmaxf(1.f - part, 0.f);
  __syncthreads();
  const int total = 1 << precision;
  int sum = 0;
  for (int i = lane; i < n; i += 64) {
    int q = static_cast<int>(rintf(p[i] * static_cast<float>(total)));
    q = max(q, 1);
    v[i] = q;
    sum += q;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  __syncthreads();
  if (sum > total) {
    rebalance<true>(p, n, sum - total, v, key, ticket, lane);
  } else if (sum < total) {
    rebalance<false>(p, n, total - sum, v, key, ticket, lane);
  }
  __syncthreads();
  int32_t* dst = out + offsets[r];
  if (lane == 0) { dst[0] = -precision; dst[1] = 0; }
  int carry = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    int x = i < n ? v[i] : 0;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (i < n) dst[i + 2] = carry + x;
    carry += __shfl(x, 63, 64);
  }
}

}  // namespace tfc

extern "C" int tfc_build_tables(const float* pmf, int64_t rows, int64_t stride, const int32_t* lengths,
                                const int64_t* offsets, int64_t max_length, int precision, int32_t* out, void* stream) {
  return tfc_build_tables_overflow(pmf, rows, stride, lengths, offsets, max_length, precision, nullptr, out, stream);
}

extern "C" int tfc_build_tables_overflow(const float* pmf, int64_t rows, int64_t stride, const int32_t* lengths,
                                         const int64_t* offsets, int64_t max_length, int precision, const float* overflow,
                                         int32_t* out, void* stream) {
  using namespace tfc;
  if (!(0 < precision && precision <= 16))
    return fail("`precision` must be in [1, 16]: %d", precision);
  if (max_length < 1 || max_length > stride) return fail("tfc_build_tables: max_length must be in [1, stride]");
  if (rows == 0) return 0;
  const size_t n = static_cast<size_t>(max_length) + 1;
  const size_t lds = n * (sizeof(double) + 2 * sizeof(int) + sizeof(float));
  if (lds > 160 * 1024)
    return fail("`pmf` rows of %lld elements exceed the on-chip table builder's limit (%d)",
                static_cast<long long>(n), static_cast<int>(160 * 1024 / 20));
  hipStream_t st = static_cast<hipStream_t>(stream);
  TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pmf_to_cdf_ragged_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(pmf_to_cdf_ragged_kernel, dim3(static_cast<unsigned>(rows)), dim3(64), lds, st, pmf, stride,
                     lengths, offsets, precision, overflow, out);
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_pmf_to_quantized_cdf(const float* pmf, int64_t rows, int64_t n, int precision,
                                        int32_t* cdf, void* stream) {
  using namespace tfc;
  if (!(0 < precision && precision <= 16))
    return fail("`precision` must be in [1, 16]: %d", precision);
  if (n <= 1) return fail("`pmf` size should be at least 2 in the last axis.");
  if (rows == 0) return 0;
  const size_t lds = static_cast<size_t>(n) * (sizeof(double) + 2 * sizeof(int));
  if (lds > 160 * 1024)
    return fail("`pmf` rows of %lld elements exceed the on-chip table builder's limit (%d)",
                static_cast<long long>(n), static_cast<int>(160 * 1024 / 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
  TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pmf_to_cdf_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(pmf_to_cdf_kernel, dim3(static_cast<unsigned>(rows)), dim3(64), lds, st, pmf,
                     rows, static_cast<int>(n), precision, cdf);
  TFC_HIP(hipGetLastError());
  return 0;
}
