// out[i] += sum over rows of partial[row * stride + i], i < n — the second stage of the deterministic
// (fixed-order, no float atomics) parameter-gradient reductions.  A workgroup owns 32 consecutive
// outputs (one 128-byte line per row) and splits the rows over kSlices slices; a thread keeps four
// independent running sums so that its loads overlap.  (One thread walking all rows of an output is a
// serial chain of dependent-latency loads: with 256 - 2048 rows that took as long as the gradient kernels.)
#pragma once
#include <hip/hip_runtime.h>

namespace tfc {

constexpr int kRowSlices = 16;

static __global__ void __launch_bounds__(32 * kRowSlices) sum_rows_kernel(const float* partial, long long rows,
                                                                   long long stride, int n, float* out) {
  __shared__ float part[kRowSlices][32];
  const int o = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + o;
  float s = 0.f;
  if (idx < n) {
    const long long per = (rows + kRowSlices - 1) / kRowSlices;
    const long long r0 = slice * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float* src = partial + idx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long r = r0;
    for (; r + 4 <= r1; r += 4) {
      a0 += src[r * stride];
      a1 += src[(r + 1) * stride];
      a2 += src[(r + 2) * stride];
      a3 += src[(r + 3) * stride];
    }
    for (; r < r1; ++r) a0 += src[r * stride];
    s = (a0 + a1) + (a2 + a3);
  }
  part[slice][o] = s;
  __syncthreads();
  if (slice == 0 && idx < n) {
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < kRowSlices; ++k) total += part[k][o];
    out[idx] += total;
  }
}

static inline void launch_sum_rows(const float* partial, long long rows, long long stride, int n, float* out,
                            hipStream_t st) {
  hipLaunchKernelGGL(sum_rows_kernel, dim3((n + 31) / 32), dim3(32 * kRowSlices), 0, st, partial, rows, stride, n,
                     out);
}

}  // namespace tfc
