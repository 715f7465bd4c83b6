// Tails and quantisation offset of a deep factorized prior, on the device in one launch.
//
// The range-coding tables of a ContinuousBatchedEntropyModel over a NoisyDeepFactorized prior need, per channel, the
// points where the logits of the cumulative reach log(t / 2 / (1 - t / 2)), its negative, and 0
// (python/distributions/deep_factorized.py:222-246 `_lower_tail` / `_upper_tail` / `_quantization_offset`), which the
// reference finds with `helpers.estimate_tails` (python/distributions/helpers.py:29-104): an Adam-like iteration on
// |logits(x) - target| run for ALL channels together until every channel's loss is below 1e-8 or every channel has made
// 100 steps past its first sign change.  As tensor ops that is a few hundred iterations of ~60 two-microsecond kernels
// plus two host read-backs each, three times per model (profiles/r04_bls2017_stats.md: 18 000 launches) — the
// "PMF/CDF-build" stage of the path is launch-bound, not compute-bound.  Here one workgroup per target runs the whole
// iteration: a thread per channel keeps (x, m, v, count, best) in registers and evaluates the channel's monotone MLP
// (deep_factorized.py:166-194) and its derivative analytically; the two global conditions of the loop are workgroup
// reductions through LDS.  Same update rule, same stopping rule, same "best iterate seen" result; the arithmetic is
// float32 like the reference's, the derivative is the forward-mode product instead of autodiff's reverse sweep (a few
// ulp apart: the result only enters the tables through floor / ceil of the tails and a rounded offset).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "../../include/tfc_hip.h"
#include "common.h"

namespace tfc {

constexpr int kTailMaxWidth = 8;        // hidden widths up to this (the reference's default is 3)
constexpr int kTailMaxIters = 100000;   // the reference's loop has no bound; a channel that never converges must not hang the GPU

// logits of the cumulative and d logits / dx at x.  params (per channel, bottleneck_ops.pack_factorized_params):
// layer 0: m[W], b[W], a[W]; layers 1 .. K-2: m[W][W] (row = output), b[W], a[W]; layer K-1: m[W], b[1] — m = softplus of
// the stored matrix, a = tanh of the stored factor.
__device__ inline void tails_mlp(const float* p, int K, int W, float x, float* f, float* df) {
  float h[kTailMaxWidth], d[kTailMaxWidth];
  for (int i = 0; i < W; ++i) {
    const float pre = p[i] * x + p[W + i];
    const float t = tanhf(pre);
    const float a = p[2 * W + i];
    h[i] = pre + a * t;
    d[i] = p[i] * (1.f + a * (1.f - t * t));
  }
  const float* q = p + 3 * W;
  for (int l = 1; l < K - 1; ++l) {
    float hn[kTailMaxWidth], dn[kTailMaxWidth];
    for (int i = 0; i < W; ++i) {
      float pre = 0.f, dpre = 0.f;
      for (int j = 0; j < W; ++j) {
        pre += q[i * W + j] * h[j];
        dpre += q[i * W + j] * d[j];
      }
      pre += q[W * W + i];
      const float t = tanhf(pre);
      const float a = q[W * W + W + i];
      hn[i] = pre + a * t;
      dn[i] = dpre * (1.f + a * (1.f - t * t));
    }
    for (int i = 0; i < W; ++i) { h[i] = hn[i]; d[i] = dn[i]; }
    q += W * W + 2 * W;
  }
  float out = 0.f, dout = 0.f;
  for (int j = 0; j < W; ++j) {
    out += q[j] * h[j];
    dout += q[j] * d[j];
  }
  *f = out + q[W];
  *df = dout;
}

// grid = targets, block = channels rounded up to a wave (<= 1024)
__global__ void __launch_bounds__(1024) deep_factorized_tails_kernel(const float* params, int channels, int P, int K, int W,
                                                                     const float* targets, float* out, int* iterations) {
  __shared__ float red_loss[16];
  __shared__ int red_count[16];
  __shared__ int go;
  const int c = threadIdx.x;
  const bool live = c < channels;
  const float target = targets[blockIdx.x];
  const float* p = params + static_cast<size_t>(live ? c : 0) * P;
  float x = 0.f, m = 0.f, v = 1.f, best_x = 0.f, best_loss = FLT_MAX, loss = FLT_MAX;
  int count = 0, it = 0;
  for (;;) {
    // while loss.max() > 1e-8 and count.min() < 100 (helpers.py:78) — over the live channels
    // (reduce_max propagates a NaN loss and `NaN > 1e-8` is false: the reference's loop ends at once and returns the
    // best so far of every channel; fmaxf would drop the NaN, so it travels as +inf's complement: a flag)
    float wl = live ? loss : 0.f;
    int wc = live ? count : 0x7FFFFFFF;
    const bool wn = __any(live && loss != loss);
    for (int off = 32; off > 0; off >>= 1) {
      wl = fmaxf(wl, __shfl_xor(wl, off, 64));
      wc = min(wc, __shfl_xor(wc, off, 64));
    }
    // (a wave's NaN flag rides in its count word: -1 is below every count, tested before the minimum is used)
    if ((threadIdx.x & 63) == 0) { red_loss[threadIdx.x >> 6] = wl; red_count[threadIdx.x >> 6] = wn ? -1 : wc; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ml = 0.f;
      int mc = 0x7FFFFFFF;
      bool nan = false;
      for (unsigned int w = 0; w < (blockDim.x + 63) / 64; ++w) {
        ml = fmaxf(ml, red_loss[w]);
        nan |= red_count[w] < 0;
        mc = min(mc, red_count[w]);
      }
      go = (!nan && ml > 1e-8f && mc < 100 && it < kTailMaxIters) ? 1 : 0;
    }
    __syncthreads();
    const int cont = go;
    __syncthreads();
    if (!cont) break;
    float f, df;
    tails_mlp(p, K, W, x, &f, &df);
    const float u = f - target;
    loss = fabsf(u);
    const float grad = (u > 0.f ? df : (u < 0.f ? -df : 0.f));      // d|u|/dx, sign(0) = 0 as autograd has it
    if (loss < best_loss) { best_x = x; best_loss = loss; }
    const float prev_m = m;
    m = (m + grad) / 2.f;
    v = (v + grad * grad) / 2.f;
    const float k = sqrtf(static_cast<float>(count + 1));
    x = x - 0.1f * m / (k * sqrtf(v) + 1e-20f);
    count = (count > 0 || prev_m * grad < 0.f) ? count + 1 : count;
    ++it;
  }
  if (live) out[static_cast<size_t>(blockIdx.x) * channels + c] = best_x;
  if (threadIdx.x == 0 && iterations) iterations[blockIdx.x] = it;
}

}  // namespace tfc

extern "C" int tfc_deep_factorized_tails(const float* params, int64_t channels, int64_t params_per_channel, int layers,
                                         int width, const float* targets, int num_targets, float* out, int* iterations,
                                         void* stream) {
  using namespace tfc;
  if (channels < 1 || channels > 1024)
    return fail("tfc_deep_factorized_tails: 1 ... 1024 channels (one workgroup iterates them together), got %lld",
                static_cast<long long>(channels));
  if (layers < 2 || width < 1 || width > kTailMaxWidth)
    return fail("tfc_deep_factorized_tails: layers >= 2 and hidden width 1 ... %d (got %d, %d)", kTailMaxWidth, layers, width);
  const long long want = 3ll * width + (layers - 2) * (static_cast<long long>(width) * width + 2 * width) + width + 1;
  if (params_per_channel != want)
    return fail("tfc_deep_factorized_tails: %lld parameters per channel, the layout has %lld",
                static_cast<long long>(params_per_channel), want);
  if (num_targets < 1) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned block = static_cast<unsigned>((channels + 63) / 64 * 64);
  hipLaunchKernelGGL(deep_factorized_tails_kernel, dim3(static_cast<unsigned>(num_targets)), dim3(block), 0, st, params,
                     static_cast<int>(channels), static_cast<int>(params_per_channel), layers, width, targets, out, iterations);
  TFC_HIP(hipGetLastError());
  return 0;
}
