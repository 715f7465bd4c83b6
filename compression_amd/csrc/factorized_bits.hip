// Training-time path of the entropy bottleneck, fused:
//   y_hat = y + u,  u ~ U(-.5, .5)                    (python/ops/math_ops.py:157-216, expected_grads=False)
//   log p(y_hat) = log( c(y_hat + .5) - c(y_hat - .5) )  (python/distributions/uniform_noise.py:117-156)
//   c = sigmoid(logits), logits = the deep factorized per-channel monotone MLP
//                                                        (python/distributions/deep_factorized.py:166-194)
//   bits[unit] = - sum_unit log p / ln 2                 (python/entropy_models/continuous_batched.py:291-322)
// optionally with the Laplace-mixture tail of continuous_base.py:298-334 (the *_tail entry points, laplace_tail.h).
// The reference runs this as a few dozen TF kernels over the whole latent tensor; here it is one
// forward kernel (y, u -> y_hat, per-block partial sums of log p) and one backward kernel
// (y_hat, dL/dbits -> dL/dy_hat and per-channel gradients of the reparameterised MLP weights,
// accumulated in registers: a thread keeps one channel for its whole life), each followed by a
// small fixed-order reduction — results do not depend on scheduling.
//
// The MLP has K layers 1 -> W -> ... -> W -> 1 (K = len(num_filters) + 1, equal hidden widths W).
// `params` holds, per channel, the REPARAMETERISED values (softplus(matrix), bias, tanh(factor));
// the chain rule through softplus / tanh is the caller's (autograd on [C x P] tensors):
//   layer 0:      m[W], b[W], a[W]
//   layers 1..K-2: m[W][W] (row = output), b[W], a[W]
//   layer K-1:    m[W], b[1]
// Roofline: VALU (about 40 transcendentals per element); HBM traffic is 3 tensors forward, 2 backward.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include "../../include/tfc_hip.h"
#include "common.h"
#include "laplace_tail.h"
#include "reduce_rows.h"

namespace tfc {

template <int K, int W> struct MlpLayout {
  static constexpr int kMid = K - 2;
  static constexpr int kParams = 3 * W + kMid * (W * W + 2 * W) + W + 1;
  __host__ __device__ static constexpr int mid(int l) { return 3 * W + l * (W * W + 2 * W); }
  __host__ __device__ static constexpr int last() { return 3 * W + kMid * (W * W + 2 * W); }
};

__device__ inline float fast_tanh(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1); exact at +-inf, ~2 ulp elsewhere
  const float e = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
// log(1 + x) for |x| <= 1 by the hardware log: absolute error <= 1e-7 (where log1p would return
// ~x for tiny x this returns 0), far below what a sum of bits resolves; libm's log1pf costs
// several dozen instructions per call and there are six per element.
__device__ inline float fast_log1p(float x) { return __logf(1.f + x); }
__device__ inline float log_sigmoid(float x) {     // log(1 / (1 + exp(-x))), stable both ways
  return fminf(x, 0.f) - fast_log1p(__expf(-fabsf(x)));
}
__device__ inline float sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// logits of the cumulative at z; optionally keeps the activations the backward pass needs:
// h[l][i] = output of layer l (after the gate), th[l][i] = tanh of its pre-gate value.
template <int K, int W, bool KEEP>
__device__ inline float mlp_forward(const float* p, float z, float (*h)[W], float (*th)[W]) {
  using L = MlpLayout<K, W>;
  float cur[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const float pre = fmaf(p[i], z, p[W + i]);
    const float t = fast_tanh(pre);
    cur[i] = fmaf(p[2 * W + i], t, pre);
    if (KEEP) { h[0][i] = cur[i]; th[0][i] = t; }
  }
#pragma unroll
  for (int l = 0; l < L::kMid; ++l) {
    const float* q = p + L::mid(l);
    float nxt[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
      float pre = q[W * W + i];
#pragma unroll
      for (int j = 0; j < W; ++j) pre = fmaf(q[i * W + j], cur[j], pre);
      const float t = fast_tanh(pre);
      nxt[i] = fmaf(q[W * W + W + i], t, pre);
      if (KEEP) { h[l + 1][i] = nxt[i]; th[l + 1][i] = t; }
    }
#pragma unroll
    for (int i = 0; i < W; ++i) cur[i] = nxt[i];
  }
  const float* q = p + L::last();
  float out = q[W];
#pragma unroll
  for (int j = 0; j < W; ++j) out = fmaf(q[j], cur[j], out);
  return out;
}

// log( c(upper) - c(lower) ) with the survival-function switch of uniform_noise.py:134-156
__device__ inline float log_interval(float upper, float lower) {
  const bool right = upper > 0.f;            // logsf(upper) < logcdf(upper)
  const float big = right ? log_sigmoid(-lower) : log_sigmoid(upper);
  const float small = right ? log_sigmoid(-upper) : log_sigmoid(lower);
  return fast_log1p(-__expf(small - big)) + big;
}

template <typename T> __device__ inline float load_as_float(const T* p, long long i);
template <> __device__ inline float load_as_float<float>(const float* p, long long i) { return p[i]; }
template <> __device__ inline float load_as_float<__hip_bfloat16>(const __hip_bfloat16* p, long long i) {
  return __bfloat162float(p[i]);
}
template <typename T> __device__ inline void store_from_float(T* p, long long i, float v);
template <> __device__ inline void store_from_float<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ inline void store_from_float<__hip_bfloat16>(__hip_bfloat16* p, long long i, float v) {
  p[i] = __float2bfloat16(v);
}

struct BitsParams {
  const void* y;
  const void* noise;
  void* y_hat;
  long long units, elems;       // elems per unit, channels innermost
  int channels;
  const float* params;
  float* log_prob;              // optional, per element
  float* partial;               // [units][blocks_per_unit]
  int blocks_per_unit;
  int threads;                  // multiple of channels
  // backward
  const float* gbits;           // [units]
  void* dy;
  float* dpartial;              // [total blocks][channels][P]
  float tail_mass;              // Laplace-mixture tail (laplace_tail.h); 0 = none
};

// A block owns a strided set of `threads`-element rows of one unit: thread t always sees channel
// t % channels, so its parameters (and, backward, their gradients) live in registers.
template <typename T, int K, int W>
__global__ void __launch_bounds__(512) factorized_forward_kernel(BitsParams p) {
  using L = MlpLayout<K, W>;
  const int t = threadIdx.x;
  const long long unit = blockIdx.x / p.blocks_per_unit;
  const int blk = blockIdx.x % p.blocks_per_unit;
  float prm[L::kParams];
  const float* src = p.params + static_cast<long long>(t % p.channels) * L::kParams;
#pragma unroll
  for (int i = 0; i < L::kParams; ++i) prm[i] = src[i];
  const T* y = static_cast<const T*>(p.y) + unit * p.elems;
  const T* nz = p.noise ? static_cast<const T*>(p.noise) + unit * p.elems : nullptr;
  T* yh = static_cast<T*>(p.y_hat) + unit * p.elems;
  float acc = 0.f;
  for (long long e = static_cast<long long>(blk) * p.threads + t; e < p.elems;
       e += static_cast<long long>(p.blocks_per_unit) * p.threads) {
    float v = load_as_float(y, e);
    if (nz) v += load_as_float(nz, e);
    store_from_float(yh, e, v);
    v = load_as_float(yh, e);                       // the value later passes see (dtype-rounded)
    const float up = mlp_forward<K, W, false>(prm, v + 0.5f, nullptr, nullptr);
    const float lo = mlp_forward<K, W, false>(prm, v - 0.5f, nullptr, nullptr);
    float lp = log_interval(up, lo);
    if (p.tail_mass > 0.f) lp = tail_mix(lp, v, p.tail_mass);
    if (p.log_prob) p.log_prob[unit * p.elems + e] = lp;
    acc += lp;
  }
  // block sum in a fixed order: wave shuffles, then lane 0 of each wave through LDS
  __shared__ float wsum[16];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((t & 63) == 0) wsum[t >> 6] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int w = 0; w < (p.threads + 63) / 64; ++w) s += wsum[w];
    p.partial[blockIdx.x] = s;
  }
}

__global__ void factorized_bits_reduce_kernel(const float* partial, int blocks_per_unit, long long units,
                                              float* bits) {
  const long long u = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (u >= units) return;
  float s = 0.f;
  for (int b = 0; b < blocks_per_unit; ++b) s += partial[u * blocks_per_unit + b];
  bits[u] = s * -1.4426950408889634f;          // / -ln 2
}

// c(upper) - c(lower), on the side of the median where it does not cancel
__device__ inline float interval_prob(float upper, float lower) {
  return upper > 0.f ? sigmoid(-lower) - sigmoid(-upper) : sigmoid(upper) - sigmoid(lower);
}
// d log_interval / d(upper, lower) = sigma'(u) * inv, -sigma'(l) * inv with inv = 1 / prob
// (with a Laplace tail: inv = (1 - m) / mixture, laplace_tail.h)
__device__ inline void log_interval_grad(float upper, float lower, float inv, float* gu, float* gl) {
  *gu = sigmoid(upper) * sigmoid(-upper) * inv;
  *gl = -sigmoid(lower) * sigmoid(-lower) * inv;
}

template <int K, int W>
__device__ inline float mlp_backward(const float* p, float z, float gout, float* dp) {
  using L = MlpLayout<K, W>;
  float h[K - 1][W], th[K - 1][W];
  (void)mlp_forward<K, W, true>(p, z, h, th);
  float dh[W];
  {
    const float* q = p + L::last();
    float* dq = dp + L::last();
    dq[W] += gout;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      dq[j] = fmaf(gout, h[K - 2][j], dq[j]);
      dh[j] = gout * q[j];
    }
  }
#pragma unroll
  for (int l = L::kMid - 1; l >= 0; --l) {
    const float* q = p + L::mid(l);
    float* dq = dp + L::mid(l);
    float dprev[W];
#pragma unroll
    for (int j = 0; j < W; ++j) dprev[j] = 0.f;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const float t = th[l + 1][i];
      dq[W * W + W + i] = fmaf(dh[i], t, dq[W * W + W + i]);                 // d factor
      const float dpre = dh[i] * fmaf(q[W * W + W + i], 1.f - t * t, 1.f);
      dq[W * W + i] += dpre;                                                  // d bias
#pragma unroll
      for (int j = 0; j < W; ++j) {
        dq[i * W + j] = fmaf(dpre, h[l][j], dq[i * W + j]);                   // d matrix
        dprev[j] = fmaf(dpre, q[i * W + j], dprev[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) dh[j] = dprev[j];
  }
  float dz = 0.f;
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const float t = th[0][i];
    dp[2 * W + i] = fmaf(dh[i], t, dp[2 * W + i]);
    const float dpre = dh[i] * fmaf(p[2 * W + i], 1.f - t * t, 1.f);
    dp[W + i] += dpre;
    dp[i] = fmaf(dpre, z, dp[i]);
    dz = fmaf(dpre, p[i], dz);
  }
  return dz;
}

// MAXT: the block-size bound the kernel is compiled for.  The wider MLPs keep ~2 x 56 parameter registers plus
// the activations of two cumulatives: within the 256 registers a 512-thread block allows they spill (300 / 650
// bytes per lane), within the 512 of a <= 256-thread block they do not.
template <typename T, int K, int W, int MAXT>
__global__ void __launch_bounds__(MAXT) factorized_backward_kernel(BitsParams p) {
  using L = MlpLayout<K, W>;
  const int t = threadIdx.x;
  const long long unit = blockIdx.x / p.blocks_per_unit;
  const int blk = blockIdx.x % p.blocks_per_unit;
  float prm[L::kParams], dprm[L::kParams];
  const float* src = p.params + static_cast<long long>(t % p.channels) * L::kParams;
#pragma unroll
  for (int i = 0; i < L::kParams; ++i) { prm[i] = src[i]; dprm[i] = 0.f; }
  const T* yh = static_cast<const T*>(p.y_hat) + unit * p.elems;
  const T* yin = p.y ? static_cast<const T*>(p.y) + unit * p.elems : nullptr;   // expected gradients: the input
  T* dy = static_cast<T*>(p.dy) + unit * p.elems;
  const float g = p.gbits[unit] * -1.4426950408889634f;     // dL/d(sum log p) of this unit
  for (long long e = static_cast<long long>(blk) * p.threads + t; e < p.elems;
       e += static_cast<long long>(p.blocks_per_unit) * p.threads) {
    const float v = load_as_float(yh, e);
    const float up = mlp_forward<K, W, false>(prm, v + 0.5f, nullptr, nullptr);
    const float lo = mlp_forward<K, W, false>(prm, v - 0.5f, nullptr, nullptr);
    const float prob = interval_prob(up, lo);
    float inv = 1.f / prob, direct = 0.f;
    if (p.tail_mass > 0.f) {
      const TailGrad tg = tail_mix_grad(prob, v, p.tail_mass);
      inv = tg.prior;
      direct = tg.direct;
    }
    float gu, gl;
    log_interval_grad(up, lo, inv, &gu, &gl);
    float dz = mlp_backward<K, W>(prm, v + 0.5f, g * gu, dprm) + mlp_backward<K, W>(prm, v - 0.5f, g * gl, dprm);
    dz = fmaf(g, direct, dz);
    if (yin) {
      // expected gradients (math_ops.py:157-216, perturb_and_apply(expected_grads=True)): the derivative
      // w.r.t. the input is E_u[d log p / dx] = log p(x + .5) - log p(x - .5) at the UNPERTURBED x, i.e.
      // the cumulative at x + 1, x, x - 1; the parameter gradients above stay those of log p(x + u).
      const float x = load_as_float(yin, e);
      const float c1 = mlp_forward<K, W, false>(prm, x + 1.f, nullptr, nullptr);
      const float c0 = mlp_forward<K, W, false>(prm, x, nullptr, nullptr);
      const float cm = mlp_forward<K, W, false>(prm, x - 1.f, nullptr, nullptr);
      float hi = log_interval(c1, c0), low = log_interval(c0, cm);
      if (p.tail_mass > 0.f) {
        hi = tail_mix(hi, x + 0.5f, p.tail_mass);
        low = tail_mix(low, x - 0.5f, p.tail_mass);
      }
      dz = g * (hi - low);
    }
    store_from_float(dy, e, dz);
  }
  // threads t, t + C, t + 2C, ... of the block share a channel: fold them through LDS in a fixed
  // order, then one partial row per (block, channel)
  extern __shared__ float fold[];                // [threads][P]
#pragma unroll
  for (int i = 0; i < L::kParams; ++i) fold[t * L::kParams + i] = dprm[i];
  __syncthreads();
  float* out = p.dpartial + static_cast<long long>(blockIdx.x) * p.channels * L::kParams;
  for (int idx = t; idx < p.channels * L::kParams; idx += p.threads) {
    const int c = idx / L::kParams, i = idx % L::kParams;
    float s = 0.f;
    for (int r = c; r < p.threads; r += p.channels) s += fold[r * L::kParams + i];
    out[idx] = s;
  }
}

int plan_blocks(long long units, long long elems, int channels, int* threads, int* blocks_per_unit) {
  if (channels < 1 || channels > 512) return fail("tfc_factorized_bits: channels must be in [1, 512]");
  if (elems % channels != 0) return fail("tfc_factorized_bits: elements per unit must be a multiple of channels");
  int t = (512 / channels) * channels;             // largest multiple of channels <= 512
  while (t > 256 && (t - channels) >= 192) t -= channels;   // keep blocks around 192..256 threads
  *threads = t;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long rows = ceil_div(elems, t);
  long long want = ceil_div(static_cast<long long>(cus) * 8, std::max<long long>(units, 1));
  *blocks_per_unit = static_cast<int>(std::max<long long>(1, std::min<long long>(rows, want)));
  if (units * *blocks_per_unit >= (1ll << 31)) return fail("tfc_factorized_bits: problem too large");
  return 0;
}

template <typename T, int K, int W>
int run_forward(BitsParams p, float* bits, hipStream_t st) {
  DevBuf partial;
  TFC_HIP(partial.alloc(sizeof(float) * p.units * p.blocks_per_unit, st));
  p.partial = partial.as<float>();
  {
    KernelTimer timer("factorized_forward", st);
    hipLaunchKernelGGL((factorized_forward_kernel<T, K, W>), dim3(static_cast<unsigned>(p.units * p.blocks_per_unit)),
                       dim3(p.threads), 0, st, p);
  }
  hipLaunchKernelGGL(factorized_bits_reduce_kernel, dim3(static_cast<unsigned>(ceil_div(p.units, 256))), dim3(256),
                     0, st, p.partial, p.blocks_per_unit, p.units, bits);
  TFC_HIP(hipGetLastError());
  return 0;
}

template <typename T, int K, int W>
int run_backward(BitsParams p, float* dparams, hipStream_t st) {
  using L = MlpLayout<K, W>;
  const long long blocks = p.units * p.blocks_per_unit;
  const int n = p.channels * L::kParams;
  DevBuf dpartial;
  TFC_HIP(dpartial.alloc(sizeof(float) * blocks * n, st));
  p.dpartial = dpartial.as<float>();
  const size_t lds = sizeof(float) * p.threads * L::kParams;
  {
    KernelTimer timer("factorized_backward", st);
    if (p.threads <= 256) {
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&factorized_backward_kernel<T, K, W, 256>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL((factorized_backward_kernel<T, K, W, 256>), dim3(static_cast<unsigned>(blocks)),
                         dim3(p.threads), lds, st, p);
    } else {
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&factorized_backward_kernel<T, K, W, 512>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL((factorized_backward_kernel<T, K, W, 512>), dim3(static_cast<unsigned>(blocks)),
                         dim3(p.threads), lds, st, p);
    }
  }
  launch_sum_rows(p.dpartial, blocks, n, n, dparams, st);   // dparams += block partials, fixed order
  TFC_HIP(hipGetLastError());
  return 0;
}

template <typename F32, typename BF16>
int dispatch(int dtype, int layers, int width, const char* who, F32 f32, BF16 bf16) {
  if (dtype != 0 && dtype != 1) return fail("%s: dtype must be 0 (float32) or 1 (bfloat16)", who);
  if (!((layers == 3 || layers == 4) && width == 3) && !(layers == 3 && width == 5))
    return fail("%s: built for num_filters (3, 3), (3, 3, 3) and (5, 5) (got %d layers of width %d)", who,
                layers, width);
  return dtype == 0 ? f32() : bf16();
}

}  // namespace tfc

#define TFC_FB_SWITCH(FN, ...)                                                                   \
  (layers == 3 && width == 3)   ? FN<TT, 3, 3>(__VA_ARGS__)                                       \
  : (layers == 4 && width == 3) ? FN<TT, 4, 3>(__VA_ARGS__)                                       \
                                : FN<TT, 3, 5>(__VA_ARGS__)

namespace {
int fb_forward(const char* who, const void* y, const void* noise, void* y_hat, int dtype, int64_t units, int64_t elems,
               int64_t channels, const float* params, int layers, int width, float tail_mass, float* log_prob,
               float* bits, void* stream) {
  using namespace tfc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (units == 0 || elems == 0) return 0;
  BitsParams p{};
  p.y = y; p.noise = noise; p.y_hat = y_hat; p.units = units; p.elems = elems;
  p.channels = static_cast<int>(channels); p.params = params; p.log_prob = log_prob; p.tail_mass = tail_mass;
  if (int rc = plan_blocks(units, elems, p.channels, &p.threads, &p.blocks_per_unit)) return rc;
  return dispatch(dtype, layers, width, who,
                  [&] { using TT = float; return TFC_FB_SWITCH(run_forward, p, bits, st); },
                  [&] { using TT = __hip_bfloat16; return TFC_FB_SWITCH(run_forward, p, bits, st); });
}

// y non-null: expected gradients (the unperturbed input)
int fb_backward(const char* who, const void* y, const void* y_hat, int dtype, int64_t units, int64_t elems,
                int64_t channels, const float* params, int layers, int width, float tail_mass, const float* gbits,
                void* dy, float* dparams, void* stream) {
  using namespace tfc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (units == 0 || elems == 0) return 0;
  BitsParams p{};
  p.y = y; p.y_hat = const_cast<void*>(y_hat); p.units = units; p.elems = elems;
  p.channels = static_cast<int>(channels); p.params = params; p.gbits = gbits; p.dy = dy; p.tail_mass = tail_mass;
  if (int rc = plan_blocks(units, elems, p.channels, &p.threads, &p.blocks_per_unit)) return rc;
  return dispatch(dtype, layers, width, who,
                  [&] { using TT = float; return TFC_FB_SWITCH(run_backward, p, dparams, st); },
                  [&] { using TT = __hip_bfloat16; return TFC_FB_SWITCH(run_backward, p, dparams, st); });
}
}  // namespace

extern "C" int tfc_factorized_bits_forward(const void* y, const void* noise, void* y_hat, int dtype,
                                           int64_t units, int64_t elems, int64_t channels,
                                           const float* params, int layers, int width, float* log_prob,
                                           float* bits, void* stream) {
  return fb_forward("tfc_factorized_bits_forward", y, noise, y_hat, dtype, units, elems, channels, params, layers,
                    width, 0.f, log_prob, bits, stream);
}

extern "C" int tfc_factorized_bits_backward(const void* y_hat, int dtype, int64_t units, int64_t elems,
                                            int64_t channels, const float* params, int layers, int width,
                                            const float* gbits, void* dy, float* dparams, void* stream) {
  return fb_backward("tfc_factorized_bits_backward", nullptr, y_hat, dtype, units, elems, channels, params, layers,
                     width, 0.f, gbits, dy, dparams, stream);
}

extern "C" int tfc_factorized_bits_backward_expected(const void* y, const void* y_hat, int dtype, int64_t units,
                                                     int64_t elems, int64_t channels, const float* params,
                                                     int layers, int width, const float* gbits, void* dy,
                                                     float* dparams, void* stream) {
  if (y == nullptr && units != 0 && elems != 0)
    return tfc::fail("tfc_factorized_bits_backward_expected: the unperturbed input is required");
  return fb_backward("tfc_factorized_bits_backward_expected", y, y_hat, dtype, units, elems, channels, params,
                     layers, width, 0.f, gbits, dy, dparams, stream);
}

extern "C" int tfc_factorized_bits_forward_tail(const void* y, const void* noise, void* y_hat, int dtype,
                                                int64_t units, int64_t elems, int64_t channels,
                                                const float* params, int layers, int width,
                                                float laplace_tail_mass, float* log_prob, float* bits,
                                                void* stream) {
  if (!tfc::tail_mass_ok(laplace_tail_mass))
    return tfc::fail("tfc_factorized_bits_forward_tail: laplace_tail_mass must be in (0, 1) (got %g)",
                     static_cast<double>(laplace_tail_mass));
  return fb_forward("tfc_factorized_bits_forward_tail", y, noise, y_hat, dtype, units, elems, channels, params,
                    layers, width, laplace_tail_mass, log_prob, bits, stream);
}

extern "C" int tfc_factorized_bits_backward_tail(const void* y, const void* y_hat, int dtype, int64_t units,
                                                 int64_t elems, int64_t channels, const float* params, int layers,
                                                 int width, float laplace_tail_mass, const float* gbits, void* dy,
                                                 float* dparams, void* stream) {
  if (!tfc::tail_mass_ok(laplace_tail_mass))
    return tfc::fail("tfc_factorized_bits_backward_tail: laplace_tail_mass must be in (0, 1) (got %g)",
                     static_cast<double>(laplace_tail_mass));
  return fb_backward("tfc_factorized_bits_backward_tail", y, y_hat, dtype, units, elems, channels, params, layers,
                     width, laplace_tail_mass, gbits, dy, dparams, stream);
}
