// Training-time path of the indexed (conditional) entropy model with a NoisyNormal prior, fused
// (python/entropy_models/continuous_indexed.py:313-353 `__call__(training=True)`,
//  python/ops/math_ops.py:157-216 perturb_and_apply, python/distributions/uniform_noise.py:117-156):
//   y_hat = y + u,  u ~ U(-.5, .5)
//   log p(y_hat | sigma) = log( Phi((y_hat + .5) / sigma) - Phi((y_hat - .5) / sigma) )
//   bits[unit] = - sum_unit log p / ln 2
// with sigma a per-element tensor (scale_fn(indexes), differentiable: in bmshj2018 the indexes come out of the
// hyper-synthesis transform).  The location is the caller's (the model shifts the bottleneck).  The
// reference runs this as ~25 TF kernels over the latent tensor; here one forward kernel (y, u, sigma ->
// y_hat, block partial sums) and one backward kernel (y_hat, sigma, dL/dbits -> dL/dy_hat, dL/dsigma), each
// element-wise: HBM-bound, 3 tensors in + 1 out forward, 2 in + 2 out backward.
//
// Numerics follow the reference's op order: the difference of the two cumulatives is taken on the side of
// the median where it does not cancel (survival functions right of it), in log space; log Phi as torch /
// TF do it: log(erfcx(-z / sqrt 2) / 2) - z^2 / 2 for z < -1, log1p(-erfc(z / sqrt 2) / 2) otherwise.
// The *_tail entry points add the Laplace-mixture tail of continuous_base.py:298-334 (laplace_tail.h); the
// Laplace component sits at 0, so they are for priors whose location the caller has already subtracted.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include "../../include/tfc_hip.h"
#include "common.h"
#include "laplace_tail.h"

namespace tfc {
namespace {

constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kLogSqrt2Pi = 0.91893853320467274f;

__device__ inline float log_ndtr(float z) {
  if (z < -1.f) return __logf(0.5f * erfcxf(-z * kInvSqrt2)) - 0.5f * z * z;
  return log1pf(-0.5f * erfcf(z * kInvSqrt2));
}

// log( Phi(zu) - Phi(zl) ), zu > zl
__device__ inline float log_interval_normal(float zu, float zl) {
  const bool right = zu > 0.f;                         // logsf(zu) < logcdf(zu)
  const float big = right ? log_ndtr(-zl) : log_ndtr(zu);
  const float small = right ? log_ndtr(-zu) : log_ndtr(zl);
  return log1pf(-__expf(small - big)) + big;
}

template <typename T> __device__ inline float ld(const T* p, long long i);
template <> __device__ inline float ld<float>(const float* p, long long i) { return p[i]; }
template <> __device__ inline float ld<__hip_bfloat16>(const __hip_bfloat16* p, long long i) { return __bfloat162float(p[i]); }
template <typename T> __device__ inline void st(T* p, long long i, float v);
template <> __device__ inline void st<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ inline void st<__hip_bfloat16>(__hip_bfloat16* p, long long i, float v) { p[i] = __float2bfloat16(v); }

struct NnParams {
  const void* y;
  const void* noise;
  const float* scale;
  void* y_hat;
  long long units, elems;
  float* partial;            // [units][blocks_per_unit]
  int blocks_per_unit;
  const float* gbits;        // backward
  void* dy;
  float* dscale;
  const void* y_in;          // expected gradients: the unperturbed input (else null)
  float tail_mass;           // Laplace-mixture tail; 0 = none
};

constexpr int kThreads = 256;

template <typename T>
__global__ void __launch_bounds__(kThreads) noisy_normal_forward_kernel(NnParams p) {
  const int t = threadIdx.x;
  const long long unit = blockIdx.x / p.blocks_per_unit;
  const int blk = blockIdx.x % p.blocks_per_unit;
  const T* y = static_cast<const T*>(p.y) + unit * p.elems;
  const T* nz = p.noise ? static_cast<const T*>(p.noise) + unit * p.elems : nullptr;
  const float* sc = p.scale + unit * p.elems;
  T* yh = static_cast<T*>(p.y_hat) + unit * p.elems;
  float acc = 0.f;
  for (long long e = static_cast<long long>(blk) * kThreads + t; e < p.elems;
       e += static_cast<long long>(p.blocks_per_unit) * kThreads) {
    float v = ld(y, e);
    if (nz) v += ld(nz, e);
    st(yh, e, v);
    v = ld(yh, e);                                    // the value later passes see (dtype-rounded)
    const float inv = 1.f / sc[e];
    float lp = log_interval_normal((v + 0.5f) * inv, (v - 0.5f) * inv);
    if (p.tail_mass > 0.f) lp = tail_mix(lp, v, p.tail_mass);
    acc += lp;
  }
  __shared__ float wsum[kThreads / 64];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((t & 63) == 0) wsum[t >> 6] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int w = 0; w < kThreads / 64; ++w) s += wsum[w];     // fixed order
    p.partial[blockIdx.x] = s;
  }
}

__global__ void noisy_normal_reduce_kernel(const float* partial, int blocks_per_unit, long long units, float* bits) {
  const long long u = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (u >= units) return;
  float s = 0.f;
  for (int b = 0; b < blocks_per_unit; ++b) s += partial[u * blocks_per_unit + b];
  bits[u] = s * -1.4426950408889634f;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) noisy_normal_backward_kernel(NnParams p) {
  const long long unit = blockIdx.x / p.blocks_per_unit;
  const int blk = blockIdx.x % p.blocks_per_unit;
  const T* yh = static_cast<const T*>(p.y_hat) + unit * p.elems;
  const T* yin = p.y_in ? static_cast<const T*>(p.y_in) + unit * p.elems : nullptr;
  const float* sc = p.scale + unit * p.elems;
  T* dy = static_cast<T*>(p.dy) + unit * p.elems;
  float* ds = p.dscale + unit * p.elems;
  const float g = p.gbits[unit] * -1.4426950408889634f;     // dL/d(sum log p) of this unit
  for (long long e = static_cast<long long>(blk) * kThreads + threadIdx.x; e < p.elems;
       e += static_cast<long long>(p.blocks_per_unit) * kThreads) {
    const float v = ld(yh, e);
    const float inv = 1.f / sc[e];
    const float zu = (v + 0.5f) * inv, zl = (v - 0.5f) * inv;
    const float lp = log_interval_normal(zu, zl);
    // d lp / d zu = phi(zu) / P, d lp / d zl = -phi(zl) / P, with P = exp(lp): ratios in log space
    float gu, gl, direct = 0.f;
    if (p.tail_mass > 0.f) {
      // d log(mixture) = (1 - m) dP / mixture + m dQ / mixture  (or d log Q where the mixture is below 1e-10)
      const TailGrad tg = tail_mix_grad(__expf(lp), v, p.tail_mass);
      gu = __expf(-0.5f * zu * zu - kLogSqrt2Pi) * tg.prior;
      gl = -__expf(-0.5f * zl * zl - kLogSqrt2Pi) * tg.prior;
      direct = tg.direct;
    } else {
      gu = __expf(-0.5f * zu * zu - kLogSqrt2Pi - lp);
      gl = -__expf(-0.5f * zl * zl - kLogSqrt2Pi - lp);
    }
    float dv = g * fmaf(gu + gl, inv, direct);
    ds[e] = -g * (gu * zu + gl * zl) * inv;
    if (yin) {
      // expected gradients (math_ops.py:157-216): log p(x + .5) - log p(x - .5) at the unperturbed x
      const float x = ld(yin, e);
      float hi = log_interval_normal((x + 1.f) * inv, x * inv), low = log_interval_normal(x * inv, (x - 1.f) * inv);
      if (p.tail_mass > 0.f) {
        hi = tail_mix(hi, x + 0.5f, p.tail_mass);
        low = tail_mix(low, x - 0.5f, p.tail_mass);
      }
      dv = g * (hi - low);
    }
    st(dy, e, dv);
  }
}

int plan(long long units, long long elems, int* blocks_per_unit) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long rows = ceil_div(elems, kThreads);
  const long long want = ceil_div(static_cast<long long>(cus) * 8, std::max<long long>(units, 1));
  *blocks_per_unit = static_cast<int>(std::max<long long>(1, std::min<long long>(rows, want)));
  if (units * *blocks_per_unit >= (1ll << 31)) return fail("tfc_noisy_normal_bits: problem too large");
  return 0;
}

}  // namespace
}  // namespace tfc

namespace {
int nn_forward(const char* who, const void* y, const void* noise, const float* scale, void* y_hat, int dtype,
               int64_t units, int64_t elems, float tail_mass, float* bits, void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("%s: dtype must be 0 (float32) or 1 (bfloat16)", who);
  if (units == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  NnParams p{};
  p.y = y; p.noise = noise; p.scale = scale; p.y_hat = y_hat; p.units = units; p.elems = elems;
  p.tail_mass = tail_mass;
  if (int rc = plan(units, elems, &p.blocks_per_unit)) return rc;
  DevBuf partial;
  TFC_HIP(partial.alloc(sizeof(float) * units * p.blocks_per_unit, st));
  p.partial = partial.as<float>();
  {
    KernelTimer timer("noisy_normal_forward", st);
    const dim3 grid(static_cast<unsigned>(units * p.blocks_per_unit));
    if (dtype == 0) hipLaunchKernelGGL(noisy_normal_forward_kernel<float>, grid, dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL(noisy_normal_forward_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, st, p);
  }
  hipLaunchKernelGGL(noisy_normal_reduce_kernel, dim3(static_cast<unsigned>(ceil_div(units, 256))), dim3(256), 0, st,
                     p.partial, p.blocks_per_unit, static_cast<long long>(units), bits);
  TFC_HIP(hipGetLastError());
  return 0;
}

int nn_backward(const char* who, const void* y_in, const void* y_hat, const float* scale, int dtype, int64_t units,
                int64_t elems, float tail_mass, const float* gbits, void* dy, float* dscale, void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("%s: dtype must be 0 (float32) or 1 (bfloat16)", who);
  if (units == 0 || elems == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  NnParams p{};
  p.y_in = y_in; p.y_hat = const_cast<void*>(y_hat); p.scale = scale; p.units = units; p.elems = elems;
  p.gbits = gbits; p.dy = dy; p.dscale = dscale; p.tail_mass = tail_mass;
  if (int rc = plan(units, elems, &p.blocks_per_unit)) return rc;
  KernelTimer timer("noisy_normal_backward", st);
  const dim3 grid(static_cast<unsigned>(units * p.blocks_per_unit));
  if (dtype == 0) hipLaunchKernelGGL(noisy_normal_backward_kernel<float>, grid, dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL(noisy_normal_backward_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, st, p);
  TFC_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int tfc_noisy_normal_bits_forward(const void* y, const void* noise, const float* scale, void* y_hat,
                                             int dtype, int64_t units, int64_t elems, float* bits, void* stream) {
  return nn_forward("tfc_noisy_normal_bits_forward", y, noise, scale, y_hat, dtype, units, elems, 0.f, bits, stream);
}

extern "C" int tfc_noisy_normal_bits_backward(const void* y_in, const void* y_hat, const float* scale, int dtype,
                                              int64_t units, int64_t elems, const float* gbits, void* dy,
                                              float* dscale, void* stream) {
  return nn_backward("tfc_noisy_normal_bits_backward", y_in, y_hat, scale, dtype, units, elems, 0.f, gbits, dy,
                     dscale, stream);
}

extern "C" int tfc_noisy_normal_bits_forward_tail(const void* y, const void* noise, const float* scale, void* y_hat,
                                                  int dtype, int64_t units, int64_t elems, float laplace_tail_mass,
                                                  float* bits, void* stream) {
  if (!tfc::tail_mass_ok(laplace_tail_mass))
    return tfc::fail("tfc_noisy_normal_bits_forward_tail: laplace_tail_mass must be in (0, 1) (got %g)",
                     static_cast<double>(laplace_tail_mass));
  return nn_forward("tfc_noisy_normal_bits_forward_tail", y, noise, scale, y_hat, dtype, units, elems,
                    laplace_tail_mass, bits, stream);
}

extern "C" int tfc_noisy_normal_bits_backward_tail(const void* y_in, const void* y_hat, const float* scale, int dtype,
                                                   int64_t units, int64_t elems, float laplace_tail_mass,
                                                   const float* gbits, void* dy, float* dscale, void* stream) {
  if (!tfc::tail_mass_ok(laplace_tail_mass))
    return tfc::fail("tfc_noisy_normal_bits_backward_tail: laplace_tail_mass must be in (0, 1) (got %g)",
                     static_cast<double>(laplace_tail_mass));
  return nn_backward("tfc_noisy_normal_bits_backward_tail", y_in, y_hat, scale, dtype, units, elems,
                     laplace_tail_mass, gbits, dy, dscale, stream);
}
