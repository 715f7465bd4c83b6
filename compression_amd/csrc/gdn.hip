// GDN / IGDN forward for gfx950 (channels-last, [pixels, C]).
//
//   u = |x|^alpha,  n_i = beta_i + sum_j gamma[j][i] u_j,  y_i = x_i / n_i^eps  (GDN)
//                                                      or  x_i * n_i^eps  (IGDN)
// python/layers/gdn.py:371-421 runs this as 4-5 separate TF kernels (abs, 1x1
// conv, bias_add, div), each streaming the whole tensor.  Here it is one kernel
// whose HBM traffic is the algorithmic minimum (read x once, write y once):
//
//   * The contraction runs TRANSPOSED on the matrix cores: N^T = Gamma^T * U^T,
//     A operand = Gamma^T (out-channel rows) from LDS, B operand = U^T whose
//     fragment for lane l is 8 (bf16) / 1 (f32) channels of ONE pixel (l & 31) —
//     i.e. plain contiguous loads from the NHWC tensor, no LDS staging.
//   * The K (input-channel) order fed to the MFMA is permuted so that the
//     channels a lane loads as B fragments are exactly the channels whose
//     outputs land in that lane's accumulator registers
//     (C/D map: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31).
//     The epilogue (add beta, reciprocal, multiply by x) therefore needs no
//     transpose, no shuffle and no second read of x; y leaves with the same
//     access pattern x came in with.
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate), gamma rounded to bf16
//     like a Keras mixed_bfloat16 policy would.  f32: v_mfma_f32_32x32x2_f32,
//     bit-exact fp32 FMA chains (the <=1e-5 parity path).
//
// Roofline: HBM-bound, 2*sizeof(dtype) bytes per element.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/tfc_hip.h"
#include "common.h"

namespace tfc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct GdnParams {
  const void* x;
  void* y;
  const float* beta;
  const float* gamma;   // [C in j][C out i]
  long long pixels;
  int C;
  int inverse, rectify, alpha2, eps_half;
  long long tiles;      // ceil(pixels / 32)
  const void* image;    // fragment-ordered Gamma^T (+ beta) built by gdn_prep_*_kernel
};

// y = x / n^eps (GDN) or x * n^eps (IGDN); hardware rcp / rsq / sqrt are 1-ulp approximations,
// far inside the 1e-5 tolerance.  Flags are template parameters so that the epilogue carries
// only the instructions of the variant in use (the kernel switches once, wave-uniformly).
template <bool INVERSE, bool EPS_HALF>
__device__ inline float gdn_apply(float x, float n) {
  if (INVERSE) return x * (EPS_HALF ? __builtin_amdgcn_sqrtf(n) : n);
  return x * (EPS_HALF ? __builtin_amdgcn_rsqf(n) : __builtin_amdgcn_rcpf(n));
}

__device__ inline float bf16_bits_to_float(unsigned int bits16) { return __uint_as_float(bits16 << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// v_cvt_pk_bf16_f32: two floats -> packed bf16 pair, round-to-nearest-even.
__device__ inline unsigned int pack_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__device__ inline unsigned int float_to_bf16_bits(float f) {
  // round to nearest even, NaN preserved (matches __float2bfloat16)
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// ---------------------------------------------------------------------------
// bf16 I/O.  KT = C / 32 output tiles, KS = C / 16 K-steps.
// LDS: A fragments of Gamma^T, fragment-ordered: [(t * KS + s) * 64 + lane][8].
//   element e of lane (i = lane & 31, h = lane >> 5) at (t, s) is
//   gamma[ch(s, h, e)][32 t + i],  ch(s, h, e) = 16 s + 4 h + (e & 3) + 8 (e >> 2).
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(512) gdn_fwd_bf16_kernel(GdnParams p) {
  constexpr int C = KT * 32;
  constexpr int KS = KT * 2;
  extern __shared__ unsigned char smem[];
  bf16x8* afrag = reinterpret_cast<bf16x8*>(smem);
  float* beta_s = reinterpret_cast<float*>(smem + sizeof(bf16x8) * KT * KS * 64);

  {
    // fragment image (built once per call by gdn_prep_bf16_kernel): linear 16-byte copy
    const u32x4* src = static_cast<const u32x4*>(p.image);
    u32x4* dstv = reinterpret_cast<u32x4*>(smem);
    constexpr int n16 = KT * KS * 64 + (C * 4) / 16;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dstv[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = static_cast<long long>(gridDim.x) * (blockDim.x >> 6);
  const unsigned short* x = static_cast<const unsigned short*>(p.x);
  unsigned short* y = static_cast<unsigned short*>(p.y);

  for (long long tile = wave; tile < p.tiles; tile += nwaves) {
    const long long pix = tile * 32 + (lane & 31);
    const bool live = pix < p.pixels;
    const long long row = (live ? pix : p.pixels - 1) * C;
    // ---- loads: after the swap, K-step s holds channels 16s+4h+{0..3} and 16s+4h+8+{0..3} ----
    u32x4 xr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // One 16-byte load per lane (channels 16s + 8h + 0..7), then v_permlane32_swap trades
      // the inner halves between lanes l and l+32 so that the lane ends up with channels
      // 16s + 4h + {0..3} and 16s + 4h + 8 + {0..3} — twice the bytes per cache line touched
      // by one load instruction compared with two 8-byte loads.
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + row + 16 * s + 8 * h);
      const auto s0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
      xr[s] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // u = |x| (clear sign bits), relu first if rectify, x*x if alpha == 2
      u32x4 u = xr[s];
      if (p.rectify || p.alpha2) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float lo = bf16_bits_to_float(u[w] & 0xFFFFu), hi = bf16_bits_to_float(u[w] >> 16);
          if (p.rectify) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          if (p.alpha2) { lo = lo * lo; hi = hi * hi; } else { lo = fabsf(lo); hi = fabsf(hi); }
          u[w] = pack_bf16(lo, hi);
        }
      } else {
        u &= 0x7FFF7FFFu;
      }
      const bf16x8 bfrag = __builtin_bit_cast(bf16x8, u);
#pragma unroll
      for (int t = 0; t < KT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[(t * KS + s) * 64 + lane], bfrag,
                                                          acc[t], 0, 0, 0);
      // keep the scheduler from hoisting every K-step's LDS fragment loads to the top
      // (72 fragments = 288 VGPRs): one K-step's fragments at a time.
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: acc[t][4q + r] is channel 32t + 8q + 4h + r of this lane's pixel ----
    auto epilogue = [&](auto inv, auto epsh) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int t = s >> 1;
        u32x4 out;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int q = 2 * (s & 1) + half;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
          float yv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned int word = xr[s][2 * half + (r >> 1)];
            float xv = __uint_as_float((r & 1) ? (word & 0xFFFF0000u) : (word << 16));
            if (p.rectify) xv = fmaxf(xv, 0.f);
            yv[r] = gdn_apply<decltype(inv)::value, decltype(epsh)::value>(xv, acc[t][4 * q + r] + b4[r]);
          }
          out[2 * half] = pack_bf16(yv[0], yv[1]);
          out[2 * half + 1] = pack_bf16(yv[2], yv[3]);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(out.x, out.z, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(out.y, out.w, false, false);
        if (live)
          *reinterpret_cast<u32x4*>(y + row + 16 * s + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (p.inverse) {
      if (p.eps_half) epilogue(T{}, T{}); else epilogue(T{}, F{});
    } else {
      if (p.eps_half) epilogue(F{}, T{}); else epilogue(F{}, F{});
    }
  }
}

// ---------------------------------------------------------------------------
// f32 I/O, exact fp32 MFMA (v_mfma_f32_32x32x2_f32: K = 2 per instruction).
// K index inside K-tile kt at step u (0..15), half h:  ch = 32 kt + 4 h + (u & 3) + 8 (u >> 2).
// LDS: Gamma^T fragments [((t * KT + kt) * 4 + u4) * 64 + lane][4]  (4 consecutive steps u).
// ---------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(256) gdn_fwd_f32_kernel(GdnParams p) {
  constexpr int C = KT * 32;
  extern __shared__ unsigned char smem[];
  f32x4* afrag = reinterpret_cast<f32x4*>(smem);
  float* beta_s = reinterpret_cast<float*>(smem + sizeof(f32x4) * KT * KT * 4 * 64);

  {
    const u32x4* src = static_cast<const u32x4*>(p.image);
    u32x4* dstv = reinterpret_cast<u32x4*>(smem);
    constexpr int n16 = KT * KT * 4 * 64 + (C * 4) / 16;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dstv[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = static_cast<long long>(gridDim.x) * (blockDim.x >> 6);
  const float* x = static_cast<const float*>(p.x);
  float* y = static_cast<float*>(p.y);

  for (long long tile = wave; tile < p.tiles; tile += nwaves) {
    const long long pix = tile * 32 + (lane & 31);
    const bool live = pix < p.pixels;
    const long long row = (live ? pix : p.pixels - 1) * C;
    f32x4 xr[KT][4];   // [K-tile][q]: channels 32kt + 8q + 4h + {0..3}
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + row + 32 * kt + 8 * q + 4 * h);
        if (p.rectify) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        xr[kt][q] = v;
      }
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 u = xr[kt][q];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = p.alpha2 ? u[e] * u[e] : fabsf(u[e]);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const f32x4 a4 = afrag[((t * KT + kt) * 4 + q) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], u[e], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    auto epilogue = [&](auto inv, auto epsh) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
          f32x4 out;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[r] = gdn_apply<decltype(inv)::value, decltype(epsh)::value>(
                xr[t][q][r], acc[t][4 * q + r] + b4[r]);
          if (live) *reinterpret_cast<f32x4*>(y + row + 32 * t + 8 * q + 4 * h) = out;
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (p.inverse) {
      if (p.eps_half) epilogue(T{}, T{}); else epilogue(T{}, F{});
    } else {
      if (p.eps_half) epilogue(F{}, T{}); else epilogue(F{}, F{});
    }
  }
}

// Builds the fragment-ordered Gamma^T image (+ beta behind it) the main kernels copy to LDS.
__global__ void gdn_prep_bf16_kernel(const float* gamma, const float* beta, int C, bf16x8* image) {
  const int KT = C / 32, KS = C / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < KT * KS * 64) {
    const int l = idx & 63, ts = idx >> 6;
    const int t = ts / KS, s = ts % KS;
    const int i = l & 31, h = l >> 5;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = 16 * s + 4 * h + (e & 3) + 8 * (e >> 2);
      v[e] = static_cast<__bf16>(gamma[ch * C + 32 * t + i]);
    }
    image[idx] = v;
  }
  float* b = reinterpret_cast<float*>(image + KT * KS * 64);
  if (idx < C) b[idx] = beta[idx];
}

__global__ void gdn_prep_f32_kernel(const float* gamma, const float* beta, int C, f32x4* image) {
  const int KT = C / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < KT * KT * 4 * 64) {
    const int l = idx & 63, rest = idx >> 6;
    const int u4 = rest & 3, tk = rest >> 2;
    const int t = tk / KT, kt = tk % KT;
    const int i = l & 31, h = l >> 5;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = 32 * kt + 4 * h + e + 8 * u4;
      v[e] = gamma[ch * C + 32 * t + i];
    }
    image[idx] = v;
  }
  float* b = reinterpret_cast<float*>(image + KT * KT * 4 * 64);
  if (idx < C) b[idx] = beta[idx];
}

template <int KT>
int launch_gdn(GdnParams p, int dtype, hipStream_t st) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // bf16: 8 waves (2 per SIMD, <= 256 VGPRs each); f32: 4 waves so that the 96 x + 96
  // accumulator registers fit the 512-entry unified file without spilling.
  const int waves_per_block = dtype == 1 ? 8 : 4;
  const long long want = ceil_div(p.tiles, waves_per_block);
  const unsigned blocks = static_cast<unsigned>(std::max<long long>(1, std::min<long long>(want, cus)));
  DevBuf image;
  if (dtype == 1) {
    const size_t lds = sizeof(bf16x8) * KT * (KT * 2) * 64 + sizeof(float) * KT * 32;
    TFC_HIP(image.alloc(lds, st));
    const int n = KT * KT * 2 * 64;
    hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                       KT * 32, image.as<bf16x8>());
    p.image = image.p;
    KernelTimer timer("gdn_forward", st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_bf16_kernel<KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((gdn_fwd_bf16_kernel<KT>), dim3(blocks), dim3(64 * waves_per_block), lds, st, p);
  } else {
    const size_t lds = sizeof(f32x4) * KT * KT * 4 * 64 + sizeof(float) * KT * 32;
    if (lds > 160 * 1024)
      return fail("tfc_gdn_forward: float32 path supports up to 192 channels (Gamma must fit in LDS)");
    TFC_HIP(image.alloc(lds, st));
    const int n = KT * KT * 4 * 64;
    hipLaunchKernelGGL(gdn_prep_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                       KT * 32, image.as<f32x4>());
    p.image = image.p;
    KernelTimer timer("gdn_forward", st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_f32_kernel<KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((gdn_fwd_f32_kernel<KT>), dim3(blocks), dim3(64 * waves_per_block), lds, st, p);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

}  // namespace tfc

extern "C" int tfc_gdn_forward(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                               const float* beta, const float* gamma, int inverse, int rectify,
                               int alpha_mode, int eps_mode, void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("tfc_gdn_forward: dtype must be 0 (float32) or 1 (bfloat16)");
  if (alpha_mode != 1 && alpha_mode != 2) return fail("tfc_gdn_forward: alpha must be 1 or 2");
  if (eps_mode != 0 && eps_mode != 1) return fail("tfc_gdn_forward: epsilon must be 1 or 0.5");
  if (channels <= 0 || channels % 32 != 0 || channels > 256)
    return fail("tfc_gdn_forward: channels must be a multiple of 32, at most 256 (got %lld)",
                static_cast<long long>(channels));
  if (pixels == 0) return 0;
  GdnParams p;
  p.x = x; p.y = y; p.beta = beta; p.gamma = gamma;
  p.pixels = pixels; p.C = static_cast<int>(channels);
  p.inverse = inverse; p.rectify = rectify; p.alpha2 = alpha_mode == 2; p.eps_half = eps_mode == 1;
  p.tiles = ceil_div(pixels, 32);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (channels / 32) {
    case 1: return launch_gdn<1>(p, dtype, st);
    case 2: return launch_gdn<2>(p, dtype, st);
    case 3: return launch_gdn<3>(p, dtype, st);
    case 4: return launch_gdn<4>(p, dtype, st);
    case 5: return launch_gdn<5>(p, dtype, st);
    case 6: return launch_gdn<6>(p, dtype, st);
    case 7: return launch_gdn<7>(p, dtype, st);
    default: return launch_gdn<8>(p, dtype, st);
  }
}
