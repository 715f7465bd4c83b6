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
  // backward passes (see the mode table above the kernels)
  const void* g;        // dL/dy                      (MODE_BWD_T)
  const void* r;        // g * n^s from pass 1        (MODE_BWD_DX)
  const void* xraw;     // the layer input            (MODE_BWD_DX)
  void* y2;             // second output: R           (MODE_BWD_T)
};

// Kernel modes.  All three run the same tile loop / MFMA contraction; only the B operand
// preparation and the epilogue differ.
//   MODE_FWD     x -> y = x * n^s,  n = beta + U Gamma,  s = -eps (GDN) / +eps (IGDN)
//   MODE_BWD_T   x, g -> T = dL/dn = s g y / n   and   R = g n^s          (image: Gamma^T, beta)
//   MODE_BWD_DX  T, R, x -> dx = R + (T Gamma^T) * d|x|^alpha/dx           (image: Gamma, no beta)
constexpr int MODE_FWD = 0, MODE_BWD_T = 1, MODE_BWD_DX = 2;

// p = n^s and the factor c with T = c * g * x:  GDN eps=1: p = 1/n, c = -p^2;  GDN eps=.5:
// p = rsqrt(n), c = -p^3/2;  IGDN eps=1: p = n, c = 1;  IGDN eps=.5: p = sqrt(n), c = 1/(2p).
template <bool INVERSE, bool EPS_HALF>
__device__ inline void gdn_grad_factors(float n, float* pw, float* c) {
  if (INVERSE) {
    if (EPS_HALF) { const float q = __builtin_amdgcn_sqrtf(n); *pw = q; *c = 0.5f * __builtin_amdgcn_rcpf(q); }
    else { *pw = n; *c = 1.f; }
  } else {
    if (EPS_HALF) { const float q = __builtin_amdgcn_rsqf(n); *pw = q; *c = -0.5f * q * q * q; }
    else { const float q = __builtin_amdgcn_rcpf(n); *pw = q; *c = -q * q; }
  }
}

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
template <int KT, int MODE>
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
      // u = |x| (clear sign bits), relu first if rectify, x*x if alpha == 2;
      // in MODE_BWD_DX the operand is T itself (signed).
      u32x4 u = xr[s];
      if (MODE == MODE_BWD_DX) {
      } else if (p.rectify || p.alpha2) {
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
    // element (s, half, r) of a lane's fragment words <-> that channel, see the load above.
    auto frag_load = [&](const unsigned short* base, int s) -> u32x4 {
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + row + 16 * s + 8 * h);
      const auto s0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
      return u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    auto frag_store = [&](unsigned short* base, int s, u32x4 out) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(out.x, out.z, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(out.y, out.w, false, false);
      if (live) *reinterpret_cast<u32x4*>(base + row + 16 * s + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    auto elem = [&](const u32x4& f, int half, int r) -> float {
      const unsigned int word = f[2 * half + (r >> 1)];
      return __uint_as_float((r & 1) ? (word & 0xFFFF0000u) : (word << 16));
    };
    auto epilogue = [&](auto inv, auto epsh) {
      constexpr bool INV = decltype(inv)::value, EPSH = decltype(epsh)::value;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int t = s >> 1;
        u32x4 out, out2;
        u32x4 gs, rs, xs;
        if (MODE == MODE_BWD_T) gs = frag_load(static_cast<const unsigned short*>(p.g), s);
        if (MODE == MODE_BWD_DX) {
          rs = frag_load(static_cast<const unsigned short*>(p.r), s);
          xs = frag_load(static_cast<const unsigned short*>(p.xraw), s);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int q = 2 * (s & 1) + half;
          f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MODE != MODE_BWD_DX) b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
          float yv[4], y2v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = acc[t][4 * q + r];
            if (MODE == MODE_FWD) {
              float xv = elem(xr[s], half, r);
              if (p.rectify) xv = fmaxf(xv, 0.f);
              yv[r] = gdn_apply<INV, EPSH>(xv, a + b4[r]);
            } else if (MODE == MODE_BWD_T) {
              float xv = elem(xr[s], half, r);
              if (p.rectify) xv = fmaxf(xv, 0.f);
              const float gv = elem(gs, half, r);
              float pw, c;
              gdn_grad_factors<INV, EPSH>(a + b4[r], &pw, &c);
              yv[r] = c * gv * xv;        // T
              y2v[r] = gv * pw;           // R
            } else {
              const float xv = elem(xs, half, r);
              float du;                    // d|x|^alpha / dx (relu'd input: zero slope below 0)
              if (p.alpha2) du = 2.f * (p.rectify ? fmaxf(xv, 0.f) : xv);
              else du = p.rectify ? (xv > 0.f ? 1.f : 0.f) : (xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f));
              float d = elem(rs, half, r) + a * du;
              if (p.rectify && !(xv > 0.f)) d = 0.f;
              yv[r] = d;
            }
          }
          out[2 * half] = pack_bf16(yv[0], yv[1]);
          out[2 * half + 1] = pack_bf16(yv[2], yv[3]);
          if (MODE == MODE_BWD_T) {
            out2[2 * half] = pack_bf16(y2v[0], y2v[1]);
            out2[2 * half + 1] = pack_bf16(y2v[2], y2v[3]);
          }
        }
        frag_store(y, s, out);
        if (MODE == MODE_BWD_T) frag_store(static_cast<unsigned short*>(p.y2), s, out2);
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
template <int KT, int MODE>
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
        if (MODE != MODE_BWD_DX && p.rectify) {
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
        if (MODE != MODE_BWD_DX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = p.alpha2 ? u[e] * u[e] : fabsf(u[e]);
        }
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
      constexpr bool INV = decltype(inv)::value, EPSH = decltype(epsh)::value;
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long at = row + 32 * t + 8 * q + 4 * h;
          f32x4 out, out2;
          if (MODE == MODE_FWD) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = gdn_apply<INV, EPSH>(xr[t][q][r], acc[t][4 * q + r] + b4[r]);
          } else if (MODE == MODE_BWD_T) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.g) + at);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pw, c;
              gdn_grad_factors<INV, EPSH>(acc[t][4 * q + r] + b4[r], &pw, &c);
              out[r] = c * g4[r] * xr[t][q][r];
              out2[r] = g4[r] * pw;
            }
            if (live) *reinterpret_cast<f32x4*>(static_cast<float*>(p.y2) + at) = out2;
          } else {
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.r) + at);
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.xraw) + at);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float xv = x4[r];
              float du;
              if (p.alpha2) du = 2.f * (p.rectify ? fmaxf(xv, 0.f) : xv);
              else du = p.rectify ? (xv > 0.f ? 1.f : 0.f) : (xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f));
              float d = r4[r] + acc[t][4 * q + r] * du;
              if (p.rectify && !(xv > 0.f)) d = 0.f;
              out[r] = d;
            }
          }
          if (live) *reinterpret_cast<f32x4*>(y + at) = out;
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
// Parameter gradients:  dgamma[j][i] = sum_p u_j[p] T_i[p],  dbeta[i] = sum_p T_i[p].
// A [C x C] = U^T T contraction over PIXELS: both MFMA operands need, per lane, consecutive
// pixels of one channel, i.e. the transpose of the channels-last tensors.  A block stages
// 64 pixels of u and T through LDS (bf16: written transposed [channel][pixel] so that a
// fragment is one ds_read_b128; f32: copied as is, fragments are conflict-free ds_read_b32),
// its four waves own the (j-tile, i-tile) pairs of one parity class each (so every fragment
// read feeds up to KT/2 MFMAs), and each block leaves a [C*C + C] partial that
// gdn_param_reduce_kernel sums in a fixed order (deterministic, no float atomics).
// ---------------------------------------------------------------------------
constexpr int PG_PIX = 64;        // pixels per LDS stage
constexpr int PG_STRIDE = 72;     // bf16 elements per transposed LDS row (144 B: b128 reads conflict-free)

template <typename T, int KT>
__global__ void __launch_bounds__(256) gdn_param_grad_kernel(GdnParams p, float* partial) {
  constexpr int C = KT * 32;
  constexpr int NH = (KT + 1) / 2;
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ unsigned char smem[];
  // bf16: uT[C][PG_STRIDE], tT[C][PG_STRIDE] (u16); f32: us[PG_PIX][C], ts[PG_PIX][C] (float)
  unsigned short* uT = reinterpret_cast<unsigned short*>(smem);
  unsigned short* tT = uT + C * PG_STRIDE;
  float* us = reinterpret_cast<float*>(smem);
  float* ts = us + PG_PIX * C;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wj = w >> 1, wi = w & 1, i32 = lane & 31, h = lane >> 5;
  const T* x = static_cast<const T*>(p.x);
  const T* tsrc = static_cast<const T*>(p.g);   // T = dL/dn from pass 1

  f32x16 acc[NH][NH];
#pragma unroll
  for (int a = 0; a < NH; ++a)
#pragma unroll
    for (int b = 0; b < NH; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;

  const long long stages = (p.pixels + PG_PIX - 1) / PG_PIX;
  for (long long st = blockIdx.x; st < stages; st += gridDim.x) {
    const long long p0 = st * PG_PIX;
    __syncthreads();
    if (BF) {
      // chunk = 8 channels of one pixel; consecutive lanes take consecutive pixels so that the
      // transposed 2-byte LDS writes of a wave fall in one 128-byte row segment.
      constexpr int chunks = PG_PIX * C / 8;
      for (int c = tid; c < chunks; c += 256) {
        const int px = c % PG_PIX, cg = c / PG_PIX;
        u32x4 xv = u32x4{0, 0, 0, 0}, tv = u32x4{0, 0, 0, 0};
        if (p0 + px < p.pixels) {
          xv = *reinterpret_cast<const u32x4*>(x + (p0 + px) * C + 8 * cg);
          tv = *reinterpret_cast<const u32x4*>(tsrc + (p0 + px) * C + 8 * cg);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int xb = (e & 1) ? (xv[e >> 1] >> 16) : (xv[e >> 1] & 0xFFFFu);
          const unsigned int tb = (e & 1) ? (tv[e >> 1] >> 16) : (tv[e >> 1] & 0xFFFFu);
          float f = bf16_bits_to_float(xb);
          if (p.rectify) f = fmaxf(f, 0.f);
          f = p.alpha2 ? f * f : fabsf(f);
          uT[(8 * cg + e) * PG_STRIDE + px] = static_cast<unsigned short>(float_to_bf16_bits(f));
          tT[(8 * cg + e) * PG_STRIDE + px] = static_cast<unsigned short>(tb);
        }
      }
    } else {
      constexpr int chunks = PG_PIX * C / 4;
      for (int c = tid; c < chunks; c += 256) {
        const int px = c / (C / 4), cg = c % (C / 4);
        f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f}, tv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p0 + px < p.pixels) {
          xv = *reinterpret_cast<const f32x4*>(x + (p0 + px) * C + 4 * cg);
          tv = *reinterpret_cast<const f32x4*>(tsrc + (p0 + px) * C + 4 * cg);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f = xv[e];
          if (p.rectify) f = fmaxf(f, 0.f);
          xv[e] = p.alpha2 ? f * f : fabsf(f);
        }
        *reinterpret_cast<f32x4*>(us + px * C + 4 * cg) = xv;
        *reinterpret_cast<f32x4*>(ts + px * C + 4 * cg) = tv;
      }
    }
    __syncthreads();
    // dbeta: thread c sums column c of the staged T tile
    if (tid < C) {
      if (BF) {
#pragma unroll
        for (int k = 0; k < PG_PIX / 8; ++k) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(tT + tid * PG_STRIDE + 8 * k);
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum += bf16_bits_to_float(v[e] & 0xFFFFu) + bf16_bits_to_float(v[e] >> 16);
        }
      } else {
#pragma unroll 8
        for (int k = 0; k < PG_PIX; ++k) bsum += ts[k * C + tid];
      }
    }
    if (BF) {
#pragma unroll
      for (int ks = 0; ks < PG_PIX / 16; ++ks) {
        bf16x8 af[NH], bfr[NH];
#pragma unroll
        for (int a = 0; a < NH; ++a) {
          const int jt = wj + 2 * a, it = wi + 2 * a;
          if (jt < KT) af[a] = *reinterpret_cast<const bf16x8*>(uT + (32 * jt + i32) * PG_STRIDE + 16 * ks + 8 * h);
          if (it < KT) bfr[a] = *reinterpret_cast<const bf16x8*>(tT + (32 * it + i32) * PG_STRIDE + 16 * ks + 8 * h);
        }
#pragma unroll
        for (int a = 0; a < NH; ++a)
#pragma unroll
          for (int b = 0; b < NH; ++b)
            if (wj + 2 * a < KT && wi + 2 * b < KT)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int k2 = 0; k2 < PG_PIX / 2; ++k2) {
        float af[NH], bfr[NH];
#pragma unroll
        for (int a = 0; a < NH; ++a) {
          const int jt = wj + 2 * a, it = wi + 2 * a;
          af[a] = jt < KT ? us[(2 * k2 + h) * C + 32 * jt + i32] : 0.f;
          bfr[a] = it < KT ? ts[(2 * k2 + h) * C + 32 * it + i32] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < NH; ++a)
#pragma unroll
          for (int b = 0; b < NH; ++b)
            if (wj + 2 * a < KT && wi + 2 * b < KT)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  float* out = partial + static_cast<size_t>(blockIdx.x) * (C * C + C);
#pragma unroll
  for (int a = 0; a < NH; ++a)
#pragma unroll
    for (int b = 0; b < NH; ++b) {
      const int jt = wj + 2 * a, it = wi + 2 * b;
      if (jt < KT && it < KT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
          out[j * C + 32 * it + i32] = acc[a][b][r];
        }
      }
    }
  if (tid < C) out[C * C + tid] = bsum;
}

// dgamma / dbeta += sum over block partials (fixed order).
__global__ void gdn_param_reduce_kernel(const float* partial, int blocks, int C, float* dgamma,
                                        float* dbeta) {
  const int n = C * C + C;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[static_cast<size_t>(b) * n + idx];
  if (idx < C * C) dgamma[idx] += s; else dbeta[idx - C * C] += s;
}

// Builds the fragment-ordered Gamma^T image (+ beta behind it) the main kernels copy to LDS.
// transposed = 1 swaps the roles of the two gamma indices (MODE_BWD_DX contracts over i).
__global__ void gdn_prep_bf16_kernel(const float* gamma, const float* beta, int C, int transposed,
                                     bf16x8* image) {
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
      v[e] = static_cast<__bf16>(transposed ? gamma[(32 * t + i) * C + ch] : gamma[ch * C + 32 * t + i]);
    }
    image[idx] = v;
  }
  float* b = reinterpret_cast<float*>(image + KT * KS * 64);
  if (idx < C) b[idx] = beta[idx];
}

__global__ void gdn_prep_f32_kernel(const float* gamma, const float* beta, int C, int transposed,
                                    f32x4* image) {
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
      v[e] = transposed ? gamma[(32 * t + i) * C + ch] : gamma[ch * C + 32 * t + i];
    }
    image[idx] = v;
  }
  float* b = reinterpret_cast<float*>(image + KT * KT * 4 * 64);
  if (idx < C) b[idx] = beta[idx];
}

template <int KT, int MODE>
int launch_gdn(GdnParams p, int dtype, hipStream_t st) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // bf16: 8 waves (2 per SIMD, <= 256 VGPRs each); f32: 4 waves so that the 96 x + 96
  // accumulator registers fit the 512-entry unified file without spilling.
  const int waves_per_block = dtype == 1 ? 8 : 4;
  const long long want = ceil_div(p.tiles, waves_per_block);
  const unsigned blocks = static_cast<unsigned>(std::max<long long>(1, std::min<long long>(want, cus)));
  const char* label = MODE == MODE_FWD ? "gdn_forward" : MODE == MODE_BWD_T ? "gdn_backward_t" : "gdn_backward_dx";
  const int transposed = MODE == MODE_BWD_DX;
  DevBuf image;
  if (dtype == 1) {
    const size_t lds = sizeof(bf16x8) * KT * (KT * 2) * 64 + sizeof(float) * KT * 32;
    TFC_HIP(image.alloc(lds, st));
    const int n = KT * KT * 2 * 64;
    hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                       KT * 32, transposed, image.as<bf16x8>());
    p.image = image.p;
    KernelTimer timer(label, st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_bf16_kernel<KT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((gdn_fwd_bf16_kernel<KT, MODE>), dim3(blocks), dim3(64 * waves_per_block), lds, st, p);
  } else {
    const size_t lds = sizeof(f32x4) * KT * KT * 4 * 64 + sizeof(float) * KT * 32;
    if (lds > 160 * 1024)
      return fail("tfc_gdn: float32 path supports up to 192 channels (Gamma must fit in LDS)");
    TFC_HIP(image.alloc(lds, st));
    const int n = KT * KT * 4 * 64;
    hipLaunchKernelGGL(gdn_prep_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                       KT * 32, transposed, image.as<f32x4>());
    p.image = image.p;
    KernelTimer timer(label, st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_f32_kernel<KT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((gdn_fwd_f32_kernel<KT, MODE>), dim3(blocks), dim3(64 * waves_per_block), lds, st, p);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

template <typename T, int KT>
int launch_param_grad(GdnParams p, float* dgamma, float* dbeta, hipStream_t st) {
  constexpr int C = KT * 32;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long stages = ceil_div(p.pixels, static_cast<long long>(PG_PIX));
  const int blocks = static_cast<int>(std::max<long long>(1, std::min<long long>(stages, cus)));
  const size_t lds = sizeof(T) == 2 ? sizeof(unsigned short) * 2 * C * PG_STRIDE : sizeof(float) * 2 * PG_PIX * C;
  DevBuf partial;
  TFC_HIP(partial.alloc(sizeof(float) * static_cast<size_t>(blocks) * (C * C + C), st));
  {
    KernelTimer timer("gdn_backward_params", st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_param_grad_kernel<T, KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((gdn_param_grad_kernel<T, KT>), dim3(blocks), dim3(256), lds, st, p, partial.as<float>());
  }
  const int n = C * C + C;
  hipLaunchKernelGGL(gdn_param_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partial.as<float>(),
                     blocks, C, dgamma, dbeta);
  TFC_HIP(hipGetLastError());
  return 0;
}

// Three passes (all reading / writing each tensor once):
//   1. MODE_BWD_T:   x, g        -> T (= dL/dn), R (= g n^s)
//   2. MODE_BWD_DX:  T, R, x     -> dx
//   3. param grads:  x, T        -> dgamma, dbeta
template <int KT>
int run_gdn_backward(GdnParams p, const void* g, void* dx, int dtype, float* dbeta, float* dgamma,
                     hipStream_t st) {
  const size_t bytes = static_cast<size_t>(p.pixels) * p.C * (dtype == 1 ? 2 : 4);
  DevBuf tbuf, rbuf;
  TFC_HIP(tbuf.alloc(bytes, st));
  TFC_HIP(rbuf.alloc(bytes, st));
  GdnParams a = p;
  a.g = g; a.y = tbuf.p; a.y2 = rbuf.p;
  if (int rc = launch_gdn<KT, MODE_BWD_T>(a, dtype, st)) return rc;
  GdnParams b = p;
  b.x = tbuf.p; b.r = rbuf.p; b.xraw = p.x; b.y = dx;
  if (int rc = launch_gdn<KT, MODE_BWD_DX>(b, dtype, st)) return rc;
  GdnParams c = p;
  c.g = tbuf.p;
  if (dtype == 1) return launch_param_grad<unsigned short, KT>(c, dgamma, dbeta, st);
  return launch_param_grad<float, KT>(c, dgamma, dbeta, st);
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
  GdnParams p{};
  p.x = x; p.y = y; p.beta = beta; p.gamma = gamma;
  p.pixels = pixels; p.C = static_cast<int>(channels);
  p.inverse = inverse; p.rectify = rectify; p.alpha2 = alpha_mode == 2; p.eps_half = eps_mode == 1;
  p.tiles = ceil_div(pixels, 32);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (channels / 32) {
    case 1: return launch_gdn<1, MODE_FWD>(p, dtype, st);
    case 2: return launch_gdn<2, MODE_FWD>(p, dtype, st);
    case 3: return launch_gdn<3, MODE_FWD>(p, dtype, st);
    case 4: return launch_gdn<4, MODE_FWD>(p, dtype, st);
    case 5: return launch_gdn<5, MODE_FWD>(p, dtype, st);
    case 6: return launch_gdn<6, MODE_FWD>(p, dtype, st);
    case 7: return launch_gdn<7, MODE_FWD>(p, dtype, st);
    default: return launch_gdn<8, MODE_FWD>(p, dtype, st);
  }
}

extern "C" int tfc_gdn_backward(const void* x, const void* g, void* dx, int dtype, int64_t pixels,
                                int64_t channels, const float* beta, const float* gamma, int inverse,
                                int rectify, int alpha_mode, int eps_mode, float* dbeta, float* dgamma,
                                void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("tfc_gdn_backward: dtype must be 0 (float32) or 1 (bfloat16)");
  if (alpha_mode != 1 && alpha_mode != 2) return fail("tfc_gdn_backward: alpha must be 1 or 2");
  if (eps_mode != 0 && eps_mode != 1) return fail("tfc_gdn_backward: epsilon must be 1 or 0.5");
  if (channels <= 0 || channels % 32 != 0 || channels > 256)
    return fail("tfc_gdn_backward: channels must be a multiple of 32, at most 256 (got %lld)",
                static_cast<long long>(channels));
  if (dtype == 0 && channels > 192)
    return fail("tfc_gdn_backward: float32 path supports up to 192 channels (Gamma must fit in LDS)");
  if (pixels == 0) return 0;
  GdnParams p{};
  p.x = x; p.beta = beta; p.gamma = gamma;
  p.pixels = pixels; p.C = static_cast<int>(channels);
  p.inverse = inverse; p.rectify = rectify; p.alpha2 = alpha_mode == 2; p.eps_half = eps_mode == 1;
  p.tiles = ceil_div(pixels, 32);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (channels / 32) {
    case 1: return run_gdn_backward<1>(p, g, dx, dtype, dbeta, dgamma, st);
    case 2: return run_gdn_backward<2>(p, g, dx, dtype, dbeta, dgamma, st);
    case 3: return run_gdn_backward<3>(p, g, dx, dtype, dbeta, dgamma, st);
    case 4: return run_gdn_backward<4>(p, g, dx, dtype, dbeta, dgamma, st);
    case 5: return run_gdn_backward<5>(p, g, dx, dtype, dbeta, dgamma, st);
    case 6: return run_gdn_backward<6>(p, g, dx, dtype, dbeta, dgamma, st);
    case 7: return run_gdn_backward<7>(p, g, dx, dtype, dbeta, dgamma, st);
    default: return run_gdn_backward<8>(p, g, dx, dtype, dbeta, dgamma, st);
  }
}
