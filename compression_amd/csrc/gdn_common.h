#pragma once
// GDN / IGDN tile kernels for gfx950 (channels-last, [pixels, C]); shared by gdn.hip (forward)
// and gdn_backward.hip.
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
//     transpose, no shuffle and no second read of x.  The forward kernel's y
//     leaves as whole 128-byte lines through a wave-private LDS area
//     (TFC_GDN_LINES, round 6); the backward kernels' outputs with the access
//     pattern x came in with.
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
  float relu_floor;     // 0 if rectify else -inf:  x' = max(x, relu_floor)
  float a2;             // 1 if alpha == 2 else 0
  // general (learned) exponents, GEN kernels only: u = x'^alpha, y = x' n^eps_s  (gdn.py:377-416, tf.pow)
  int gen;
  float alpha;          // >= 1
  float eps_s;          // -epsilon (GDN) / +epsilon (IGDN)
  float negf;           // x'^alpha for x' < 0 is |x'|^alpha * negf: +1 / -1 for an even / odd integer alpha, else NaN
  float negf_e;         // the same for n^epsilon
  long long tiles;      // ceil(pixels / 32)
  const void* image;    // fragment-ordered Gamma^T (+ beta) built by gdn_prep_*_kernel
  const void* prepared; // != null: the caller's image of these parameters (tfc_gdn_params): no prep launch
  int nt_store;         // forward, whole-line stores: non-temporal (the tensors are larger than the caches)
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

// The kernels come in two flavours: PLAIN (alpha = 1, no rectify — the layer's defaults, where
// |x| is a sign-bit clear on the packed words) and the general one, which folds rectify and
// alpha into arithmetic on wave-uniform constants (no per-element branches either way).
// u = |x'|^alpha for x' = max(x, relu_floor)
__device__ inline float gdn_u(float xe, float a2) {
  const float ax = fabsf(xe);
  return ax * fmaf(ax - 1.f, a2, 1.f);
}
// x'^alpha for any alpha > 0 as tf.pow gives it: v_log_f32 / v_exp_f32 (base 2, 1 ulp); 0 at x' = 0
__device__ inline float gdn_u_gen(float xe, float alpha, float negf) {
  const float m = __builtin_amdgcn_exp2f(alpha * __builtin_amdgcn_logf(fabsf(xe)));
  return xe < 0.f ? m * negf : m;
}
// x' n^s, a negative n treated like a negative x' above
__device__ inline float gdn_apply_gen(float x, float n, float eps_s, float negf_e) {
  const float m = __builtin_amdgcn_exp2f(eps_s * __builtin_amdgcn_logf(fabsf(n)));
  return x * (n < 0.f ? m * negf_e : m);
}
// d|x'|^alpha / dx':  sign(x') for alpha = 1, 2 x' for alpha = 2
__device__ inline float gdn_du(float xe, float a2) {
  float sg = xe > 0.f ? 1.f : 0.f;
  sg = xe < 0.f ? -1.f : sg;
  return sg * fmaf(2.f * fabsf(xe) - 1.f, a2, 1.f);
}
// dx = R + a * du, zero where the rectifier is closed
template <bool PLAIN>
__device__ inline float gdn_dx(float xv, float r, float a, float relu_floor, float a2) {
  if (PLAIN) {
    const float sa = __uint_as_float(__float_as_uint(a) ^ (__float_as_uint(xv) & 0x80000000u));
    return r + (xv == 0.f ? 0.f : sa);
  }
  const float xe = fmaxf(xv, relu_floor);
  const float d = fmaf(a, gdn_du(xe, a2), r);
  return xv > relu_floor ? d : 0.f;
}

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
// TFC_GDN_LINES (round 6): the forward kernel's y leaves as WHOLE 128-byte lines.  Straight from the accumulators' layout a
// store instruction is 32 pixels x 32 bytes — four instructions fill a line, each a partial write on its way through L2 —
// and with them in flight the kernel's reads ran at 60 % of what they reach alone (builds without the stores 32 us, without
// the loads 28, with both 54.5 on [262144, 192]: profiles/r06_notes.md).  Instead four K steps' results (64 channels, one
// line per pixel) go to a wave-private 4 KB of LDS as [pixel][8 granules], granule g at g ^ (pixel / 2 & 7) (all banks
// once per 16 lanes), and leave 8 lanes to a line, 8 pixels to an instruction.
#ifndef TFC_GDN_LINES
#define TFC_GDN_LINES 1
#endif
// (Measured in round 6 and not kept, profiles/r06_notes.md: global accesses by lane pairs with ds_bpermute_b32 between memory
// order and the MFMA's, either direction; the x tile by linear buffer_load ... lds into a wave-private buffer; non-temporal
// loads.)  TFC_GDN_EXP (timing builds, wrong results): 1 no stores, 2 no loads.
template <int KT, int MODE, bool PLAIN, bool GEN = false>
__global__ void __launch_bounds__(512) gdn_fwd_bf16_kernel(GdnParams p) {
  static_assert(!GEN || (MODE == MODE_FWD && !PLAIN), "general exponents: forward only");
  constexpr int C = KT * 32;
  constexpr int KS = KT * 2;
  extern __shared__ unsigned char smem[];
  bf16x8* afrag = reinterpret_cast<bf16x8*>(smem);
  float* beta_s = reinterpret_cast<float*>(smem + sizeof(bf16x8) * KT * KS * 64);

  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  // (the wave's number as a scalar: tile numbers, the tile's base addresses and bounds then live in SGPRs)
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) +
                         __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const long long nwaves = static_cast<long long>(gridDim.x) * (blockDim.x >> 6);
  const unsigned short* x = static_cast<const unsigned short*>(p.x);
  unsigned short* y = static_cast<unsigned short*>(p.y);

  // Forward, C <= 192: the next tile's x is fetched into spare registers while this tile is
  // contracted (+4 % on C3), and the first tile's x is requested before the fragment image is
  // copied into LDS, so the HBM latency of the first tile overlaps that copy.  Measured
  // alternatives that did NOT pay (profiles/r01_e_gdn_notes.md): double-buffered A fragments, and
  // fully coalesced tile I/O staged through LDS.
  constexpr bool PREFETCH = MODE == MODE_FWD && (KT <= 5 || (PLAIN && KT == 6));
  constexpr bool LINES = TFC_GDN_LINES != 0 && MODE == MODE_FWD && KT <= 7;      // (256 channels: the image leaves no room)
  constexpr int IMG_BYTES = static_cast<int>(sizeof(bf16x8)) * KT * KS * 64 + C * 4;
  u32x4 xn[PREFETCH ? KS : 1];
  auto fetch = [&](long long tile) {
#if defined(TFC_GDN_EXP) && (TFC_GDN_EXP & 2)
#pragma unroll
    for (int s = 0; s < KS; ++s) xn[PREFETCH ? s : 0] = u32x4{lane + 0x3f803f80u, static_cast<unsigned int>(tile), 0x3f803f80u, 0x40004000u};   // (timing: no loads)
    return;
#endif
    const long long pix = tile * 32 + (lane & 31);
    const long long row = (pix < p.pixels ? pix : p.pixels - 1) * C;
#pragma unroll
    for (int s = 0; s < KS; ++s) xn[PREFETCH ? s : 0] = *reinterpret_cast<const u32x4*>(x + row + 16 * s + 8 * h);
  };
  {
    // fragment image (built once per call by gdn_prep_bf16_kernel): linear 16-byte copy.  Its loads are
    // issued first and the first tile's x right behind them, so the wait before the LDS writes covers
    // the image only (loads complete in order) and the x latency runs under the copy.
    const u32x4* src = static_cast<const u32x4*>(p.image);
    u32x4* dstv = reinterpret_cast<u32x4*>(smem);
    constexpr int n16 = KT * KS * 64 + (C * 4) / 16;
    constexpr int PER = (n16 + 511) / 512;   // the kernel is launched with 512 threads
    u32x4 img[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = threadIdx.x + k * 512;
      if (i < n16) img[k] = src[i];
    }
    if (PREFETCH && wave < p.tiles) fetch(wave);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = threadIdx.x + k * 512;
      if (i < n16) dstv[i] = img[k];
    }
  }
  __syncthreads();

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
      const u32x4 v = PREFETCH ? xn[PREFETCH ? s : 0] : *reinterpret_cast<const u32x4*>(x + row + 16 * s + 8 * h);
      const auto s0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
      xr[s] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
    if (PREFETCH && tile + nwaves < p.tiles) fetch(tile + nwaves);
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
      if (MODE != MODE_BWD_DX) {
        if (PLAIN) {
          u &= 0x7FFF7FFFu;
        } else {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float lo = fmaxf(bf16_bits_to_float(u[w] & 0xFFFFu), p.relu_floor);
            const float hi = fmaxf(__uint_as_float(u[w] & 0xFFFF0000u), p.relu_floor);
            u[w] = GEN ? pack_bf16(gdn_u_gen(lo, p.alpha, p.negf), gdn_u_gen(hi, p.alpha, p.negf))
                       : pack_bf16(gdn_u(lo, p.a2), gdn_u(hi, p.a2));
          }
        }
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
      if constexpr (LINES) {
        // (this lane: channels 16 s + 8 h + {0 .. 7} of pixel lane & 31 = granule 2 (s & 3) + h of the pixel's line s >> 2)
        unsigned char* const ost = smem + ((IMG_BYTES + 15) & ~15) + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) * 4096;
        auto slot = [&](int pix, int g) -> unsigned char* { return ost + pix * 128 + ((g ^ ((pix >> 1) & 7)) << 4); };
        auto wave_sync = [&]() __attribute__((always_inline)) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        *reinterpret_cast<u32x4*>(slot(lane & 31, 2 * (s & 3) + h)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        if ((s & 3) == 3 || s == KS - 1) {
          const int granules = 2 * ((s & 3) + 1);            // of this line (a last line of 32 or 96 channels: fewer)
          wave_sync();
          unsigned short* const tbase = base + tile * (32 * C);                       // (wave-uniform)
          const int loff = (lane >> 3) * C + 8 * (lane & 7);
          const long long left = p.pixels - tile * 32;
          const int room = left < 32 ? static_cast<int>(left) : 32;                   // pixels of this tile inside the tensor
          // (pixels 8 k + lane / 8: the swizzle of k + 2 is that of k — two LDS addresses, + 2 KB)
          const unsigned char* const rd[2] = {slot(lane >> 3, lane & 7), slot(8 + (lane >> 3), lane & 7)};
          const bool on = (lane & 7) < granules;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(rd[k & 1] + 2048 * (k >> 1));
#if defined(TFC_GDN_EXP) && (TFC_GDN_EXP & 1)
            if (on && 8 * k + (lane >> 3) < room && v.x == 0x12345u) *reinterpret_cast<u32x4*>(tbase + loff + 8 * k * C + 64 * (s >> 2)) = v;      // (timing: no stores)
#else
            // whole lines, non-temporal where the tensors cannot stay in the caches anyway (p.nt_store, launch_gdn_variant):
            // 46.8 -> 41.8 us on [262144, 192].  (With the 32-byte partial stores of rounds 1-5 non-temporal was 2-3x slower;
            // non-temporal LOADS — still 32 bytes of a line per instruction — 57 us)
            if (on && 8 * k + (lane >> 3) < room) {
              u32x4* const dst = reinterpret_cast<u32x4*>(tbase + loff + 8 * k * C + 64 * (s >> 2));
              // (as an instruction by hand: with __builtin_nontemporal_store in one branch and a plain store in the other the
              // optimiser merges the two into ONE plain store)
              if (p.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
              else *dst = v;
            }
#endif
            __builtin_amdgcn_sched_barrier(0);        // (one granule at a time: registers)
          }
          wave_sync();
        }
        return;
      }
#if defined(TFC_GDN_EXP) && (TFC_GDN_EXP & 1)
      if (live && s0[0] == 0x12345u) *reinterpret_cast<u32x4*>(base + row + 16 * s + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};   // (timing: no stores)
#else
      if (live) *reinterpret_cast<u32x4*>(base + row + 16 * s + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
#endif
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
              if (!PLAIN) xv = fmaxf(xv, p.relu_floor);
              yv[r] = GEN ? gdn_apply_gen(xv, a + b4[r], p.eps_s, p.negf_e) : gdn_apply<INV, EPSH>(xv, a + b4[r]);
            } else if (MODE == MODE_BWD_T) {
              float xv = elem(xr[s], half, r);
              if (!PLAIN) xv = fmaxf(xv, p.relu_floor);
              const float gv = elem(gs, half, r);
              float pw, c;
              gdn_grad_factors<INV, EPSH>(a + b4[r], &pw, &c);
              yv[r] = c * gv * xv;        // T
              y2v[r] = gv * pw;           // R
            } else {
              yv[r] = gdn_dx<PLAIN>(elem(xs, half, r), elem(rs, half, r), a, p.relu_floor, p.a2);
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
    if (GEN) {
      epilogue(F{}, F{});
    } else if (p.inverse) {
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
template <int KT, int MODE, bool PLAIN, bool GEN = false>
__global__ void __launch_bounds__(256) gdn_fwd_f32_kernel(GdnParams p) {
  static_assert(!GEN || (MODE == MODE_FWD && !PLAIN), "general exponents: forward only");
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
        if (MODE != MODE_BWD_DX && !PLAIN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], p.relu_floor);
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
          for (int e = 0; e < 4; ++e)
            u[e] = PLAIN ? fabsf(u[e]) : GEN ? gdn_u_gen(u[e], p.alpha, p.negf) : gdn_u(u[e], p.a2);
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
            for (int r = 0; r < 4; ++r)
              out[r] = GEN ? gdn_apply_gen(xr[t][q][r], acc[t][4 * q + r] + b4[r], p.eps_s, p.negf_e)
                           : gdn_apply<INV, EPSH>(xr[t][q][r], acc[t][4 * q + r] + b4[r]);
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
            for (int r = 0; r < 4; ++r)
              out[r] = gdn_dx<PLAIN>(x4[r], r4[r], acc[t][4 * q + r], p.relu_floor, p.a2);
          }
          if (live) *reinterpret_cast<f32x4*>(y + at) = out;
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (GEN) {
      epilogue(F{}, F{});
    } else if (p.inverse) {
      if (p.eps_half) epilogue(T{}, T{}); else epilogue(T{}, F{});
    } else {
      if (p.eps_half) epilogue(F{}, T{}); else epilogue(F{}, F{});
    }
  }
}

// Builds the fragment-ordered Gamma^T image (+ beta behind it) the main kernels copy to LDS.
// transposed = 1 swaps the roles of the two gamma indices (MODE_BWD_DX contracts over i).
static __global__ void gdn_prep_bf16_kernel(const float* gamma, const float* beta, int C, int transposed,
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

static __global__ void gdn_prep_f32_kernel(const float* gamma, const float* beta, int C, int transposed,
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

// DTYPES: bit 0 = instantiate the float32 kernels, bit 1 = the bfloat16 ones.
template <int KT, int MODE, bool PLAIN, int DTYPES, bool GEN = false>
int launch_gdn_variant(GdnParams p, int dtype, hipStream_t st) {
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
   if constexpr ((DTYPES & 2) != 0) {
    const size_t lds = sizeof(bf16x8) * KT * (KT * 2) * 64 + sizeof(float) * KT * 32;
    if (p.prepared) {
      p.image = p.prepared;
    } else {
      TFC_HIP(image.alloc(lds, st));
      const int n = KT * KT * 2 * 64;
      hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                         KT * 32, transposed, image.as<bf16x8>());
      p.image = image.p;
    }
    KernelTimer timer(label, st);
    // x + y beyond half the 256 MB Infinity Cache: y's lines go out non-temporal (TFC_GDN_NT = 0 / 1: never / always)
    {
      static const int nt_env = [] { const char* e = std::getenv("TFC_GDN_NT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
      p.nt_store = nt_env >= 0 ? nt_env : (static_cast<long long>(p.pixels) * KT * 32 * 4 > (128ll << 20) ? 1 : 0);
    }
    const size_t lds_lines = TFC_GDN_LINES != 0 && MODE == MODE_FWD && KT <= 7 ? ((lds + 15) & ~size_t{15}) + static_cast<size_t>(waves_per_block) * 4096 : lds;
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_bf16_kernel<KT, MODE, PLAIN, GEN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_lines)));
    hipLaunchKernelGGL((gdn_fwd_bf16_kernel<KT, MODE, PLAIN, GEN>), dim3(blocks), dim3(64 * waves_per_block), lds_lines,
                       st, p);
   } else {
    return fail("tfc_gdn: bfloat16 kernel not built for this configuration");
   }
  } else {
    if constexpr (KT <= 6 && (DTYPES & 1) != 0) {
      const size_t lds = sizeof(f32x4) * KT * KT * 4 * 64 + sizeof(float) * KT * 32;
      if (p.prepared) {
        p.image = p.prepared;
      } else {
        TFC_HIP(image.alloc(lds, st));
        const int n = KT * KT * 4 * 64;
        hipLaunchKernelGGL(gdn_prep_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                           KT * 32, transposed, image.as<f32x4>());
        p.image = image.p;
      }
      KernelTimer timer(label, st);
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_fwd_f32_kernel<KT, MODE, PLAIN, GEN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL((gdn_fwd_f32_kernel<KT, MODE, PLAIN, GEN>), dim3(blocks), dim3(64 * waves_per_block), lds,
                         st, p);
    } else {
      return fail("tfc_gdn: float32 path supports up to 192 channels (Gamma must fit in LDS)");
    }
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

template <int KT, int MODE, int DTYPES = 3>
int launch_gdn(GdnParams p, int dtype, hipStream_t st) {
  p.relu_floor = p.rectify ? 0.f : -__builtin_inff();
  p.a2 = p.alpha2 ? 1.f : 0.f;
  if constexpr (MODE == MODE_FWD) {
    if (p.gen) return launch_gdn_variant<KT, MODE, false, DTYPES, true>(p, dtype, st);
  }
  if (!p.rectify && !p.alpha2) return launch_gdn_variant<KT, MODE, true, DTYPES>(p, dtype, st);
  return launch_gdn_variant<KT, MODE, false, DTYPES>(p, dtype, st);
}

}  // namespace tfc
