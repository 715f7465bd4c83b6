// SignalConv2D weight gradient for gfx950 (the input gradient of either direction is the
// forward kernel of the other direction with the kernel's channel axes swapped; see
// layers/functional.py).  One contraction serves both directions:
//
//   G[t][a][b] = sum over n, q of  A[n, q*s + t - k/2, a] * B[n, q, b]        (zero outside A)
//
//   analysis  y = corr_down(x, w, s):  dw[t][ci][co] = G with A = x,  B = dy          (python/layers/
//   synthesis y = conv_up(x, w, s):    dw[t][ci][co] = G^T with A = dy, B = x          signal_conv.py:663-690, 778-847)
//
// Per kernel tap t this is an [a x b] = A_t^T B contraction over PIXELS, the shape of the GDN
// parameter gradient (gdn_backward.hip): a workgroup owns one tap and a strided set of 64-pixel
// stages of B's grid, stages both operands through LDS transposed ([channel][pixel]; bf16 as
// pixel pairs, software-pipelined), its four waves own the (a-tile, b-tile) parity classes, and
// it leaves one [a x b] partial; a second kernel sums the partials of a tap in a fixed order.
// Channel counts: multiples of 32 up to 256, or <= 4 (the image side of the first / last layer:
// one zero-padded 32-row tile, element-wise staging).  bf16: v_mfma_f32_32x32x16_bf16; f32:
// v_mfma_f32_32x32x2_f32 (exact products, fp32 accumulation either way).
// Roofline: MFMA-bound, 2 * taps * M * a * b FLOP — the forward pass's count.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/tfc_hip.h"
#include "common.h"

namespace tfc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int WG_PIX = 64;       // pixels per LDS stage
constexpr int WG_STRIDE = 72;    // bf16 elements per transposed LDS row (144 B: conflict-free b128 reads)

struct WgradGeom {
  const void* A;
  const void* B;
  long long N;
  int HA, WA, CA;        // A: the tensor read at q*s + t - k/2
  int HB, WB, CB;        // B: the tensor on the q grid
  int kh, kw, stride;
  int chunks;            // workgroups per tap
  float* partial;        // [kh*kw][chunks][CA][CB]
  unsigned int wb_mul, wb_sh, hb_mul, hb_sh;      // n / WB, n / HB as multiplications (wg_fast_div)
  int tr;                // bf16, both tensors wide, < 2^31 pixels: row-major stages + ds_read_b64_tr_b16 (below)
};

// n / d for 0 <= n < 2^31 as a multiplication (signal_conv.hip fast_div): mul = ceil(2^(31 + s) / d), s = ceil(log2 d)
inline void wg_fast_div_setup(unsigned int d, unsigned int* mul, unsigned int* sh) {
  if (d <= 1) { *mul = 0; *sh = 0; return; }
  unsigned int s = 0;
  while ((1ull << s) < d) ++s;
  *mul = static_cast<unsigned int>(((1ull << (31 + s)) + d - 1) / d);
  *sh = s - 1;
}
__device__ inline unsigned int wg_fast_div(unsigned int n, unsigned int mul, unsigned int sh) {
  return mul ? __umulhi(n, mul) >> sh : n;
}
// Row stride (elements) of a row-major bf16 stage in LDS whose rows ds_read_b64_tr_b16 reads without bank conflicts:
// stride in dwords = 16 mod 32 (gdn_backward.hip pg_row_elems).
template <int C>
constexpr int wg_row_elems() { return C + 2 * (((16 - (C / 2) % 32) + 32) % 32); }
typedef __attribute__((ext_vector_type(2))) unsigned int wg_u32x2;

// KTA / KTB = 32-channel tiles of A / B (1 with CA <= 4 = narrow tensor)
template <typename T, int KTA, int KTB, bool TRB = false>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradGeom g) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int RA = KTA * 32, RB = KTB * 32;          // LDS rows
  constexpr int NHA = (KTA + 1) / 2, NHB = (KTB + 1) / 2;
  extern __shared__ unsigned char smem[];
  // bf16: aT[RA][WG_STRIDE], bT[RB][WG_STRIDE] (u16); f32: as[WG_PIX][RA], bs[WG_PIX][RB]
  unsigned short* aT = reinterpret_cast<unsigned short*>(smem);
  unsigned short* bT = aT + RA * WG_STRIDE;
  float* as = reinterpret_cast<float*>(smem);
  float* bs = as + WG_PIX * RA;
  // TR (round 6; g.tr): the stages stay ROW-MAJOR in LDS — a thread's 16-byte pieces go global -> registers -> LDS as they
  // are, a load instruction 1 KB contiguous wherever the tap leaves A's rows contiguous — and the MFMA operands, whose K runs
  // over pixels, are read with ds_read_b64_tr_b16 (gdn_backward.hip TFC_GDN_PG_TR; tools/ubench/tr_read_probe.hip).  The
  // pixel -> (image, row, column) arithmetic is 32-bit with the divisions as multiplications (it was three 64-bit
  // divisions per stage and thread).
  constexpr int RSA = wg_row_elems<RA>(), RSB = wg_row_elems<RB>();
  unsigned short* const asr = reinterpret_cast<unsigned short*>(smem);
  unsigned short* const bsr = asr + WG_PIX * RSA;
  constexpr bool tr = TRB;          // (a build of its own: with both stagings in one kernel the registers ran out)
  static_assert(!TRB || BF, "transposing reads: bfloat16");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wa = w >> 1, wb = w & 1, i32 = lane & 31, h = lane >> 5;
  const int tap = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
  const int ty = tap / g.kw, tx = tap % g.kw;
  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  const long long M = g.N * g.HB * g.WB;
  const bool narrowA = g.CA < 32, narrowB = g.CB < 32;
  const bool small = M < (1ll << 31);

  f32x16 acc[NHA][NHB];
#pragma unroll
  for (int a = 0; a < NHA; ++a)
#pragma unroll
    for (int b = 0; b < NHB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // source row of pixel m of the stage: B row m; A row at the tap's offset (or -1: zero)
  auto rows = [&](long long m, long long* rowA, long long* rowB) {
    *rowB = m < M ? m : -1;
    *rowA = -1;
    if (m < M) {
      int qx, qy;
      long long n;
      if (small) {          // (< 2^31 pixels: the divisions as multiplications — a 64-bit division is ~150 instructions)
        const unsigned int m32 = static_cast<unsigned int>(m);
        const unsigned int r1 = wg_fast_div(m32, g.wb_mul, g.wb_sh);
        qx = static_cast<int>(m32 - r1 * static_cast<unsigned int>(g.WB));
        const unsigned int n1 = wg_fast_div(r1, g.hb_mul, g.hb_sh);
        qy = static_cast<int>(r1 - n1 * static_cast<unsigned int>(g.HB));
        n = n1;
      } else {
        qx = static_cast<int>(m % g.WB);
        qy = static_cast<int>((m / g.WB) % g.HB);
        n = m / (static_cast<long long>(g.WB) * g.HB);
      }
      const int iy = qy * g.stride + ty - g.kh / 2, ix = qx * g.stride + tx - g.kw / 2;
      if (iy >= 0 && iy < g.HA && ix >= 0 && ix < g.WA) *rowA = (n * g.HA + iy) * g.WA + ix;
    }
  };

  // wide tensors: 16-byte chunks held in registers one stage ahead.  bf16: a QUAD of lanes holds two neighbouring channel
  // groups of two neighbouring pixels (lane r of quad tid >> 2: pixel quad_px, channel group quad_cg(k) in round k), so
  // that a lane pair reads 32 contiguous bytes — 32 lines per load instruction instead of the 64 that consecutive
  // lanes = consecutive pixels touch (the CU's address unit takes ~4 cycles per line: gdn_backward.hip, profiles/r06_notes.md)
  // — and the pixel pair that shares a word of the transposed image is two lanes apart.
  const int quad_px = 2 * ((tid >> 2) & 31) + ((tid >> 1) & 1);
  auto quad_cg = [&](int k) -> int { return 2 * ((tid >> 7) + 2 * k) + (tid & 1); };
  constexpr int NA = BF ? KTA : 2 * KTA, NB = BF ? KTB : 2 * KTB;
  u32x4 aq[NA], bq[NB];
  auto fetch = [&](long long st) {
    const long long m0 = st * WG_PIX;
    if constexpr (BF && tr) {
      const unsigned int m32 = static_cast<unsigned int>(m0), M32 = static_cast<unsigned int>(M);
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int c = tid + 256 * k;
        const unsigned int m = m32 + static_cast<unsigned int>(c / (RA / 8));
        const int cg = c % (RA / 8);
        long long ra = -1;
        if (m < M32) {
          const unsigned int r1 = wg_fast_div(m, g.wb_mul, g.wb_sh);              // n * HB + qy
          const int qx = static_cast<int>(m - r1 * static_cast<unsigned int>(g.WB));
          const unsigned int n1 = wg_fast_div(r1, g.hb_mul, g.hb_sh);
          const int qy = static_cast<int>(r1 - n1 * static_cast<unsigned int>(g.HB));
          const int iy = qy * g.stride + ty - g.kh / 2, ix = qx * g.stride + tx - g.kw / 2;
          if (iy >= 0 && iy < g.HA && ix >= 0 && ix < g.WA) ra = (static_cast<long long>(n1) * g.HA + iy) * g.WA + ix;
        }
        aq[k] = ra >= 0 ? *reinterpret_cast<const u32x4*>(A + ra * g.CA + 8 * cg) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int c = tid + 256 * k;
        const unsigned int m = m32 + static_cast<unsigned int>(c / (RB / 8));
        bq[k] = m < M32 ? *reinterpret_cast<const u32x4*>(B + static_cast<long long>(m) * g.CB + 8 * (c % (RB / 8))) : u32x4{0, 0, 0, 0};
      }
      return;
    }
    if (!narrowA) {
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int c = tid + 256 * k;
        const int px = BF ? quad_px : (c / (RA / 4));
        const int off = BF ? 8 * quad_cg(k) : 4 * (c % (RA / 4));
        long long ra, rb;
        rows(m0 + px, &ra, &rb);
        aq[k] = ra >= 0 ? *reinterpret_cast<const u32x4*>(A + ra * g.CA + off) : u32x4{0, 0, 0, 0};
      }
    }
    if (!narrowB) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int c = tid + 256 * k;
        const int px = BF ? quad_px : (c / (RB / 4));
        const int off = BF ? 8 * quad_cg(k) : 4 * (c % (RB / 4));
        long long ra, rb;
        rows(m0 + px, &ra, &rb);
        bq[k] = rb >= 0 ? *reinterpret_cast<const u32x4*>(B + rb * g.CB + off) : u32x4{0, 0, 0, 0};
      }
    }
  };
  // pixel pairs -> 4-byte transposed LDS writes (see gdn_backward.hip)
  auto pair_store = [&](const u32x4& v, unsigned short* base, int px, int cg) {
    const bool odd = px & 1;
    const unsigned int send0 = odd ? v[0] : v[2], send1 = odd ? v[1] : v[3];
    const unsigned int keep0 = odd ? v[2] : v[0], keep1 = odd ? v[3] : v[1];
    unsigned int recv0 = __builtin_amdgcn_update_dpp(0u, send0, 0x4E, 0xF, 0xF, false);      // quad_perm [2, 3, 0, 1]
    unsigned int recv1 = __builtin_amdgcn_update_dpp(0u, send1, 0x4E, 0xF, 0xF, false);
    asm volatile("" : "+v"(recv0), "+v"(recv1));
    const unsigned int e0 = odd ? recv0 : keep0, o0 = odd ? keep0 : recv0;
    const unsigned int e1 = odd ? recv1 : keep1, o1 = odd ? keep1 : recv1;
    unsigned int* dst = reinterpret_cast<unsigned int*>(base + (8 * cg + (odd ? 4 : 0)) * WG_STRIDE + (px & ~1));
    dst[0 * WG_STRIDE / 2] = __builtin_amdgcn_perm(o0, e0, 0x05040100u);
    dst[1 * WG_STRIDE / 2] = __builtin_amdgcn_perm(o0, e0, 0x07060302u);
    dst[2 * WG_STRIDE / 2] = __builtin_amdgcn_perm(o1, e1, 0x05040100u);
    dst[3 * WG_STRIDE / 2] = __builtin_amdgcn_perm(o1, e1, 0x07060302u);
  };
  // narrow tensor (<= 4 channels): one element per thread-iteration, rows >= C stay zero
  auto stage_narrow = [&](const T* src, int C, bool isA, long long st) {
    const long long m0 = st * WG_PIX;
    for (int idx = tid; idx < WG_PIX * 32; idx += 256) {
      const int px = idx % WG_PIX, ch = idx / WG_PIX;
      long long ra, rb;
      rows(m0 + px, &ra, &rb);
      const long long r = isA ? ra : rb;
      T v = T(0.f);
      if (ch < C && r >= 0) v = src[r * C + ch];
      if constexpr (BF) {
        (isA ? aT : bT)[ch * WG_STRIDE + px] = __builtin_bit_cast(unsigned short, v);
      } else {
        (isA ? as : bs)[px * (isA ? RA : RB) + ch] = static_cast<float>(v);
      }
    }
  };
  auto stage_to_lds = [&](long long st) {
    if constexpr (BF && tr) {
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int c = tid + 256 * k;
        *reinterpret_cast<u32x4*>(asr + (c / (RA / 8)) * RSA + 8 * (c % (RA / 8))) = aq[k];
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int c = tid + 256 * k;
        *reinterpret_cast<u32x4*>(bsr + (c / (RB / 8)) * RSB + 8 * (c % (RB / 8))) = bq[k];
      }
      return;
    }
    if (narrowA) {
      stage_narrow(A, g.CA, true, st);
    } else {
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int c = tid + 256 * k;
        if (BF) pair_store(aq[k], aT, quad_px, quad_cg(k));
        else *reinterpret_cast<u32x4*>(as + (c / (RA / 4)) * RA + 4 * (c % (RA / 4))) = aq[k];
      }
    }
    if (narrowB) {
      stage_narrow(B, g.CB, false, st);
    } else {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int c = tid + 256 * k;
        if (BF) pair_store(bq[k], bT, quad_px, quad_cg(k));
        else *reinterpret_cast<u32x4*>(bs + (c / (RB / 4)) * RB + 4 * (c % (RB / 4))) = bq[k];
      }
    }
  };

  const long long stages = (M + WG_PIX - 1) / WG_PIX;
  if (chunk < stages) fetch(chunk);
  for (long long st = chunk; st < stages; st += g.chunks) {
    __syncthreads();
    stage_to_lds(st);
    __syncthreads();
    if (st + g.chunks < stages) fetch(st + g.chunks);
    if constexpr (BF && tr) {
      // this lane's chunk of tile 0, K step 0, first half: pixel 8 (grp >> 1) + (j >> 2), channels 16 (grp & 1) + 4 (j & 3) ...
      const int grp = lane >> 4, j = lane & 15;
      const unsigned int abase = static_cast<unsigned int>(reinterpret_cast<size_t>(asr)) +
                                 static_cast<unsigned int>(((8 * (grp >> 1) + (j >> 2)) * RSA + 16 * (grp & 1) + 4 * (j & 3)) * 2);
      const unsigned int bbase = static_cast<unsigned int>(reinterpret_cast<size_t>(bsr)) +
                                 static_cast<unsigned int>(((8 * (grp >> 1) + (j >> 2)) * RSB + 16 * (grp & 1) + 4 * (j & 3)) * 2);
      wg_u32x2 ra2[2][NHA][2], rb2[2][NHB][2];
      // (the reads of K step ks + 1 are issued in front of the MFMAs of K step ks; the compiler does not count these reads:
      // a wait per K step, tied to the registers they fill.  Plain unrolled code: a generic lambda does not capture
      // variables that only inline-asm operands name)
#define TFC_WG_REQUEST(S, KS)                                                                                             \
      _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                                  \
        _Pragma("unroll") for (int a = 0; a < NHA; ++a) {                                                                 \
          ra2[S][a][hf] = wg_u32x2{0u, 0u};                                                                               \
          if (wa + 2 * a < KTA)                                                                                           \
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ra2[S][a][hf])                                               \
                         : "v"(abase + static_cast<unsigned int>(((16 * (KS) + 4 * hf) * RSA) * 2) + 64u * (wa + 2 * a))); \
        }                                                                                                                 \
        _Pragma("unroll") for (int b = 0; b < NHB; ++b) {                                                                 \
          rb2[S][b][hf] = wg_u32x2{0u, 0u};                                                                               \
          if (wb + 2 * b < KTB)                                                                                           \
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rb2[S][b][hf])                                               \
                         : "v"(bbase + static_cast<unsigned int>(((16 * (KS) + 4 * hf) * RSB) * 2) + 64u * (wb + 2 * b))); \
        }                                                                                                                 \
      }
      TFC_WG_REQUEST(0, 0)
#pragma unroll
      for (int ks = 0; ks < WG_PIX / 16; ++ks) {
        const int S = ks & 1;
#pragma unroll
        for (int a = 0; a < NHA; ++a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra2[S][a][0]), "+v"(ra2[S][a][1]) : : "memory");
#pragma unroll
        for (int b = 0; b < NHB; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb2[S][b][0]), "+v"(rb2[S][b][1]) : : "memory");
        if (ks + 1 < WG_PIX / 16) { TFC_WG_REQUEST(S ^ 1, ks + 1) }
        bf16x8 af[NHA], bfr[NHB];
#pragma unroll
        for (int a = 0; a < NHA; ++a)
          af[a] = __builtin_bit_cast(bf16x8, u32x4{ra2[S][a][0].x, ra2[S][a][0].y, ra2[S][a][1].x, ra2[S][a][1].y});
#pragma unroll
        for (int b = 0; b < NHB; ++b)
          bfr[b] = __builtin_bit_cast(bf16x8, u32x4{rb2[S][b][0].x, rb2[S][b][0].y, rb2[S][b][1].x, rb2[S][b][1].y});
#pragma unroll
        for (int a = 0; a < NHA; ++a)
#pragma unroll
          for (int b = 0; b < NHB; ++b)
            if (wa + 2 * a < KTA && wb + 2 * b < KTB)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
#undef TFC_WG_REQUEST
    } else if (BF) {
#pragma unroll
      for (int ks = 0; ks < WG_PIX / 16; ++ks) {
        bf16x8 af[NHA], bfr[NHB];
#pragma unroll
        for (int a = 0; a < NHA; ++a)
          if (wa + 2 * a < KTA)
            af[a] = *reinterpret_cast<const bf16x8*>(aT + (32 * (wa + 2 * a) + i32) * WG_STRIDE + 16 * ks + 8 * h);
#pragma unroll
        for (int b = 0; b < NHB; ++b)
          if (wb + 2 * b < KTB)
            bfr[b] = *reinterpret_cast<const bf16x8*>(bT + (32 * (wb + 2 * b) + i32) * WG_STRIDE + 16 * ks + 8 * h);
#pragma unroll
        for (int a = 0; a < NHA; ++a)
#pragma unroll
          for (int b = 0; b < NHB; ++b)
            if (wa + 2 * a < KTA && wb + 2 * b < KTB)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int k2 = 0; k2 < WG_PIX / 2; ++k2) {
        float af[NHA], bfr[NHB];
#pragma unroll
        for (int a = 0; a < NHA; ++a)
          af[a] = wa + 2 * a < KTA ? as[(2 * k2 + h) * RA + 32 * (wa + 2 * a) + i32] : 0.f;
#pragma unroll
        for (int b = 0; b < NHB; ++b)
          bfr[b] = wb + 2 * b < KTB ? bs[(2 * k2 + h) * RB + 32 * (wb + 2 * b) + i32] : 0.f;
#pragma unroll
        for (int a = 0; a < NHA; ++a)
#pragma unroll
          for (int b = 0; b < NHB; ++b)
            if (wa + 2 * a < KTA && wb + 2 * b < KTB)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  float* out = g.partial + (static_cast<size_t>(tap) * g.chunks + chunk) * g.CA * g.CB;
#pragma unroll
  for (int a = 0; a < NHA; ++a)
#pragma unroll
    for (int b = 0; b < NHB; ++b) {
      const int at = wa + 2 * a, bt = wb + 2 * b;
      if (at < KTA && bt < KTB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ca = 32 * at + (r & 3) + 8 * (r >> 2) + 4 * h, cb = 32 * bt + i32;
          if (ca < g.CA && cb < g.CB) out[ca * g.CB + cb] = acc[a][b][r];
        }
      }
    }
}

// dw[t][ci][co] += sum over chunks of G (or its transpose)
__global__ void conv_wgrad_reduce_kernel(const float* partial, int taps, int chunks, int CA, int CB,
                                         int transpose, float* dw) {
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  const long long n = static_cast<long long>(taps) * CA * CB;
  if (idx >= n) return;
  const int t = static_cast<int>(idx / (CA * CB));
  const int rem = static_cast<int>(idx % (CA * CB));
  const int ca = rem / CB, cb = rem % CB;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(static_cast<size_t>(t) * chunks + c) * CA * CB + rem];
  if (transpose) dw[(static_cast<long long>(t) * CB + cb) * CA + ca] += s;
  else dw[idx] += s;
}

template <typename T, int KTA, int KTB>
int launch_wgrad(WgradGeom g, int transpose, float* dw, hipStream_t st) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int taps = g.kh * g.kw;
  const long long stages = ceil_div(g.N * g.HB * g.WB, static_cast<long long>(WG_PIX));
  // taps x chunks workgroups: at most two per CU, and not a few more than that (525 workgroups on 256 CUs, one resident
  // per CU, were three rounds of which the last ran 13)
  g.chunks = static_cast<int>(std::max<long long>(1, std::min<long long>(stages, (2 * cus) / taps)));
  DevBuf partial;
  TFC_HIP(partial.alloc(sizeof(float) * static_cast<size_t>(taps) * g.chunks * g.CA * g.CB, st));
  g.partial = partial.as<float>();
  size_t lds = sizeof(T) == 2 ? sizeof(unsigned short) * (KTA + KTB) * 32 * WG_STRIDE
                              : sizeof(float) * WG_PIX * (KTA + KTB) * 32;
  {
    // TFC_WGRAD_TR = 0: the transpose by hand (the round-5 staging)
    static const bool tr_on = [] { const char* e = std::getenv("TFC_WGRAD_TR"); return !(e && e[0] == '0'); }();
    g.tr = tr_on && sizeof(T) == 2 && g.CA >= 32 && g.CB >= 32 && g.N * g.HB * g.WB < (1ll << 31) ? 1 : 0;
    wg_fast_div_setup(static_cast<unsigned int>(g.WB), &g.wb_mul, &g.wb_sh);
    wg_fast_div_setup(static_cast<unsigned int>(g.HB), &g.hb_mul, &g.hb_sh);
    if (g.tr)
      lds = std::max(lds, sizeof(unsigned short) * WG_PIX * (wg_row_elems<KTA * 32>() + wg_row_elems<KTB * 32>()));
  }
  {
    KernelTimer timer("conv2d_wgrad", st);
    if constexpr (sizeof(T) == 2) {
      if (g.tr) {
        TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, KTA, KTB, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL((conv_wgrad_kernel<T, KTA, KTB, true>), dim3(static_cast<unsigned>(taps * g.chunks)), dim3(256),
                           lds, st, g);
      }
    }
    if (!g.tr) {
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, KTA, KTB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL((conv_wgrad_kernel<T, KTA, KTB>), dim3(static_cast<unsigned>(taps * g.chunks)), dim3(256),
                         lds, st, g);
    }
  }
  const long long n = static_cast<long long>(taps) * g.CA * g.CB;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, st,
                     g.partial, taps, g.chunks, g.CA, g.CB, transpose, dw);
  TFC_HIP(hipGetLastError());
  return 0;
}

inline int tiles_of(int64_t c) { return c <= 4 ? 1 : static_cast<int>(c / 32); }

template <typename T>
int dispatch_wgrad(WgradGeom g, int transpose, float* dw, hipStream_t st) {
  const int ka = tiles_of(g.CA), kb = tiles_of(g.CB);
#define TFC_WG(KA, KB) if (ka == KA && kb == KB) return launch_wgrad<T, KA, KB>(g, transpose, dw, st)
#define TFC_WG_ROW(KA) TFC_WG(KA, 1); TFC_WG(KA, 2); TFC_WG(KA, 4); TFC_WG(KA, 6); TFC_WG(KA, 8)
  TFC_WG_ROW(1); TFC_WG_ROW(2); TFC_WG_ROW(4); TFC_WG_ROW(6); TFC_WG_ROW(8);
#undef TFC_WG_ROW
#undef TFC_WG
  return fail("tfc_conv2d_wgrad: channel pair (%d, %d) is not built (each side: <= 4 channels, or 32, 64, 128, "
              "192 or 256)", g.CA, g.CB);
}

}  // namespace tfc

extern "C" int tfc_conv2d_wgrad(const void* a, const void* b, float* dw, int dtype, int64_t n, int64_t ha,
                                int64_t wa, int64_t ca, int64_t hb, int64_t wb, int64_t cb, int kh, int kw,
                                int stride, int transpose, void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("tfc_conv2d_wgrad: dtype must be 0 (float32) or 1 (bfloat16)");
  if (kh < 1 || kw < 1 || stride < 1 || ca < 1 || cb < 1) return fail("tfc_conv2d_wgrad: bad geometry");
  auto ok = [](int64_t c) { return c <= 4 || (c % 32 == 0 && c <= 256); };
  if (!ok(ca) || !ok(cb))
    return fail("tfc_conv2d_wgrad: channel counts must be <= 4 or multiples of 32 up to 256 (got %lld, %lld)",
                static_cast<long long>(ca), static_cast<long long>(cb));
  if (n == 0 || hb == 0 || wb == 0) return 0;
  WgradGeom g{};
  g.A = a; g.B = b; g.N = n;
  g.HA = static_cast<int>(ha); g.WA = static_cast<int>(wa); g.CA = static_cast<int>(ca);
  g.HB = static_cast<int>(hb); g.WB = static_cast<int>(wb); g.CB = static_cast<int>(cb);
  g.kh = kh; g.kw = kw; g.stride = stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == 1 ? dispatch_wgrad<__bf16>(g, transpose, dw, st) : dispatch_wgrad<float>(g, transpose, dw, st);
}
