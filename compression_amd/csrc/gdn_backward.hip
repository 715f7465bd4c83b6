// GDN / IGDN backward for gfx950: dx, dbeta, dgamma of python/layers/gdn.py:371-421 (the
// reference differentiates those ops with TF autodiff; there is no hand-written gradient).
//
//   n = beta + U Gamma,  y = x n^s  (s = -eps GDN / +eps IGDN),  g = dL/dy
//   T = dL/dn = s g y / n,   R = g n^s
//   dx = R + (T Gamma^T) d|x|^alpha/dx,   dbeta = sum_p T,   dgamma = U^T T
#include "gdn_common.h"
#include "reduce_rows.h"

namespace tfc {

// ---------------------------------------------------------------------------
// bf16 backward, passes 1 and 2 fused (C <= 192: both fragment images fit the 160 KB LDS).
// T leaves pass 1 in exactly the register layout the MFMA wants as its B operand (that is what
// the permuted K order buys), so dx = R + (T Gamma^T) du/dx is contracted straight from
// registers: x and g are read once, T (for the parameter gradients) and dx written once.
// The second contraction runs in two halves of the output tiles to stay under 256 VGPRs.
// LDS: [image of Gamma^T | image of Gamma | beta].
// ---------------------------------------------------------------------------
// Build-time variants of the fused kernel, measured on C3 ([262144, 192] bf16, MI355X; tools/gdn_bwd_variants.sh,
// profiles/r02_notes.md): 512 threads, n all at once (round 1): 143 us, 62 registers per lane spilled to scratch
// (+1.2 tensors of HBM reads and +1.1 of writes); 512 threads, n in two groups: 108 us, no spills; 256 threads
// (512 registers per lane, one wave per SIMD) + next tile's x, g prefetched: 105 us; + A fragments one K-step
// ahead: 100 us (spills at 512 threads).
#ifndef TFC_GDN_BWD_THREADS
#define TFC_GDN_BWD_THREADS 256
#endif
#ifndef TFC_GDN_BWD_SPLIT1
#define TFC_GDN_BWD_SPLIT1 1
#endif
#ifndef TFC_GDN_BWD_PREFETCH
#define TFC_GDN_BWD_PREFETCH 1
#endif
#ifndef TFC_GDN_BWD_APIPE
#define TFC_GDN_BWD_APIPE 1
#endif
template <int KT, bool PLAIN>
__global__ void __launch_bounds__(TFC_GDN_BWD_THREADS) gdn_bwd_fused_bf16_kernel(GdnParams p) {
  constexpr int C = KT * 32;
  constexpr int KS = KT * 2;
  constexpr int IMG = KT * KS * 64;       // fragments per image
  extern __shared__ unsigned char smem[];
  const bf16x8* afrag = reinterpret_cast<const bf16x8*>(smem);
  const bf16x8* bfrag2 = afrag + IMG;
  const float* beta_s = reinterpret_cast<const float*>(smem + sizeof(bf16x8) * 2 * IMG);
  {
    const u32x4* src = static_cast<const u32x4*>(p.image);
    u32x4* dstv = reinterpret_cast<u32x4*>(smem);
    constexpr int n16 = 2 * IMG + (C * 4) / 16;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dstv[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = static_cast<long long>(gridDim.x) * (blockDim.x >> 6);
  const unsigned short* x = static_cast<const unsigned short*>(p.x);
  const unsigned short* g = static_cast<const unsigned short*>(p.g);
  unsigned short* tout = static_cast<unsigned short*>(p.y2);
  unsigned short* dx = static_cast<unsigned short*>(p.y);

  // TFC_GDN_BWD_PREFETCH: x and g of the wave's NEXT tile are requested (raw, 96 registers) before the
  // work on the current one starts; needs the 512-register budget of a 256-thread block.
  auto row_of = [&](long long tile) -> long long {
    const long long pix = tile * 32 + (lane & 31);
    return (pix < p.pixels ? pix : p.pixels - 1) * C;
  };
  auto to_frag = [&](const u32x4& v) -> u32x4 {
    const auto s0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
  };
  u32x4 xn[KS], gn[KS];
  if (TFC_GDN_BWD_PREFETCH && wave < p.tiles) {
    const long long row1 = row_of(wave);
#pragma unroll
    for (int s = 0; s < KS; ++s) xn[s] = *reinterpret_cast<const u32x4*>(x + row1 + 16 * s + 8 * h);
#pragma unroll
    for (int s = 0; s < KS; ++s) gn[s] = *reinterpret_cast<const u32x4*>(g + row1 + 16 * s + 8 * h);
  }

  auto one_tile = [&](const long long tile) __attribute__((always_inline)) {
    const long long pix = tile * 32 + (lane & 31);
    const bool live = pix < p.pixels;
    const long long row = (live ? pix : p.pixels - 1) * C;
    auto frag_load = [&](const unsigned short* base, int s) -> u32x4 {
      return to_frag(*reinterpret_cast<const u32x4*>(base + row + 16 * s + 8 * h));
    };
    auto frag_store = [&](unsigned short* base, int s, u32x4 out) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(out.x, out.z, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(out.y, out.w, false, false);
      // unconditional: lanes past the end hold a copy of the last pixel (clamped row) and store the same
      // values to the same place.  A store under `if (live)` is a branch around it, and the compiler's
      // s_waitcnt for the prefetched loads then assumes the path WITHOUT the stores behind them: it waits
      // for everything in flight at the top of every tile.
      *reinterpret_cast<u32x4*>(base + row + 16 * s + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    auto elem = [&](const u32x4& f, int half, int r) -> float {
      const unsigned int word = f[2 * half + (r >> 1)];
      return __uint_as_float((r & 1) ? (word & 0xFFFF0000u) : (word << 16));
    };
    u32x4 xr[KS], gr[KS];
    asm volatile("" ::: "memory");   // keep the (tile-invariant) LDS fragment loads inside the loop
    if (TFC_GDN_BWD_PREFETCH) {
#pragma unroll
      for (int s = 0; s < KS; ++s) xr[s] = to_frag(xn[s]);
#pragma unroll
      for (int s = 0; s < KS; ++s) gr[s] = to_frag(gn[s]);
      if (tile + nwaves < p.tiles) {
        const long long row1 = row_of(tile + nwaves);
#pragma unroll
        for (int s = 0; s < KS; ++s) xn[s] = *reinterpret_cast<const u32x4*>(x + row1 + 16 * s + 8 * h);
#pragma unroll
        for (int s = 0; s < KS; ++s) gn[s] = *reinterpret_cast<const u32x4*>(g + row1 + 16 * s + 8 * h);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) xr[s] = frag_load(x, s);
    }

    // ---- contraction 1: n = beta + U Gamma, and T / R from it; in two groups of output tiles
    // (TFC_GDN_BWD_SPLIT1: all of n at once needs 96 accumulator registers next to x, g and R
    // and spills ~60 registers per lane to scratch: +1 tensor of HBM writes and reads) ----
    u32x4 rr[KS];
    constexpr int H0 = TFC_GDN_BWD_SPLIT1 ? (KT + 1) / 2 : KT;
#pragma unroll
    for (int grp = 0; grp < (TFC_GDN_BWD_SPLIT1 ? 2 : 1); ++grp) {
      const int t0 = grp == 0 ? 0 : H0;
      const int nt = grp == 0 ? H0 : KT - H0;
      if (nt <= 0) continue;
      f32x16 acc[H0];
      bf16x8 a_cur[H0], a_nxt[H0];
#pragma unroll
      for (int t = 0; t < H0; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 u = xr[s];
        if (PLAIN) {
          u &= 0x7FFF7FFFu;
        } else {
          asm volatile("" : "+v"(u));   // no CSE of this unpack with the epilogues' (live-range bloat)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float lo = fmaxf(bf16_bits_to_float(u[w] & 0xFFFFu), p.relu_floor);
            const float hi = fmaxf(__uint_as_float(u[w] & 0xFFFF0000u), p.relu_floor);
            u[w] = pack_bf16(gdn_u(lo, p.a2), gdn_u(hi, p.a2));
          }
        }
        const bf16x8 b = __builtin_bit_cast(bf16x8, u);
        // g is fetched late (its 48 registers do not fit next to x, the accumulators and the A
        // fragments for the whole contraction): the loads fly under the last K-steps' MFMAs.
        if (!TFC_GDN_BWD_PREFETCH && grp == 0 && s == (KS > 3 ? KS - 3 : 0)) {
#pragma unroll
          for (int k = 0; k < KS; ++k) gr[k] = frag_load(g, k);
        }
        if (TFC_GDN_BWD_APIPE) {
          // A fragments one K-step ahead: their LDS latency runs under this step's MFMAs
          if (s == 0) {
#pragma unroll
            for (int t = 0; t < H0; ++t) if (t < nt) a_nxt[t] = afrag[((t0 + t) * KS) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < H0; ++t) a_cur[t] = a_nxt[t];
          if (s + 1 < KS) {
#pragma unroll
            for (int t = 0; t < H0; ++t) if (t < nt) a_nxt[t] = afrag[((t0 + t) * KS + s + 1) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < H0; ++t)
            if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[t], b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < H0; ++t)
            if (t < nt)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[((t0 + t) * KS + s) * 64 + lane], b, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // n = acc + beta, in place and ahead of the variant switch (the loads are common to the four
      // variants; left inside, they are hoisted above the switch all at once)
#pragma unroll
      for (int t = 0; t < H0; ++t) {
        if (t >= nt) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * (t0 + t) + 8 * q + 4 * h);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][4 * q + r] += b4[r];
        }
        // pin the sums here: IR-level sinking would otherwise keep the beta loads live and
        // redo the add at each use
        asm volatile("" : "+v"(acc[t]));
      }
      // ---- T (overwrites g in registers) and R (kept packed) ----
      auto pass1 = [&](auto inv, auto epsh) {
        constexpr bool INV = decltype(inv)::value, EPSH = decltype(epsh)::value;
#pragma unroll
        for (int sl = 0; sl < 2 * H0; ++sl) {
          if (sl >= 2 * nt) continue;
          const int s = 2 * t0 + sl;
          const int t = sl >> 1;
          u32x4 tq, rq;
          // Opaque copies: without them the unpacking of all elements, identical in the four
          // variants, is hoisted above the variant switch and costs ~190 live registers.
          u32x4 xs = xr[s], gs = gr[s];
          asm volatile("" : "+v"(xs), "+v"(gs));
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int q = 2 * (s & 1) + half;
            float tv[4], rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float xv = elem(xs, half, r);
              if (!PLAIN) xv = fmaxf(xv, p.relu_floor);
              const float gv = elem(gs, half, r);
              float pw, c;
              gdn_grad_factors<INV, EPSH>(acc[t][4 * q + r], &pw, &c);
              tv[r] = c * gv * xv;
              rv[r] = gv * pw;
            }
            tq[2 * half] = pack_bf16(tv[0], tv[1]);
            tq[2 * half + 1] = pack_bf16(tv[2], tv[3]);
            rq[2 * half] = pack_bf16(rv[0], rv[1]);
            rq[2 * half + 1] = pack_bf16(rv[2], rv[3]);
          }
          // pin T and R as packed words now (otherwise R = g * pw is sunk to its use in the dx
          // epilogue and the unpacked pw / g floats stay live across the second contraction)
          asm volatile("" : "+v"(tq), "+v"(rq));
          gr[s] = tq;
          rr[s] = rq;
        }
      };
      using TT = std::true_type;
      using FF = std::false_type;
      if (p.inverse) {
        if (p.eps_half) pass1(TT{}, TT{}); else pass1(TT{}, FF{});
      } else {
        if (p.eps_half) pass1(FF{}, TT{}); else pass1(FF{}, FF{});
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) frag_store(tout, s, gr[s]);

    // ---- contraction 2: (T Gamma^T), output tiles in two groups ----
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      constexpr int G0 = (KT + 1) / 2;
      const int t0 = grp == 0 ? 0 : G0;
      const int nt = grp == 0 ? G0 : KT - G0;
      f32x16 acc2[G0];
#pragma unroll
      for (int t = 0; t < G0; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
      bf16x8 a2_cur[G0], a2_nxt[G0];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, gr[s]);
        if (TFC_GDN_BWD_APIPE) {
          if (s == 0) {
#pragma unroll
            for (int t = 0; t < G0; ++t) if (t < nt) a2_nxt[t] = bfrag2[((t0 + t) * KS) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < G0; ++t) a2_cur[t] = a2_nxt[t];
          if (s + 1 < KS) {
#pragma unroll
            for (int t = 0; t < G0; ++t) if (t < nt) a2_nxt[t] = bfrag2[((t0 + t) * KS + s + 1) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < G0; ++t)
            if (t < nt) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_cur[t], b, acc2[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < G0; ++t)
            if (t < nt)
              acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag2[((t0 + t) * KS + s) * 64 + lane], b,
                                                                 acc2[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < G0; ++t) {
        if (t >= nt) continue;
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
          const int s = 2 * (t0 + t) + sh;
          u32x4 out;
          u32x4 xs = xr[s], rs = rr[s];
          asm volatile("" : "+v"(xs), "+v"(rs));
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int q = 2 * sh + half;
            float dv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              dv[r] = gdn_dx<PLAIN>(elem(xs, half, r), elem(rs, half, r), acc2[t][4 * q + r],
                                    p.relu_floor, p.a2);
            out[2 * half] = pack_bf16(dv[0], dv[1]);
            out[2 * half + 1] = pack_bf16(dv[2], dv[3]);
          }
          asm volatile("" : "+v"(out));
          rr[s] = out;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) frag_store(dx, s, rr[s]);
  };
  if (TFC_GDN_BWD_PREFETCH) {
    // The first tile apart: the loop is then entered with the same memory operations in flight as it is
    // repeated with (the next tile's loads, then this tile's stores), and the compiler's s_waitcnt for the
    // loads at the top of an iteration leaves the stores behind them alone.
    if (wave < p.tiles) {
      one_tile(wave);
      for (long long tile = wave + nwaves; tile < p.tiles; tile += nwaves) one_tile(tile);
    }
  } else {
    for (long long tile = wave; tile < p.tiles; tile += nwaves) one_tile(tile);
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
// sum_rows_kernel (reduce_rows.h) sums in a fixed order (deterministic, no float atomics).
// ---------------------------------------------------------------------------
constexpr int PG_PIX = 64;        // pixels per LDS stage
constexpr int PG_STRIDE = 72;     // bf16 elements per transposed LDS row (144 B: b128 reads conflict-free)

// Workgroups per CU the kernel is built for.  Two (bf16, up to 192 channels: 7 registers spill) were measured
// SLOWER on C3, 97 us against 82 us: twice the block partials to write and to reduce outweigh the overlap
// of the two barriers per stage.
#ifndef TFC_GDN_PG_WGS
#define TFC_GDN_PG_WGS 1
#endif
template <typename T, int KT>
constexpr int pg_wgs() { return (sizeof(T) == 2 && KT <= 6) ? TFC_GDN_PG_WGS : 1; }
// TFC_GDN_PG_TR (round 6, bf16): the stage stays ROW-MAJOR in LDS ([pixel][channel], rows padded so that the reads below are
// conflict-free) — a thread's 16-byte pieces go global -> registers -> LDS as they are, linear on both sides — and the MFMA
// operands, whose K runs over pixels, are read with ds_read_b64_tr_b16: every lane gives the address of four consecutive
// channels of one pixel, and within a group of 16 lanes (4 pixels x 16 channels) lane c receives the 4 pixels of channel c
// (checked on the device: tools/ubench/tr_read_probe.hip).  Two reads = the 8 K values of a 32x32x16 operand.  dbeta comes
// out of the matrix cores too: a row of ones against the T operand.  (Before: the transpose by hand, per 16-byte piece two
// DPP moves, four byte permutes, eight selects and four ds_write_b32, and a gather of 32 lines per load instruction.)
#ifndef TFC_GDN_PG_TR
#define TFC_GDN_PG_TR 1
#endif
template <int C>
constexpr int pg_row_elems() { return C + 2 * (((16 - (C / 2) % 32) + 32) % 32); }      // row stride in dwords = 16 mod 32
template <typename T, int KT>
__global__ void __launch_bounds__(256, (pg_wgs<T, KT>())) gdn_param_grad_kernel(GdnParams p, float* partial) {
  constexpr int C = KT * 32;
  constexpr int NH = (KT + 1) / 2;
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ unsigned char smem[];
  // bf16: uT[C][PG_STRIDE], tT[C][PG_STRIDE] (u16); f32: us[PG_PIX][C], ts[PG_PIX][C] (float)
  unsigned short* uT = reinterpret_cast<unsigned short*>(smem);
  unsigned short* tT = uT + C * PG_STRIDE;
  constexpr bool TR = BF && TFC_GDN_PG_TR != 0 && KT <= 6;      // (224 / 256 channels: no registers for the ones-row products)
  constexpr int RS = pg_row_elems<C>();                 // TR: us[PG_PIX][RS], ts[PG_PIX][RS] (u16)
  unsigned short* const usr = reinterpret_cast<unsigned short*>(smem);
  unsigned short* const tsr = usr + PG_PIX * RS;
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
  f32x16 accb[NH];          // (TR) the ones-row products: every row of accb[b] is the column sums of T's tile wi + 2 b
#pragma unroll
  for (int b = 0; b < NH; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[b][r] = 0.f;

  // Software pipeline: the next stage's global loads are issued into registers before the MFMAs
  // of the current one and written to LDS after them.
  //   bf16: chunk = 8 channels (16 bytes) of one pixel.  A QUAD of lanes holds two neighbouring channel groups of two
  //         neighbouring pixels — lane r of quad qd = tid >> 2: pixel 2 (qd & 31) + (r >> 1), channel group
  //         2 ((qd >> 5) + 2 k) + (r & 1) in round k < KT — so that a lane PAIR reads 32 contiguous bytes and a load
  //         instruction touches 32 lines, and the pixel pair whose values share a 4-byte word of the transposed image
  //         is two lanes apart (DPP quad_perm).  (Round 6.  Before: consecutive lanes = consecutive pixels of one
  //         channel group, 64 lines per instruction; the CU's address unit takes ~4 cycles per line an instruction
  //         touches — measured in the convolution kernel, profiles/r06_notes.md: 83.8 -> 77.0 us for the pass on
  //         [262144, 192].  Two stages requested ahead instead of one: 78.2, as in round 5 — not the loads' latency.)
  //   f32:  chunk = 4 channels of one pixel, channel group fastest (coalesced, conflict-free).
  constexpr int NCH = BF ? KT : 2 * KT;
  u32x4 xq[NCH], tq[NCH];
  auto fetch = [&](long long st) {
    const long long p0 = st * PG_PIX;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = tid + 256 * k;
      const int px = TR ? c / (C / 8) : BF ? 2 * ((tid >> 2) & 31) + ((tid >> 1) & 1) : (c / (C / 4));
      const int off = TR ? 8 * (c % (C / 8)) : BF ? 8 * (2 * ((tid >> 7) + 2 * k) + (tid & 1)) : 4 * (c % (C / 4));
      xq[k] = u32x4{0, 0, 0, 0};
      tq[k] = u32x4{0, 0, 0, 0};
      if (p0 + px < p.pixels) {
        xq[k] = *reinterpret_cast<const u32x4*>(x + (p0 + px) * C + off);
        tq[k] = *reinterpret_cast<const u32x4*>(tsrc + (p0 + px) * C + off);
      }
    }
  };
  const bool plain = !p.rectify && !p.alpha2;
  const float relu_floor = p.rectify ? 0.f : -__builtin_inff();
  const float a2 = p.alpha2 ? 1.f : 0.f;
  auto stage_to_lds = [&]() {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = tid + 256 * k;
      if (TR) {
        const int px = c / (C / 8), cg = c % (C / 8);
        u32x4 uv = xq[k];
        if (plain) {
          uv &= 0x7FFF7FFFu;
        } else {
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            const float lo = fmaxf(bf16_bits_to_float(uv[w2] & 0xFFFFu), relu_floor);
            const float hi = fmaxf(__uint_as_float(uv[w2] & 0xFFFF0000u), relu_floor);
            uv[w2] = pack_bf16(gdn_u(lo, a2), gdn_u(hi, a2));
          }
        }
        *reinterpret_cast<u32x4*>(usr + px * RS + 8 * cg) = uv;
        *reinterpret_cast<u32x4*>(tsr + px * RS + 8 * cg) = tq[k];
      } else if (BF) {
        const int px = 2 * ((tid >> 2) & 31) + ((tid >> 1) & 1), cg = 2 * ((tid >> 7) + 2 * k) + (tid & 1);
        u32x4 uv = xq[k];
        if (plain) {
          uv &= 0x7FFF7FFFu;
        } else {
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            const float lo = fmaxf(bf16_bits_to_float(uv[w2] & 0xFFFFu), relu_floor);
            const float hi = fmaxf(__uint_as_float(uv[w2] & 0xFFFF0000u), relu_floor);
            uv[w2] = pack_bf16(gdn_u(lo, a2), gdn_u(hi, a2));
          }
        }
        // Pixel pairs: the even lane of a pair ends up with channels 0..3 of both pixels, the odd
        // lane with channels 4..7, each as (even pixel | odd pixel << 16) words: four 4-byte LDS
        // writes per tensor instead of eight 2-byte ones.
        const bool odd = px & 1;
        auto pair_store = [&](const u32x4& v, unsigned short* base) {
          const unsigned int send0 = odd ? v[0] : v[2], send1 = odd ? v[1] : v[3];
          const unsigned int keep0 = odd ? v[2] : v[0], keep1 = odd ? v[3] : v[1];
          unsigned int recv0 = __builtin_amdgcn_update_dpp(0u, send0, 0x4E, 0xF, 0xF, false);      // quad_perm [2, 3, 0, 1]
          unsigned int recv1 = __builtin_amdgcn_update_dpp(0u, send1, 0x4E, 0xF, 0xF, false);
          asm volatile("" : "+v"(recv0), "+v"(recv1));
          const unsigned int e0 = odd ? recv0 : keep0, o0 = odd ? keep0 : recv0;
          const unsigned int e1 = odd ? recv1 : keep1, o1 = odd ? keep1 : recv1;
          unsigned int* dst = reinterpret_cast<unsigned int*>(base + (8 * cg + (odd ? 4 : 0)) * PG_STRIDE + (px & ~1));
          dst[0 * PG_STRIDE / 2] = __builtin_amdgcn_perm(o0, e0, 0x05040100u);
          dst[1 * PG_STRIDE / 2] = __builtin_amdgcn_perm(o0, e0, 0x07060302u);
          dst[2 * PG_STRIDE / 2] = __builtin_amdgcn_perm(o1, e1, 0x05040100u);
          dst[3 * PG_STRIDE / 2] = __builtin_amdgcn_perm(o1, e1, 0x07060302u);
        };
        pair_store(uv, uT);
        pair_store(tq[k], tT);
      } else {
        const int px = c / (C / 4), cg = c % (C / 4);
        f32x4 xv = __builtin_bit_cast(f32x4, xq[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = plain ? fabsf(xv[e]) : gdn_u(fmaxf(xv[e], relu_floor), a2);
        *reinterpret_cast<f32x4*>(us + px * C + 4 * cg) = xv;
        *reinterpret_cast<u32x4*>(ts + px * C + 4 * cg) = tq[k];
      }
    }
  };

  const long long stages = (p.pixels + PG_PIX - 1) / PG_PIX;
  if (blockIdx.x < stages) fetch(blockIdx.x);
  for (long long st = blockIdx.x; st < stages; st += gridDim.x) {
    __syncthreads();          // the previous stage's fragment reads are done
    stage_to_lds();
    __syncthreads();
    if (st + gridDim.x < stages) fetch(st + gridDim.x);
    // dbeta: thread c sums column c of the staged T tile (TR: the matrix cores do, below)
    if (!TR && tid < C) {
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
    if (TR) {
      // this lane's chunk of tile 0, K step 0, first half: pixel 8 (g >> 1) + (j >> 2), channels 16 (g & 1) + 4 (j & 3) ...
      const int g = lane >> 4, j = lane & 15;
      const unsigned int lane_off = static_cast<unsigned int>(((8 * (g >> 1) + (j >> 2)) * RS + 16 * (g & 1) + 4 * (j & 3)) * 2);
      const unsigned int ubase = static_cast<unsigned int>(reinterpret_cast<size_t>(usr)) + lane_off;
      const unsigned int tbase = static_cast<unsigned int>(reinterpret_cast<size_t>(tsr)) + lane_off;
      // (the reads of K step ks + 1 are issued in front of the MFMAs of K step ks; the compiler does not count these reads:
      // one wait per K step, tied to the registers they fill)
      u32x2 ra[2][NH][2], rb[2][NH][2];
      // (plain unrolled code: a generic lambda does not capture variables that only inline-asm operands name)
#define TFC_PG_REQUEST(S, KS)                                                                                        \
      _Pragma("unroll") for (int a = 0; a < NH; ++a)                                                                 \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                           \
          const int jt = wj + 2 * a, it = wi + 2 * a;                                                                \
          const unsigned int rowoff = static_cast<unsigned int>(((16 * (KS) + 4 * hf) * RS) * 2);                    \
          ra[S][a][hf] = u32x2{0u, 0u};                                                                              \
          rb[S][a][hf] = u32x2{0u, 0u};                                                                              \
          if (jt < KT) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ra[S][a][hf]) : "v"(ubase + rowoff + 64u * jt)); \
          if (it < KT) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rb[S][a][hf]) : "v"(tbase + rowoff + 64u * it)); \
        }
      TFC_PG_REQUEST(0, 0)
#pragma unroll
      for (int ks = 0; ks < PG_PIX / 16; ++ks) {
        const int S = ks & 1;
#pragma unroll
        for (int a = 0; a < NH; ++a)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[S][a][0]), "+v"(ra[S][a][1]), "+v"(rb[S][a][0]), "+v"(rb[S][a][1]) : : "memory");
        if (ks + 1 < PG_PIX / 16) { TFC_PG_REQUEST(S ^ 1, ks + 1) }
        bf16x8 af[NH], bfr[NH];
#pragma unroll
        for (int a = 0; a < NH; ++a) {
          af[a] = __builtin_bit_cast(bf16x8, u32x4{ra[S][a][0].x, ra[S][a][0].y, ra[S][a][1].x, ra[S][a][1].y});
          bfr[a] = __builtin_bit_cast(bf16x8, u32x4{rb[S][a][0].x, rb[S][a][0].y, rb[S][a][1].x, rb[S][a][1].y});
        }
#pragma unroll
        for (int a = 0; a < NH; ++a)
#pragma unroll
          for (int b = 0; b < NH; ++b)
            if (wj + 2 * a < KT && wi + 2 * b < KT)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        // dbeta[i] = sum over the pixels of T[., i]: a row of ones as the A operand (the waves of the first j parity)
        if (wj == 0) {
          const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
#pragma unroll
          for (int b = 0; b < NH; ++b)
            if (wi + 2 * b < KT) accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[b], accb[b], 0, 0, 0);
        }
      }
#undef TFC_PG_REQUEST
    } else if (BF) {
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
  if (TR) {
    if (wj == 0 && h == 0) {
#pragma unroll
      for (int b = 0; b < NH; ++b)
        if (wi + 2 * b < KT) out[C * C + 32 * (wi + 2 * b) + i32] = accb[b][0];
    }
  } else if (tid < C) {
    out[C * C + tid] = bsum;
  }
}

template <typename T, int KT>
int launch_param_grad(GdnParams p, float* dgamma, float* dbeta, hipStream_t st) {
  constexpr int C = KT * 32;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long stages = ceil_div(p.pixels, static_cast<long long>(PG_PIX));
  const int blocks = static_cast<int>(std::max<long long>(1, std::min<long long>(stages, pg_wgs<T, KT>() * cus)));
  const size_t lds = sizeof(T) == 2 ? (TFC_GDN_PG_TR != 0 && KT <= 6 ? sizeof(unsigned short) * 2 * PG_PIX * pg_row_elems<C>()
                                                           : sizeof(unsigned short) * 2 * C * PG_STRIDE)
                                    : sizeof(float) * 2 * PG_PIX * C;
  DevBuf partial;
  TFC_HIP(partial.alloc(sizeof(float) * static_cast<size_t>(blocks) * (C * C + C), st));
  KernelTimer timer("gdn_backward_params", st);   // gradient kernel + the reduction of its partials
  TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_param_grad_kernel<T, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL((gdn_param_grad_kernel<T, KT>), dim3(blocks), dim3(256), lds, st, p, partial.as<float>());
  // dgamma / dbeta += the block partials, summed in a fixed order
  launch_sum_rows(partial.as<float>(), blocks, C * C + C, C * C, dgamma, st);
  launch_sum_rows(partial.as<float>() + C * C, blocks, C * C + C, C, dbeta, st);
  TFC_HIP(hipGetLastError());
  return 0;
}

template <int KT, bool PLAIN>
int launch_gdn_bwd_fused_variant(GdnParams p, hipStream_t st) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long want = ceil_div(p.tiles, TFC_GDN_BWD_THREADS / 64);
  const unsigned blocks = static_cast<unsigned>(std::max<long long>(1, std::min<long long>(want, cus)));
  constexpr int IMG = KT * KT * 2 * 64;
  const size_t lds = sizeof(bf16x8) * 2 * IMG + sizeof(float) * KT * 32;
  DevBuf image;
  TFC_HIP(image.alloc(lds, st));
  // Gamma^T image, then the Gamma image; beta lands behind the second one
  hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((IMG + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                     KT * 32, 0, image.as<bf16x8>());
  hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((IMG + 255) / 256), dim3(256), 0, st, p.gamma, p.beta,
                     KT * 32, 1, image.as<bf16x8>() + IMG);
  p.image = image.p;
  KernelTimer timer("gdn_backward_fused", st);
  TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_bwd_fused_bf16_kernel<KT, PLAIN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL((gdn_bwd_fused_bf16_kernel<KT, PLAIN>), dim3(blocks), dim3(TFC_GDN_BWD_THREADS), lds, st, p);
  TFC_HIP(hipGetLastError());
  return 0;
}

template <int KT>
int launch_gdn_bwd_fused(GdnParams p, hipStream_t st) {
  p.relu_floor = p.rectify ? 0.f : -__builtin_inff();
  p.a2 = p.alpha2 ? 1.f : 0.f;
  if (!p.rectify && !p.alpha2) return launch_gdn_bwd_fused_variant<KT, true>(p, st);
  return launch_gdn_bwd_fused_variant<KT, false>(p, st);
}

// bf16 with C <= 192: fused kernel (x, g -> T, dx) + parameter pass.  Otherwise three passes
// (all reading / writing each tensor once):
//   1. MODE_BWD_T:   x, g        -> T (= dL/dn), R (= g n^s)
//   2. MODE_BWD_DX:  T, R, x     -> dx
//   3. param grads:  x, T        -> dgamma, dbeta
template <int KT>
int run_gdn_backward(GdnParams p, const void* g, void* dx, int dtype, float* dbeta, float* dgamma,
                     hipStream_t st) {
  const size_t bytes = static_cast<size_t>(p.pixels) * p.C * (dtype == 1 ? 2 : 4);
  DevBuf tbuf, rbuf;
  TFC_HIP(tbuf.alloc(bytes, st));
  if constexpr (KT <= 6) {
    if (dtype == 1) {
      GdnParams f = p;
      f.g = g; f.y = dx; f.y2 = tbuf.p;
      if (int rc = launch_gdn_bwd_fused<KT>(f, st)) return rc;
      GdnParams c = p;
      c.g = tbuf.p;
      return launch_param_grad<unsigned short, KT>(c, dgamma, dbeta, st);
    }
  }
  if constexpr (KT > 6) {
    if (dtype != 1) return fail("tfc_gdn_backward: float32 path supports up to 192 channels");
  }
  TFC_HIP(rbuf.alloc(bytes, st));
  GdnParams a = p;
  a.g = g; a.y = tbuf.p; a.y2 = rbuf.p;
  constexpr int DT = KT <= 6 ? 1 : 2;   // two-pass path: f32 (C <= 192) or bf16 with C > 192
  if (int rc = launch_gdn<KT, MODE_BWD_T, DT>(a, dtype, st)) return rc;
  GdnParams b = p;
  b.x = tbuf.p; b.r = rbuf.p; b.xraw = p.x; b.y = dx;
  if (int rc = launch_gdn<KT, MODE_BWD_DX, DT>(b, dtype, st)) return rc;
  GdnParams c = p;
  c.g = tbuf.p;
  if constexpr (KT <= 6) return launch_param_grad<float, KT>(c, dgamma, dbeta, st);
  else return launch_param_grad<unsigned short, KT>(c, dgamma, dbeta, st);
}

}  // namespace tfc

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
