// Multi-stream range DECODER, four streams per wave — included by range_coder.hip.
//
// dec_fast_kernel (range_decoder_fast.h) spends a whole wave on one symbol: 64 lanes hold 64 candidate
// symbols, 27.9 vector instructions per symbol, hand-scheduled for the latency of a lone wave.  With many
// steps in flight the coder is bound by VALU issue, and what counts is the work per issued
// instruction.  Here a wave carries FOUR streams, one per 16-lane row: the 16 lanes of a row hold 16
// candidates, a row of more than 16 symbols is searched in stages of stride 256 / 16 / 1 over the SAME
// table (no pivot arrays), the winner of a row is found from the row's 16-bit slice of one ballot
// (v_ffbl, computed redundantly by every lane of the row — no scalar code), and the winner's successor
// state travels to the row's lanes with ds_bpermute.  Nothing here is scheduled by hand: the latency of a
// step is several LDS round trips, covered by the other waves of the SIMD (8 waves share one LDS table
// copy), which is the right trade only when there are enough waves — hence throughput mode only.
//
// Arithmetic: identical to dec_fast_kernel / the reference (cc/lib/range_coder.h:224-271):
//   B_k = ((s + 1) * hi_k) >> 16, b_k = B_k - 1, A_k = B_{k-1} (0 before the row's first symbol),
//   symbol = first k with D <= b_k; D' = D - A_k, s' = b_k - A_k; if s' < 2^16 one 16-bit digit is
//   shifted in.  hi_k is the row's cdf entry pre-scaled to 16-bit precision (the decoder image).
//
// STATUS: experimental, off by default (TFC_DEC_QUAD=1 with throughput mode).  It is bit-exact (the GPU
// tests pass with it) and needs 3-4x fewer vector instructions per symbol, but one step is still ~800
// cycles per symbol and row (multiply -> ballot -> two ds_bpermute per search stage, 1.7 stages per symbol
// on the bench tables), and 128 waves per launch cannot hide that: 16.7 ms for the bench step against
// 4.3 ms (profiles/r01_o_notes.md).  It pays below ~250 cycles per symbol and row.
//
// Escape symbols are not decoded here: a stream that meets one is flagged in `redo`, keeps its start
// state, and is decoded again by dec_fast_kernel (launched right behind with the flags as a mask).
#pragma once
#include <type_traits>

namespace tfc {

template <int N> using QuadIdx = std::integral_constant<int, N>;

struct QuadDigits {          // per lane: one digit of the row's window
  const uint8_t* src;        // row-uniform
  long long len;             // row-uniform
  unsigned int wbase;        // digit index held by lane 0 of the row (row-uniform)
  int reg;                   // digit wbase + i
  unsigned int next_hi, next_lo;   // raw bytes of digit wbase + 16 + i, requested one batch ahead
  bool next_hi_ok, next_lo_ok;
};

__device__ inline void quad_request(QuadDigits& w, unsigned int first, int i) {
  const long long b = 2ll * (static_cast<long long>(first) + i);
  const long long last = w.len - 1;                       // w.src is readable on [0, max(len, 1))
  w.next_hi = w.src[b < last ? b : (last < 0 ? 0 : last)];
  w.next_lo = w.src[b + 1 < last ? b + 1 : (last < 0 ? 0 : last)];
  w.next_hi_ok = b < w.len;
  w.next_lo_ok = b + 1 < w.len;
}
__device__ inline int quad_next(const QuadDigits& w) {
  return static_cast<int>(((w.next_hi_ok ? w.next_hi : 0u) << 8) | (w.next_lo_ok ? w.next_lo : 0u));
}

template <typename Dst>
__global__ void dec_quad_kernel(DecParams p, Dst dst, unsigned int* redo) {
  extern __shared__ int32_t lds[];
  const int waves = blockDim.x >> 6;
  // the table part of the decoder image (entries scaled to 16 bits, then 64 entries of 65536) and the
  // row directory; the 64-pivot arrays behind it are not needed
  const int words = p.tab.total + 64;
  int32_t* tab = lds;
  DecRow* dir = reinterpret_cast<DecRow*>(lds + ((words + 3) & ~3));
  for (int k = threadIdx.x; k < words; k += blockDim.x) tab[k] = p.tab.dec_image[k];
  for (int k = threadIdx.x; k < p.tab.ntab; k += blockDim.x) dir[k] = p.tab.dec_dir[k];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int seg = lane >> 4, i = lane & 15;
  const int rowbase4 = (lane & 48) << 2;                  // byte index of the row's lane 0 for ds_bpermute
  const int wid = threadIdx.x >> 6;
  const int64_t s0 = (static_cast<int64_t>(blockIdx.x) * waves + wid) * 4;
  if (s0 >= p.streams) return;
  const int64_t s = s0 + seg;
  const bool live = s < p.streams;
  const int64_t sc = live ? s : p.streams - 1;            // idle rows shadow the last stream, store nothing

  const uint4 st0 = p.state[sc];
  unsigned int span = st0.y;                              // span - 1
  unsigned int D = st0.z - st0.x;                         // window - base
  unsigned int pos = 0;                                   // digits consumed since wbase
  QuadDigits w;
  const long long o0 = p.off[sc];
  w.len = p.off[sc + 1] - o0;
  w.src = w.len > 0 ? p.blob + o0 : reinterpret_cast<const uint8_t*>(p.off);
  w.wbase = st0.w;
  quad_request(w, w.wbase, i);
  w.reg = quad_next(w);
  quad_request(w, w.wbase + 16u, i);

  const int ntab = p.tab.ntab;
  const int shift = seg * 16;
  unsigned int ch0 = 0;
  bool escaped = false;

  for (int64_t j0 = 0; j0 < p.elems; j0 += 16) {
    // ---- per batch: the window moves on by the digits the last batch consumed (<= 16) ----------
    if (j0 != 0) {
      const int idx = i + static_cast<int>(pos);
      const int from_reg = __builtin_amdgcn_ds_bpermute(rowbase4 + ((idx & 15) << 2), w.reg);
      const int from_next = __builtin_amdgcn_ds_bpermute(rowbase4 + ((idx & 15) << 2), quad_next(w));
      w.wbase += pos;
      w.reg = idx < 16 ? from_reg : from_next;
      pos = 0;
      quad_request(w, w.wbase + 16u, i);
    }
    // ---- table of every symbol of the batch (lane i: symbol j0 + i) ---------------------------
    const int64_t j = j0 + i;
    const bool valid = j < p.elems;
    int t = 0;
    if (valid) {
      if (p.index) {
        t = p.index[sc * p.elems + j];
        if (t < 0 || t >= ntab) {
          if (live) atomicMin(p.first_error, static_cast<unsigned long long>(s * p.elems + j));
          t = 0;
        }
      } else {
        const unsigned int c = ch0 + static_cast<unsigned int>(i);
        t = static_cast<int>(c >= static_cast<unsigned int>(ntab) ? c % static_cast<unsigned int>(ntab) : c);
      }
    }
    ch0 = (ch0 + 16u) % static_cast<unsigned int>(ntab);
    const int cnt = static_cast<int>(min<int64_t>(16, p.elems - j0));
    int outv = 0;

    // ---- off the per-symbol chain: the rows of the batch and their first-stage candidates --------
    // Lane i reads the directory entry of ITS symbol; entry n reaches the row's lanes through
    // row_newbcast:n (a DPP move, no LDS round trip).  The first-stage candidates of all 16 symbols
    // depend on the tables only, not on the coder state, so their 16 LDS reads are issued here, back to
    // back; the per-symbol chain below then consists of the multiply, one ballot and two ds_bpermute,
    // plus one dependent table read per further stage of a wide row.
    const DecRow mine = dir[t];
    int cdf0s[16], nsyms[16];
    unsigned int hi1[16];
    auto prefetch = [&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value;
      cdf0s[n] = __builtin_amdgcn_update_dpp(0, mine.y, 0x150 + n, 0xF, 0xF, false);
      nsyms[n] = __builtin_amdgcn_update_dpp(0, mine.z, 0x150 + n, 0xF, 0xF, false) & 0xFFFF;
      const int stride = nsyms[n] > 256 ? 256 : (nsyms[n] > 16 ? 16 : 1);
      hi1[n] = static_cast<unsigned int>(tab[cdf0s[n] + min((i + 1) * stride, nsyms[n])]);
    };
    const int escs = mine.w;

    // one candidate evaluation: the row's first lane whose upper bound covers D, its A and b
    auto evaluate = [&](unsigned int hi, unsigned int A0, int* win, unsigned int* Aw, unsigned int* bw)
                        __attribute__((always_inline)) {
      const unsigned long long prod = static_cast<unsigned long long>(span) * hi + hi;   // (span + 1) * hi
      const unsigned int B = static_cast<unsigned int>(prod >> 16);
      const unsigned int b = B - 1u;
      const unsigned int m16 = static_cast<unsigned int>(__ballot(D <= b) >> shift) & 0xFFFFu;
      *win = m16 ? __builtin_ctz(m16) : 15;           // no candidate: damaged input only
      const unsigned int A = __builtin_amdgcn_update_dpp(A0, B, 0x111, 0xF, 0xF, false);   // row_shr:1
      *Aw = static_cast<unsigned int>(__builtin_amdgcn_ds_bpermute(rowbase4 + (*win << 2), static_cast<int>(A)));
      *bw = static_cast<unsigned int>(__builtin_amdgcn_ds_bpermute(rowbase4 + (*win << 2), static_cast<int>(b)));
    };

    auto step = [&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value;
      const int nsym = nsyms[n], cdf0 = cdf0s[n];
      const unsigned int dig = static_cast<unsigned int>(
          __builtin_amdgcn_ds_bpermute(rowbase4 + ((pos & 15u) << 2), w.reg));
      int stride = nsym > 256 ? 256 : (nsym > 16 ? 16 : 1);
      int win;
      unsigned int Aw, bw;
      evaluate(hi1[n], 0u, &win, &Aw, &bw);
      int lo = 0;
      // further stages (stride 16, then 1) for the rows that need them; the others keep their result
      while (__ballot(stride > 1) != 0) {
        const bool more = stride > 1;
        const int lo2 = lo + win * stride;
        const int stride2 = stride >> 4;
        const int k = min(lo2 + (i + 1) * stride2, nsym);
        const unsigned int hi = static_cast<unsigned int>(tab[cdf0 + (more ? k : 0)]);
        int win2;
        unsigned int Aw2, bw2;
        evaluate(hi, Aw, &win2, &Aw2, &bw2);
        if (more) {
          lo = lo2;
          stride = stride2;
          win = win2;
          Aw = Aw2;
          bw = bw2;
        }
      }
      const int sym = min(lo + win, nsym - 1);
      // ---- successor state ---------------------------------------------------------------------
      const unsigned int Dn = D - Aw;
      const unsigned int t1 = bw - Aw;
      const bool ren = t1 < 65536u;
      D = ren ? ((Dn << 16) | dig) : Dn;
      span = ren ? ((t1 << 16) | 0xFFFFu) : t1;
      pos += ren ? 1u : 0u;
      const int esc_n = __builtin_amdgcn_update_dpp(0, escs, 0x150 + n, 0xF, 0xF, false);
      escaped |= sym == esc_n;
      if (i == n) outv = sym;
    };
    auto guarded = [&](auto nc) __attribute__((always_inline)) {
      if (decltype(nc)::value < cnt) step(nc);
    };
#define TFC_QUAD_EACH(F)                                                                     \
    F(QuadIdx<0>{}); F(QuadIdx<1>{}); F(QuadIdx<2>{}); F(QuadIdx<3>{}); F(QuadIdx<4>{});         \
    F(QuadIdx<5>{}); F(QuadIdx<6>{}); F(QuadIdx<7>{}); F(QuadIdx<8>{}); F(QuadIdx<9>{});         \
    F(QuadIdx<10>{}); F(QuadIdx<11>{}); F(QuadIdx<12>{}); F(QuadIdx<13>{}); F(QuadIdx<14>{});    \
    F(QuadIdx<15>{});
    TFC_QUAD_EACH(prefetch)
    if (cnt == 16) {
      TFC_QUAD_EACH(step)
    } else {
      TFC_QUAD_EACH(guarded)
    }
#undef TFC_QUAD_EACH
    if (valid && live && !escaped) dst.store(s * p.elems + j, t, outv);
  }

  if (escaped) {
    // decoded again from the untouched start state by dec_fast_kernel
    if (live && i == 0) redo[s] = 1u;
    return;
  }
  w.wbase += pos;
  if (live && i == 0) {
    // back to the (base, span-1, window, digits pulled) form the other kernels use
    const long long b = 2ll * w.wbase;
    unsigned int window = 0;
    for (int q4 = -4; q4 < 0; ++q4) {
      const long long q = b + q4;
      window = (window << 8) | ((q >= 0 && q < w.len) ? w.src[q] : 0u);
    }
    p.state[s] = make_uint4(window - D, span, window, w.wbase);
  }
}

}  // namespace tfc
