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
// tests pass with it) and needs 3-4x fewer vector instructions per symbol, but one step is ~1000 cycles
// of dependent LDS round trips per symbol and row, and 128 waves per launch cannot hide that: 20 ms for
// the bench step against 4.3 ms (profiles/r01_o_notes.md).  A latency-engineered version (row
// information prefetched per batch, winner state through DPP instead of ds_bpermute) is the next step.
//
// Escape symbols are not decoded here: a stream that meets one is flagged in `redo`, keeps its start
// state, and is decoded again by dec_fast_kernel (launched right behind with the flags as a mask).
#pragma once

namespace tfc {

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

    for (int n = 0; n < cnt; ++n) {
      const int tn = __builtin_amdgcn_ds_bpermute(rowbase4 + (n << 2), t);    // the row's symbol n
      const DecRow row = dir[tn];
      const int nsym = row.z & 0xFFFF;
      const int cdf0 = row.y;
      // ---- staged search: stride 256, 16, 1 ------------------------------------------------
      int stride = nsym > 256 ? 256 : (nsym > 16 ? 16 : 1);
      int lo = 0;                                   // symbols known to lie below the hit
      unsigned int A0 = 0;                          // B of entry `lo`
      unsigned int Aw, bw;
      int win;
      while (true) {
        const int k = min(lo + (i + 1) * stride, nsym);
        const unsigned int hi = static_cast<unsigned int>(tab[cdf0 + k]);
        const unsigned long long prod = (static_cast<unsigned long long>(span) + 1ull) * hi;
        const unsigned int B = static_cast<unsigned int>(prod >> 16);
        const unsigned int b = B - 1u;
        const unsigned int m16 = static_cast<unsigned int>(__ballot(D <= b) >> shift) & 0xFFFFu;
        win = m16 ? __builtin_ctz(m16) : 15;        // no candidate: damaged input only
        const unsigned int A = __builtin_amdgcn_update_dpp(A0, B, 0x111, 0xF, 0xF, false);   // row_shr:1
        Aw = static_cast<unsigned int>(__builtin_amdgcn_ds_bpermute(rowbase4 + (win << 2), static_cast<int>(A)));
        bw = static_cast<unsigned int>(__builtin_amdgcn_ds_bpermute(rowbase4 + (win << 2), static_cast<int>(b)));
        if (__ballot(stride > 1) == 0) break;       // every row is at its last stage
        // rows still searching narrow down; rows already at stride 1 repeat the same result
        if (stride > 1) {
          lo += win * stride;
          A0 = Aw;
          stride >>= 4;
        }
      }
      const int sym = min(lo + win, nsym - 1);
      // ---- successor state -------------------------------------------------------------------
      const unsigned int dig = static_cast<unsigned int>(
          __builtin_amdgcn_ds_bpermute(rowbase4 + ((pos & 15u) << 2), w.reg));
      const unsigned int Dn = D - Aw;
      const unsigned int t1 = bw - Aw;
      const bool ren = t1 < 65536u;
      D = ren ? ((Dn << 16) | dig) : Dn;
      span = ren ? ((t1 << 16) | 0xFFFFu) : t1;
      pos += ren ? 1u : 0u;
      escaped |= sym == row.w;
      if (i == n) outv = sym;
    }
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
