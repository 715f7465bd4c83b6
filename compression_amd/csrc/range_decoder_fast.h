// Fast multi-stream range DECODER for gfx950 — included by range_coder.hip.
//
// One wave per stream.  State is kept in "offset" form: D = window - base and
// s = span - 1 (the decoder never needs base and window separately,
// cc/lib/range_coder.h:224-271; they are rebuilt when the state is stored).
//
// Per symbol every lane k holds ONE candidate symbol of the row (its upper cdf
// bound, pre-scaled to 16-bit precision, read from the LDS-resident table one
// symbol ahead) and computes that candidate's complete successor state
//     B_k = ((s+1) * hi_k) >> 16, A_k = B_{k-1} (DPP wave_shr), span' = B_k - A_k - 1,
//     D'  = D - A_k, both renormalised with the next input digit if span' < 2^16,
// speculatively; the symbol is the first lane with D <= B_k - 1 (the reference's
// binary search finds the same position, range_coder.h:204-222), chosen by one
// ballot + s_ff1, and two v_readlane fetch the winning state.  Rows wider than
// 64 symbols go through 64 pivots first (coarse), then 64 entries of the chosen
// chunk (fine).
//
// A batch of 64 symbols whose rows are all narrow runs fully unrolled with no
// branch; escapes are detected afterwards with one vector compare, in which
// case the batch is replayed from its saved start state by the checked loop
// (which also serves wide rows and partial batches).
#pragma once

namespace tfc {

// Row directory entry of the decoder's LDS image (built on the host).
//   x: index of the first stage-1 upper bound (narrow: cdf0 + 1, wide: pivot array)
//   y: index of cdf[0]
//   z: nsym | chunk << 16   (chunk = symbols per pivot; 1 for narrow rows)
//   w: escape symbol index (nsym - 1) if the row has negative precision, else -1
struct DecRow { int x, y, z, w; };

struct FastDecState {   // wave-uniform
  unsigned int D;       // window - base
  unsigned int s;       // span - 1
  unsigned int pos;     // digits consumed since the window register was loaded
};

struct DecWindow {
  const uint8_t* src;
  long long len;
  unsigned int wbase;   // digit index held by lane 0 of reg
  int reg;              // digits wbase .. wbase + 63
  int next;             // digits wbase + 64 .. wbase + 127, fetched one batch ahead
};

__device__ inline int fast_window_fetch(const DecWindow& w, unsigned int first, int lane) {
  const long long b = 2ll * (static_cast<long long>(first) + lane);
  const unsigned int hi = b < w.len ? w.src[b] : 0u;
  const unsigned int lo = b + 1 < w.len ? w.src[b + 1] : 0u;
  return static_cast<int>((hi << 8) | lo);
}

// Synchronous (re)load of both registers at wbase.
__device__ inline void fast_window_load(DecWindow& w, int lane) {
  w.reg = fast_window_fetch(w, w.wbase, lane);
  w.next = fast_window_fetch(w, w.wbase + 64u, lane);
}

// Advances the window by `shift` (0..64) consumed digits without waiting for HBM: the new
// register is stitched from reg/next with two ds_bpermute, and the following 64 digits are
// requested now, to be used a whole batch later.
__device__ inline void fast_window_advance(DecWindow& w, unsigned int shift, int lane) {
  const int idx = (lane + static_cast<int>(shift)) & 63;
  const int from_reg = __builtin_amdgcn_ds_bpermute(idx << 2, w.reg);
  const int from_next = __builtin_amdgcn_ds_bpermute(idx << 2, w.next);
  const bool in_reg = lane + static_cast<int>(shift) < 64;
  // shift == 64 moves `next` into place unchanged (idx == lane)
  const int stitched = in_reg ? from_reg : from_next;
  w.wbase += shift;
  w.reg = stitched;
  // digits wbase+64.. : lanes that still come from the old `next` plus newly fetched ones
  const int fresh = fast_window_fetch(w, w.wbase + 64u, lane);
  w.next = fresh;
}

// One candidate-per-lane selection step.  hi = upper bound held by this lane,
// a0 = lower offset of lane 0's candidate (0 for a whole narrow row).
// Returns the winning lane; updates the state.
// (span * hi) >> 16 with span = s + 1, as ONE v_mad_u64_u32 (s * hi + hi) + v_alignbit.  The
// addend is made opaque so that LLVM does not refactor it into (s + 1) * hi, which needs a
// 33-bit multiplicand and two multiplies.
__device__ inline unsigned int scale_bound(unsigned int s, unsigned int hi) {
  unsigned int add = hi;
  asm volatile("" : "+v"(add));
  const unsigned long long P = static_cast<unsigned long long>(s) * hi + add;
  return static_cast<unsigned int>(P >> 16);
}

__device__ inline int select_step(FastDecState& st, unsigned int hi, unsigned int a0,
                                  unsigned int dig) {
  const unsigned int Bq = scale_bound(st.s, hi);
  const unsigned long long PB = static_cast<unsigned long long>(Bq) << 16;
  const unsigned int B = static_cast<unsigned int>(PB >> 16);
  unsigned int A = static_cast<unsigned int>(
      __builtin_amdgcn_update_dpp(static_cast<int>(a0), static_cast<int>(B), 0x138, 0xF, 0xF, false));
  // Keep the wave_shr move a separate v_mov_b32_dpp: when a0 is the constant 0, LLVM's DPP
  // combine folds it into the consumers as `v_subrev_u32_dpp ... wave_shr:1 bound_ctrl:1`,
  // which returned wrong lanes on gfx950 (ROCm 7.2) — caught by the K3/K4/K6 vectors.
  asm volatile("" : "+v"(A));
  const unsigned int b = B - 1u;
  const unsigned int t1 = b - A;
  const unsigned int Dn = st.D - A;
  const bool ren = t1 < 65536u;
  const unsigned int s2 = ren ? ((t1 << 16) | 0xFFFFu) : t1;
  const unsigned int D2 = ren ? ((Dn << 16) | dig) : Dn;
  const unsigned long long renmask = __ballot(ren);
  const unsigned long long hit = __ballot(st.D <= b) | (1ull << 63);   // damaged input: take lane 63
  const int L = __builtin_ctzll(hit);
  st.s = __builtin_amdgcn_readlane(static_cast<int>(s2), L);
  st.D = __builtin_amdgcn_readlane(static_cast<int>(D2), L);
  st.pos += static_cast<unsigned int>((renmask >> L) & 1ull);
  return L;
}

// Coarse step over pivots: finds the chunk, no state update.  Returns chunk
// index and the chunk's lower offset (B of the previous pivot, 0 for chunk 0).
__device__ inline int pivot_step(const FastDecState& st, unsigned int pivot, unsigned int* a0) {
  const unsigned int B = scale_bound(st.s, pivot);
  const unsigned long long hit = __ballot(st.D <= B - 1u) | (1ull << 63);
  const int L = __builtin_ctzll(hit);
  // lower offset of chunk L = B of the previous pivot (0 for chunk 0): shift B down one lane
  // first, then one readlane — no branch on L == 0.
  unsigned int prev = static_cast<unsigned int>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(B), 0x138, 0xF, 0xF, false));
  asm volatile("" : "+v"(prev));
  *a0 = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(prev), L));
  return L;
}

// Binary digit with the uniform cdf {0,1,2} at precision 1
// (range_coder_kernels.cc:449-471 via DecodeLinearly), on the offset state.
__device__ inline int fast_bit(FastDecState& st, const DecWindow& w) {
  const unsigned long long span = static_cast<unsigned long long>(st.s) + 1;
  const unsigned long long target = (static_cast<unsigned long long>(st.D) + 1) << 1;
  const unsigned int bit = target <= span ? 0u : 1u;
  const unsigned int A = static_cast<unsigned int>((span * bit) >> 1);
  const unsigned int b = static_cast<unsigned int>(((span * (bit + 1)) >> 1) - 1);
  st.D -= A;
  st.s = b - A;
  if ((st.s >> 16) == 0) {
    const unsigned int dig = __builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u));
    st.s = (st.s << 16) | 0xFFFFu;
    st.D = (st.D << 16) | dig;
    ++st.pos;
  }
  return static_cast<int>(bit);
}

template <typename Dst>
__global__ void dec_fast_kernel(DecParams p, Dst dst) {
  extern __shared__ int32_t lds[];
  const int waves = blockDim.x >> 6;
  int32_t* tab = lds;                                        // p.tab.dec_words ints
  DecRow* dir = reinterpret_cast<DecRow*>(lds + ((p.tab.dec_words + 3) & ~3));
  for (int i = threadIdx.x; i < p.tab.dec_words; i += blockDim.x) tab[i] = p.tab.dec_image[i];
  for (int i = threadIdx.x; i < p.tab.ntab; i += blockDim.x) dir[i] = p.tab.dec_dir[i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t s = static_cast<int64_t>(blockIdx.x) * waves + wid;
  if (s >= p.streams) return;

  const uint4 st0 = p.state[s];
  FastDecState st;
  st.s = __builtin_amdgcn_readfirstlane(st0.y);
  st.D = __builtin_amdgcn_readfirstlane(st0.z) - __builtin_amdgcn_readfirstlane(st0.x);
  st.pos = 0;
  DecWindow w;
  const long long o0 = p.off[s];
  w.src = p.blob + o0;
  w.len = p.off[s + 1] - o0;
  w.wbase = __builtin_amdgcn_readfirstlane(st0.w);
  const int ntab = p.tab.ntab;
  int ch0 = 0;

  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
    // ---- vector phase: row of every symbol of the batch ---------------------
    const int64_t j = j0 + lane;
    const bool valid = j < p.elems;
    int t = 0;
    if (valid) {
      if (p.index) {
        t = p.index[s * p.elems + j];
        if (t < 0 || t >= ntab) {
          atomicMin(p.first_error, static_cast<unsigned long long>(s * p.elems + j));
          t = 0;
        }
      } else {
        t = static_cast<int>((static_cast<unsigned int>(ch0) + static_cast<unsigned int>(lane)) %
                             static_cast<unsigned int>(ntab));
      }
    }
    ch0 = static_cast<int>((static_cast<unsigned int>(ch0) + 64u) % static_cast<unsigned int>(ntab));
    const DecRow row = dir[t];
    const int cnt = static_cast<int>(min<int64_t>(64, p.elems - j0));
    const bool anywide = __ballot(valid && (row.z >> 16) > 1) != 0;

    if (j0 == 0) fast_window_load(w, lane); else fast_window_advance(w, st.pos, lane);
    st.pos = 0;
    int outv = 0;
    // stage-1 bounds (the row itself, or its pivots) are fetched one symbol ahead
    unsigned int hi_cur = static_cast<unsigned int>(tab[__builtin_amdgcn_readlane(row.x, 0) + lane]);

    // ---- checked loop: wide rows, escapes, partial batches -------------------
    auto checked = [&](int n0, int n1) {
      for (int n = n0; n < n1; ++n) {
        const unsigned int hi_next = static_cast<unsigned int>(
            tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
        const int z = __builtin_amdgcn_readlane(row.z, n);
        const int escsym = __builtin_amdgcn_readlane(row.w, n);
        const unsigned int dig =
            static_cast<unsigned int>(__builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u)));
        int sym;
        const int chunk = z >> 16;
        if (chunk <= 1) {
          sym = select_step(st, hi_cur, 0u, dig);
        } else {
          unsigned int a0;
          const int c = pivot_step(st, hi_cur, &a0);
          const int cdf0 = __builtin_amdgcn_readlane(row.y, n);
          const unsigned int hi2 = static_cast<unsigned int>(tab[cdf0 + c * chunk + 1 + lane]);
          sym = c * chunk + select_step(st, hi2, a0, dig);
        }
        if (sym == escsym) {
          // Elias-gamma escape (range_coder_kernels.cc:449-471); the unary prefix
          // is bounded so that damaged input cannot spin.
          int nb = 0;
          while (nb < 31 && fast_bit(st, w) == 0) ++nb;
          int v = 1 << nb;
          while (--nb >= 0) v |= fast_bit(st, w) << nb;
          const int neg = fast_bit(st, w);
          sym = neg ? -v : v + escsym - 1;
          if (st.pos >= 40u) {          // keep digits ahead in the window register
            fast_window_advance(w, st.pos, lane);
            st.pos = 0;
          }
        }
        outv = tfc_writelane(sym, n, outv);
        hi_cur = hi_next;
      }
    };

    if (cnt == 64) {
      // Level 1: the whole batch speculatively, fully unrolled (immediate lane indices, no
      // branch).  Level 2, only if an escape symbol turned up: blocks of 8 symbols with a
      // check after each block; a block containing an escape is replayed from its saved
      // start state by the checked loop (level 3), which decodes the Elias-gamma bits.
      const FastDecState saved64 = st;
      const unsigned int hi_saved64 = hi_cur;
      if (!anywide) {
#pragma unroll
        for (int n = 0; n < 64; ++n) {
          const unsigned int hi_next = static_cast<unsigned int>(
              tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
          const unsigned int dig = static_cast<unsigned int>(
              __builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u)));
          const int L = select_step(st, hi_cur, 0u, dig);
          outv = tfc_writelane(L, n, outv);
          hi_cur = hi_next;
        }
      } else {
#pragma unroll
        for (int n = 0; n < 64; ++n) {
          const unsigned int hi_next = static_cast<unsigned int>(
              tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
          const unsigned int dig = static_cast<unsigned int>(
              __builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u)));
          unsigned int a0;
          const int c = pivot_step(st, hi_cur, &a0);
          const int chunk = __builtin_amdgcn_readlane(row.z, n) >> 16;
          const int first = __builtin_amdgcn_readlane(row.y, n) + c * chunk;
          const unsigned int hi2 = static_cast<unsigned int>(tab[first + 1 + lane]);
          const int L = select_step(st, hi2, a0, dig);
          outv = tfc_writelane(c * chunk + L, n, outv);
          hi_cur = hi_next;
        }
      }
      if (__ballot(outv == row.w) != 0) {
        st = saved64;
        hi_cur = hi_saved64;
        for (int blk = 0; blk < 8; ++blk) {
          const int n0 = blk * 8;
          const FastDecState saved = st;
          const unsigned int hi_saved = hi_cur;
          if (!anywide) {
  #pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int n = n0 + i;
              const unsigned int hi_next = static_cast<unsigned int>(
                  tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
              const unsigned int dig = static_cast<unsigned int>(
                  __builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u)));
              const int L = select_step(st, hi_cur, 0u, dig);
              outv = tfc_writelane(L, n, outv);
              hi_cur = hi_next;
            }
          } else {
            // every symbol takes the two-stage route (pivots -> chunk -> entries); for a narrow
            // row the "pivots" are the row itself with chunk 1, so one code path serves both.
  #pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int n = n0 + i;
              const unsigned int hi_next = static_cast<unsigned int>(
                  tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
              const unsigned int dig = static_cast<unsigned int>(
                  __builtin_amdgcn_readlane(w.reg, static_cast<int>(st.pos & 63u)));
              unsigned int a0;
              const int c = pivot_step(st, hi_cur, &a0);
              const int chunk = __builtin_amdgcn_readlane(row.z, n) >> 16;
              const int first = __builtin_amdgcn_readlane(row.y, n) + c * chunk;
              const unsigned int hi2 = static_cast<unsigned int>(tab[first + 1 + lane]);
              const int L = select_step(st, hi2, a0, dig);
              outv = tfc_writelane(c * chunk + L, n, outv);
              hi_cur = hi_next;
            }
          }
          const unsigned long long hits = __ballot(outv == row.w) & (0xFFull << n0);
          if (hits != 0) {
            st = saved;
            hi_cur = hi_saved;
            checked(n0, n0 + 8);
          }
        }
      }
    } else {
      checked(0, cnt);
    }
    if (valid) dst.store(s * p.elems + j, t, outv);
  }

  w.wbase += st.pos;
  if (lane == 0) {
    // back to the (base, span-1, window, digits pulled) form the other kernels use
    const long long b = 2ll * w.wbase;
    unsigned int window = 0;
    for (int i = -4; i < 0; ++i) {
      const long long q = b + i;
      window = (window << 8) | ((q >= 0 && q < w.len) ? w.src[q] : 0u);
    }
    p.state[s] = make_uint4(window - st.D, st.s, window, w.wbase);
  }
}

}  // namespace tfc
