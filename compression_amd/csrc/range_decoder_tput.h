// Multi-stream range DECODER for throughput mode — included by range_coder.hip.
//
// dec_fast_kernel (range_decoder_fast.h) makes every one of the 64 candidate lanes compute its complete
// successor state BEFORE the winner is known and picks the winner through EXEC: no scalar instruction
// ever touches a value that came out of a vector instruction, which is what a lone wave needs (158
// cycles per symbol) — and costs 27.9 vector instructions per symbol.  With many steps in flight the
// coder is bound by VALU issue, not by the latency of one wave, so this kernel does the
// opposite: the lanes compute only their upper bound, the winner is found the textbook way (ballot,
// s_ff1, v_readlane — each hop pays the VALU->SALU hazard the other kernel avoids) and the successor
// state is computed ONCE, on the scalar unit, which issues beside the vector instructions of the other
// waves.  ~10 vector instructions per symbol.  One stream per wave as before, but 16 waves share one LDS
// copy of the decoder image, so that a CU holds 32 waves (8 per SIMD) to cover the longer chain.
//
// STATUS: experimental, off by default (TFC_DEC_TPUT=1 with throughput mode).  Bit-exact (the GPU tests
// pass with it), 214 vector but 378 scalar instructions in the kernel: the scalar unit issues at the same
// one-instruction-per-4-cycles-per-SIMD rate as the vector unit, so moving the successor state there only
// moves the bound, and the compiler's branchy code is ~730 cycles per symbol: 14.9 ms for the bench step
// alone against 4.3 ms, 21-23 ms per launch with 8-12 steps in flight (profiles/r01_o_notes.md).  The
// lesson for the next version: count ALL issue slots (VALU + SALU), keep the step branch-free, and write
// it by hand like dec_fast_kernel.
//
// Same arithmetic as dec_fast_kernel / cc/lib/range_coder.h:224-271; same image (64 pivots per row of
// more than 64 symbols), same window handling, same state format; escape codes are decoded in line.
#pragma once

namespace tfc {

struct TputState {          // wave-uniform (SGPRs)
  unsigned int D;           // window - base
  unsigned int span;        // span - 1
  unsigned int pos;         // digits consumed since the window register was loaded
};

// One search stage: the first lane whose upper bound covers D; returns the lane, writes B of the
// previous lane (A0 for lane 0) and the lane's own b = B - 1.
__device__ inline int tput_stage(const TputState& st, unsigned int hi, unsigned int A0, unsigned int* Aw,
                                 unsigned int* bw) {
  const unsigned long long prod = static_cast<unsigned long long>(st.span) * hi + hi;   // (span + 1) * hi
  const unsigned int B = static_cast<unsigned int>(prod >> 16);
  const unsigned int b = B - 1u;
  unsigned int A;
  // written by hand: LLVM's DPP combine has folded this pattern wrongly on gfx950
  asm("v_mov_b32 %0, %2\n\t"
      "s_nop 1\n\t"
      "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf"
      : "=&v"(A) : "v"(B), "s"(A0));
  const unsigned long long mask = __ballot(st.D <= b);
  const int win = mask ? __builtin_ctzll(mask) : 63;        // no candidate: damaged input only
  *Aw = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(A), win));
  *bw = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(b), win));
  return win;
}

__device__ inline void tput_advance(TputState& st, unsigned int Aw, unsigned int bw, int window_reg) {
  const unsigned int Dn = st.D - Aw;
  const unsigned int t1 = bw - Aw;
  if (t1 < 65536u) {
    const unsigned int dig = static_cast<unsigned int>(__builtin_amdgcn_readlane(window_reg, st.pos & 63u));
    st.D = (Dn << 16) | dig;
    st.span = (t1 << 16) | 0xFFFFu;
    ++st.pos;
  } else {
    st.D = Dn;
    st.span = t1;
  }
}

// Binary digit with the uniform cdf {0,1,2} at precision 1 (range_coder_kernels.cc:449-471).
__device__ inline int tput_bit(TputState& st, int window_reg) {
  const unsigned long long span = static_cast<unsigned long long>(st.span) + 1;
  const unsigned long long target = (static_cast<unsigned long long>(st.D) + 1) << 1;
  const unsigned int bit = target <= span ? 0u : 1u;
  const unsigned int A = static_cast<unsigned int>((span * bit) >> 1);
  const unsigned int b = static_cast<unsigned int>(((span * (bit + 1)) >> 1) - 1);
  tput_advance(st, A, b, window_reg);
  return static_cast<int>(bit);
}

template <typename Dst>
__global__ void __launch_bounds__(1024, 2) dec_tput_kernel(DecParams p, Dst dst) {
  extern __shared__ int32_t lds[];
  const int waves = blockDim.x >> 6;
  int32_t* tab = lds;                                        // p.tab.dec_words ints
  DecRow* dir = reinterpret_cast<DecRow*>(lds + ((p.tab.dec_words + 3) & ~3));
  for (int k = threadIdx.x; k < p.tab.dec_words; k += blockDim.x) tab[k] = p.tab.dec_image[k];
  for (int k = threadIdx.x; k < p.tab.ntab; k += blockDim.x) dir[k] = p.tab.dec_dir[k];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t s = static_cast<int64_t>(blockIdx.x) * waves + wid;
  if (s >= p.streams) return;

  const uint4 st0 = p.state[s];
  TputState st;
  st.span = __builtin_amdgcn_readfirstlane(st0.y);
  st.D = __builtin_amdgcn_readfirstlane(st0.z) - __builtin_amdgcn_readfirstlane(st0.x);
  st.pos = 0;
  DecWindow w;
  const long long o0 = p.off[s];
  w.len = p.off[s + 1] - o0;
  w.src = w.len > 0 ? p.blob + o0 : reinterpret_cast<const uint8_t*>(p.off);
  w.wbase = __builtin_amdgcn_readfirstlane(st0.w);
  const int ntab = p.tab.ntab;
  int ch0 = 0;

  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
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
    if (j0 == 0) fast_window_load(w, lane); else fast_window_advance(w, st.pos, lane);
    st.pos = 0;
    int outv = 0;
    const int chunkv = row.z >> 16;        // symbols per pivot (1: narrow row)
    const int first1v = row.y + 1;         // table index of the row's first upper bound

    // stage-1 candidates (the row itself, or its 64 pivots) are read one symbol ahead: they do not
    // depend on the coder state
    unsigned int hi_cur = static_cast<unsigned int>(tab[__builtin_amdgcn_readlane(row.x, 0) + lane]);
    for (int n = 0; n < cnt; ++n) {
      const unsigned int hi_next =
          static_cast<unsigned int>(tab[__builtin_amdgcn_readlane(row.x, (n + 1) & 63) + lane]);
      const int chunk = __builtin_amdgcn_readlane(chunkv, n);
      const int escsym = __builtin_amdgcn_readlane(row.w, n);
      unsigned int Aw, bw;
      int sym = tput_stage(st, hi_cur, 0u, &Aw, &bw);
      if (chunk > 1) {
        // wide row: `sym` is the pivot, i.e. the chunk; its entries are the second stage
        const int first1 = __builtin_amdgcn_readlane(first1v, n);
        const int cstart = sym * chunk;
        const unsigned int hi2 = static_cast<unsigned int>(tab[first1 + cstart + lane]);
        sym = cstart + tput_stage(st, hi2, Aw, &Aw, &bw);
      }
      tput_advance(st, Aw, bw, w.reg);
      if (sym == escsym) {
        // Elias-gamma escape (range_coder_kernels.cc:449-471); the unary prefix is bounded so that
        // damaged input cannot spin
        int nb = 0;
        while (nb < 31 && tput_bit(st, w.reg) == 0) ++nb;
        int v = 1 << nb;
        while (--nb >= 0) v |= tput_bit(st, w.reg) << nb;
        const int neg = tput_bit(st, w.reg);
        sym = neg ? -v : v + escsym - 1;
      }
      if (st.pos >= 40u) {          // keep digits ahead in the window register (an escape takes <= 63)
        fast_window_advance(w, st.pos, lane);
        st.pos = 0;
      }
      outv = tfc_writelane(sym, n, outv);
      hi_cur = hi_next;
    }
    if (valid) dst.store(s * p.elems + j, t, outv);
  }

  w.wbase += st.pos;
  if (lane == 0) {
    // back to the (base, span-1, window, digits pulled) form the other kernels use
    const long long b = 2ll * w.wbase;
    unsigned int window = 0;
    for (int q4 = -4; q4 < 0; ++q4) {
      const long long q = b + q4;
      window = (window << 8) | ((q >= 0 && q < w.len) ? w.src[q] : 0u);
    }
    p.state[s] = make_uint4(window - st.D, st.span, window, w.wbase);
  }
}

}  // namespace tfc
