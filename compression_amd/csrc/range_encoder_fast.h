// Fast multi-stream range ENCODER for gfx950 — included by range_coder.hip.
//
// Cost model (tools/ubench/chain.hip, MI355X): one wave issues one instruction
// per ~4.1 cycles whatever its type, so cycles/symbol ~= 4 x instructions on the
// serial path.  This kernel keeps exactly the interval recurrence serial and
// makes everything else wave-parallel:
//
//   1. vector phase   64 symbols -> coder calls (main interval, plus the
//                     Elias-gamma bit calls of escapes) appended to a per-wave
//                     LDS queue, one packed word per call;
//   2. chain phase    64 calls at a time, "systolic": lane n owns call n and
//                     receives (span, base) from lane n-1 through a DPP
//                     wave_shr:1 move; after n+1 sweeps lane n is final.  One
//                     sweep is 13 VALU instructions, no SALU, no branch;
//   3. digit phase    per 64 calls: emitted 16-bit digits, carry events and the
//                     0xFFFF-propagation are three ballots; the carry ripple is
//                     one 64-bit scalar addition (generate/propagate trick);
//                     finished digits leave with one store per lane.
//
// Delayed carries: instead of RangeEncoder's (delay_) encoding we hold back the
// last digit and the 0xFFFF run behind it ("pending digit + run"); a carry can
// only ever reach those (cc/lib/range_coder.cc:167-245 is the same statement),
// so the emitted byte string is identical; tests/ check it byte for byte.
#pragma once

namespace tfc {

constexpr int kMaxCallsPerSymbol = 64;                         // 1 + 2*30 + 2 rounded up
// Call queue of a wave: < 64 calls left over from the previous batch plus what is being queued.  It is
// deliberately small — a batch whose escape codes do not fit is queued in several passes — because
// LDS, not registers, decides how many workgroups share a CU: with a worst-case queue (64 x 64 calls,
// 16.6 KB per wave) only ONE encoder workgroup fitted beside the tables and encode-only launches in
// flight stopped at 57 % of the VALU-issue floor.
constexpr int kRingWords = 256;
static_assert(kRingWords >= 64 + kMaxCallsPerSymbol, "one symbol's calls must fit behind the leftovers");

struct FastEncState {       // wave-uniform
  unsigned int base;
  unsigned int span_m1;
  unsigned int pend;        // bit 31: valid, bits 0..15: held digit
  unsigned int run;         // 0xFFFF digits held behind the pending digit
};

struct FastSink {
  uint8_t* dst;
  unsigned int ndig;        // digits already stored
  unsigned int cap_dig;
  unsigned int overflow;
#ifdef TFC_PHASE_TIMING
  bool is0;
#endif
};

__device__ inline unsigned short be16(unsigned int d) {
  return static_cast<unsigned short>(((d & 0xFF) << 8) | ((d >> 8) & 0xFF));
}

// Stores `first` followed by `count` copies of `fill` (all 16-bit digits).
__device__ inline void sink_run(FastSink& o, unsigned int first, unsigned int fill,
                                unsigned int count, int lane) {
  const unsigned int total = count + 1;
  if (o.ndig + total > o.cap_dig) {
    o.overflow = 1;
  } else {
    unsigned short* p = reinterpret_cast<unsigned short*>(o.dst) + o.ndig;
    for (unsigned int i = lane; i < total; i += 64) p[i] = be16(i == 0 ? first : fill);
  }
  o.ndig += total;
}

__device__ inline unsigned long long brev64(unsigned long long x) { return __builtin_bitreverse64(x); }

#ifdef TFC_PHASE_TIMING   // debug: cycles per phase of stream 0, printed at kernel end
__device__ unsigned long long g_enc_phase[4];
#define TFC_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define TFC_ACC(slot, a, b) do { if (o.is0 && lane == 0) g_enc_phase[slot] += (b) - (a); } while (0)
#else
#define TFC_TICK(var)
#define TFC_ACC(slot, a, b)
#endif

// Chain + digit phase for m (1..64) queued calls; lane n holds call n in `w`.
template <bool FULL>
__device__ __attribute__((always_inline)) inline void consume_calls(FastEncState& st, FastSink& o, unsigned int w, int m,
                                     int lane) {
  const unsigned int lo = w & 0xFFFFu;
  const unsigned int hi = (w >> 16) + 1u;
  const unsigned long long addA = lo;
  const unsigned long long addB =
      static_cast<unsigned long long>(static_cast<long long>(hi) - 65536ll);

  TFC_TICK(tc0);
  unsigned int s_in = st.span_m1, b_in = st.base;   // only lane 0's copy is used as is
  unsigned int s_out = 0, b_out = 0, A = 0, bs = 0, t1 = 0;
  auto sweep = [&]() {
    s_in = __builtin_amdgcn_update_dpp(s_in, s_out, 0x138, 0xF, 0xF, false);  // wave_shr:1
    b_in = __builtin_amdgcn_update_dpp(b_in, b_out, 0x138, 0xF, 0xF, false);
    const unsigned long long PA = static_cast<unsigned long long>(s_in) * lo + addA;
    const unsigned long long PB = static_cast<unsigned long long>(s_in) * hi + addB;
    A = static_cast<unsigned int>(PA >> 16);
    const unsigned int bq = static_cast<unsigned int>(PB >> 16);
    t1 = bq - A;                       // new span - 1 before renormalisation
    bs = b_in + A;                     // new base before renormalisation (wraps)
    const bool renorm = t1 < 65536u;
    s_out = renorm ? ((t1 << 16) | 0xFFFFu) : t1;
    b_out = renorm ? (bs << 16) : bs;
  };
  if (FULL) {
#pragma unroll
    for (int it = 0; it < 64; ++it) sweep();
  } else {
    for (int it = 0; it < m; ++it) sweep();
  }

  TFC_TICK(tc1);
  TFC_ACC(1, tc0, tc1);
  // ---- digit phase --------------------------------------------------------
  const bool act = lane < m;
  const bool flag = act && (t1 < 65536u);          // this call shifted a digit out
  const bool carry = act && (bs < A);              // base + A overflowed 2^32
  const unsigned int e = bs >> 16;                 // the digit, where flag
  const unsigned long long Fm = __ballot(flag);
  const unsigned long long G = __ballot(carry);
  const unsigned long long P = __ballot(!flag || e == 0xFFFFu);
  // carries travel from a lane to the nearest digit below it and on through
  // 0xFFFF digits: position i of the reversed masks = lane 63 - i.
  const unsigned long long g = brev64(G);
  const unsigned long long pr = brev64(P) & ~g;
  const unsigned long long a = g | pr;
  const unsigned long long sum = a + g;
  const bool cout = sum < a;                       // leaves below lane 0: hits the held digits
  const unsigned long long cin = brev64(sum ^ pr);
  const unsigned int e2 = (e + static_cast<unsigned int>((cin >> lane) & 1ull)) & 0xFFFFu;
  const unsigned long long solid = __ballot(flag && e2 != 0xFFFFu);
  const unsigned int k = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned int>(Fm >> 32),
                         __builtin_amdgcn_mbcnt_lo(static_cast<unsigned int>(Fm), 0u));
  const unsigned int K = static_cast<unsigned int>(__popcll(Fm));
  const bool had = (st.pend >> 31) != 0;
  const unsigned int X = (st.pend + (cout ? 1u : 0u)) & 0xFFFFu;
  const unsigned int fill_in = cout ? 0u : 0xFFFFu;

  if (solid != 0) {
    const int top = 63 - __builtin_clzll(solid);   // last digit that is not 0xFFFF
    if (had) sink_run(o, X, fill_in, st.run, lane);
    const unsigned int below = __builtin_amdgcn_readlane(static_cast<int>(k), top);
    if (o.ndig + below > o.cap_dig) {
      o.overflow = 1;
    } else if (flag && lane < top) {
      reinterpret_cast<unsigned short*>(o.dst)[o.ndig + k] = be16(e2);
    }
    o.ndig += below;
    st.pend = 0x80000000u | static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(e2), top));
    st.run = K - below - 1u;
  } else if (K != 0 || cout) {
    // every new digit is 0xFFFF (or there is none)
    if (!had) {
      if (K != 0) {                                // nothing held yet: first 0xFFFF becomes the held digit
        st.pend = 0x80000000u | 0xFFFFu;
        st.run = K - 1u;
      }
    } else if (!cout) {
      st.run += K;
    } else if (st.run == 0) {
      if (K == 0) {
        sink_run(o, X, 0u, 0u, lane);
        st.pend = 0;
      } else {
        st.pend = 0x80000000u | X;
        st.run = K;
      }
    } else {
      // held [X, 0xFFFF x run] became [X+1, 0 x run]
      if (K == 0) {
        sink_run(o, X, 0u, st.run, lane);
        st.pend = 0;
        st.run = 0;
      } else {
        sink_run(o, X, 0u, st.run - 1u, lane);
        st.pend = 0x80000000u;                     // held digit 0x0000
        st.run = K;
      }
    }
  }
  st.span_m1 = __builtin_amdgcn_readlane(static_cast<int>(s_out), m - 1);
  st.base = __builtin_amdgcn_readlane(static_cast<int>(b_out), m - 1);
  TFC_TICK(tc2);
  TFC_ACC(2, tc1, tc2);
}

template <typename Src>
__global__ void enc_fast_kernel(EncParams p, Src src) {
  extern __shared__ int32_t lds[];
  const int waves = blockDim.x >> 6;
  // 16-bit table image, row directory, one call queue per wave
  const int words = (p.tab.total + 3) >> 2 << 1;            // image size in 32-bit words (8-byte multiple)
  const uint16_t* tab = reinterpret_cast<const uint16_t*>(lds);
  int2* rows = reinterpret_cast<int2*>(lds + words);
  unsigned int* rings = reinterpret_cast<unsigned int*>(rows + p.tab.ntab);
  {
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(p.tab.fast16);
    const int pairs = (p.tab.total + 1) >> 1;               // the device buffer holds total (>= 1) entries
    for (int i = threadIdx.x; i < pairs; i += blockDim.x) {
      // the last pair of an odd-sized image would read 2 bytes past the buffer: fetch that entry alone
      lds[i] = (2 * i + 1 < p.tab.total) ? static_cast<int32_t>(src32[i]) : static_cast<int32_t>(p.tab.fast16[2 * i]);
    }
  }
  for (int i = threadIdx.x; i < p.tab.ntab; i += blockDim.x) rows[i] = p.tab.rows_fast[i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t s = static_cast<int64_t>(blockIdx.x) * waves + wid;
  if (s >= p.streams) return;
  if (enc_guard_skips(p, s, lane)) return;
  unsigned int* ring = rings + wid * kRingWords;

  const uint4 st0 = p.state[s];
  FastEncState st;
  st.base = __builtin_amdgcn_readfirstlane(st0.x);
  st.span_m1 = __builtin_amdgcn_readfirstlane(st0.y);
  st.pend = __builtin_amdgcn_readfirstlane(st0.z);
  st.run = __builtin_amdgcn_readfirstlane(st0.w);
  FastSink o;
  const long long off0 = p.chunk_off[s];
  o.dst = p.chunk + off0;
  o.cap_dig = static_cast<unsigned int>((p.chunk_off[s + 1] - off0) >> 1);
  o.ndig = 0;
  o.overflow = 0;
#ifdef TFC_PHASE_TIMING
  o.is0 = s == 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

  int count = 0;                                   // calls queued in ring[0..count)
  const int ntab = p.tab.ntab;
  int ch0 = 0;                                     // channel of the batch's first symbol
  // Symbols (and table indices) are fetched one batch ahead so that the HBM latency of the
  // next 64 symbols hides behind the chain phase of the current ones.
  auto fetch = [&](int64_t j0, int ch, int* t_out) -> int32_t {
    const int64_t j = j0 + lane;
    if (j >= p.elems) { *t_out = 0; return 0; }
    const int64_t pos = s * p.elems + j;
    int t;
    if (p.index) {
      t = min(max(p.index[pos], 0), ntab - 1);     // range errors were reported by the counting pass
    } else {
      t = static_cast<int>((static_cast<unsigned int>(ch) + static_cast<unsigned int>(lane)) %
                           static_cast<unsigned int>(ntab));
    }
    *t_out = t;
    return src.load(pos, t);
  };
  int t_next = 0;
  int32_t v_next = fetch(0, 0, &t_next);
  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
    // ---- vector phase ----------------------------------------------------
    TFC_TICK(tv0);
    const int64_t j = j0 + lane;
    const bool valid = j < p.elems;
    const int t = t_next;
    const int32_t v = v_next;
    Call c;
    c.lo16 = 0; c.hi16 = 1; c.gamma = 0; c.neg = 0; c.bad = 0;
    if (valid) c = classify_fast(tab, rows[t], v);
    ch0 = static_cast<int>((static_cast<unsigned int>(ch0) + 64u) % static_cast<unsigned int>(ntab));
    // The next batch's symbols are requested only now, after this batch's have been consumed:
    // the wait for THESE symbols is an `s_waitcnt vmcnt(0)` (stores of the digit phase make the
    // outstanding count unknown), which would otherwise also wait for the request just issued
    // and expose a full memory latency per batch.  The chain phase below covers it.
    v_next = fetch(j0 + 64, ch0, &t_next);
    const unsigned int word = static_cast<unsigned int>(c.lo16) |
                              ((static_cast<unsigned int>(c.hi16) - 1u) << 16);
    const unsigned long long esc = __ballot(valid && c.gamma > 0);
    const int cnt = static_cast<int>(min<int64_t>(64, p.elems - j0));
    if (esc == 0 && count == 0 && cnt == 64) {
      // common case: exactly one call per symbol and nothing queued — feed the chain phase
      // straight from registers, no trip through the LDS queue
      TFC_TICK(tv1);
      TFC_ACC(0, tv0, tv1);
      consume_calls<true>(st, o, word, 64, lane);
      continue;
    }
    // Queue this batch's calls.  Without escapes that is one call per symbol; with escapes the lanes are
    // queued in order, as many per pass as the queue has room for (incl is monotone, so the lanes that
    // fit are a prefix of the remaining ones; at least one fits: fewer than 64 calls are left queued
    // and a symbol makes at most 63).  After every pass the queue is drained 64 calls at a time.
    int nb = 0, ncalls = 0, incl = 0;
    int first = 64, before = 0;                    // lanes below `first` are queued; their calls
    if (esc == 0) {
      if (valid) ring[count + lane] = word;
      count += cnt;
    } else {
      nb = c.gamma > 0 ? 31 - __clz(c.gamma) : 0;
      ncalls = valid ? (c.gamma > 0 ? 2 * nb + 3 : 1) : 0;
      incl = ncalls;
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      first = 0;
    }
    do {
      if (first < 64) {
        const int room = kRingWords - count;
        const unsigned long long fits = __ballot(lane >= first && incl - before <= room);
        const int upto = first + __popcll(fits);
        unsigned int* q = ring + count + (incl - ncalls - before);
        if (valid && lane >= first && lane < upto) {
          q[0] = word;
          if (c.gamma > 0) {
            // Elias gamma: nb zeros, then the nb+1 bits of gamma MSB first, then the sign;
            // every bit is a call [bit, bit+1) / 2  ==  [bit<<15, (bit+1)<<15) / 2^16.
            for (int i = 0; i < nb; ++i) q[1 + i] = 0x7FFFu << 16;
            for (int i = nb; i >= 0; --i) {
              const unsigned int bit = (static_cast<unsigned int>(c.gamma) >> i) & 1u;
              q[1 + nb + (nb - i)] = bit ? (0x8000u | (0xFFFFu << 16)) : (0x7FFFu << 16);
            }
            q[2 * nb + 2] = c.neg ? (0x8000u | (0xFFFFu << 16)) : (0x7FFFu << 16);
          }
        }
        const int queued = __shfl(incl, upto - 1, 64) - before;
        count += queued;
        before += queued;
        first = upto;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- chain + digit phases, 64 calls at a time; fewer than 64 stay queued ----
      int head = 0;
      while (count - head >= 64) {
        consume_calls<true>(st, o, ring[head + lane], 64, lane);
        head += 64;
      }
      if (head != 0) {
        const int left = count - head;
        const unsigned int keep = lane < left ? ring[head + lane] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < left) ring[lane] = keep;
        count = left;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } while (first < 64);
  }
  if (count > 0) consume_calls<false>(st, o, lane < count ? ring[lane] : 0x00000000u, count, lane);

#ifdef TFC_PHASE_TIMING
  if (s == 0 && lane == 0)
    printf("enc stream 0: total %llu cycles for %lld symbols: vector %llu, chain %llu, digit %llu\n",
           __builtin_readcyclecounter() - t_begin, (long long)p.elems, g_enc_phase[0], g_enc_phase[1], g_enc_phase[2]);
#endif
  if (lane == 0) {
    p.state[s] = make_uint4(st.base, st.span_m1, st.pend, st.run);
    p.chunk_len[s] = 2u * o.ndig;
    if (o.overflow) atomicOr(p.overflow_flag, 1u);
  }
}

}  // namespace tfc
