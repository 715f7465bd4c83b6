// Multi-stream range ENCODER, four streams per wave — included by range_coder.hip.
//
// enc_fast_kernel (range_encoder_fast.h) resolves the interval recurrence of ONE stream with a 64-lane
// systolic chain: 64 sweeps of 13 vector instructions finalise 64 calls, i.e. every call costs a whole
// wave-wide sweep.  With several steps in flight the coder is bound by VALU issue, so the
// useful work per issued instruction is what counts.  Here a wave carries FOUR streams, one per 16-lane
// DPP row: `row_shr:1` hands (span, base) from lane i-1 to lane i inside a row and leaves lane 0 of each
// row with its own stream's state, so 16 sweeps finalise 16 calls of each of the four streams — the
// chain costs 3.25 instructions per call instead of 13.  The digit phase (emitted 16-bit digits, carry
// ripple, the held digit and its 0xFFFF run — same statement as cc/lib/range_coder.cc:167-245, see
// range_encoder_fast.h) runs per row on 16-bit slices of the same three ballots; the per-stream
// bookkeeping that is wave-uniform in enc_fast_kernel is row-uniform here and lives in VGPRs.
//
// A group is 16 symbols per stream.  Without escape codes that is 16 calls per row and they go from
// registers straight into one round (16 sweeps + digit phase).  A group in which some row has an escape
// (its Elias-gamma bits are up to 62 extra calls) goes through a small per-row call queue in LDS: the
// row's lanes are queued in order, as many as fit, and every row takes rounds of up to 16 calls from its
// queue until all rows are through; rows that are done idle (a round with 0 calls changes nothing).  The
// queues are empty again at the end of the group, so the next escape-free group is back on the direct
// path.  The stream state written back is the one enc_fast_kernel uses, so calls handled by either
// kernel can follow each other on one handle.
#pragma once

namespace tfc {

constexpr int kQuadQueue = 96;     // calls a row can queue: one symbol makes at most 63
static_assert(kQuadQueue >= kMaxCallsPerSymbol, "one symbol's calls must fit the row queue");

struct QuadSink {           // row-uniform
  unsigned short* dst;
  unsigned int ndig;        // digits already stored
  unsigned int cap_dig;
  unsigned int overflow;
};

// Stores `first` followed by `count` copies of `fill`; the row's 16 lanes share the work.
__device__ inline void quad_sink_run(QuadSink& o, unsigned int first, unsigned int fill, unsigned int count,
                                     int i) {
  const unsigned int total = count + 1;
  if (o.ndig + total > o.cap_dig) {
    o.overflow = 1;
  } else {
    unsigned short* p = o.dst + o.ndig;
    for (unsigned int k = i; k < total; k += 16) p[k] = be16(k == 0 ? first : fill);
  }
  o.ndig += total;
}

__device__ inline unsigned int brev16(unsigned int x) { return __builtin_bitreverse32(x) >> 16; }

template <typename Src>
__global__ void enc_quad_kernel(EncParams p, Src src) {
  extern __shared__ int32_t lds[];
  const int waves = blockDim.x >> 6;
  // 16-bit table image + row directory, as in enc_fast_kernel
  const int words = (p.tab.total + 3) >> 2 << 1;
  const uint16_t* tab = reinterpret_cast<const uint16_t*>(lds);
  int2* rows = reinterpret_cast<int2*>(lds + words);
  unsigned int* queues = reinterpret_cast<unsigned int*>(rows + p.tab.ntab);   // [waves][4][kQuadQueue]
  {
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(p.tab.fast16);
    const int pairs = (p.tab.total + 1) >> 1;
    for (int k = threadIdx.x; k < pairs; k += blockDim.x)
      lds[k] = (2 * k + 1 < p.tab.total) ? static_cast<int32_t>(src32[k]) : static_cast<int32_t>(p.tab.fast16[2 * k]);
  }
  for (int k = threadIdx.x; k < p.tab.ntab; k += blockDim.x) rows[k] = p.tab.rows_fast[k];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int seg = lane >> 4, i = lane & 15;
  const int wid = threadIdx.x >> 6;
  const int64_t s = (static_cast<int64_t>(blockIdx.x) * waves + wid) * 4 + seg;   // this row's stream
  if ((static_cast<int64_t>(blockIdx.x) * waves + wid) * 4 >= p.streams) return;  // whole wave idle
  const bool live = s < p.streams;
  const int64_t sc = live ? s : p.streams - 1;      // idle rows shadow the last stream and store nothing

  const uint4 st0 = p.state[sc];
  unsigned int base = st0.x, span = st0.y, pend = st0.z, run = st0.w;   // row-uniform
  QuadSink o;
  const long long off0 = p.chunk_off[sc];
  o.dst = reinterpret_cast<unsigned short*>(p.chunk + off0);
  o.cap_dig = static_cast<unsigned int>((p.chunk_off[sc + 1] - off0) >> 1);
  o.ndig = 0;
  o.overflow = 0;

  const int ntab = p.tab.ntab;
  auto fetch = [&](int64_t j0, int ch, int* t_out) -> int32_t {
    const int64_t j = j0 + i;
    if (j >= p.elems) { *t_out = 0; return 0; }
    const int64_t pos = sc * p.elems + j;
    int t;
    if (p.index) {
      t = min(max(p.index[pos], 0), ntab - 1);      // range errors were reported by the counting pass
    } else {
      t = static_cast<int>((static_cast<unsigned int>(ch) + static_cast<unsigned int>(i)) %
                           static_cast<unsigned int>(ntab));
    }
    *t_out = t;
    return src.load(pos, t);
  };
  // An iteration covers 32 symbols per stream = two rounds of (16 sweeps + digit phase).  The symbols
  // of the next iteration are requested right after this iteration's have been turned into table
  // entries: the digit stores make the outstanding-load count unknowable, so the wait before that is
  // `s_waitcnt vmcnt(0)` and must find only loads that are a whole iteration (~4500 cycles, more than
  // an HBM latency) old.
  auto bounds = [&](int64_t j, int32_t v, int t, unsigned int* lo, unsigned int* hi, int* gamma, int* neg) {
    *lo = 0;
    *hi = 1;
    *gamma = 0;
    *neg = 0;
    if (live && j < p.elems) {
      const Call c = classify_fast(tab, rows[t], v);
      *lo = static_cast<unsigned int>(c.lo16);
      *hi = ((static_cast<unsigned int>(c.hi16) - 1u) & 0xFFFFu) + 1u;    // upper bound, 65536 restored
      *gamma = c.gamma;
      *neg = c.neg;
    }
  };
  const unsigned int ch_step = 16u % static_cast<unsigned int>(ntab);
  auto advance = [&](unsigned int ch) { ch += ch_step; return ch >= static_cast<unsigned int>(ntab) ? ch - ntab : ch; };
  unsigned int chA = 0u, chB = advance(0u);              // channel of the first symbol of round A / B
  int tA = 0, tB = 0;
  int32_t vA = fetch(0, static_cast<int>(chA), &tA);
  int32_t vB = fetch(16, static_cast<int>(chB), &tB);
  const int shift = seg * 16;
  const unsigned int below_me = (1u << i) - 1u;

  unsigned int* rq = queues + ((threadIdx.x >> 6) * 4 + seg) * kQuadQueue;      // this row's call queue

  // one round: chain + digit phase for the first `cnt` (row-uniform, 0..16) calls (lo, hi) of every row
  auto round = [&](unsigned int lo, unsigned int hi, int cnt) __attribute__((always_inline)) {
    const bool valid = i < cnt;
    // ---- chain phase: 16 sweeps, row_shr:1 ----------------------------------------------
    const unsigned long long addA = lo;
    const unsigned long long addB = static_cast<unsigned long long>(static_cast<long long>(hi) - 65536ll);
    unsigned int s_in = span, b_in = base;          // lane 0 of the row keeps these
    unsigned int s_out = 0, b_out = 0, A = 0, bs = 0, t1 = 0;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      s_in = __builtin_amdgcn_update_dpp(s_in, s_out, 0x111, 0xF, 0xF, false);   // row_shr:1
      b_in = __builtin_amdgcn_update_dpp(b_in, b_out, 0x111, 0xF, 0xF, false);
      const unsigned long long PA = static_cast<unsigned long long>(s_in) * lo + addA;
      const unsigned long long PB = static_cast<unsigned long long>(s_in) * hi + addB;
      A = static_cast<unsigned int>(PA >> 16);
      const unsigned int bq = static_cast<unsigned int>(PB >> 16);
      t1 = bq - A;                       // new span - 1 before renormalisation
      bs = b_in + A;                     // new base before renormalisation (wraps)
      const bool renorm = t1 < 65536u;
      s_out = renorm ? ((t1 << 16) | 0xFFFFu) : t1;
      b_out = renorm ? (bs << 16) : bs;
    }

    // ---- digit phase, per row -------------------------------------------------------------
    const bool act = valid;
    const bool flag = act && (t1 < 65536u);          // this call shifted a digit out
    const bool carry = act && (bs < A);              // base + A overflowed 2^32
    const unsigned int e = bs >> 16;                 // the digit, where flag
    const unsigned int Fs = static_cast<unsigned int>(__ballot(flag) >> shift) & 0xFFFFu;
    const unsigned int Gs = static_cast<unsigned int>(__ballot(carry) >> shift) & 0xFFFFu;
    const unsigned int Ps = static_cast<unsigned int>(__ballot(!flag || e == 0xFFFFu) >> shift) & 0xFFFFu;
    // carries travel from a lane to the nearest digit below it and on through 0xFFFF digits:
    // position r of the reversed 16-bit masks = lane 15 - r of the row
    const unsigned int g = brev16(Gs);
    const unsigned int pr = brev16(Ps) & ~g;
    const unsigned int a = g | pr;
    const unsigned int sum = a + g;
    const bool cout = (sum >> 16) != 0;              // leaves below lane 0: hits the held digits
    const unsigned int cin = brev16((sum ^ pr) & 0xFFFFu);
    const unsigned int e2 = (e + ((cin >> i) & 1u)) & 0xFFFFu;
    const unsigned int Ss = static_cast<unsigned int>(__ballot(flag && e2 != 0xFFFFu) >> shift) & 0xFFFFu;
    const unsigned int k = __popc(Fs & below_me);
    const unsigned int K = __popc(Fs);
    const bool had = (pend >> 31) != 0;
    const unsigned int X = (pend + (cout ? 1u : 0u)) & 0xFFFFu;
    const unsigned int fill_in = cout ? 0u : 0xFFFFu;

    if (Ss != 0) {
      const int top = 31 - __clz(static_cast<int>(Ss));      // last digit that is not 0xFFFF
      if (had) quad_sink_run(o, X, fill_in, run, i);
      const unsigned int below = __popc(Fs & ((1u << top) - 1u));
      if (o.ndig + below > o.cap_dig) {
        o.overflow = 1;
      } else if (flag && i < top) {
        o.dst[o.ndig + k] = be16(e2);
      }
      o.ndig += below;
      pend = 0x80000000u | static_cast<unsigned int>(__shfl(static_cast<int>(e2), (lane & 48) + top, 64));
      run = K - below - 1u;
    } else if (K != 0 || cout) {
      // every new digit is 0xFFFF (or there is none)
      if (!had) {
        if (K != 0) {                                // nothing held yet: first 0xFFFF becomes the held digit
          pend = 0x80000000u | 0xFFFFu;
          run = K - 1u;
        }
      } else if (!cout) {
        run += K;
      } else if (run == 0) {
        if (K == 0) {
          quad_sink_run(o, X, 0u, 0u, i);
          pend = 0;
        } else {
          pend = 0x80000000u | X;
          run = K;
        }
      } else {
        // held [X, 0xFFFF x run] became [X+1, 0 x run]
        if (K == 0) {
          quad_sink_run(o, X, 0u, run, i);
          pend = 0;
          run = 0;
        } else {
          quad_sink_run(o, X, 0u, run - 1u, i);
          pend = 0x80000000u;                        // held digit 0x0000
          run = K;
        }
      }
    }
    if (__ballot(cnt != 16) == 0) {   // the row's last lane, broadcast inside the row (row_newbcast:15)
      span = __builtin_amdgcn_update_dpp(0u, s_out, 0x15F, 0xF, 0xF, false);
      base = __builtin_amdgcn_update_dpp(0u, b_out, 0x15F, 0xF, 0xF, false);
    } else {                          // rows with fewer calls: their last call's lane; none: unchanged
      const int last = (lane & 48) + max(cnt, 1) - 1;
      const unsigned int sn = static_cast<unsigned int>(__shfl(static_cast<int>(s_out), last, 64));
      const unsigned int bn = static_cast<unsigned int>(__shfl(static_cast<int>(b_out), last, 64));
      span = cnt > 0 ? sn : span;
      base = cnt > 0 ? bn : base;
    }
  };

  // the 16 symbols j0 .. j0 + 15 of every row: one round straight from registers, or through the queues
  auto emit_group = [&](int64_t j0, unsigned int lo, unsigned int hi, int gamma, int neg)
                        __attribute__((always_inline)) {
    const bool valid = live && j0 + i < p.elems;
    const bool queued_mode = __ballot(valid && gamma > 0) != 0;
    unsigned int rlo = lo, rhi = hi;
    int rcnt = live ? static_cast<int>(min<int64_t>(16, max<int64_t>(0, p.elems - j0))) : 0;
    int nb = 0, ncalls = 0, incl = 0;
    int first = 16, before = 0;                 // lanes below `first` are queued; their calls
    int qcount = 0, qhead = 0;                  // row queue: calls queued / taken
    if (queued_mode) {
      nb = gamma > 0 ? 31 - __clz(gamma) : 0;
      ncalls = valid ? (gamma > 0 ? 2 * nb + 3 : 1) : 0;
      incl = ncalls;                            // inclusive scan inside the row (zeros shifted in)
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
      first = 0;
    }
    while (true) {
      if (queued_mode) {
        // rows whose queue is used up take their next lanes: a prefix of the remaining ones (incl is
        // monotone), at least one (a symbol makes at most 63 calls)
        const bool refill = qhead == qcount && first < 16;
        const unsigned int fits = static_cast<unsigned int>(
            __ballot(refill && i >= first && incl - before <= kQuadQueue) >> shift) & 0xFFFFu;
        const int upto = first + __popc(fits);
        if (refill && valid && i >= first && i < upto) {
          unsigned int* q = rq + (incl - ncalls - before);
          q[0] = (lo & 0xFFFFu) | ((hi - 1u) << 16);
          if (gamma > 0) {
            // Elias gamma: nb zeros, then the nb+1 bits of gamma MSB first, then the sign; every bit is
            // a call [bit, bit+1) / 2  ==  [bit<<15, (bit+1)<<15) / 2^16 (range_coder_kernels.cc:290-322)
            for (int k = 0; k < nb; ++k) q[1 + k] = 0x7FFFu << 16;
            for (int k = nb; k >= 0; --k) {
              const unsigned int bit = (static_cast<unsigned int>(gamma) >> k) & 1u;
              q[1 + nb + (nb - k)] = bit ? (0x8000u | (0xFFFFu << 16)) : (0x7FFFu << 16);
            }
            q[2 * nb + 2] = neg ? (0x8000u | (0xFFFFu << 16)) : (0x7FFFu << 16);
          }
        }
        const int got = __shfl(incl, (lane & 48) + max(upto, 1) - 1, 64) - before;   // calls of lanes < upto
        if (refill) {
          qcount = upto > first ? got : 0;
          qhead = 0;
          before += qcount;
          first = upto;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rcnt = min(16, qcount - qhead);
        const unsigned int word = i < rcnt ? rq[qhead + i] : 0u;
        rlo = word & 0xFFFFu;
        rhi = (word >> 16) + 1u;
        qhead += rcnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (__ballot(rcnt > 0) == 0) break;     // every row is through
      }
      round(rlo, rhi, rcnt);
      if (!queued_mode) break;
    }
  };

  for (int64_t j0 = 0; j0 < p.elems; j0 += 32) {
    unsigned int loA, hiA, loB, hiB;
    int gA, nA, gB, nB;
    bounds(j0 + i, vA, tA, &loA, &hiA, &gA, &nA);
    bounds(j0 + 16 + i, vB, tB, &loB, &hiB, &gB, &nB);
    chA = advance(advance(chA));
    chB = advance(advance(chB));
    vA = fetch(j0 + 32, static_cast<int>(chA), &tA);
    vB = fetch(j0 + 48, static_cast<int>(chB), &tB);
#pragma clang loop unroll(disable)
    for (int g = 0; g < 2; ++g) {               // one copy of the round's code for both groups
      if (j0 + 16 * g < p.elems)
        emit_group(j0 + 16 * g, g ? loB : loA, g ? hiB : hiA, g ? gB : gA, g ? nB : nA);
    }
  }

  if (live && i == 0) {
    p.state[s] = make_uint4(base, span, pend, run);
    p.chunk_len[s] = 2u * o.ndig;
    if (o.overflow) atomicOr(p.overflow_flag, 1u);
  }
}

}  // namespace tfc
