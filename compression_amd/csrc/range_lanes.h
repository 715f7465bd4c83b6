// Lane-per-stream range coder for gfx950 — included by range_coder.hip.
//
// The wave-per-stream kernels (range_encoder_fast.h / range_decoder_fast.h) spend a whole 64-lane
// instruction on every step of ONE stream's chain: 15 (encode) / 28 (decode) vector instructions per
// symbol, so the vector-issue slots of the chip bound the aggregate rate however many independent
// calls are in flight.  Here every LANE owns a stream and runs the reference's scalar algorithm
// (cc/lib/range_coder.cc:37-264, cc/lib/range_coder.h:224-271, escape codes of
// cc/kernels/range_coder_kernels.cc:290-322, 449-471) on per-lane registers: one vector instruction
// advances 64 streams, ~1 instruction per symbol instead of 15-28.  A lone call is slower this way
// (its latency is elems x the per-symbol chain, and 512 streams are only 8 waves), so the host picks
// this family for throughput-oriented handles and large stream counts (see select_family()).
//
// Decoder: symbol first, successor state afterwards.
//   * quotient estimate  q ~ (D + 1/2) * 2^p / span  in fp32 (v_rcp_f32); |error| < 1 unit of 2^-16;
//   * rank of q among the row's cdf entries: the row's boundaries are a 2^p-bit bitmap in LDS with a
//     running count per 64-bit word, so  s = cum[w] + popcount(bits[w] & below(q)) - 1  is ONE LDS
//     round trip whatever the row width (rows must be strictly increasing, checked on the host);
//   * exact verification with the two bounds the state update needs anyway,
//     A = (span * cdf[s]) >> p <= D < B = (span * cdf[s+1]) >> p  — the reference's search condition
//     (range_coder.h:204-222, 249-258); the estimate is off for ~1e-5 of the symbols
//     (tools/lanes_proto.py), which a wave-uniform rare branch corrects by stepping s.
//   The Elias-gamma escape bits are decoded by the same step on a built-in binary row, with a small
//   per-lane mode machine, so a lane that meets an escape falls behind its neighbours instead of
//   stalling them (lanes run their streams at their own pace; the wave ends with its slowest lane).
//
// Encoder: the reference's state machine (base, span - 1, delay) per lane; a call emits 0, 1 or 2
// digits (a resolved delayed digit and/or the renormalisation digit; longer delayed runs take a rare
// loop), each a 2-byte store at the lane's own cursor.  The slab of a stream is sized on the host from
// a bound that needs no counting pass (lanes_slab_bytes() in range_coder.hip).
//
// LDS image (built by tfc_tables_create): row directory, 16-bit scaled cdf entries, then (decoder
// only) the boundary bitmaps and their running counts.
#pragma once

namespace tfc {

struct LaneArgs {
  const uint32_t* image;       // device copy of the LDS image
  int bytes;                   // bytes of it this kernel needs (encoder: directory + cdf entries)
  int ntab;
  unsigned int cap;            // encoder: slab bytes per stream
};

__device__ inline void lanes_load_image(unsigned char* lds, const LaneArgs& a) {
  const uint4* src = reinterpret_cast<const uint4*>(a.image);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < (a.bytes + 15) / 16; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

__device__ inline unsigned int lds_u16(const unsigned char* lds, unsigned int off) {
  return *reinterpret_cast<const unsigned short*>(lds + off);
}

// (span * c) >> 16 for span = s1 + 1 <= 2^32 and c <= 2^16, as s1 * c + c.
__device__ inline unsigned int scale16(unsigned int s1, unsigned int c) {
  return static_cast<unsigned int>((static_cast<unsigned long long>(s1) * c + c) >> 16);
}

// ---------------------------------------------------------------------------------------------
// Encoder
// ---------------------------------------------------------------------------------------------

struct LaneEmit {
  unsigned char* out;      // this lane's slab
  unsigned int wpos, cap;
  unsigned int overflow;
};

__device__ inline void lane_emit16(LaneEmit& o, unsigned int digit) {
  if (o.wpos + 2u <= o.cap) {
    const unsigned short be = static_cast<unsigned short>(((digit & 0xFFu) << 8) | ((digit >> 8) & 0xFFu));
    __builtin_memcpy(o.out + o.wpos, &be, 2);
  } else {
    o.overflow = 1u;
  }
  o.wpos += 2u;
}

// One RangeEncoder::Encode call (range_coder.cc:37-264) on the interval [lo, hi) / 2^16.
// pd = delay_ & 0xFFFF (0: state 0), pb = delay_ >> 16.
__device__ inline void lane_encode(unsigned int& base, unsigned int& s1, unsigned int& pd, unsigned int& pb,
                                   unsigned int lo, unsigned int hi, LaneEmit& o) {
  const unsigned int a = scale16(s1, lo);
  const unsigned int b = scale16(s1, hi) - 1u;
  base += a;
  s1 = b - a;
  const bool wrapped = base < a;
  const bool ren = (s1 >> 16) == 0;
  if (static_cast<unsigned int>(base + s1) < base) {          // state 1 (the carry is undecided)
    if (ren) {
      base <<= 16;
      s1 = (s1 << 16) | 0xFFFFu;
      pb += 2u;
    }
    return;
  }
  if (pd != 0u) {                                              // state 1 -> 0: the delayed digit is decided
    lane_emit16(o, wrapped ? pd : pd - 1u);
    const unsigned int fill = wrapped ? 0u : 0xFFFFu;
    for (unsigned int k = 0; k < pb; k += 2u) lane_emit16(o, fill);
    pd = 0u;
    pb = 0u;
  }
  if (ren) {
    const unsigned int top = base >> 16;
    base <<= 16;
    s1 = (s1 << 16) | 0xFFFFu;
    if (base <= static_cast<unsigned int>(base + s1)) lane_emit16(o, top);
    else pd = top + 1u;
  }
}

template <typename Src>
__global__ void __launch_bounds__(512) enc_lanes_kernel(EncParams p, Src src, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  lanes_load_image(lanes_lds, la);
  const LaneRow* dir = reinterpret_cast<const LaneRow*>(lanes_lds);

  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = s < p.streams;
  const unsigned int elems = live ? static_cast<unsigned int>(p.elems) : 0u;
  const int64_t pos0 = (live ? s : 0) * p.elems;

  uint4 st = live ? p.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
  unsigned int base = st.x, s1 = st.y, pd = st.z, pb = st.w;
  LaneEmit o;
  o.out = p.chunk + (live ? s : 0) * static_cast<int64_t>(la.cap);
  o.wpos = 0u;
  o.cap = la.cap;
  o.overflow = 0u;

  const unsigned int ntab = static_cast<unsigned int>(la.ntab);
  unsigned int j = 0u;              // next symbol to take
  unsigned int tch = 0u;            // its table in channel mode (j mod ntab)
  unsigned int qn = 0u, g = 0u, neg = 0u;   // escape bits still to code: qn of them, from g then the sign
  decltype(src.raw(0)) raw{};       // symbol j, requested one step ahead
  int tix = 0;
  if (elems != 0u) {
    raw = src.raw(pos0);
    if (p.index) tix = p.index[pos0];
  }

  while (__any(j < elems || qn != 0u)) {
    if (j < elems || qn != 0u) {
      unsigned int lo, hi;
      if (qn == 0u) {
        int t = static_cast<int>(tch);
        if (p.index) {
          t = tix;
          if (t < 0 || t >= la.ntab) {
            atomicMin(p.first_error, static_cast<unsigned long long>(pos0 + j));
            t = 0;
          }
        }
        const int32_t v = src.quant(raw, t);
        const LaneRow row = dir[t];
        const int nsym = static_cast<int>(row.info & 0xFFFFu);
        int sym = v;
        if (row.info >> 31) {
          const int vmax = nsym - 1;           // the last interval is the escape symbol
          if (v < 0 || v >= vmax) {
            neg = v < 0 ? 1u : 0u;
            g = v < 0 ? 0u - static_cast<unsigned int>(v) : static_cast<unsigned int>(v - vmax) + 1u;
            sym = vmax;
            qn = 2u * static_cast<unsigned int>(31 - __clz(static_cast<int>(g))) + 2u;
          }
        } else if (v < 0 || v >= nsym) {
          atomicMin(p.first_error, static_cast<unsigned long long>(pos0 + j));
          sym = 0;
        }
        lo = lds_u16(lanes_lds, row.cdf + 2u * static_cast<unsigned int>(sym));
        hi = lds_u16(lanes_lds, row.cdf + 2u * static_cast<unsigned int>(sym) + 2u);
        if (hi == 0u) hi = 65536u;
        ++j;
        ++tch;
        if (tch == ntab) tch = 0u;
        if (j < elems) {
          raw = src.raw(pos0 + j);
          if (p.index) tix = p.index[pos0 + j];
        }
      } else {
        // Elias-gamma code of g (floor(log2 g) zeros, the bits of g), then the sign bit
        // (range_coder_kernels.cc:304-321), each a call with the uniform binary cdf at precision 1.
        --qn;
        const unsigned int sft = qn - 1u;      // qn = 0: the sign
        const unsigned int bit = qn == 0u ? neg : (sft < 32u ? (g >> sft) & 1u : 0u);
        lo = bit << 15;
        hi = (bit + 1u) << 15;
      }
      lane_encode(base, s1, pd, pb, lo, hi, o);
    }
  }
  if (live) {
    p.state[s] = make_uint4(base, s1, pd, pb);
    p.chunk_len[s] = o.wpos;
    if (o.overflow) atomicOr(p.overflow_flag, 1u);
  }
}

// ---------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------

// Bytes [pos, pos + 4) of the stream as a little-endian word; bytes past the end read as zero
// (Read16BitValue, range_coder.h:273-282).
__device__ inline unsigned int lane_window(const unsigned char* src, unsigned int pos, unsigned int len) {
  unsigned int w;
  if (pos + 4u <= len) {
    __builtin_memcpy(&w, src + pos, 4);
  } else {
    w = 0u;
    for (unsigned int k = 0; k < 4u; ++k)
      if (pos + k < len) w |= static_cast<unsigned int>(src[pos + k]) << (8u * k);
  }
  return w;
}

template <typename Dst>
__global__ void __launch_bounds__(512) dec_lanes_kernel(DecParams p, Dst dst, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  lanes_load_image(lanes_lds, la);
  const LaneRow* dir = reinterpret_cast<const LaneRow*>(lanes_lds);

  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = s < p.streams;
  const unsigned int elems = live ? static_cast<unsigned int>(p.elems) : 0u;
  const int64_t pos0 = (live ? s : 0) * p.elems;

  const uint4 st = live ? p.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 2u);
  unsigned int D = st.z - st.x;      // window - base
  unsigned int s1 = st.y;            // span - 1
  const long long o0 = live ? p.off[s] : 0;
  const unsigned int len = live ? static_cast<unsigned int>(p.off[s + 1] - o0) : 0u;
  const unsigned char* srcp = p.blob + o0;
  unsigned int pos = 2u * st.w;      // bytes consumed
  unsigned int win = lane_window(srcp, pos, len);
  unsigned int woff = 0u;            // bit offset in `win` of the next digit (0 or 16)

  const int ntab = la.ntab;
  const LaneRow bin = dir[ntab];     // uniform binary row {0, 1/2, 1} for the escape bits
  unsigned int j = 0u;
  int t = 0, tn = 1 % ntab;          // table of symbol j and of symbol j + 1 (channel mode)
  int ixn = 0;                       // index mode: table index of symbol j + 1
  if (p.index && elems != 0u) {
    t = p.index[pos0];
    if (t < 0 || t >= ntab) {
      atomicMin(p.first_error, static_cast<unsigned long long>(pos0));
      t = 0;
    }
    ixn = elems > 1u ? p.index[pos0 + 1] : 0;
  }
  LaneRow row = dir[t];
  unsigned int mode = 0u;            // 0 symbol, 1 unary prefix, 2 payload bits, 3 sign
  unsigned int nb = 0u, val = 0u;

  while (__any(j < elems)) {
    if (j < elems) {
      // requests for the NEXT step: the directory entry of symbol j + 1 and the input window at the
      // current position (the next digit is at offset 0 or 2 in it)
      int tnext = tn;
      if (p.index) {
        tnext = ixn;
        if (tnext < 0 || tnext >= ntab) {
          if (j + 1u < elems) atomicMin(p.first_error, static_cast<unsigned long long>(pos0 + j + 1u));
          tnext = 0;
        }
      }
      const LaneRow rown = dir[tnext];
      const unsigned int winn = lane_window(srcp, pos, len);
      int ix2 = 0;
      if (p.index && j + 2u < elems) ix2 = p.index[pos0 + j + 2u];

      const LaneRow R = mode != 0u ? bin : row;
      const unsigned int nsym = R.info & 0xFFFFu;
      const unsigned int sh = (R.info >> 16) & 31u;
      // ---- symbol first: quotient estimate -> rank ---------------------------------------
      const float fq = (static_cast<float>(D) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(s1)) * 65536.0f;
      unsigned int cp = static_cast<unsigned int>(fq);
      cp = min(cp, 65535u) >> sh;
      const unsigned int w = cp >> 6;
      const unsigned long long word = *reinterpret_cast<const unsigned long long*>(lanes_lds + R.bits + 8u * w);
      const unsigned int cum = lds_u16(lanes_lds, R.cum + 2u * w);
      const unsigned long long below = ~0ull >> (63u - (cp & 63u));
      unsigned int sym = cum + static_cast<unsigned int>(__popcll(word & below)) - 1u;
      // ---- exact bounds, verification ----------------------------------------------------
      unsigned int lo = lds_u16(lanes_lds, R.cdf + 2u * sym);
      unsigned int hi = lds_u16(lanes_lds, R.cdf + 2u * sym + 2u);
      if (hi == 0u) hi = 65536u;
      unsigned int A = scale16(s1, lo);
      unsigned int b = scale16(s1, hi) - 1u;      // B - 1; B = 2^32 wraps to 0 (K3 pins this)
      bool bad = D < A || D > b;
      if (__any(bad)) {
        for (int it = 0; it < 4 && __any(bad); ++it) {
          if (bad) {
            if (D < A) sym = sym > 0u ? sym - 1u : 0u;
            else sym = sym + 1u < nsym ? sym + 1u : nsym - 1u;
            lo = lds_u16(lanes_lds, R.cdf + 2u * sym);
            hi = lds_u16(lanes_lds, R.cdf + 2u * sym + 2u);
            if (hi == 0u) hi = 65536u;
            A = scale16(s1, lo);
            b = scale16(s1, hi) - 1u;
            bad = D < A || D > b;
          }
        }
        // still bad: damaged input (offset outside the interval); like the wave-per-stream
        // kernels the step is taken with the clamped symbol and never leaves the tables
      }
      // ---- successor state ---------------------------------------------------------------
      D -= A;
      s1 = b - A;
      const unsigned int dig0 = (win >> woff) & 0xFFFFu;                 // bytes (hi, lo) little-endian
      const unsigned int dig = ((dig0 & 0xFFu) << 8) | (dig0 >> 8);
      if ((s1 >> 16) == 0u) {
        D = (D << 16) | dig;
        s1 = (s1 << 16) | 0xFFFFu;
        pos += 2u;
        woff = 16u;
      } else {
        woff = 0u;
      }
      win = winn;
      // ---- what the decoded value means --------------------------------------------------
      bool done = false;
      int outv = static_cast<int>(sym);
      if (mode == 0u) {
        if ((row.info >> 31) && sym == nsym - 1u) {
          mode = 1u;
          nb = 0u;
        } else {
          done = true;
        }
      } else if (mode == 1u) {
        // unary prefix, bounded so that damaged input cannot spin (range_coder_kernels.cc:449-471)
        if (sym == 0u) {
          ++nb;
          if (nb == 31u) { val = 1u << 31; mode = 2u; }
        } else {
          val = 1u << nb;
          mode = nb != 0u ? 2u : 3u;
        }
      } else if (mode == 2u) {
        --nb;
        val |= sym << nb;
        if (nb == 0u) mode = 3u;
      } else {
        const int escsym = static_cast<int>(row.info & 0xFFFFu) - 1;
        outv = sym != 0u ? -static_cast<int>(val) : static_cast<int>(val) + escsym - 1;
        mode = 0u;
        done = true;
      }
      if (done) {
        dst.store(pos0 + j, t, outv);
        ++j;
        t = tnext;
        row = rown;
        tn = tnext + 1 == ntab ? 0 : tnext + 1;
        ixn = ix2;
      }
    }
  }

  if (live) {
    // back to the (base, span - 1, window, digits pulled) form shared with the other kernels
    unsigned int window = 0u;
    for (int i = -4; i < 0; ++i) {
      const long long q = static_cast<long long>(pos) + i;
      window = (window << 8) | ((q >= 0 && q < static_cast<long long>(len)) ? srcp[q] : 0u);
    }
    p.state[s] = make_uint4(window - D, s1, window, pos >> 1);
  }
}

}  // namespace tfc
