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
// Memory: a lane's accesses are 64 different cache lines per wave instruction, and anything a step
// waits for sits on the chain (first version, one 4-byte load and store per step: ~1300 cycles per step,
// almost all of it memory latency — and hipcc's s_waitcnt vmcnt(0) for a pending load also waits for
// every store issued before it).  So ALL global traffic of a wave is issued in a "memory phase" every
// C steps, and the steps in between touch LDS only:
//   * inputs (symbols / bottleneck values, index, the decoder's code bytes): per lane a window of 2 C
//     steps' worth of bytes, requested at one memory phase (dwordx4 loads into registers), parked in LDS
//     at the next and read from there for C steps — by then the lane has moved at most C elements, so
//     every element it can need is inside; lanes stay free to run at their own pace;
//   * outputs (the encoder's digits, the decoder's elements): collected per lane in LDS, stored 16 bytes
//     at a time at the next memory phase.
//   A lane's staging area is contiguous with a stride of 8 (mod 16) bytes between lanes, so that the
//   8-byte accesses of a phase and the 2/4-byte accesses of a step are at most two lanes per bank.
//   Whatever the phase issues has C steps (>= 1000 cycles) to complete before anything waits for it.
//
// Instruction count is what a lone wave pays for (~4.7 cycles per instruction of any kind, vector or
// scalar, measured with the SQ counters): the steps are written branch-free for the common case —
// state updates by select, speculative LDS writes that only count when their cursor advances — with the
// rare cases (escape bits, estimate corrections, delayed-digit runs) behind wave-uniform branches.
// All rows of the tables must have the same precision (the quotient scale is a scalar constant).
//
// LDS image (built by tfc_tables_create): row directory, 16-bit scaled cdf entries, then (decoder
// only) the boundary bitmaps and their running counts.
#pragma once

namespace tfc {

struct LaneArgs {
  const uint32_t* image;       // device copy of the LDS image
  int bytes;                   // bytes of it this kernel needs (encoder: directory + cdf entries)
  int ntab;
  int precision;               // of every row
  unsigned int cap;            // encoder: slab bytes per stream
  int lds_image;               // LDS bytes reserved for the image (multiple of 1024)
  int lds_wave;                // LDS bytes of every wave's private area behind it
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
  unsigned int add = c;
  asm volatile("" : "+v"(add));     // opaque: else LLVM rewrites it as (s1 + 1) * c, a 33-bit multiplicand, two multiplies
  return static_cast<unsigned int>((static_cast<unsigned long long>(s1) * c + add) >> 16);
}

// Several independent calls (same tables, same geometry, different handles) run as ONE launch: the
// hardware overlaps at most ~8 kernels however many HIP streams carry them, and a 512-stream call is
// only 8 waves — so the way to fill the chip with independent calls is to put them into one grid.  The
// jobs travel as a kernel argument (captured at launch, no device copy to keep alive).
constexpr int kLaneJobArgBytes = 3584;     // of the 4 KB a kernel's arguments may take
template <typename Src>
struct EncLaneJob {
  Src src;
  const int32_t* index;             // null: channel mode
  uint4* state;
  uint8_t* chunk;
  unsigned int* chunk_len;
  unsigned long long* first_error;
  unsigned int* overflow_flag;
};
template <typename Src>
struct EncLaneJobs {
  int64_t streams, elems;           // of every job
  int blocks_per_job, n;
  static constexpr int kMax = kLaneJobArgBytes / sizeof(EncLaneJob<Src>) < 64 ? kLaneJobArgBytes / sizeof(EncLaneJob<Src>) : 64;
  EncLaneJob<Src> job[kMax];
};
template <typename Dst>
struct DecLaneJob {
  Dst dst;
  const int32_t* index;
  const uint8_t* blob;
  const long long* off;
  uint4* state;
  unsigned long long* first_error;
};
template <typename Dst>
struct DecLaneJobs {
  int64_t streams, elems;
  int blocks_per_job, n;
  static constexpr int kMax = kLaneJobArgBytes / sizeof(DecLaneJob<Dst>) < 64 ? kLaneJobArgBytes / sizeof(DecLaneJob<Dst>) : 64;
  DecLaneJob<Dst> job[kMax];
};

// Global memory through address-space-1 pointers, at any alignment.  The pointers of a job come out of
// the kernel-argument struct by a dynamic index, so hipcc takes them for generic ("flat") pointers —
// and a flat access counts on lgkmcnt as well: every LDS read of the steps after a memory phase would
// wait for the phase's global loads.
typedef unsigned int lanes_u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
struct __attribute__((packed)) LanePacked { T v; };
#define TFC_AS1 __attribute__((address_space(1)))
__device__ inline uint4 lanes_gload16(const void* p) {
  const lanes_u32x4 v = reinterpret_cast<const TFC_AS1 LanePacked<lanes_u32x4>*>((const TFC_AS1 void*)p)->v;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline unsigned int lanes_gload8(const void* p) {
  return *reinterpret_cast<const TFC_AS1 unsigned char*>((const TFC_AS1 void*)p);
}
__device__ inline void lanes_gstore16(void* p, const uint2& lo, const uint2& hi) {
  lanes_u32x4 v = {lo.x, lo.y, hi.x, hi.y};
  reinterpret_cast<TFC_AS1 LanePacked<lanes_u32x4>*>((TFC_AS1 void*)p)->v = v;
}
// one element of 2 or 4 bytes, by its bits
template <typename T>
__device__ inline void lanes_gstore_elem(T* p, const T& v) {
  if (sizeof(T) == 2) {
    unsigned short bits;
    __builtin_memcpy(&bits, &v, 2);
    reinterpret_cast<TFC_AS1 LanePacked<unsigned short>*>((TFC_AS1 void*)p)->v = bits;
  } else {
    unsigned int bits;
    __builtin_memcpy(&bits, &v, 4);
    reinterpret_cast<TFC_AS1 LanePacked<unsigned int>*>((TFC_AS1 void*)p)->v = bits;
  }
}

// Values loaded before the main loop are "used" here, so that hipcc waits for them HERE: it places
// the s_waitcnt of a pending load at its first use, and a first use inside the loop means an
// s_waitcnt vmcnt(0) — which also waits for every store in flight — in every iteration.
__device__ inline void lanes_pin(unsigned int& a, unsigned int& b, unsigned int& c, unsigned int& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// ---------------------------------------------------------------------------------------------
// Per-lane input window
// ---------------------------------------------------------------------------------------------

// 8 * WORDS bytes of this lane's byte stream starting at an arbitrary position, fetched by request()
// into registers and parked by commit() in this lane's LDS area; bytes [base, base + 8 WORDS) are then
// at lds[0 ..).  Bytes past the stream's end read as zero.
template <int WORDS>
struct LaneWindow {
  unsigned char* lds;        // this lane's staging area (8-byte aligned)
  const unsigned char* g;    // this lane's stream
  unsigned int len;          // its length in bytes
  unsigned int base;         // stream position of the parked window
  unsigned int pbase;        // ... of the requested one
  uint2 pend[WORDS];

  __device__ __noinline__ unsigned int tail_word(unsigned int q0) const {
    unsigned int word = 0u;
#pragma nounroll
    for (unsigned int k = 0; k < 4u; ++k)
      if (q0 + k < len) word |= static_cast<unsigned int>(lanes_gload8(g + q0 + k)) << (8u * k);
    return word;
  }
  __device__ void request(unsigned int pos) {
    pbase = pos;
    if (pos + 8u * WORDS <= len) {
#pragma unroll
      for (int w = 0; w < WORDS; w += 2) {
        const uint4 v = lanes_gload16(g + pos + 8u * w);
        pend[w] = make_uint2(v.x, v.y);
        pend[w + 1] = make_uint2(v.z, v.w);
      }
    } else {
      // the stream ends inside the window (its last phases only): byte by byte, zero behind the end
      // (the word loops are unrolled so that `pend` stays in registers)
#pragma unroll
      for (int w = 0; w < WORDS; ++w) {
        pend[w].x = tail_word(pos + 8u * w);
        pend[w].y = tail_word(pos + 8u * w + 4u);
      }
    }
  }
  __device__ void commit() {
#pragma unroll
    for (int w = 0; w < WORDS; ++w) reinterpret_cast<uint2*>(lds)[w] = pend[w];
    base = pbase;
  }
};

// byte stride between the staging areas of neighbouring lanes: payload rounded up to 8, plus 8 when that
// is a multiple of 16 (then two lanes share a bank instead of 4 or 8)
__host__ __device__ constexpr int lane_stride(int payload) {
  return ((payload + 7) & ~7) + ((((payload + 7) & ~7) % 16 == 0) ? 8 : 0);
}

// ---------------------------------------------------------------------------------------------
// Encoder
// ---------------------------------------------------------------------------------------------

constexpr unsigned int kEncCadence = 16;        // steps between memory phases
constexpr unsigned int kEncDigitBytes = 64;     // digit bytes staged per lane between two phases

// LDS of one encoder wave: per lane digits, value window (2 cadences of elements), index window
template <typename Raw>
struct EncWaveLds {
  static constexpr int kValueWords = 2 * kEncCadence * sizeof(Raw) / 8;
  static constexpr int kIndexWords = 2 * kEncCadence * 4 / 8;
  static constexpr int kDigitStride = lane_stride(kEncDigitBytes + 8);   // + the speculative write behind a full area
  static constexpr int kValueStride = lane_stride(8 * kValueWords);
  static constexpr int kIndexStride = lane_stride(8 * kIndexWords);
  static constexpr int kDigits = 0;
  static constexpr int kValue = kDigits + 64 * kDigitStride;
  static constexpr int kIndex = kValue + 64 * kValueStride;
  static constexpr int kBytes = kIndex + 64 * kIndexStride;
};

template <bool INDEXED, typename Src>
__global__ void __launch_bounds__(512) enc_lanes_kernel(const EncLaneJobs<Src> jobs, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  lanes_load_image(lanes_lds, la);
  using Raw = typename Src::raw_type;
  using L = EncWaveLds<Raw>;
  constexpr unsigned int kRaw = sizeof(Raw);

  const EncLaneJob<Src>& J = jobs.job[blockIdx.x / static_cast<unsigned int>(jobs.blocks_per_job)];
  const Src src = J.src;
  const int32_t* const index = J.index;
  unsigned long long* const first_error = J.first_error;
  const unsigned int lane = threadIdx.x & 63u;
  const int64_t s = static_cast<int64_t>(blockIdx.x % static_cast<unsigned int>(jobs.blocks_per_job)) * blockDim.x + threadIdx.x;
  const bool live = s < jobs.streams;
  const unsigned int elems = live ? static_cast<unsigned int>(jobs.elems) : 0u;
  const int64_t pos0 = (live ? s : 0) * jobs.elems;

  uint4 st = live ? J.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
  unsigned int base = st.x, s1 = st.y, pd = st.z, pb = st.w;
  lanes_pin(base, s1, pd, pb);

  unsigned char* const wave_lds = lanes_lds + la.lds_image + (threadIdx.x >> 6) * la.lds_wave;
  unsigned char* const dstage = wave_lds + L::kDigits + L::kDigitStride * lane;   // this lane's digit bytes
  unsigned char* const out = J.chunk + (live ? s : 0) * static_cast<int64_t>(la.cap);
  unsigned int wpos = 0u;          // slab bytes written by earlier phases
  unsigned int n = 0u;             // digit bytes produced since
  unsigned int overflow = 0u;
  LaneWindow<L::kValueWords> vw;
  vw.lds = wave_lds + L::kValue + L::kValueStride * lane;
  vw.g = reinterpret_cast<const unsigned char*>(src.base() + pos0);
  vw.len = elems * kRaw;
  vw.request(0u);
  LaneWindow<L::kIndexWords> iw;
  if (INDEXED) {
    iw.lds = wave_lds + L::kIndex + L::kIndexStride * lane;
    iw.g = reinterpret_cast<const unsigned char*>(index + pos0);
    iw.len = elems * 4u;
    iw.request(0u);
  }

  const unsigned int dir_end = 16u * static_cast<unsigned int>(la.ntab);
  unsigned int j = 0u;              // next symbol to take
  unsigned int dirp = 0u;           // channel mode: LDS offset of its directory entry
  unsigned int qn = 0u, g = 0u, neg = 0u;   // escape bits still to code: qn of them, from g then the sign

  // Digit `d` into the staging area, speculatively: it counts only if `on` advances the cursor.  The
  // area holds what the kEncCadence steps between two phases can produce (two digits each); the run of
  // digits behind a resolved long delay goes through put_run, which empties the area first.
  auto put = [&](unsigned int d, bool on) {
    *reinterpret_cast<unsigned short*>(dstage + n) = __builtin_bswap16(static_cast<unsigned short>(d));
    n += on ? 2u : 0u;
  };
  auto flush = [&]() {
#pragma unroll
    for (unsigned int c = 0; c < kEncDigitBytes / 16u; ++c) {
      if (16u * c < n) {
        uint2 v[2];
        v[0] = reinterpret_cast<const uint2*>(dstage)[2 * c];
        v[1] = reinterpret_cast<const uint2*>(dstage)[2 * c + 1];
        if (wpos + 16u * c + 16u <= la.cap) lanes_gstore16(out + wpos + 16u * c, v[0], v[1]);
        else overflow = 1u;
      }
    }
    wpos += n;
    n = 0u;
  };
  auto put_run = [&](unsigned int fill, unsigned int bytes) {
    flush();
    for (unsigned int k = 0; k < bytes; k += 2u) {
      const unsigned short be = static_cast<unsigned short>(fill);       // 0x0000 / 0xFFFF: no byte order
      if (wpos + 2u <= la.cap) lanes_gstore_elem(reinterpret_cast<unsigned short*>(out + wpos), be);
      else overflow = 1u;
      wpos += 2u;
    }
  };

  for (unsigned int it = 0u; __any(j < elems || qn != 0u); ++it) {
    if ((it & (kEncCadence - 1u)) == 0u) {
      // memory phase: park what the previous phase requested, request from the current position, store
      // the digits of the last kEncCadence steps
      vw.commit();
      vw.request(j * kRaw);
      if (INDEXED) {
        iw.commit();
        iw.request(j * 4u);
      }
      flush();
    }
    // ---- the call of this step: speculatively the next symbol as a plain one (reads stay inside the
    // lane's window and the directory whatever j is); escapes, escape bits, range errors and idle lanes
    // are sorted out behind a wave-uniform branch ---------------------------------------------------
    unsigned int dp = dirp;
    if (INDEXED) {
      int t = *reinterpret_cast<const int*>(iw.lds + (j * 4u - iw.base));
      t = (t < 0 || t >= la.ntab) ? -1 : t;
      dp = t < 0 ? 0u : 16u * static_cast<unsigned int>(t);
      if (__any(t < 0 && qn == 0u && j < elems)) {
        if (t < 0 && qn == 0u && j < elems) atomicMin(first_error, static_cast<unsigned long long>(pos0 + j));
      }
    }
    const int32_t v = src.quant(*reinterpret_cast<const Raw*>(vw.lds + (j * kRaw - vw.base)), static_cast<int>(dp >> 4));
    const uint2 row = *reinterpret_cast<const uint2*>(lanes_lds + dp);   // cdf offset, limit | escape << 31
    const unsigned int limit = row.y & 0x7FFFFFFFu;                      // first value that is not a plain symbol
    const bool take = qn == 0u && j < elems;
    const bool plain = static_cast<unsigned int>(v) < limit;             // negative values are not
    unsigned int sym = plain ? static_cast<unsigned int>(v) : limit;
    bool act = take;                 // this lane makes a coder call in this step
    bool adv = take;                 // ... and moves on to the next symbol
    unsigned int lo = 0u, hi = 0u;
    if (__any(!(take && plain) && (qn != 0u || j < elems))) {
      if (take && !plain) {
        if (row.y >> 31) {
          // escape: the row's last interval now, the Elias-gamma code of the excess in the next steps
          neg = v < 0 ? 1u : 0u;
          g = v < 0 ? 0u - static_cast<unsigned int>(v) : static_cast<unsigned int>(v) - limit + 1u;
          qn = 2u * static_cast<unsigned int>(31 - __clz(static_cast<int>(g))) + 2u;
        } else {
          atomicMin(first_error, static_cast<unsigned long long>(pos0 + j));
          sym = 0u;
        }
      } else if (!take && qn != 0u) {
        // Elias-gamma code of g (floor(log2 g) zeros, the bits of g), then the sign bit
        // (range_coder_kernels.cc:304-321), each a call with the uniform binary cdf at precision 1.
        --qn;
        const unsigned int sft = qn - 1u;      // qn = 0: the sign
        const unsigned int bit = qn == 0u ? neg : (sft < 32u ? (g >> sft) & 1u : 0u);
        lo = bit << 15;
        hi = (bit + 1u) << 15;
        act = true;
      }
    }
    {
      const unsigned int tlo = lds_u16(lanes_lds, row.x + 2u * sym);
      const unsigned int thi = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
      lo = take ? tlo : lo;
      hi = take ? (thi == 0u ? 65536u : thi) : hi;
    }
    j += adv ? 1u : 0u;
    if (!INDEXED) {
      const unsigned int nd = dirp + 16u == dir_end ? 0u : dirp + 16u;
      dirp = adv ? nd : dirp;
    }
    // ---- RangeEncoder::Encode (range_coder.cc:37-264) on [lo, hi) / 2^16; pd = delay_ & 0xFFFF
    // (0: state 0), pb = delay_ >> 16; every update is a select on `act` --------------------------------
    const unsigned int a = scale16(s1, lo);
    const unsigned int b = scale16(s1, hi) - 1u;
    const unsigned int base1 = base + a;
    const unsigned int s11 = b - a;
    const bool wrapped = base1 < a;
    const bool st1 = static_cast<unsigned int>(base1 + s11) < base1;      // the carry is (still) undecided
    const bool ren = act && (s11 >> 16) == 0u;
    // state 1 -> 0: the delayed digit is decided (and the run of 0x0000 / 0xFFFF digits behind it)
    const bool resolve = act && !st1 && pd != 0u;
    put(wrapped ? pd : pd - 1u, resolve);
    if (__any(resolve && pb != 0u)) {
      if (resolve && pb != 0u) put_run(wrapped ? 0u : 0xFFFFu, pb);
    }
    pd = resolve ? 0u : pd;
    pb = resolve ? 0u : pb;
    // renormalisation
    const unsigned int top = base1 >> 16;
    const unsigned int base2 = ren ? base1 << 16 : base1;
    const unsigned int s12 = ren ? (s11 << 16) | 0xFFFFu : s11;
    const bool st1r = static_cast<unsigned int>(base2 + s12) < base2;     // state after the shift
    put(top, ren && !st1 && !st1r);
    pd = (ren && !st1 && st1r) ? top + 1u : pd;
    pb = (ren && st1) ? pb + 2u : pb;
    base = act ? base2 : base;
    s1 = act ? s12 : s1;
  }
  flush();
  if (live) {
    J.state[s] = make_uint4(base, s1, pd, pb);
    J.chunk_len[s] = wpos;
    if (overflow) atomicOr(J.overflow_flag, 1u);
  }
}

// ---------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------

constexpr unsigned int kDecCadence = 8;

// LDS of one decoder wave: per lane the code-byte window (a step consumes <= 2 bytes), the decoded
// elements of one cadence, and the index window.
template <typename Elem>
struct DecWaveLds {
  static constexpr int kCodeWords = 2 * kDecCadence * 2 / 8;
  static constexpr int kIndexWords = 2 * kDecCadence * 4 / 8;
  static constexpr int kOutBytes = kDecCadence * sizeof(Elem);
  static constexpr int kCodeStride = lane_stride(8 * kCodeWords);
  static constexpr int kOutStride = lane_stride(kOutBytes);
  static constexpr int kIndexStride = lane_stride(8 * kIndexWords);
  static constexpr int kCodes = 0;
  static constexpr int kOut = kCodes + 64 * kCodeStride;
  static constexpr int kIndex = kOut + 64 * kOutStride;
  static constexpr int kBytes = kIndex + 64 * kIndexStride;
};

template <bool INDEXED, typename Dst>
__global__ void __launch_bounds__(512) dec_lanes_kernel(const DecLaneJobs<Dst> jobs, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  lanes_load_image(lanes_lds, la);
  using Elem = typename Dst::elem;
  using L = DecWaveLds<Elem>;
  constexpr unsigned int kEs = sizeof(Elem);

  const DecLaneJob<Dst>& J = jobs.job[blockIdx.x / static_cast<unsigned int>(jobs.blocks_per_job)];
  const Dst dst = J.dst;
  const int32_t* const index = J.index;
  unsigned long long* const first_error = J.first_error;
  const unsigned int lane = threadIdx.x & 63u;
  const int64_t s = static_cast<int64_t>(blockIdx.x % static_cast<unsigned int>(jobs.blocks_per_job)) * blockDim.x + threadIdx.x;
  const bool live = s < jobs.streams;
  const unsigned int elems = live ? static_cast<unsigned int>(jobs.elems) : 0u;
  const int64_t pos0 = (live ? s : 0) * jobs.elems;

  const uint4 st = live ? J.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 2u);
  unsigned int D = st.z - st.x;      // window - base
  unsigned int s1 = st.y;            // span - 1
  const long long o0 = live ? J.off[s] : 0;
  unsigned int len = live ? static_cast<unsigned int>(J.off[s + 1] - o0) : 0u;
  unsigned int pos_start = 2u * st.w;      // bytes consumed
  lanes_pin(D, s1, len, pos_start);

  unsigned char* const wave_lds = lanes_lds + la.lds_image + (threadIdx.x >> 6) * la.lds_wave;
  LaneWindow<L::kCodeWords> cw;
  cw.lds = wave_lds + L::kCodes + L::kCodeStride * lane;
  cw.g = J.blob + o0;
  cw.len = len;
  cw.request(pos_start);
  cw.base = pos_start;
  const unsigned char* cp = cw.lds;      // LDS address of the next code digit: stream position cw.base + (cp - cw.lds)
  unsigned char* const outq = wave_lds + L::kOut + L::kOutStride * lane;
  unsigned int ko = 0u;                  // bytes of decoded elements waiting in outq
  LaneWindow<L::kIndexWords> iw;
  if (INDEXED) {
    iw.lds = wave_lds + L::kIndex + L::kIndexStride * lane;
    iw.g = reinterpret_cast<const unsigned char*>(index + pos0);
    iw.len = elems * 4u;
    iw.request(0u);
  }

  const float scale = static_cast<float>(1u << la.precision);     // quotient scale: 2^precision
  const unsigned int cp_max = (1u << la.precision) - 1u;
  const unsigned int dir_end = 16u * static_cast<unsigned int>(la.ntab);
  unsigned int j = 0u;
  unsigned int dirp = 0u;            // channel mode: LDS offset of the directory entry of symbol j
  unsigned int mode = 0u;            // 0 symbol, 1 unary prefix, 2 payload bits, 3 sign
  unsigned int nb = 0u, val = 0u, esc_limit = 0u;

  // elements [j - ko / kEs, j) leave the staging area: a full cadence as 16-byte stores
  auto flush = [&]() {
    Elem* const to = dst.ptr() + (pos0 + j - ko / kEs);
    if (ko == static_cast<unsigned int>(L::kOutBytes)) {
#pragma unroll
      for (int c = 0; c < L::kOutBytes / 16; ++c) {
        uint2 v[2];
        v[0] = reinterpret_cast<const uint2*>(outq)[2 * c];
        v[1] = reinterpret_cast<const uint2*>(outq)[2 * c + 1];
        lanes_gstore16(reinterpret_cast<unsigned char*>(to) + 16 * c, v[0], v[1]);
      }
    } else {
      for (unsigned int e = 0; e < ko / kEs; ++e) lanes_gstore_elem(to + e, reinterpret_cast<const Elem*>(outq)[e]);
    }
    ko = 0u;
  };

  for (unsigned int it = 0u; __any(j < elems); ++it) {
    if ((it & (kDecCadence - 1u)) == 0u) {
      // memory phase: park the code bytes requested at the previous phase, request from the current
      // position, store the elements of the last kDecCadence steps
      const unsigned int pos = cw.base + static_cast<unsigned int>(cp - cw.lds);
      cw.commit();
      cp = cw.lds + (pos - cw.base);
      cw.request(pos);
      if (INDEXED) {
        iw.commit();
        iw.request(j * 4u);
      }
      flush();
    }
    if (j < elems) {
      if (mode == 0u) {
        unsigned int dp = dirp;
        if (INDEXED) {
          int t = *reinterpret_cast<const int*>(iw.lds + (j * 4u - iw.base));
          if (t < 0 || t >= la.ntab) {
            atomicMin(first_error, static_cast<unsigned long long>(pos0 + j));
            t = 0;
          }
          dp = 16u * static_cast<unsigned int>(t);
        }
        const uint4 row = *reinterpret_cast<const uint4*>(lanes_lds + dp);   // cdf, limit | escape << 31, bits, cum
        const unsigned int dig = __builtin_bswap16(*reinterpret_cast<const unsigned short*>(cp));
        // ---- symbol first: quotient estimate -> rank among the row's boundaries ------------------
        const float fq = (static_cast<float>(D) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(s1)) * scale;
        const unsigned int q = min(static_cast<unsigned int>(fq), cp_max);
        const unsigned int w = q >> 6;
        const unsigned long long word = *reinterpret_cast<const unsigned long long*>(lanes_lds + row.z + 8u * w);
        const unsigned int cum = lds_u16(lanes_lds, row.w + 2u * w);
        const unsigned long long below = ~0ull >> (63u - (q & 63u));
        unsigned int sym = cum + static_cast<unsigned int>(__popcll(word & below)) - 1u;
        // ---- exact bounds; the reference's search condition A <= D < B verifies the estimate ----
        unsigned int lo = lds_u16(lanes_lds, row.x + 2u * sym);
        unsigned int hi = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
        unsigned int A = scale16(s1, lo);
        unsigned int b = scale16(s1, hi) - 1u;      // B - 1
        b = hi == 0u ? s1 : b;                      // the row's last entry, 2^16, is stored as 0: B = span
        if (__any(D - A > b - A)) {
          // the estimate was one boundary off (~1e-5 of the symbols), or the input is damaged (offset
          // outside the interval: the step is then taken with the clamped symbol)
          const unsigned int nsym = (row.y & 0x7FFFFFFFu) + (row.y >> 31);
          for (int fix = 0; fix < 4; ++fix) {
            if (D - A > b - A) {
              if (D < A) sym = sym > 0u ? sym - 1u : 0u;
              else sym = sym + 1u < nsym ? sym + 1u : nsym - 1u;
              lo = lds_u16(lanes_lds, row.x + 2u * sym);
              hi = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
              A = scale16(s1, lo);
              b = scale16(s1, hi) - 1u;
              b = hi == 0u ? s1 : b;
            }
          }
        }
        // ---- successor state ---------------------------------------------------------------------
        D -= A;
        s1 = b - A;
        const bool ren = (s1 >> 16) == 0u;
        D = ren ? (D << 16) | dig : D;
        s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
        cp += ren ? 2 : 0;
        // ---- the element (written speculatively: it counts only if the cursors advance) ------------
        const bool esc = sym == (row.y ^ 0x80000000u);      // the escape symbol of a row that has one
        *reinterpret_cast<Elem*>(outq + ko) = dst.make(static_cast<int>(dp >> 4), static_cast<int>(sym));
        ko += esc ? 0u : kEs;
        j += esc ? 0u : 1u;
        mode = esc ? 1u : 0u;
        nb = 0u;
        esc_limit = sym;
        if (!INDEXED) {
          const unsigned int nd = dirp + 16u == dir_end ? 0u : dirp + 16u;
          dirp = esc ? dirp : nd;
        }
      } else {
        // ---- one bit of an Elias-gamma escape code (range_coder_kernels.cc:449-471): the uniform
        // binary cdf {0, 1, 2} at precision 1 needs no table ------------------------------------------
        const unsigned int dig = __builtin_bswap16(*reinterpret_cast<const unsigned short*>(cp));
        const unsigned int half = scale16(s1, 32768u);          // B of the first interval
        const unsigned int bit = D >= half ? 1u : 0u;
        const unsigned int A = bit ? half : 0u;
        const unsigned int b = bit ? s1 : half - 1u;
        D -= A;
        s1 = b - A;
        const bool ren = (s1 >> 16) == 0u;
        D = ren ? (D << 16) | dig : D;
        s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
        cp += ren ? 2 : 0;
        bool done = false;
        if (mode == 1u) {
          // unary prefix, bounded so that damaged input cannot spin
          if (bit == 0u) {
            ++nb;
            if (nb == 31u) { val = 1u << 31; mode = 2u; }
          } else {
            val = 1u << nb;
            mode = nb != 0u ? 2u : 3u;
          }
        } else if (mode == 2u) {
          --nb;
          val |= bit << nb;
          if (nb == 0u) mode = 3u;
        } else {
          done = true;
        }
        if (done) {
          const unsigned int dp = INDEXED ? 16u * static_cast<unsigned int>(min(max(*reinterpret_cast<const int*>(iw.lds + (j * 4u - iw.base)), 0), la.ntab - 1)) : dirp;
          const int outv = bit != 0u ? -static_cast<int>(val) : static_cast<int>(val) + static_cast<int>(esc_limit) - 1;
          *reinterpret_cast<Elem*>(outq + ko) = dst.make(static_cast<int>(dp >> 4), outv);
          ko += kEs;
          ++j;
          mode = 0u;
          if (!INDEXED) dirp = dirp + 16u == dir_end ? 0u : dirp + 16u;
        }
      }
    }
  }
  flush();

  if (live) {
    // back to the (base, span - 1, window, digits pulled) form shared with the other kernels
    const unsigned int pos = cw.base + static_cast<unsigned int>(cp - cw.lds);
    const unsigned char* srcp = J.blob + o0;
    unsigned int window = 0u;
    for (int i = -4; i < 0; ++i) {
      const long long q = static_cast<long long>(pos) + i;
      window = (window << 8) | ((q >= 0 && q < static_cast<long long>(len)) ? srcp[q] : 0u);
    }
    J.state[s] = make_uint4(window - D, s1, window, pos >> 1);
  }
}

}  // namespace tfc
