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
//     (range_coder.h:204-222, 249-258); the estimate is off for ~1e-6 of the symbols
//     (tools/lanes_proto.py), which a wave-uniform rare branch corrects by stepping s.
//   The Elias-gamma escape bits are decoded by the same step on a built-in binary row, with a small
//   per-lane mode machine, so a lane that meets an escape falls behind its neighbours instead of
//   stalling them (lanes run their streams at their own pace; the wave ends with its slowest lane).
//
// Encoder: base and span - 1 per lane as in the reference; its delayed-carry bookkeeping (delay_,
// range_coder.cc:167-263) is kept in the form the wave-per-stream encoder uses — the last digit is HELD
// back (plus a count of 0xFFFF digits behind it), because a carry can only ever reach those — which makes
// the common step "add the carry to the held digit, and if the call renormalises, store the held digit
// and hold the new one": one digit per call at most, 26 vector instructions instead of the 40 the delay_
// arithmetic took (profiles/r02_d_sq_lanes.md -> r03).  The finalize kernel maps (held digit, run) to the
// reference's Finalize bytes (enc_tail_one, fast_state); the bytes are identical, tests/ check them
// against the compiled reference for every golden vector.  A 2-byte store at the lane's own cursor per digit.  The slab of a stream is sized on the host from
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

// Workgroups of the lane-per-stream kernels that ran as the FALLBACK of the pipelined kernels (range_pipe.h): a job
// the latter gave up on.  Read by tfc_pipe_counters (tests: the fast path really is the one that ran).
__device__ unsigned long long g_pipe_fallback_blocks;
// (measurement aid, tools/chain_clock_probe.py) core-clock cycles and 100 MHz ticks the first chain workgroup of the
// last pipelined encode / decode launch ran for: the clock the chain ran at
__device__ unsigned long long g_pipe_clock[8];
// (TFC_PIPE_TIMING builds) the encoder chain's hand-scheduled blocks: cycles inside them, their number, blocks repeated
// call by call, cycles of those repetitions
__device__ unsigned long long g_enc_clock[4];

struct LaneArgs {
  const uint32_t* image;       // device copy of the LDS image
  int bytes;                   // bytes of it this kernel needs (encoder: directory + cdf entries)
  int ntab;
  int precision;               // of every row
  unsigned int cap;            // encoder: slab bytes per stream
  int lds_image;               // LDS bytes reserved for the image (multiple of 1024)
  int lds_wave;                // LDS bytes of every wave's private area behind it
  int defer;                   // blocks a wave may run with lanes parked in front of an escape before it codes them
  const unsigned int* guard;   // null, or one flag per job: the job is coded only if its flag is set (fallback of range_pipe.h)
};

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

__device__ inline void lanes_load_image(unsigned char* lds, const LaneArgs& a) {
  const uint4* src = reinterpret_cast<const uint4*>(a.image);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  // eight loads in flight per thread: a one-wave workgroup (the pipelined decoder on config 2's tables: 150 KB by 64
  // threads) otherwise pays a memory latency per 1 KB
  const int n = (a.bytes + 15) / 16, step = static_cast<int>(blockDim.x);
  int i = threadIdx.x;
  for (; i + 7 * step < n; i += 8 * step) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = lanes_gload16(src + i + u * step);
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * step] = v[u];
  }
  for (; i < n; i += step) dst[i] = src[i];
  __syncthreads();
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
#if defined(TFC_PDEC_ABL) && (TFC_PDEC_ABL & 2)
        // (timing experiment, results wrong: one 16-byte load per request instead of WORDS / 2)
        const uint4 v = w == 0 ? lanes_gload16(g + pos) : make_uint4(pend[0].x, pend[0].y, pend[1].x, pend[1].y);
#else
        const uint4 v = lanes_gload16(g + pos + 8u * w);
#endif
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

constexpr unsigned int kLaneDirRepeat = 16;     // directory entries repeated behind its end (tfc_tables_create, kDirRepeat)
constexpr unsigned int kEncCadence = 16;        // steps between memory phases (= steps of the hand-scheduled block)
constexpr unsigned int kEncDigitBytes = 32;     // digit bytes staged per lane between two phases (<= 2 digits per step)

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

// ---- kEncCadence encoder steps, hand-scheduled ------------------------------------------------
// The common case only: every lane codes a plain int32 symbol of a channel-mode row in every step.  Step
// k + 1's table lookups (row -> cdf entries) are issued before step k's interval arithmetic, so the chain
// never waits for LDS.  Interval update and digit bookkeeping per call, all selects (c, r = 0/1):
//   a = (span lo) >> 16, b = ((span hi) >> 16) - 1, bs = base + a (carry c), t1 = b - a, r = t1 < 2^16
//   X = H + c                          the held digit with the carry applied
//   emit X  iff  had & (c | r)         a carry settles the held digit for good; a renormalisation pushes it out
//   r: H = bs >> 16, had = 1 (the new digit is held), base = bs << 16, span - 1 = (t1 << 16) | 0xFFFF
//   !r: had &= !c, base = bs, span - 1 = t1
// The digit is written to the staging area speculatively; the cursor NA only moves when it counts.
// Not covered, FLAG is bumped and the caller repeats the block from the saved lane state with the generic
// steps: escapes and values out of range, a new digit 0xFFFF (it opens a run a later carry may have to
// ripple through) and lanes that come in with such a run (RN != 0, tested by the caller).
// Fixed temporaries v140-v175:
// v[124:139] the block's values, v[148:149] / v[150:151] rows (cdf - 2, info), v[152:153] v[154:155] /
// v[156:157] v[158:159] (lo, 0) (hi, 0) of even / odd steps.
#define TFC_LENC_A(VAL, R0, R1, LO, HI, NEXTROW, NP)                                        \
  NEXTROW                                                                                 \
  "v_and_b32 v171, 0x7fffffff, v" #R1 "\n\t"                                              \
  "v_cmp_ge_u32 " NP(VAL)                                                                 \
  "v_min_u32 v172, v" #VAL ", v171\n\t"                                                   \
  "v_lshl_add_u32 v172, v172, 1, v" #R0 "\n\t"                                            \
  "ds_read_u16 v" #LO ", v172 offset:2\n\t"                                               \
  "ds_read_u16 v" #HI ", v172 offset:4\n\t"
#define TFC_LENC_B(LO, LOH, HI, HIH, PRE)                                                 \
  PRE                                                                                     \
  "v_mad_u64_u32 v[160:161], s[52:53], v" #LO ", %[S], v[" #LO ":" #LOH "]\n\t"             \
  "v_mad_u64_u32 v[162:163], s[52:53], v" #HI ", %[S], v[" #HI ":" #HIH "]\n\t"             \
  "v_alignbit_b32 v160, v161, v160, 16\n\t"                                               \
  "v_alignbit_b32 v162, v163, v162, 16\n\t"                                               \
  "v_add_u32 v162, -1, v162\n\t"                                                          \
  "v_min_u32 v162, v162, %[S]\n\t"                                                        \
  "v_add_co_u32 v164, vcc, %[BASE], v160\n\t"                                             \
  "v_addc_co_u32 v167, vcc, 0, %[H], vcc\n\t"                                             \
  "v_sub_u32 v165, v162, v160\n\t"                                                        \
  "v_sub_u32 v173, v167, %[H]\n\t"                                                        \
  "v_cmp_gt_u32 vcc, %[K64K], v165\n\t"                                                   \
  "v_cndmask_b32 v174, 0, 1, vcc\n\t"                                                     \
  "v_perm_b32 v166, 0, v167, %[PERM]\n\t"                                                 \
  "ds_write_b16 %[NA], v166\n\t"                                                          \
  "v_or_b32 v168, v173, v174\n\t"                                                         \
  "v_and_b32 v168, v168, %[HAD]\n\t"                                                      \
  "v_lshl_add_u32 %[NA], v168, 1, %[NA]\n\t"                                              \
  "v_lshrrev_b32 v169, 16, v164\n\t"                                                      \
  "v_lshlrev_b32 v170, 16, v164\n\t"                                                      \
  "v_lshl_or_b32 v171, v165, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32 %[BASE], v164, v170, vcc\n\t"                                            \
  "v_cndmask_b32 %[S], v165, v171, vcc\n\t"                                               \
  "v_cndmask_b32 %[H], %[H], v169, vcc\n\t"                                               \
  "v_or_b32 v175, %[HAD], v174\n\t"                                                       \
  "v_bfi_b32 %[HAD], v173, v174, v175\n\t"                                                \
  "v_cmp_eq_u32 vcc, %[KFFFF], %[H]\n\t"                                                  \
  "v_addc_co_u32 %[FLAG], vcc, 0, %[FLAG], vcc\n\t"
#define TFC_LENC_ROW_A(OFF) "ds_read_b64 v[148:149], %[DIRP] offset:" #OFF "\n\t"
#define TFC_LENC_ROW_B(OFF) "ds_read_b64 v[150:151], %[DIRP] offset:" #OFF "\n\t"
// A value that is not a plain symbol: (plain variant) bump FLAG, the block is repeated generically /
// (freezing variant) the lane is taken out of this and the following steps and counts the steps it made
#define TFC_LENC_NP_PLAIN(VAL) "vcc, v" #VAL ", v171\n\tv_addc_co_u32 %[FLAG], vcc, 0, %[FLAG], vcc\n\t"
#define TFC_LENC_PRE_PLAIN ""
#define TFC_LENC_NP_FREEZE0(VAL) "s[58:59], v" #VAL ", v171\n\t"
#define TFC_LENC_NP_FREEZE1(VAL) "s[60:61], v" #VAL ", v171\n\t"
#define TFC_LENC_NP_FREEZE2(VAL) "s[62:63], v" #VAL ", v171\n\t"
#define TFC_LENC_PRE_FREEZE0 "s_andn2_b64 exec, exec, s[58:59]\n\tv_add_u32 %[CNT], 1, %[CNT]\n\t"
#define TFC_LENC_PRE_FREEZE1 "s_andn2_b64 exec, exec, s[60:61]\n\tv_add_u32 %[CNT], 1, %[CNT]\n\t"
#define TFC_LENC_PRE_FREEZE2 "s_andn2_b64 exec, exec, s[62:63]\n\tv_add_u32 %[CNT], 1, %[CNT]\n\t"
// generated by tools/gen_lenc_block.py 16: part A of a step two steps ahead of its part B, three (lo, hi) sets;
// the block's kEncCadence values in v[124:139]
#define TFC_LENC_BLOCK(NP0, NP1, NP2, PRE0, PRE1, PRE2)                                        \
  "s_mov_b64 s[56:57], exec\n\t" \
  "ds_read2_b32 v[124:125], %[VP] offset0:0 offset1:1\n\t" \
  "ds_read2_b32 v[126:127], %[VP] offset0:2 offset1:3\n\t" \
  "ds_read2_b32 v[128:129], %[VP] offset0:4 offset1:5\n\t" \
  "ds_read2_b32 v[130:131], %[VP] offset0:6 offset1:7\n\t" \
  "ds_read2_b32 v[132:133], %[VP] offset0:8 offset1:9\n\t" \
  "ds_read2_b32 v[134:135], %[VP] offset0:10 offset1:11\n\t" \
  "ds_read2_b32 v[136:137], %[VP] offset0:12 offset1:13\n\t" \
  "ds_read2_b32 v[138:139], %[VP] offset0:14 offset1:15\n\t" \
  TFC_LENC_ROW_A(0) \
  "v_mov_b32 v153, 0\n\tv_mov_b32 v155, 0\n\tv_mov_b32 v157, 0\n\tv_mov_b32 v159, 0\n\t" \
  "v_mov_b32 v177, 0\n\tv_mov_b32 v179, 0\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  TFC_LENC_A(124, 148, 149, 152, 154, TFC_LENC_ROW_B(16), NP0) \
  "s_waitcnt lgkmcnt(2)\n\t" \
  TFC_LENC_A(125, 150, 151, 156, 158, TFC_LENC_ROW_A(32), NP1) \
  "s_waitcnt lgkmcnt(2)\n\t" \
  TFC_LENC_A(126, 148, 149, 176, 178, TFC_LENC_ROW_B(48), NP2) \
  "s_waitcnt lgkmcnt(6)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(127, 150, 151, 152, 154, TFC_LENC_ROW_A(64), NP0) \
  "s_waitcnt lgkmcnt(7)\n\t" \
  TFC_LENC_B(156, 157, 158, 159, PRE1) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(128, 148, 149, 156, 158, TFC_LENC_ROW_B(80), NP1) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(176, 177, 178, 179, PRE2) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(129, 150, 151, 176, 178, TFC_LENC_ROW_A(96), NP2) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(130, 148, 149, 152, 154, TFC_LENC_ROW_B(112), NP0) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(156, 157, 158, 159, PRE1) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(131, 150, 151, 156, 158, TFC_LENC_ROW_A(128), NP1) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(176, 177, 178, 179, PRE2) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(132, 148, 149, 176, 178, TFC_LENC_ROW_B(144), NP2) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(133, 150, 151, 152, 154, TFC_LENC_ROW_A(160), NP0) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(156, 157, 158, 159, PRE1) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(134, 148, 149, 156, 158, TFC_LENC_ROW_B(176), NP1) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(176, 177, 178, 179, PRE2) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(135, 150, 151, 176, 178, TFC_LENC_ROW_A(192), NP2) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(136, 148, 149, 152, 154, TFC_LENC_ROW_B(208), NP0) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(156, 157, 158, 159, PRE1) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(137, 150, 151, 156, 158, TFC_LENC_ROW_A(224), NP1) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(176, 177, 178, 179, PRE2) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(138, 148, 149, 176, 178, TFC_LENC_ROW_B(240), NP2) \
  "s_waitcnt lgkmcnt(8)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_waitcnt lgkmcnt(3)\n\t" \
  TFC_LENC_A(139, 150, 151, 152, 154, "", NP0) \
  "s_waitcnt lgkmcnt(7)\n\t" \
  TFC_LENC_B(156, 157, 158, 159, PRE1) \
  "s_waitcnt lgkmcnt(4)\n\t" \
  TFC_LENC_B(176, 177, 178, 179, PRE2) \
  "s_waitcnt lgkmcnt(2)\n\t" \
  TFC_LENC_B(152, 153, 154, 155, PRE0) \
  "s_mov_b64 exec, s[56:57]\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t"

template <bool INDEXED, typename Src>
__global__ void __launch_bounds__(512) enc_lanes_kernel(const EncLaneJobs<Src> jobs, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  if (la.guard) {
    if (la.guard[blockIdx.x / static_cast<unsigned int>(jobs.blocks_per_job)] == 0u) return;
    if (threadIdx.x == 0) atomicAdd(&g_pipe_fallback_blocks, 1ull);
  }
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
  // (base, span - 1, held digit | valid << 31, 0xFFFF digits held behind it): the wave-per-stream fast
  // encoder's state form (range_encoder_fast.h, FastEncState)
  unsigned int base = st.x, s1 = st.y, hd = st.z & 0xFFFFu, had = st.z >> 31, rn = st.w;
  lanes_pin(base, s1, hd, rn);

  // the kernel's dynamic LDS starts at LDS address lds0 (0 unless static LDS ever gets added)
  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) unsigned char*)lanes_lds));
  const unsigned int wave_off = static_cast<unsigned int>(la.lds_image) + (threadIdx.x >> 6) * static_cast<unsigned int>(la.lds_wave);
  const unsigned int ds_off = wave_off + L::kDigits + L::kDigitStride * lane;      // this lane's digit bytes
  unsigned char* const dstage = lanes_lds + ds_off;
  unsigned char* const out = J.chunk + (live ? s : 0) * static_cast<int64_t>(la.cap);
  unsigned int wpos = 0u;          // slab bytes written by earlier phases
  unsigned int n = 0u;             // digit bytes produced since
  unsigned int overflow = 0u;
  LaneWindow<L::kValueWords> vw;
  const unsigned int vw_off = wave_off + L::kValue + L::kValueStride * lane;
  vw.lds = lanes_lds + vw_off;
  vw.g = reinterpret_cast<const unsigned char*>(src.base() + pos0);
  vw.len = elems * kRaw;
  vw.request(0u);
  LaneWindow<L::kIndexWords> iw;
  if (INDEXED) {
    iw.lds = lanes_lds + wave_off + L::kIndex + L::kIndexStride * lane;
    iw.g = reinterpret_cast<const unsigned char*>(index + pos0);
    iw.len = elems * 4u;
    iw.request(0u);
  }

  const unsigned int dir_end = 16u * static_cast<unsigned int>(la.ntab);
  const unsigned int dir_step = (16u * kEncCadence) % dir_end;
  unsigned int j = 0u;              // next symbol to take
  unsigned int dirp = 0u;           // channel mode: LDS offset of its directory entry
  unsigned int qn = 0u, g = 0u, neg = 0u;   // escape bits still to code: qn of them, from g then the sign
  unsigned int saw_exception = 0u;          // this lane met a value that is not a plain symbol

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

  // one coder call on [lo, hi) / 2^16 for the lanes with `act`
  auto call = [&](unsigned int lo, unsigned int hi, bool act) __attribute__((always_inline)) {
    // ---- RangeEncoder::Encode (range_coder.cc:37-264) on [lo, hi) / 2^16 with the delayed digits kept as
    // (held digit hd, had, run rn of 0xFFFF digits behind it) — the single-call case of consume_calls() in
    // range_encoder_fast.h, every update a select on `act` --------------------------------------------
    const unsigned int a = scale16(s1, lo);
    const unsigned int b = scale16(s1, hi) - 1u;
    const unsigned int bs = base + a;
    const unsigned int t1 = b - a;
    const bool carry = act && bs < a;                   // base + a left 32 bits: +1 into the held digits
    const bool ren = act && (t1 >> 16) == 0u;
    const unsigned int e = bs >> 16;                    // the digit a renormalisation shifts out
    const bool solid = ren && e != 0xFFFFu;
    const bool ffff = ren && e == 0xFFFFu;
    const bool held = had != 0u;
    const unsigned int X = (hd + (carry ? 1u : 0u)) & 0xFFFFu;
    // X leaves when a digit that cannot pass a carry on arrives behind it, or when the carry has arrived
    // (a new 0xFFFF digit directly behind a held digit without a run only joins it: [X][FFFF])
    const bool emit = held && (solid || (carry && (!ffff || rn != 0u)));
    put(X, emit);
    // ... followed by the run behind it: 0xFFFF digits as they are, or 0x0000 if the carry went through
    // them (with a new 0xFFFF digit arriving, the last of them stays held as 0x0000)
    const unsigned int run_out = !held ? 0u : solid ? rn : (carry && rn != 0u) ? (ffff ? rn - 1u : rn) : 0u;
    if (__any(run_out != 0u)) {
      if (run_out != 0u) put_run(carry ? 0u : 0xFFFFu, 2u * run_out);
    }
    if (solid) {
      hd = e; had = 1u; rn = 0u;
    } else if (ffff) {
      if (!held) { hd = 0xFFFFu; had = 1u; rn = 0u; }
      else if (!carry) { rn += 1u; }
      else if (rn == 0u) { hd = X; rn = 1u; }
      else { hd = 0u; rn = 1u; }
    } else if (carry && held) {
      had = 0u; rn = 0u;
    }
    base = act ? (ren ? bs << 16 : bs) : base;
    s1 = act ? (ren ? (t1 << 16) | 0xFFFFu : t1) : s1;
  };

  // one generic step: any lane state
  auto step = [&]() {
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
    const uint2 row = *reinterpret_cast<const uint2*>(lanes_lds + dp);   // cdf offset - 2, limit | escape << 31
    const unsigned int limit = row.y & 0x7FFFFFFFu;                      // first value that is not a plain symbol
    const bool take = qn == 0u && j < elems;
    const bool plain = static_cast<unsigned int>(v) < limit;             // negative values are not
    unsigned int sym = plain ? static_cast<unsigned int>(v) : limit;
    bool act = take;                 // this lane makes a coder call in this step
    const bool adv = take;           // ... and moves on to the next symbol
    unsigned int lo = 0u, hi = 0u;
    if (__any(!(take && plain) && (qn != 0u || j < elems))) {
      if (take && !plain) {
        saw_exception = 1u;
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
      const unsigned int tlo = lds_u16(lanes_lds, row.x + 2u * sym + 2u);       // row.x = offset of cdf[0] - 2
      const unsigned int thi = lds_u16(lanes_lds, row.x + 2u * sym + 4u);
      lo = take ? tlo : lo;
      hi = take ? (thi == 0u ? 65536u : thi) : hi;
    }
    j += adv ? 1u : 0u;
    if (!INDEXED) {
      const unsigned int nd = dirp + 16u == dir_end ? 0u : dirp + 16u;
      dirp = adv ? nd : dirp;
    }
    call(lo, hi, act);
  };

  // The main loop exists twice: with the plain block until the wave meets its first value that is not a
  // plain symbol, with the freezing block (~5 % slower per step) from there on, so that each of the two
  // hot loops is one contiguous piece of code.  Only such a value switches loops: the other exceptions
  // (a long carry run here, a wrong rank estimate in the decoder) happen in every stream now and then.
  constexpr bool kFastBlock = !INDEXED && std::is_same<Src, SymInt32>::value;
  auto run = [&](auto freeze_tag) __attribute__((always_inline)) {
  constexpr bool kFreeze = decltype(freeze_tag)::value;
  int deferred = 0;                 // blocks since the stopped lanes' escape codes were last taken
  while (__any(j < elems || qn != 0u)) {
    {
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
    const bool busy = j < elems || qn != 0u;
    if (__builtin_expect(kFastBlock && lds0 == 0u && !__any(busy && (j + kEncCadence > elems || qn != 0u)), 1)) {
      const unsigned int base0 = base, s10 = s1, hd0 = hd, had0 = had;
      unsigned int flag = rn, na = ds_off, cnt = 0u;           // a lane inside a 0xFFFF run: generic steps
      if (busy) {
        const unsigned int vp = vw_off + (j * 4u - vw.base);
        // the plain block until this wave has met its first exception, the freezing one afterwards
#define TFC_LENC_OPERANDS                                                                                          \
                     : [BASE] "+v"(base), [S] "+v"(s1), [H] "+v"(hd), [HAD] "+v"(had), [NA] "+v"(na), [FLAG] "+v"(flag), \
                       [CNT] "+v"(cnt)                                                                                \
                     : [VP] "v"(vp), [DIRP] "v"(dirp), [DIR0] "v"(0u), [K64K] "s"(0x10000u),                          \
                       [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x0c0c0001u)                                                  \
                     : "vcc", "memory", "s52", "s53", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "v176", "v177", \
                       "v178", "v179", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", \
                       "v136", "v137", "v138", "v139", "v140", "v141", "v142", \
                       "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", \
                       "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", \
                       "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175"
        if constexpr (kFreeze) {
          asm volatile(TFC_LENC_BLOCK(TFC_LENC_NP_FREEZE0, TFC_LENC_NP_FREEZE1, TFC_LENC_NP_FREEZE2,
                                      TFC_LENC_PRE_FREEZE0, TFC_LENC_PRE_FREEZE1, TFC_LENC_PRE_FREEZE2)
                       TFC_LENC_OPERANDS);
        } else {
          asm volatile(TFC_LENC_BLOCK(TFC_LENC_NP_PLAIN, TFC_LENC_NP_PLAIN, TFC_LENC_NP_PLAIN,
                                      TFC_LENC_PRE_PLAIN, TFC_LENC_PRE_PLAIN, TFC_LENC_PRE_PLAIN)
                       TFC_LENC_OPERANDS);
          cnt = kEncCadence;
        }
#undef TFC_LENC_OPERANDS
      }
      if (__builtin_expect(!__any(flag != 0u), 1)) {
        if (__builtin_expect(!__any(busy && cnt != kEncCadence), 1)) {
          if (busy) {
            j += kEncCadence;
            n = na - ds_off;
            dirp += dir_step;                          // kEncCadence entries further, modulo the table count
            dirp -= dirp >= dir_end ? dir_end : 0u;
          }
          continue;
        }
        // Some lanes stopped in front of a value that is not a plain symbol, after `cnt` steps.  Its
        // calls now — the row's escape symbol, then the bits of the Elias-gamma code — in a loop of bare
        // coder calls while the other lanes wait, as long as the digit area has room for two more digits
        // (a code that does not fit goes on in the generic steps behind the next memory phase).
        if (busy) {
          j += cnt;
          n = na - ds_off;
          dirp += 16u * cnt;
        }
        while (__any(busy && dirp >= dir_end)) dirp -= dirp >= dir_end ? dir_end : 0u;
        // The escape codes of the stopped lanes cost the whole wave a loop of ~10 rounds; a stopped lane simply
        // stops again at step 0 of the next block (its value is still not a plain symbol).  So the codes are
        // taken together, every la.defer-th block, as long as somebody still gets a whole block done.
        if (deferred + 1 < la.defer && __any(busy && cnt == kEncCadence)) {
          ++deferred;
          continue;
        }
        deferred = 0;
        const bool stopped = busy && cnt != kEncCadence;
        unsigned int lo = 0u, hi = 0u;
        if (stopped) {
          const int32_t v = *reinterpret_cast<const int32_t*>(vw.lds + (j * 4u - vw.base));
          const uint2 row = *reinterpret_cast<const uint2*>(lanes_lds + dirp);
          const unsigned int limit = row.y & 0x7FFFFFFFu;
          unsigned int sym = limit;
          if (row.y >> 31) {
            neg = v < 0 ? 1u : 0u;
            g = v < 0 ? 0u - static_cast<unsigned int>(v) : static_cast<unsigned int>(v) - limit + 1u;
            qn = 2u * static_cast<unsigned int>(31 - __clz(static_cast<int>(g))) + 2u;
          } else {
            atomicMin(first_error, static_cast<unsigned long long>(pos0 + j));
            sym = 0u;
          }
          lo = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
          hi = lds_u16(lanes_lds, row.x + 2u * sym + 4u);
          hi = hi == 0u ? 65536u : hi;
          ++j;
          dirp = dirp + 16u == dir_end ? 0u : dirp + 16u;
        }
        call(lo, hi, stopped);
        // The bits: one call per round on [0, 2^15) or [2^15, 2^16), i.e. a = 0 / half, b = half - 1 / s1 with
        // half = (s1 + 1) >> 1; the interval update is the block's (TFC_LENC_B).  Hand-written: the compiler's
        // version of this loop takes about twice the instructions.  A round is computed into temporaries and
        // committed under EXEC: a lane whose round would shift out a 0xFFFF digit (it opens a run) drops out
        // BEFORE the commit and takes that bit in the generic steps; EXEC also drops a lane when its code is
        // complete or its digit area has no room for another digit.  A lane inside a run does not enter.
        if (qn != 0u && n + 4u <= kEncDigitBytes && rn == 0u) {
          unsigned int na2 = ds_off + n;
          asm volatile(
              "s_mov_b64 s[56:57], exec\n\t"
              "v_mov_b32 v108, %[G]\n\t"
              "v_mov_b32 v109, 0\n\t"
              "1:\n\t"
              "v_add_u32 v106, -1, %[QN]\n\t"                        // qn - 1 (committed below)
              "v_add_u32 v110, -1, v106\n\t"
              "v_lshrrev_b64 v[110:111], v110, v[108:109]\n\t"       // bit qn - 2 of g (0 from bit 32 on)
              "v_and_b32 v110, 1, v110\n\t"
              "v_cmp_eq_u32 vcc, 0, v106\n\t"
              "v_cndmask_b32 v110, v110, %[NEG], vcc\n\t"            // ... the sign at the end
              "v_lshrrev_b32 v101, 1, %[S]\n\t"
              "v_and_b32 v102, 1, %[S]\n\t"
              "v_add_u32 v101, v101, v102\n\t"                       // half
              "v_cmp_ne_u32 vcc, 0, v110\n\t"
              "v_cndmask_b32 v160, 0, v101, vcc\n\t"                 // a
              "v_add_u32 v162, -1, v101\n\t"
              "v_cndmask_b32 v162, v162, %[S], vcc\n\t"              // b
              "v_add_co_u32 v164, vcc, %[BASE], v160\n\t"            // bs, carry
              "v_addc_co_u32 v167, vcc, 0, %[H], vcc\n\t"            // X = H + c
              "v_sub_u32 v165, v162, v160\n\t"                       // t1
              "v_sub_u32 v173, v167, %[H]\n\t"                       // c
              "v_lshrrev_b32 v169, 16, v164\n\t"                     // e
              "v_cmp_gt_u32 vcc, %[K64K], v165\n\t"                  // r
              "v_cmp_eq_u32 s[54:55], %[KFFFF], v169\n\t"
              "s_and_b64 s[54:55], s[54:55], vcc\n\t"                // r & e == 0xFFFF: not in here
              "s_andn2_b64 exec, exec, s[54:55]\n\t"
              "v_cndmask_b32 v174, 0, 1, vcc\n\t"
              "v_perm_b32 v166, 0, v167, %[PERM]\n\t"
              "ds_write_b16 %[NA], v166\n\t"
              "v_or_b32 v168, v173, v174\n\t"
              "v_and_b32 v168, v168, %[HAD]\n\t"
              "v_lshl_add_u32 %[NA], v168, 1, %[NA]\n\t"
              "v_lshlrev_b32 v170, 16, v164\n\t"
              "v_lshl_or_b32 v171, v165, 16, %[KFFFF]\n\t"
              "v_cndmask_b32 %[BASE], v164, v170, vcc\n\t"
              "v_cndmask_b32 %[S], v165, v171, vcc\n\t"
              "v_cndmask_b32 %[H], %[H], v169, vcc\n\t"
              "v_or_b32 v175, %[HAD], v174\n\t"
              "v_bfi_b32 %[HAD], v173, v174, v175\n\t"
              "v_mov_b32 %[QN], v106\n\t"
              "v_cmp_ne_u32 s[54:55], 0, %[QN]\n\t"
              "v_sub_u32 v107, %[NA], %[DSOFF]\n\t"
              "v_cmp_ge_u32 vcc, %[ROOM], v107\n\t"
              "s_and_b64 s[54:55], s[54:55], vcc\n\t"
              "s_and_b64 exec, exec, s[54:55]\n\t"
              "s_cbranch_execnz 1b\n\t"
              "s_mov_b64 exec, s[56:57]\n\t"
              "s_waitcnt lgkmcnt(0)\n\t"
              : [BASE] "+v"(base), [S] "+v"(s1), [H] "+v"(hd), [HAD] "+v"(had), [NA] "+v"(na2), [QN] "+v"(qn)
              : [G] "v"(g), [NEG] "v"(neg), [DSOFF] "v"(ds_off), [ROOM] "s"(kEncDigitBytes - 4u),
                [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x0c0c0001u)
              : "vcc", "memory", "s54", "s55", "s56", "s57", "v101", "v102", "v106", "v107", "v108", "v109", "v110",
                "v111", "v160", "v162", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v173", "v174",
                "v175");
          n = na2 - ds_off;
        }
        continue;
      }
      base = base0; s1 = s10; hd = hd0; had = had0;                 // an exception somewhere in the wave (rn is not touched by the block)
    }
#pragma nounroll
    for (unsigned int k = 0; k < kEncCadence; ++k) step();
    if (kFastBlock && !kFreeze && __any(saw_exception != 0u)) return;      // such values are to be expected from here on
  }
  };
  run(std::false_type{});
  if constexpr (kFastBlock) run(std::true_type{});
  flush();
  if (live) {
    J.state[s] = make_uint4(base, s1, (hd & 0xFFFFu) | (had << 31), rn);
    // (past the slab's end wpos only counts, nothing is stored: the piece gets length 0 so that finalize packs
    // nothing from behind the slab — the handle is flagged, and the caller codes it again)
    J.chunk_len[s] = overflow ? 0u : wpos;
    if (overflow) atomicOr(J.overflow_flag, 1u);
  }
}

// ---------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------

constexpr unsigned int kDecCadence = 16;
static_assert(kEncCadence <= kLaneDirRepeat && kDecCadence <= kLaneDirRepeat, "a block reads cadence consecutive directory entries");

// LDS of one decoder wave: per lane the code-byte window (a step consumes <= 2 bytes), the decoded
// elements of one cadence, and the index window.
template <typename Elem>
struct DecWaveLds {
  // two cadences of digits: the window parked at a memory phase starts at the position of the phase
  // before it, and between two phases a lane consumes at most kDecCadence digits
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

// ---- kDecCadence decoder steps, hand-scheduled ------------------------------------------------
// The common case only: every lane decodes a plain symbol of a channel-mode row in every step.  A lone
// wave pays ~4 cycles per instruction of any kind, so the block is written for instruction count
// (~55 per step against ~100 from the compiler) and for LDS latency: the directory entry of step k + 1
// and the code digit are requested before the quotient arithmetic of step k.  Anything else — an
// estimate that the verification rejects, damaged input — only bumps FLAG; the caller then restores the
// lane state it saved and repeats the block with the generic steps.  A lane that decodes an ESCAPE symbol
// takes that step's state update and then sits out the rest of the block (EXEC): CNT is the number of
// symbols a lane completed; the caller decodes the escape's bits in a short loop of its own and the lane
// carries on, a few symbols behind its neighbours.
// LDS operands are absolute LDS addresses (the kernel's dynamic LDS starts at 0, checked by the
// caller).  Temporaries are the fixed registers v100-v132 (register pairs and the 4-register row
// buffers need known numbers): v[100:103] / v[104:107] rows (cdf - 2, info, bits, cum), v[120:121] = -1,
// v[122:123] = (lo, 0), v[124:125] = (hi, 0).
#define TFC_LDEC_STEP(ROW0, ROW1, ROW2, ROW3, PREFETCH, OUTOFF, ESC1, ESC2)                \
  PREFETCH                                                                                \
  "ds_read_u16 v109, %[CP]\n\t"                                                           \
  "v_cvt_f32_u32 v111, %[S]\n\t"                                                          \
  "v_add_f32 v111, 1.0, v111\n\t"                                                         \
  "v_rcp_f32 v111, v111\n\t"                                                              \
  "v_cvt_f32_u32 v110, %[D]\n\t"                                                          \
  "v_fma_f32 v110, v110, %[SCALE], %[HSCALE]\n\t"                                         \
  "v_mul_f32 v110, v110, v111\n\t"                                                        \
  "v_cvt_u32_f32 v110, v110\n\t"                                                          \
  "v_min_u32 v110, %[QMAX], v110\n\t"                                                     \
  "v_lshrrev_b32 v111, 6, v110\n\t"                                                       \
  "v_lshl_add_u32 v112, v111, 3, v" #ROW2 "\n\t"                                          \
  "v_lshl_add_u32 v113, v111, 1, v" #ROW3 "\n\t"                                          \
  "ds_read_b64 v[114:115], v112\n\t"                                                      \
  "ds_read_i16 v116, v113\n\t"                                                            \
  "v_not_b32 v110, v110\n\t"                                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                              \
  "v_lshlrev_b64 v[118:119], v110, v[114:115]\n\t"                                        \
  "v_bcnt_u32_b32 v116, v118, v116\n\t"                                                   \
  "v_bcnt_u32_b32 v117, v119, v116\n\t"                                                   \
  "v_lshl_add_u32 v112, v117, 1, v" #ROW0 "\n\t"                                          \
  "ds_read_u16 v122, v112 offset:2\n\t"                                                   \
  "ds_read_u16 v124, v112 offset:4\n\t"                                                   \
  "v_xad_u32 v113, v117, v" #ROW1 ", %[K31]\n\t"                                          \
  ESC1                                                                                    \
  "ds_write_b32 %[OQ], v117 offset:" #OUTOFF "\n\t"                                       \
  "v_perm_b32 v109, 0, v109, %[PERM]\n\t"                                                 \
  "s_waitcnt lgkmcnt(1)\n\t"                                                              \
  "v_mad_u64_u32 v[126:127], vcc, v122, %[S], v[122:123]\n\t"                             \
  "v_mad_u64_u32 v[128:129], vcc, v124, %[S], v[124:125]\n\t"                             \
  "v_alignbit_b32 v126, v127, v126, 16\n\t"                                               \
  "v_alignbit_b32 v128, v129, v128, 16\n\t"                                               \
  "v_add_u32 v128, -1, v128\n\t"                                                          \
  "v_min_u32 v128, v128, %[S]\n\t"                                                        \
  "v_sub_u32 v130, %[D], v126\n\t"                                                        \
  "v_sub_u32 v131, v128, v126\n\t"                                                        \
  "v_cmp_gt_u32 vcc, v130, v131\n\t"                                                      \
  "v_addc_co_u32 %[FLAG], vcc, 0, %[FLAG], vcc\n\t"                                       \
  "v_cmp_gt_u32 vcc, %[K64K], v131\n\t"                                                   \
  "v_lshl_or_b32 v132, v130, 16, v109\n\t"                                                \
  "v_cndmask_b32 %[D], v130, v132, vcc\n\t"                                               \
  "v_lshl_or_b32 v132, v131, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32 %[S], v131, v132, vcc\n\t"                                               \
  "v_cndmask_b32 v132, 0, 2, vcc\n\t"                                                     \
  "v_add_u32 %[CP], %[CP], v132\n\t"                                                      \
  ESC2
// an escape symbol: (plain variant) bump FLAG, the whole block is repeated generically / (freezing variant)
// remember the lanes, let them finish this step, then take them out of the rest of the block
#define TFC_LDEC_ESC1_PLAIN "v_min_u32 %[ACC], %[ACC], v113\n\t"
#define TFC_LDEC_ESC2_PLAIN ""
#define TFC_LDEC_ESC1_FREEZE "v_cmp_eq_u32 s[54:55], 0, v113\n\t"
#define TFC_LDEC_ESC2_FREEZE "s_andn2_b64 exec, exec, s[54:55]\n\tv_add_u32 %[CNT], 1, %[CNT]\n\t"
#define TFC_LDEC_READ_A(OFF) "ds_read_b128 v[100:103], %[DIRP] offset:" #OFF "\n\t"
#define TFC_LDEC_READ_B(OFF) "ds_read_b128 v[104:107], %[DIRP] offset:" #OFF "\n\t"
// kDecCadence steps (rows alternate between the two 4-register buffers, the next row is requested a step ahead)
#define TFC_LDEC_PARK_FREEZE "s_andn2_b64 exec, exec, %[PARK]\n\t"
#define TFC_LDEC_BLOCK(ESC1, ESC2, PARKPRE)                                               \
  "s_mov_b64 s[56:57], exec\n\t"                                                          \
  PARKPRE                                                                                 \
  "v_mov_b32 v123, 0\n\tv_mov_b32 v125, 0\n\t"                                           \
  TFC_LDEC_READ_A(0) "s_waitcnt lgkmcnt(0)\n\t"                                           \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(16), 0, ESC1, ESC2)                   \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(32), 4, ESC1, ESC2)                   \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(48), 8, ESC1, ESC2)                   \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(64), 12, ESC1, ESC2)                  \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(80), 16, ESC1, ESC2)                  \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(96), 20, ESC1, ESC2)                  \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(112), 24, ESC1, ESC2)                 \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(128), 28, ESC1, ESC2)                 \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(144), 32, ESC1, ESC2)                 \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(160), 36, ESC1, ESC2)                 \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(176), 40, ESC1, ESC2)                 \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(192), 44, ESC1, ESC2)                 \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(208), 48, ESC1, ESC2)                 \
  TFC_LDEC_STEP(104, 105, 106, 107, TFC_LDEC_READ_A(224), 52, ESC1, ESC2)                 \
  TFC_LDEC_STEP(100, 101, 102, 103, TFC_LDEC_READ_B(240), 56, ESC1, ESC2)                 \
  TFC_LDEC_STEP(104, 105, 106, 107, "", 60, ESC1, ESC2)                                   \
  "s_mov_b64 exec, s[56:57]\n\t"                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"

template <bool INDEXED, typename Dst>
__global__ void __launch_bounds__(512) dec_lanes_kernel(const DecLaneJobs<Dst> jobs, LaneArgs la) {
  extern __shared__ unsigned char lanes_lds[];
  if (la.guard) {
    if (la.guard[blockIdx.x / static_cast<unsigned int>(jobs.blocks_per_job)] == 0u) return;
    if (threadIdx.x == 0) atomicAdd(&g_pipe_fallback_blocks, 1ull);
  }
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

  // the kernel's dynamic LDS starts at LDS address lds0 (0 unless static LDS ever gets added)
  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) unsigned char*)lanes_lds));
  const unsigned int wave_off = static_cast<unsigned int>(la.lds_image) + (threadIdx.x >> 6) * static_cast<unsigned int>(la.lds_wave);
  LaneWindow<L::kCodeWords> cw;
  const unsigned int cw_off = wave_off + L::kCodes + L::kCodeStride * lane;
  cw.lds = lanes_lds + cw_off;
  cw.g = J.blob + o0;
  cw.len = len;
  cw.request(pos_start);
  cw.base = pos_start;
  unsigned int cp = cw_off;              // LDS offset of the next code digit: stream position cw.base + (cp - cw_off)
  const unsigned int oq_off = wave_off + L::kOut + L::kOutStride * lane;
  unsigned char* const outq = lanes_lds + oq_off;
  unsigned int ko = 0u;                  // bytes of decoded elements waiting in outq
  LaneWindow<L::kIndexWords> iw;
  if (INDEXED) {
    iw.lds = lanes_lds + wave_off + L::kIndex + L::kIndexStride * lane;
    iw.g = reinterpret_cast<const unsigned char*>(index + pos0);
    iw.len = elems * 4u;
    iw.request(0u);
  }

  const float scale = static_cast<float>(1u << la.precision);     // quotient scale: 2^precision
  // The estimate is the float image of the exact quotient (D + 1) 2^p / (S + 1) — the symbol is the rank of its ceiling
  // minus one (range_pipe.h, "Round 5, second pass": it lands across a boundary for ~1e-6 of the steps; (D + 1/2) 2^p / S,
  // rounds 2 - 4, for 3e-5, a block of generic steps each time) — so the addend of the numerator is the scale itself.
  float hscale = scale;                                           // (in a vector register for the block)
  unsigned int k31 = 0x80000000u;
  asm volatile("" : "+v"(hscale), "+v"(k31));
  const unsigned int cp_max = (1u << la.precision) - 1u;
  const unsigned int dir_end = 16u * static_cast<unsigned int>(la.ntab);
  const unsigned int dir_step = (16u * kDecCadence) % dir_end;
  unsigned int j = 0u;
  unsigned int dirp = 0u;            // channel mode: LDS offset of the directory entry of symbol j
  unsigned int mode = 0u;            // 0 symbol, 1 unary prefix, 2 payload bits, 3 sign
  unsigned int nb = 0u, val = 0u, esc_limit = 0u;
  unsigned int saw_escape = 0u;      // this lane met an escape symbol

  // elements [j - ko / kEs, j) leave the staging area: a full cadence as 16-byte stores
  auto flush = [&]() {
    Elem* const to = dst.ptr() + (pos0 + j - ko / kEs);
    if (__builtin_expect(ko == static_cast<unsigned int>(L::kOutBytes), 1)) {
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

  // one bit of an Elias-gamma escape code, for the lanes that are inside one (mode != 0)
  auto bit_step = [&]() {
    if (j < elems && mode != 0u) {
        // ---- one bit of an Elias-gamma escape code (range_coder_kernels.cc:449-471): the uniform
        // binary cdf {0, 1, 2} at precision 1 needs no table ------------------------------------------
        const unsigned int dig = __builtin_bswap16(*reinterpret_cast<const unsigned short*>(lanes_lds + cp));
        const unsigned int half = scale16(s1, 32768u);          // B of the first interval
        const unsigned int bit = D >= half ? 1u : 0u;
        const unsigned int A = bit ? half : 0u;
        const unsigned int b = bit ? s1 : half - 1u;
        D -= A;
        s1 = b - A;
        const bool ren = (s1 >> 16) == 0u;
        D = ren ? (D << 16) | dig : D;
        s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
        cp += ren ? 2u : 0u;
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
  };

  // one generic step: any mode, any lane state
  auto step = [&]() {
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
        const uint4 row = *reinterpret_cast<const uint4*>(lanes_lds + dp);   // cdf - 2, limit | escape << 31, bits, cum
        const unsigned int dig = __builtin_bswap16(*reinterpret_cast<const unsigned short*>(lanes_lds + cp));
        // ---- symbol first: quotient estimate -> rank among the row's boundaries ------------------
        const float fq = (static_cast<float>(D) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(s1)) * scale;
        const unsigned int q = min(static_cast<unsigned int>(fq), cp_max);
        const unsigned int w = q >> 6;
        const unsigned long long word = *reinterpret_cast<const unsigned long long*>(lanes_lds + row.z + 8u * w);
        // boundaries before word w, minus one (int16): symbol = that + the boundaries up to q in the word
        const int cum_m1 = *reinterpret_cast<const short*>(lanes_lds + row.w + 2u * w);
        const unsigned long long below = ~0ull >> (63u - (q & 63u));
        unsigned int sym = static_cast<unsigned int>(cum_m1 + __popcll(word & below));
        // ---- exact bounds; the reference's search condition A <= D < B verifies the estimate ----
        unsigned int lo = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
        unsigned int hi = lds_u16(lanes_lds, row.x + 2u * sym + 4u);
        unsigned int A = scale16(s1, lo);
        unsigned int b = scale16(s1, hi) - 1u;      // B - 1
        b = hi == 0u ? s1 : b;                      // the row's last entry, 2^16, is stored as 0: B = span
        if (__any(D - A > b - A)) {
          // the estimate was one boundary off (~1e-6 of the symbols), or the input is damaged (offset
          // outside the interval: the step is then taken with the clamped symbol)
          const unsigned int nsym = (row.y & 0x7FFFFFFFu) + (row.y >> 31);
          for (int fix = 0; fix < 4; ++fix) {
            if (D - A > b - A) {
              if (D < A) sym = sym > 0u ? sym - 1u : 0u;
              else sym = sym + 1u < nsym ? sym + 1u : nsym - 1u;
              lo = lds_u16(lanes_lds, row.x + 2u * sym + 2u);
              hi = lds_u16(lanes_lds, row.x + 2u * sym + 4u);
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
        cp += ren ? 2u : 0u;
        // ---- the element (written speculatively: it counts only if the cursors advance) ------------
        const bool esc = sym == (row.y ^ 0x80000000u);      // the escape symbol of a row that has one
        saw_escape |= esc ? 1u : 0u;
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
        bit_step();
      }
    }
  };

  // the main loop twice, as in the encoder: plain block until the first escape symbol, freezing block after it
  constexpr bool kFastBlock = !INDEXED && std::is_same<Dst, OutInt32>::value;
  auto run = [&](auto freeze_tag) __attribute__((always_inline)) {
  constexpr bool kFreeze = decltype(freeze_tag)::value;
  int deferred = 0;                 // blocks since the parked lanes' escape codes were last read
  while (__any(j < elems)) {
    {
      // memory phase: park the code bytes requested at the previous phase, request from the current
      // position, store the elements of the last kDecCadence steps
      const unsigned int pos = cw.base + (cp - cw_off);
      cw.commit();
      cp = cw_off + (pos - cw.base);
      cw.request(pos);
      if (INDEXED) {
        iw.commit();
        iw.request(j * 4u);
      }
      flush();
    }
    // A lane that has decoded an escape symbol and not yet a bit of its code (mode 1, no zeros seen) may sit
    // out blocks ("parked": the freezing block takes it out of EXEC at its top); any other lane inside a
    // code goes through the generic steps.
    if (__builtin_expect(kFastBlock && lds0 == 0u &&
                         !__any(j < elems && (j + kDecCadence > elems ||
                                              (mode != 0u && (!kFreeze || mode != 1u || nb != 0u)))), 1)) {
      const unsigned long long park = __ballot(j < elems && mode != 0u);
      const unsigned int D0 = D, s10 = s1, cp0 = cp;
      unsigned int flag = 0u, cnt = 0u;
      unsigned int acc = 0xFFFFFFFFu;      // plain block: min over the steps of (symbol ^ row info) + 2^31, 0 = an escape symbol
      if (j < elems) {
        // the plain block until this wave has met its first escape symbol, the freezing one afterwards
        // (the freeze costs ~10 % per step: a scalar EXEC update behind a vector compare, and the count)
#define TFC_LDEC_OPERANDS                                                                                       \
                     : [D] "+v"(D), [S] "+v"(s1), [CP] "+v"(cp), [FLAG] "+v"(flag), [CNT] "+v"(cnt), [ACC] "+v"(acc) \
                     : [DIRP] "v"(dirp), [OQ] "v"(oq_off), [SCALE] "s"(scale), [HSCALE] "v"(hscale), [QMAX] "s"(cp_max), \
                       [K31] "v"(k31), [PARK] "s"(park),                                                           \
                       [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x0c0c0001u)                       \
                     : "vcc", "memory", "s54", "s55", "s56", "s57", "v100", "v101", "v102", "v103", "v104", "v105", \
                       "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
                       "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", \
                       "v130", "v131", "v132"
        if constexpr (kFreeze) {
          asm volatile(TFC_LDEC_BLOCK(TFC_LDEC_ESC1_FREEZE, TFC_LDEC_ESC2_FREEZE, TFC_LDEC_PARK_FREEZE) TFC_LDEC_OPERANDS);
        } else {
          asm volatile(TFC_LDEC_BLOCK(TFC_LDEC_ESC1_PLAIN, TFC_LDEC_ESC2_PLAIN, "") TFC_LDEC_OPERANDS);
          cnt = kDecCadence;
          flag += acc == 0u ? 1u : 0u;
        }
#undef TFC_LDEC_OPERANDS
      }
      if (__builtin_expect(!__any(flag != 0u), 1)) {
        if (__builtin_expect(!__any(j < elems && cnt != kDecCadence), 1)) {
          if (j < elems) {
            j += kDecCadence;
            ko = kDecCadence * kEs;
            dirp += dir_step;                        // kDecCadence entries further, modulo the table count
            dirp -= dirp >= dir_end ? dir_end : 0u;
          }
          continue;
        }
        // some lanes met an escape symbol after `cnt` symbols: their Elias-gamma bits now, in a loop of
        // bit steps only (the other lanes wait: ~10 short steps), then everybody goes on from its own j
        if (j < elems) {
          j += cnt;
          ko = cnt * kEs;
          dirp += 16u * cnt;
          if (cnt != kDecCadence && mode == 0u) {        // (a parked lane keeps the code it is waiting to read)
            mode = 1u;
            nb = 0u;
            esc_limit = *reinterpret_cast<const unsigned int*>(lanes_lds + dirp + 4u) & 0x7FFFFFFFu;
          }
        }
        while (__any(j < elems && dirp >= dir_end)) dirp -= dirp >= dir_end ? dir_end : 0u;
        // the codes are read together every la.defer-th block (the loop below costs the whole wave ~10 rounds),
        // as long as somebody still gets a whole block done
        if (deferred + 1 < la.defer && __any(j < elems && cnt == kDecCadence)) {
          ++deferred;
          continue;
        }
        deferred = 0;
        // The code as a bit string: z zeros, the z + 1 bits of the magnitude, the sign.  acc collects the
        // bits behind the zeros, k counts all of them; the code is complete at k = 2 z + 2.  A lane stays
        // in the loop (hand-written: the compiler's version of it takes twice the instructions) while the
        // digits of this phase fit its code window; a long code behind a run of rare symbols goes on in
        // the generic steps of the next phase, and so does a damaged one with 31 zeros.
        {
          unsigned int acc = 0u, k = 0u, z = 0u;
          const bool pending = j < elems && mode != 0u;
          const bool in = pending && cp - cp0 < 2u * kDecCadence;
          if (in) {
            // One binary call per round (the uniform cdf {0, 1, 2} at precision 1,
            // range_coder_kernels.cc:449-471): half = B of the first interval = (s1 + 1) >> 1.  With acc = 0
            // z follows k, so k = 2 z + 2 only happens with the leading one in acc.  EXEC drops a lane when
            // its code is complete, its digits would leave the window, or it has seen 31 zeros.
            asm volatile(
                "s_mov_b64 s[56:57], exec\n\t"
                "1:\n\t"
                "ds_read_u16 v100, %[CP]\n\t"
                "v_lshrrev_b32 v101, 1, %[S]\n\t"
                "v_and_b32 v102, 1, %[S]\n\t"
                "v_add_u32 v101, v101, v102\n\t"              // half
                "v_cmp_ge_u32 vcc, %[D], v101\n\t"            // the bit
                "v_cndmask_b32 v103, 0, v101, vcc\n\t"        // A
                "v_add_u32 v104, -1, v101\n\t"
                "v_cndmask_b32 v104, v104, %[S], vcc\n\t"     // B - 1
                "v_sub_u32 %[D], %[D], v103\n\t"
                "v_sub_u32 v104, v104, v103\n\t"              // span' - 1
                "v_cndmask_b32 v105, 0, 1, vcc\n\t"
                "v_lshl_or_b32 %[ACC], %[ACC], 1, v105\n\t"
                "v_add_u32 %[K], 1, %[K]\n\t"
                "v_cmp_gt_u32 vcc, %[K64K], v104\n\t"         // renormalise
                "v_lshl_or_b32 v106, v104, 16, %[KFFFF]\n\t"
                "v_cndmask_b32 %[S], v104, v106, vcc\n\t"
                "v_cndmask_b32 v106, 0, 2, vcc\n\t"
                "v_add_u32 %[CP], %[CP], v106\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_perm_b32 v100, 0, v100, %[PERM]\n\t"
                "v_lshl_or_b32 v106, %[D], 16, v100\n\t"
                "v_cndmask_b32 %[D], %[D], v106, vcc\n\t"
                "v_cmp_eq_u32 vcc, 0, %[ACC]\n\t"
                "v_cndmask_b32 %[Z], %[Z], %[K], vcc\n\t"
                "v_lshl_add_u32 v107, %[Z], 1, 2\n\t"
                "v_cmp_ne_u32 s[54:55], %[K], v107\n\t"       // code not complete
                "v_sub_u32 v107, %[CP], %[CP0]\n\t"
                "v_cmp_gt_u32 vcc, 16, v107\n\t"              // digits of this phase < kDecCadence
                "s_and_b64 s[54:55], s[54:55], vcc\n\t"
                "v_cmp_ne_u32 vcc, 31, %[Z]\n\t"
                "s_and_b64 s[54:55], s[54:55], vcc\n\t"
                "s_and_b64 exec, exec, s[54:55]\n\t"
                "s_cbranch_execnz 1b\n\t"
                "s_mov_b64 exec, s[56:57]\n\t"
                : [D] "+v"(D), [S] "+v"(s1), [CP] "+v"(cp), [ACC] "+v"(acc), [K] "+v"(k), [Z] "+v"(z)
                : [CP0] "v"(cp0), [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x0c0c0001u)
                : "vcc", "memory", "s54", "s55", "s56", "s57", "v100", "v101", "v102", "v103", "v104", "v105",
                  "v106", "v107");
          }
          const bool fin = in && k == 2u * z + 2u;
          if (fin) {
            const unsigned int val = acc >> 1;
            const int outv = (acc & 1u) ? -static_cast<int>(val) : static_cast<int>(val) + static_cast<int>(esc_limit) - 1;
            *reinterpret_cast<Elem*>(outq + ko) = dst.make(static_cast<int>(dirp >> 4), outv);
            ko += kEs;
            ++j;
            mode = 0u;
            dirp = dirp + 16u == dir_end ? 0u : dirp + 16u;
          } else if (pending) {
            // unfinished: over to the generic steps' form (mode, bits to go, value so far)
            if (acc == 0u) {
              mode = k == 31u ? 2u : 1u;
              nb = k;
              val = 1u << 31;
            } else {
              const unsigned int togo = z - (k - z - 1u);       // magnitude bits still to come
              mode = togo != 0u ? 2u : 3u;
              nb = togo;
              val = acc << togo;
            }
          }
        }
        continue;
      }
      D = D0; s1 = s10; cp = cp0;                    // an exception somewhere in the wave: the generic steps
    }
#pragma nounroll
    for (unsigned int k = 0; k < kDecCadence; ++k) step();
    if (kFastBlock && !kFreeze && __any(saw_escape != 0u)) return;         // escapes are to be expected from here on
  }
  };
  run(std::false_type{});
  if constexpr (kFastBlock) run(std::true_type{});
  flush();

  if (live) {
    // back to the (base, span - 1, window, digits pulled) form shared with the other kernels
    const unsigned int pos = cw.base + (cp - cw_off);
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
