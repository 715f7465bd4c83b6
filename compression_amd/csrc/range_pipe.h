// Pipelined lane-per-stream range coder for gfx950 — included by range_coder.hip behind range_lanes.h.
//
// range_lanes.h gives every LANE a stream, so the time of a launch is (symbols per stream) x (cycles of one
// step of a lone wave) whatever the number of streams: everything a step does costs chain latency, and with
// BASELINE config 2 (20 x 512 streams = 160 waves) 85 % of the chip's SIMDs hold no wave at all.  Here the
// work of a step that is NOT on the chain leaves the chain wave and runs chip-wide in kernels of its own:
//
//   encode   enc_expand_kernel   parallel over symbols: quantise, table index, row lookup, range check, escape
//                                codes expanded into their binary calls -> one 32-bit CALL WORD per coder call
//                                (lo | hi << 16 on the 2^16 scale), laid out [tile][row][lane] so that the
//                                chain wave reads row k of all its 64 streams with ONE coalesced load
//            enc_chain_kernel    lane per stream: RangeEncoder::Encode (cc/lib/range_coder.cc:37-264) on the
//                                call words — interval update + held-digit bookkeeping of range_lanes.h's block,
//                                no table, no value window, no escape logic (an escape is just more rows)
//   decode   dec_rows_kernel     (index mode) table index -> LDS address of its row, range-checked
//            dec_chain_kernel    lane per stream: RangeDecoder::Decode (cc/lib/range_coder.h:224-271) with the
//                                quotient-estimate / bitmap-rank step of range_lanes.h; the Elias-gamma bits of
//                                an escape (cc/kernels/range_coder_kernels.cc:449-471) are ordinary steps on a
//                                built-in binary row, selected per lane by a small mode counter M — a lane that
//                                meets an escape falls a few steps behind its neighbours and nobody waits;
//                                every step's raw result goes to [row][lane] with one coalesced store
//            dec_parse_kernel    parallel over rows: transposes the raw results back to [stream][element],
//                                assembles escape values from their bit rows, adds cdf_offset / dequantises
//
// The lanes of a wave advance in lockstep over ROWS (coder calls), not over symbols: a stream with an escape
// simply has more rows.  The encoder re-aligns its 64 streams at every tile of kPipeTile symbols (the rows of
// a tile start at row 0 for every lane; lanes with fewer calls sit out the tile's last rows under EXEC).
//
// Bytes and symbols are the reference's: the chain arithmetic is the one range_lanes.h already runs (tests
// parametrise over both families), the expansion restates classify_fast() / escape_calls().
// When the pipelined kernels cannot take a job (more escape codes in one tile than they plan for, a decoder
// that runs out of rows) they raise the job's fallback flag and leave its state untouched; the lane-per-stream
// kernel of range_lanes.h is launched behind them under that flag (LaneArgs::guard) and codes the job.
#pragma once

// TFC_PIPE_TIMING (build switch, tools/chain_clock_probe.py): the chain waves time their waits and their hand-scheduled
// blocks (g_pipe_clock); off in the shipped library — a clock read is an s_memtime and a wait for it.
#ifndef TFC_PIPE_TIMING
#define TFC_PIPE_TIMING 0
#endif

namespace tfc {

constexpr int kPipeTile = 256;        // symbols of a stream per expansion workgroup
constexpr int kPipeEscMax = 32;       // escape codes per stream and tile the expansion plans for
constexpr unsigned int kPipeBlock = 16;          // rows per hand-scheduled block of the chain kernels
// A call word that no table produces (lower bound 0xFFFF above upper bound 1): "this lane has no call in this row"
constexpr unsigned int kPipeNoCall = 0x0001FFFFu;
// Status word of a (tile, stream) of the expansion: kind << 30 | rows.  A tile first publishes the rows (coder calls)
// its own symbols take, then — once it has looked back over the tiles before it — the rows of the stream up to and
// including itself.
constexpr unsigned int kPipeAggregate = 1u << 30;
constexpr unsigned int kPipePrefix = 2u << 30;
constexpr unsigned int kPipeRowsMask = (1u << 30) - 1u;

struct PipeEncArgs {
  const uint16_t* fast16;       // tables scaled to 16 bits (tfc_tables::d_fast)
  const int2* rows_fast;        // (offset, length | escape row << 31) per table
  int ntab;
  int tab_entries;              // entries of fast16 when the image fits the expansion's LDS copy, else 0
  unsigned int* calls;          // [group][rows / 32][lane][32]: call word r of a stream at [r / 32][lane][r % 32]
  unsigned int* status;         // [group][tile][lane]: see kPipeAggregate / kPipePrefix (0: not yet known)
  unsigned int* done;           // [group][tile]: 1 once the tile's call words are in memory
  unsigned int* fallback;       // [job]
  unsigned int* started;        // chain workgroups that are running (enc_gate_kernel)
  uint4* stage_state;           // [group][lane]: the chain's successor state ...
  uint2* stage_out;             // ... bytes written, slab outgrown; committed by enc_commit_kernel unless the job fell back
  int nt;                       // tiles per stream
  int rows;                     // row capacity of a stream
  int groups_per_job;           // groups of 64 streams per job
  int groups;                   // of the launch
  unsigned int cap;             // slab bytes per stream (chain kernel)
  long long poll_ticks;         // wall_clock64() ticks the chain waits for a tile before it gives the job up
};

// ---------------------------------------------------------------------------------------------
// Encoder, stage 1: symbols -> call words
// ---------------------------------------------------------------------------------------------

// Number of extra (binary) calls of an escape code and its k-th call word, k = 1 ... extra
// (range_coder_kernels.cc:304-321: floor(log2 g) zeros, the bits of g, the sign; each a call on the uniform
// binary cdf at precision 1 = [0, 2^15) or [2^15, 2^16) on the 2^16 scale).
__device__ inline unsigned int pipe_escape_extra(unsigned int g) { return 2u * static_cast<unsigned int>(31 - __clz(static_cast<int>(g))) + 2u; }
__device__ inline unsigned int pipe_escape_word(unsigned int g, unsigned int neg, unsigned int extra, unsigned int k) {
  const unsigned int left = extra - k;            // calls still to come behind this one
  const unsigned int sft = left - 1u;
  const unsigned int bit = left == 0u ? neg : (sft < 32u ? (g >> sft) & 1u : 0u);
  return bit ? 0x00008000u : 0x80000000u;         // lo | hi << 16, hi = 2^16 stored as 0
}

// One workgroup per (tile of kPipeTile symbols, group of 64 streams); blockIdx = tile * groups + group, so that the
// tiles of one group are dispatched in order and far apart.  The call words of a stream are contiguous in memory (the
// chain's lane reads its own stream, 16 rows per request), so a tile has to know how many rows the tiles before it
// took — per stream: a decoupled look-back over the status words (every workgroup publishes its own row counts as
// soon as it has them, and needs its predecessors' only when it starts writing; the 64 lanes of one wave look back
// for the 64 streams side by side).
constexpr int kExpandThreads = 512;
constexpr int kExpandTabBytes = 32 * 1024;     // table images up to this size are staged in LDS (two workgroups per CU)
// TABLDS (= pa.tab_entries != 0): a template parameter since round 6 — as a run-time condition, phase C's word came either
// out of LDS or from a global load, and at the point where the two paths meet hipcc waited with s_waitcnt vmcnt(0) for
// the load that may be pending in the word's register: in front of every store of the loop, so every store also waited
// for the one before it to reach memory (a wave had ONE store in flight: phase C was two thirds of the kernel).
template <bool INDEXED, typename Src, bool TABLDS>
__global__ void __launch_bounds__(kExpandThreads, 4) enc_expand_kernel(const EncLaneJobs<Src> jobs, const PipeEncArgs pa) {
  constexpr int kRow = kPipeTile;
  constexpr int kHalves = kExpandThreads / kPipeTile;     // thread = (half, symbol): half h takes streams h, h + kHalves, ...
  // table entry of every (stream, symbol) of the tile: offset of the symbol's lower bound in the 16-bit table image
  // (the call word is the 4 bytes there: lo | hi << 16)
  __shared__ unsigned short W[64 * kRow];
  // the table image, when it fits (pa.tab_entries != 0): phase C's look-ups are LDS reads instead of 4-byte gathers
  // from 64 different cache lines per wave instruction — measured, those gathers were what the kernel's time went
  // into, and what made the memory system slow for everybody else
  __shared__ __attribute__((aligned(16))) unsigned short tab[kExpandTabBytes / 2];
  // escape codes per stream (appended in phase A, put in order in phase B):
  // position | neg << 8 | extra calls << 9 | low 16 bits of gamma << 16 (all of gamma when extra < 34)
  __shared__ unsigned int esc[64][kPipeEscMax];
  __shared__ unsigned short ecum[64][kPipeEscMax];        // extra calls of the stream's escape codes before this one
  __shared__ unsigned int ecount[64], nesc[64], start[64];

  const unsigned int gi = blockIdx.x % static_cast<unsigned int>(pa.groups);
  const unsigned int T = blockIdx.x / static_cast<unsigned int>(pa.groups);
  const unsigned int job = gi / static_cast<unsigned int>(pa.groups_per_job);
  const unsigned int gidx = gi % static_cast<unsigned int>(pa.groups_per_job);
  const EncLaneJob<Src>& J = jobs.job[job];
  const Src src = J.src;
  const int32_t* const index = J.index;
  const unsigned int tid = threadIdx.x, lane = tid & 63u;
  const unsigned int elems = static_cast<unsigned int>(jobs.elems);
  const int64_t s0 = static_cast<int64_t>(gidx) * 64;
  const unsigned int ntab = static_cast<unsigned int>(pa.ntab);
  if (tid < 64u) ecount[tid] = 0u;
  constexpr bool tab_lds = TABLDS;
  if (tab_lds) {
    // (in flight next to phase A's loads; first read behind the barriers in front of phase C)
    const uint4* const src16 = reinterpret_cast<const uint4*>(pa.fast16);
    for (unsigned int i = tid; i < (static_cast<unsigned int>(pa.tab_entries) + 7u) / 8u; i += kExpandThreads)
      reinterpret_cast<uint4*>(tab)[i] = src16[i];
  }
  __syncthreads();

  // table and value of symbol `at` of stream s (phase C looks again at the rare escape code of 2^16 and more)
  auto escape_of = [&](int64_t s, unsigned int at, unsigned int& g, unsigned int& neg) {
    const int64_t pos = s * jobs.elems + at;
    int t = static_cast<int>(at % ntab);
    if (INDEXED) {
      t = tfc_gload(index + pos);
      if (t < 0 || t >= pa.ntab) t = 0;
    }
    const int2 row = pa.rows_fast[t];
    const int32_t v = src.load(pos, t);
    const int32_t vmax = (row.y & 0x7FFFFFFF) - 3;
    neg = v < 0 ? 1u : 0u;
    g = v < 0 ? 0u - static_cast<unsigned int>(v) : static_cast<unsigned int>(v - vmax) + 1u;
  };

  // ---- phase A: one call word per symbol (an escape: the word of its row's escape symbol).  Written in stages
  // over register arrays — all loads of a stage are in flight together (interleaved with the LDS stores, hipcc
  // waits for every single load: a load through a pointer out of the kernel arguments is a flat load, which may
  // alias LDS) -------------------------------------------------------------------------------------------
  {
    constexpr int N = 64 / kHalves;       // streams per thread ...
    constexpr int NB = INDEXED ? 8 : 16;  // ... taken NB at a time (registers: two workgroups per CU, nothing spilled)
    const unsigned int p = tid % kPipeTile, half = tid / kPipeTile;
    const unsigned int j = T * kPipeTile + p;             // this thread's symbol of every stream it handles
    const bool inpos = j < elems;
    const unsigned int jc = inpos ? j : 0u;
    const int tch = static_cast<int>(jc % ntab);
    const int2 rowc = pa.rows_fast[tch];
    unsigned long long badpos = ~0ull;
    auto position = [&](int i) {
      const int64_t s = s0 + i * kHalves + static_cast<int>(half);
      return (s < jobs.streams ? s : 0) * jobs.elems + jc;       // (a clamped address: loaded, not used)
    };
#pragma nounroll
    for (int i0 = 0; i0 < N; i0 += NB) {
      int32_t val[NB];
      int tt[NB];
      int2 rw[NB];
      unsigned int entry[NB];
      if (INDEXED) {
#pragma unroll
        for (int i = 0; i < NB; ++i) tt[i] = tfc_gload(index + position(i0 + i));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (tt[i] < 0 || tt[i] >= pa.ntab) {
            if (inpos && s0 + (i0 + i) * kHalves + static_cast<int>(half) < jobs.streams)
              badpos = min(badpos, static_cast<unsigned long long>(position(i0 + i)));
            tt[i] = 0;
          }
          rw[i] = pa.rows_fast[tt[i]];
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) { tt[i] = tch; rw[i] = rowc; }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) val[i] = src.load(position(i0 + i), tt[i]);
      // classify_fast() without branches: plain symbols are 0 ... nplain - 1; anything else is the row's escape
      // symbol (rows that have one) or a range error; the call word is the 4 bytes lo | hi << 16 of the table image
      unsigned int escbits = 0u;          // bit i: stream i0 + i of this thread takes an escape code
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int len = rw[i].y & 0x7FFFFFFF;
        const bool hasesc = rw[i].y < 0;
        const int nplain = hasesc ? len - 3 : len - 2;
        const bool plain = static_cast<unsigned int>(val[i]) < static_cast<unsigned int>(nplain);
        const int sym = plain ? val[i] : (hasesc ? len - 3 : 0);
        entry[i] = static_cast<unsigned int>(rw[i].x + 1 + sym);
        const bool live = inpos && s0 + (i0 + i) * kHalves + static_cast<int>(half) < jobs.streams;
        if (live && !plain && !hasesc) badpos = min(badpos, static_cast<unsigned long long>(position(i0 + i)));
        escbits |= (live && !plain && hasesc) ? 1u << i : 0u;
        entry[i] = live ? entry[i] : 0u;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int sl = (i0 + i) * kHalves + static_cast<int>(half);
        W[sl * kRow + p] = static_cast<unsigned short>(entry[i]);
        if ((escbits >> i) & 1u) {
          // the code's magnitude (range_coder_kernels.cc:296-303) next to the symbol's position
          const int32_t vmax = (rw[i].y & 0x7FFFFFFF) - 3;
          const unsigned int neg = val[i] < 0 ? 1u : 0u;
          const unsigned int g = val[i] < 0 ? 0u - static_cast<unsigned int>(val[i]) : static_cast<unsigned int>(val[i] - vmax) + 1u;
          const unsigned int slot = atomicAdd(&ecount[sl], 1u);
          if (slot < static_cast<unsigned int>(kPipeEscMax)) esc[sl][slot] = p | (neg << 8) | (pipe_escape_extra(g) << 9) | (g << 16);
        }
      }
    }
    if (badpos != ~0ull) atomicMin(J.first_error, badpos);
  }
  __syncthreads();

  const unsigned int valid = min(static_cast<unsigned int>(kPipeTile), elems > T * kPipeTile ? elems - T * kPipeTile : 0u);

  // ---- phase B: a thread per stream puts its escape codes in order, counts its rows and finds where they start ----
  if (tid < 64u) {
    const int64_t s = s0 + tid;
    unsigned int n = ecount[tid], cum = 0u;
    const unsigned int nsym = s < jobs.streams ? valid : 0u;
    if (n > static_cast<unsigned int>(kPipeEscMax)) {
      atomicOr(&pa.fallback[job], 1u);      // more escape codes than planned for: the lane-per-stream kernel codes this job
      n = 0u;
    }
    for (unsigned int a = 1; a < n; ++a) {      // insertion sort by position (a handful of entries)
      const unsigned int x = esc[tid][a];
      unsigned int b = a;
      for (; b > 0u && (esc[tid][b - 1u] & 0xFFu) > (x & 0xFFu); --b) esc[tid][b] = esc[tid][b - 1u];
      esc[tid][b] = x;
    }
    for (unsigned int a = 0; a < n; ++a) {
      ecum[tid][a] = static_cast<unsigned short>(cum);
      cum += (esc[tid][a] >> 9) & 0x7Fu;
    }
    const unsigned int cnt = nsym + cum;
    nesc[tid] = n;
    // rows of the stream before this tile: its predecessors' counts, back to the first one that already knows its own
    // prefix (tiles are dispatched in order, so every predecessor is running or done).  Relaxed atomics: a status word
    // carries everything its reader needs.
    unsigned int* const st = pa.status + static_cast<size_t>(gi) * pa.nt * 64 + tid;
    __hip_atomic_store(st + static_cast<size_t>(T) * 64, kPipeAggregate | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (A predecessor that has not published yet is running or about to be dispatched — workgroups are dispatched in
    // blockIdx order in practice, which HIP does not promise: after a short spin the wave sleeps between polls, and a
    // predecessor that does not show up within `poll_ticks` gives the job to the lane-per-stream kernel — this tile
    // then publishes "does not fit", which its successors pass on.)
    unsigned int before = 0u;
    bool stuck = false;
    long long t0 = 0;
    for (unsigned int t = T; t-- > 0u && !stuck;) {
      unsigned int v, spins = 0u;
      while (((v = __hip_atomic_load(st + static_cast<size_t>(t) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) {
        if (++spins < 64u) continue;
        __builtin_amdgcn_s_sleep(16);
        const long long now = static_cast<long long>(wall_clock64());
        if (t0 == 0) t0 = now;
        else if (now - t0 > pa.poll_ticks) { stuck = true; break; }
      }
      if (stuck) break;
      before += v & kPipeRowsMask;
      if (v & kPipePrefix) break;
    }
    const unsigned int limit = static_cast<unsigned int>(pa.rows);
    const bool fits = !stuck && before + cnt <= limit;
    if (!fits) atomicOr(&pa.fallback[job], 1u);     // more rows than the launch planned for (escape codes far beyond the tables' tail mass)
    __hip_atomic_store(st + static_cast<size_t>(T) * 64, kPipePrefix | (fits ? before + cnt : limit + 1u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    start[tid] = fits ? before : 0xFFFFFFFFu;
  }
  __syncthreads();

  // ---- phase C: call words out, a wave per stream at a time (64 consecutive words: one coalesced store): the
  // plain words from their symbol's position to its row (position + extra calls of the escape codes before it),
  // then the bit rows of the stream's escape codes (a lane per code) ------------------------------------
  {
    constexpr unsigned int kWaves = kExpandThreads / 64;
    const unsigned int q = tid >> 6;
    for (unsigned int sl = q; sl < 64u; sl += kWaves) {
      const unsigned int first = start[sl];
      if (first == 0xFFFFFFFFu || s0 + sl >= jobs.streams) continue;
      // call word r of the stream: 32 consecutive rows of a stream are 128 contiguous bytes, the 64 streams' side by
      // side (what the chain's loaders read per iteration is one contiguous 8 KB)
      unsigned int* const gbase = pa.calls + static_cast<size_t>(gi) * 64 * pa.rows;
      auto out = [&](unsigned int r) -> unsigned int& {
        return gbase[((static_cast<size_t>(r >> 5) * 64 + sl) << 5) + (r & 31u)];
      };
      const unsigned int ne = nesc[sl];
      for (unsigned int p = lane; p < valid; p += 64u) {
        unsigned int cum = 0u;
        for (unsigned int i = 0; i < ne; ++i) {
          const unsigned int e = esc[sl][i];
          cum += (e & 0xFFu) < p ? (e >> 9) & 0x7Fu : 0u;
        }
        const unsigned int at = W[sl * kRow + p];
        // (lo | hi << 16 as they lie in the table; from global memory one 4-byte load at 2-byte alignment)
        unsigned int word;
        if constexpr (tab_lds) word = static_cast<unsigned int>(tab[at]) | (static_cast<unsigned int>(tab[at + 1u]) << 16);
        else word = reinterpret_cast<const TFC_AS1 LanePacked<unsigned int>*>((const TFC_AS1 void*)(pa.fast16 + at))->v;
        // Stores a wave keeps in flight in this loop: TWO (TFC_EXPAND_INFLIGHT = 1 behind the current one).  Round 5's
        // loop had one — hipcc's s_waitcnt vmcnt(0) where the LDS and the global path of the look-up met (see TABLDS) —
        // and without any bound the kernel is fastest alone, but the chain next to it, whose helpers' loads queue behind
        // these stores, loses more than the expansion gains.  20 / 32 batches per launch, encode call (expansion, chain
        // beside it), ms: one 3.42 (3.31, 3.36) / 6.14; two 3.41 (3.11, 3.36) / 5.51; four 4.05 (3.27, 3.99) / 5.56;
        // unbounded 4.11 (3.28, 4.06) / 5.84 (tools/r06_ab_sat.sh, profiles/r06_notes.md).
#ifndef TFC_EXPAND_INFLIGHT
#define TFC_EXPAND_INFLIGHT 1
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TFC_EXPAND_INFLIGHT) : "memory");
        out(first + p + cum) = word;
      }
      if (lane < ne) {
        const unsigned int e = esc[sl][lane];
        const unsigned int epos = e & 0xFFu, neg = (e >> 8) & 1u, extra = (e >> 9) & 0x7Fu;
        unsigned int g = e >> 16, ng;
        if (extra >= 34u) escape_of(s0 + sl, T * kPipeTile + epos, g, ng);      // 2^16 and more: the value again
        const unsigned int row = epos + ecum[sl][lane];                          // the escape symbol's own call
        for (unsigned int k = 1; k <= extra; ++k) out(first + row + k) = pipe_escape_word(g, neg, extra, k);
      }
    }
  }
  // the tile's words are in memory: the barrier orders every thread's stores before thread 0's release
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&pa.done[static_cast<size_t>(gi) * pa.nt + T], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Encoder, stage 2: the chain
// ---------------------------------------------------------------------------------------------

// One row: a lane whose word is "no call" leaves EXEC for the rest of the block (a stream's calls are the first rows
// of its last block, so nothing follows a "no call" inside a block); the call word is unpacked into the (lo, 0) /
// (hi, 0) register pairs TFC_LENC_B multiplies from.
// (TFC_LENC_B of range_lanes.h with its temporaries at v100-v119: a chain workgroup is eight waves (sixteen before round 5), two per SIMD,
// 128 registers each)
// Three forms of the row's last part:
//  * TFC_PENC_STEP (enc_chain_direct_kernel): the digit is stored big-endian, FLAG counts the steps that leave 0xFFFF held;
//  * TFC_PENC_STEP_H (enc_chain_kernel, round 5): 23 instructions instead of 30 — the digit goes to LDS as it is (the storer
//    wave swaps the bytes of what it copies); the new held digit is selected out of bs's upper half by the v_cndmask itself
//    (SDWA); a lane that holds 0xFFFF or more after a step leaves EXEC like one that meets "no call" (the caller looks at H
//    behind the block: the wave repeats the block call by call either way); and a carry does not push the held digit out
//    (the reference's form: the digit is final, nothing is held behind it) but stays in it, X = H + c held on — the digit
//    leaves with the next renormalisation, the same bytes in the same order, no second carry can reach it, and
//    (held, X) is a state every other kernel and finalize read the same way — which takes the carry out of the
//    bookkeeping: emit = r & held, held' = held | r, with `held` kept as 0 / 2 (the cursor's increment; v119 = 2).
#define TFC_PENC_ROW(W)                                                                    \
  "v_cmpx_ne_u32 vcc, %[NOCALL], %[" #W "]\n\t"                                            \
  "v_and_b32 v100, %[KFFFF], %[" #W "]\n\t"                                                \
  "v_lshrrev_b32 v102, 16, %[" #W "]\n\t"                                                  \
  "v_mad_u64_u32 v[104:105], s[52:53], v100, %[S], v[100:101]\n\t"                         \
  "v_mad_u64_u32 v[106:107], s[52:53], v102, %[S], v[102:103]\n\t"                         \
  "v_alignbit_b32 v104, v105, v104, 16\n\t"                                               \
  "v_alignbit_b32 v106, v107, v106, 16\n\t"                                               \
  "v_add_u32 v106, -1, v106\n\t"                                                          \
  "v_min_u32 v106, v106, %[S]\n\t"                                                        \
  "v_add_co_u32 v108, vcc, %[BASE], v104\n\t"                                             \
  "v_addc_co_u32 v111, vcc, 0, %[H], vcc\n\t"                                             \
  "v_sub_u32 v109, v106, v104\n\t"                                                        \
  "v_sub_u32 v117, v111, %[H]\n\t"                                                        \
  "v_cmp_gt_u32 vcc, %[K64K], v109\n\t"                                                   \
  "v_cndmask_b32 v118, 0, 1, vcc\n\t"
#define TFC_PENC_STEP(W)                                                                   \
  TFC_PENC_ROW(W)                                                                          \
  "v_perm_b32 v110, 0, v111, %[PERM]\n\t"                                                 \
  "ds_write_b16 %[NA], v110\n\t"                                                          \
  "v_or_b32 v112, v117, v118\n\t"                                                         \
  "v_and_b32 v112, v112, %[HAD]\n\t"                                                      \
  "v_lshl_add_u32 %[NA], v112, 1, %[NA]\n\t"                                              \
  "v_lshrrev_b32 v113, 16, v108\n\t"                                                      \
  "v_lshlrev_b32 v114, 16, v108\n\t"                                                      \
  "v_lshl_or_b32 v115, v109, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32 %[BASE], v108, v114, vcc\n\t"                                            \
  "v_cndmask_b32 %[S], v109, v115, vcc\n\t"                                               \
  "v_cndmask_b32 %[H], %[H], v113, vcc\n\t"                                               \
  "v_or_b32 v119, %[HAD], v118\n\t"                                                       \
  "v_bfi_b32 %[HAD], v117, v118, v119\n\t"                                                \
  "v_cmp_eq_u32 vcc, %[KFFFF], %[H]\n\t"                                                  \
  "v_addc_co_u32 %[FLAG], vcc, 0, %[FLAG], vcc\n\t"
#define TFC_PENC_STEP_H(W)                                                                 \
  "v_cmpx_ne_u32 vcc, %[NOCALL], %[" #W "]\n\t"                                            \
  "v_and_b32 v100, %[KFFFF], %[" #W "]\n\t"                                                \
  "v_lshrrev_b32 v102, 16, %[" #W "]\n\t"                                                  \
  "v_mad_u64_u32 v[104:105], s[52:53], v100, %[S], v[100:101]\n\t"                         \
  "v_mad_u64_u32 v[106:107], s[52:53], v102, %[S], v[102:103]\n\t"                         \
  "v_alignbit_b32 v104, v105, v104, 16\n\t"                                               \
  "v_alignbit_b32 v106, v107, v106, 16\n\t"                                               \
  "v_add_u32 v106, -1, v106\n\t"                                                          \
  "v_min_u32 v106, v106, %[S]\n\t"                                                        \
  "v_add_co_u32 v108, vcc, %[BASE], v104\n\t"                                             \
  "v_addc_co_u32 v111, vcc, 0, %[H], vcc\n\t"                                             \
  "v_sub_u32 v109, v106, v104\n\t"                                                        \
  "v_cmp_gt_u32 vcc, %[K64K], v109\n\t"                                                   \
  "ds_write_b16 %[NA], v111\n\t"                                                          \
  "v_cndmask_b32 v112, 0, %[HAD], vcc\n\t"                                                \
  "v_add_u32 %[NA], %[NA], v112\n\t"                                                      \
  "v_cndmask_b32 %[HAD], %[HAD], v119, vcc\n\t"                                           \
  "v_lshlrev_b32 v114, 16, v108\n\t"                                                      \
  "v_lshl_or_b32 v115, v109, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32 %[BASE], v108, v114, vcc\n\t"                                            \
  "v_cndmask_b32 %[S], v109, v115, vcc\n\t"                                               \
  "v_cndmask_b32_sdwa %[H], v111, v108, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
  "v_cmpx_gt_u32 vcc, %[KFFFF], %[H]\n\t"

// LDS of one group of a chain workgroup: call words in, digits out (see enc_chain_kernel)
struct PipeEncChainLds {
  static constexpr unsigned int kRows = 2 * kPipeBlock;      // rows per iteration: two hand-scheduled blocks
  // Round 5: what the chain lost beside the expansion (4.7 ms against 3.6 alone) was waiting for its helpers — 0.6 ms for
  // call words, 0.5-0.8 ms for a free digit slot: their loads and stores queue behind the expansion's — and the cure is
  // depth: four iterations of call words in LDS (each loader has its next one in registers) and six of digits.  That is
  // 74 KB per group, so a workgroup is TWO groups (eight waves) instead of four: 80 CUs to the chains of the 20-batch
  // launch instead of 40 (the expansion beside them 3.2 -> 3.5 ms), the chain 4.66 -> 3.82 ms.  Measured
  // (tools/r05_line_ab.sh, encode call of the 20-batch group): 4 groups x (2, 2) slots 4.79 ms; 4 x (2, 3) 4.71;
  // 3 x (2, 5) 4.32; 3 x (3, 4) 4.18; 2 x (3, 8) 3.89; 2 x (4, 6) 3.88.
#ifndef TFC_ENC_SLOTS
#define TFC_ENC_SLOTS 4
#endif
  static constexpr unsigned int kSlots = TFC_ENC_SLOTS;      // iterations of call words in LDS
#ifndef TFC_ENC_DIGSLOTS
#define TFC_ENC_DIGSLOTS 6
#endif
  static constexpr unsigned int kDigSlots = TFC_ENC_DIGSLOTS;   // iterations of digits the storer may be behind
  static constexpr unsigned int kDigits = 2 * kRows + 16;    // digit bytes of a lane and iteration: one digit per call at most, and
                                                             // room for runs of 0xFFFF digits that settle (longer: the fallback)
  static constexpr int kCallStride = 4 * kRows + 16;         // a lane's words of an iteration, 16-byte accesses without bank conflicts
  static constexpr int kCallSlot = 64 * kCallStride;
  static constexpr int kDigStride = lane_stride(kDigits + 8);   // + the speculative write behind a full area
  static constexpr int kDigSlot = 64 * kDigStride;
  static constexpr int kCalls = 0;
  static constexpr int kDig = kCalls + kSlots * kCallSlot;
  static constexpr int kRec = kDig + kDigSlots * kDigSlot;   // (position in the slab, bytes) of every lane's digits, per digit slot
  static constexpr int kSync = kRec + kDigSlots * 64 * 8;
  static constexpr int kGroup = kSync + 64;
#ifndef TFC_ENC_GROUPS
#define TFC_ENC_GROUPS 2
#endif
  static constexpr int kGroups = TFC_ENC_GROUPS;             // groups (chain waves) of a large launch's workgroups: one per SIMD
  // words at kSync
  static constexpr int kSeq = 0;            // [kSlots] iteration + 1 whose call words the slot holds | kLast | kBail
  static constexpr int kConsumed = 4;       // iterations the chain has read (kSeq takes kSlots <= 4 words)
  static constexpr int kDigPub = 5;         // iterations whose digits the chain has handed over
  static constexpr int kDigDone = 6;        // ... the helper has stored
  static constexpr int kExit = 7;           // the chain has left its loop
  static constexpr unsigned int kLast = 0x80000000u, kBail = 0x40000000u;
};
static_assert(PipeEncChainLds::kGroups * PipeEncChainLds::kGroup <= 160 * 1024 && PipeEncChainLds::kSlots <= 4, "a chain workgroup's LDS");
struct PipeChainJob { uint4* state; uint8_t* chunk; unsigned int* chunk_len; unsigned int* overflow_flag; };
struct PipeChainJobs {
  int64_t streams;
  PipeChainJob job[64];
};

// The chain of a group of 64 streams, lane per stream: row k of the launch is call k of every stream (a stream with
// escape codes has more rows; the shorter streams sit out the last ones).  It may run WHILE the expansion is still
// writing (the host launches it on a stream of its own), and next to anything else that keeps the memory system busy —
// which a lone wave whose time is the latency of its instruction chain must not feel: measured with the expansion
// running, every global load or store the chain wave issued itself held it up at ISSUE (full request queues), 3.7 ->
// 5.9 ms for BASELINE config 2 — and one helper wave doing all of it in turn did not keep up (the same 5.9).  So a
// workgroup is kGroups groups (two; four before round 5) of four waves, wave kGroups r + g of group g, the three helpers asleep most of the time:
//   loaders (2)  make sure, tile by tile, that the expansion has released the rows they are about to request (`done`,
//                acquire; the stream's row count up to that tile from the status words) and load the call words of every
//                other iteration — 128 bytes of every lane's own stream — into an LDS ring up to kSlots iterations
//                ahead ("no call" behind a stream's last row);
//   storer       stores the digits of finished iterations from LDS to the slabs, exactly the bytes;
//   chain        touches LDS only: 32 call words per lane in, the hand-scheduled blocks, digits out.
// Hand-over by counters in LDS (one writer each; LDS executes a wave's accesses in order).  A tile that does not arrive
// within `poll_ticks` (the two kernels were not scheduled side by side after all and this one went first) gives the job
// to the fallback.  Successor states are staged: enc_commit_kernel, behind both kernels, hands them to the handles
// unless the job fell back.  A large launch's workgroups are two groups: their LDS (148 KB) also keeps other kernels'
// workgroups off the CU; a small launch's (a model step's few groups, next to convolutions that leave no CU empty)
// are one group each and fit in anywhere.
__global__ void __launch_bounds__(1024) enc_chain_kernel(const PipeChainJobs jobs, const PipeEncArgs pa) {
  using L = PipeEncChainLds;
  constexpr unsigned int kRows = L::kRows;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_lds[];
  const unsigned int G = blockDim.x >> 8;            // groups of this workgroup: four waves each
  if (threadIdx.x == 0u) __hip_atomic_fetch_add(pa.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned int i = threadIdx.x; i < G * 16u; i += blockDim.x)
    reinterpret_cast<unsigned int*>(chain_lds + (i / 16u) * L::kGroup + L::kSync)[i % 16u] = 0u;
  __syncthreads();

  const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned int gi = blockIdx.x * G + (wave % G);
  if (gi >= static_cast<unsigned int>(pa.groups)) return;
  const unsigned int role = wave / G;                // 0: chain; 1, 2: loaders (even / odd iterations); 3: the digits' way out
  unsigned char* const area = chain_lds + (wave % G) * L::kGroup;
  // (the hand-over counters through LDS-address-space pointers: a volatile access through a generic pointer is a FLAT
  // instruction, which goes down the vector-memory path the chain wave has to stay out of)
  typedef __attribute__((address_space(3))) unsigned int lds_u32;
  volatile lds_u32* const sync = reinterpret_cast<volatile lds_u32*>(
      (__attribute__((address_space(3))) unsigned char*)chain_lds + (wave % G) * L::kGroup + L::kSync);
  const unsigned int area_off = static_cast<unsigned int>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)area));

  const unsigned int job = gi / static_cast<unsigned int>(pa.groups_per_job);
  const unsigned int gidx = gi % static_cast<unsigned int>(pa.groups_per_job);
  const PipeChainJob& J = jobs.job[job];
  const int64_t s = static_cast<int64_t>(gidx) * 64 + lane;
  const bool live = s < jobs.streams;
  unsigned char* const out = J.chunk + (live ? s : 0) * static_cast<int64_t>(pa.cap);

  if (role == 1u || role == 2u) {
    // ================================ loader wave: iterations role - 1, role + 1, ... ================================
    const unsigned int nt = static_cast<unsigned int>(pa.nt);
    const unsigned int* const status = pa.status + static_cast<size_t>(gi) * nt * 64 + lane;
    const unsigned int* const done = pa.done + static_cast<size_t>(gi) * nt;
    const unsigned char* const words = reinterpret_cast<const unsigned char*>(pa.calls + static_cast<size_t>(gi) * 64 * pa.rows);
    unsigned int avail = 0u;      // rows of this lane's stream that are in memory; all of them once tn == nt
    unsigned int tn = 0u;         // tiles [0, tn) are in
    bool bail = __hip_atomic_load(&pa.fallback[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    // rows [0, need) of every stream in memory (or all the stream has)
    auto ensure = [&](unsigned int need) {
      while (tn < nt && __any(live && avail < need)) {
        const unsigned int* const flag = done + tn;
        long long t0 = 0;
        bool timing = false;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          const long long now = static_cast<long long>(wall_clock64());
          if (!timing) { t0 = now; timing = true; }
          else if (now - t0 > pa.poll_ticks) { bail = true; return; }
          __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        avail = __hip_atomic_load(status + static_cast<size_t>(tn) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kPipeRowsMask;
        ++tn;
        // a tile that did not fit, or too many escape codes somewhere in the job: nothing of it is kept
        if (__hip_atomic_load(&pa.fallback[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bail = true; return; }
      }
    };
    for (unsigned int b = role - 1u;; b += 2u) {
      const unsigned int slot = b % L::kSlots;
      if (!bail) ensure((b + 1u) * kRows);
      if (!bail && tn == nt && !__any(b * kRows < avail)) break;       // no such iteration: the other loader had the last one
      uint4 v[kRows / 4u];
      if (!bail) {
        // requested before the slot is free: the words arrive while the chain is still reading the slot's last ones.
        // The iteration's words are 8 KB in a row, [stream][32]; request c of this lane is words 4 (lane % 8) ... of
        // stream 8 c + lane / 8 (coalesced: eight cache lines per request instead of 64)
        const unsigned char* const p = words + static_cast<size_t>(b) * (64u * kRows * 4u) + 16u * lane;
#pragma unroll
        for (unsigned int c = 0; c < kRows / 4u; ++c) v[c] = lanes_gload16(p + 1024u * c);
        if (__any(avail < (b + 1u) * kRows)) {
          // the last rows of the shorter streams (every tile is in by now): "no call" behind a stream's last
          const unsigned int r0 = b * kRows + 4u * (lane & 7u);
#pragma unroll
          for (unsigned int c = 0; c < kRows / 4u; ++c) {
            const unsigned int theirs = static_cast<unsigned int>(__shfl(static_cast<int>(avail), static_cast<int>(8u * c + (lane >> 3))));
            v[c].x = r0 + 0u < theirs ? v[c].x : kPipeNoCall;
            v[c].y = r0 + 1u < theirs ? v[c].y : kPipeNoCall;
            v[c].z = r0 + 2u < theirs ? v[c].z : kPipeNoCall;
            v[c].w = r0 + 3u < theirs ? v[c].w : kPipeNoCall;
          }
        }
      }
      // a free slot (the chain leaves its loop behind the last iteration, or when the other loader gave up)
      bool gone = false;
      while (b >= sync[L::kConsumed] + L::kSlots) {
        if (sync[L::kExit] != 0u) { gone = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (gone) break;
      if (bail) {
        if (lane == 0u) {
          sync[L::kSeq + slot] = L::kBail;
          atomicOr(&pa.fallback[job], 1u);
        }
        break;
      }
#pragma unroll
      for (unsigned int c = 0; c < kRows / 4u; ++c)
        *reinterpret_cast<uint4*>(area + L::kCalls + slot * L::kCallSlot + L::kCallStride * (8u * c + (lane >> 3)) + 16u * (lane & 7u)) = v[c];
      const bool last = tn == nt && !__any((b + 1u) * kRows < avail);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the words are in the slot
      if (lane == 0u) sync[L::kSeq + slot] = (b + 1u) | (last ? L::kLast : 0u);
      if (last) break;
    }
    return;
  }

  if (role == 3u) {
    // ================================ digits of finished iterations: LDS -> slab =================================
    for (unsigned int b_store = 0u;;) {
      if (b_store < sync[L::kDigPub]) {
        const unsigned int ds = b_store % L::kDigSlots;
        const uint2 rec = reinterpret_cast<const uint2*>(area + L::kRec)[ds * 64u + lane];      // (position, bytes)
        const unsigned char* const src = area + L::kDig + ds * L::kDigSlot + L::kDigStride * lane;
        // 16 bytes at a time from the stream's position; what a piece carries beyond the iteration's bytes is overwritten
        // by the next iteration's first piece (this wave stores all of a stream's bytes, in order).  A stream that
        // outgrows its slab: nothing is stored beyond it; the chain raises the handle's flag.
        unsigned char* const dst = out + rec.x;
#pragma unroll
        for (unsigned int c = 0; c < L::kDigits / 16u; ++c) {
          if (16u * c < rec.y && rec.x + 16u * c + 16u <= pa.cap) {
            // (the chain leaves its digits in LDS as 16-bit values; a stream is big-endian: a byte swap per half, here,
            // off the chain)
            uint2 lo = reinterpret_cast<const uint2*>(src)[2 * c], hi = reinterpret_cast<const uint2*>(src)[2 * c + 1];
            lo.x = __builtin_amdgcn_perm(lo.x, lo.x, 0x02030001u); lo.y = __builtin_amdgcn_perm(lo.y, lo.y, 0x02030001u);
            hi.x = __builtin_amdgcn_perm(hi.x, hi.x, 0x02030001u); hi.y = __builtin_amdgcn_perm(hi.y, hi.y, 0x02030001u);
            lanes_gstore16(dst + 16u * c, lo, hi);
          }
        }
        ++b_store;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot has been read
        if (lane == 0u) sync[L::kDigDone] = b_store;
      } else {
        // (the chain publishes its last digits, then its exit)
        if (sync[L::kExit] != 0u && b_store == sync[L::kDigPub]) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    return;
  }

  // ================================ chain wave ================================
  __builtin_amdgcn_s_setprio(3);
  const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
  uint4 st = live ? J.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
  unsigned int base = st.x, s1 = st.y, hd = st.z & 0xFFFFu, had = st.z >> 31, rn = st.w;
  lanes_pin(base, s1, hd, rn);

  unsigned int wpos = 0u, n = 0u, overflow = 0u;
  unsigned char* dstage = area + L::kDig + L::kDigStride * lane;               // this lane's digits of the iteration
  unsigned int ds_off = area_off + L::kDig + L::kDigStride * lane;             // ... as an LDS address

  auto put = [&](unsigned int d, bool on) {
    *reinterpret_cast<unsigned short*>(dstage + n) = static_cast<unsigned short>(d);      // (as it is: the storer swaps the bytes)
    n += on ? 2u : 0u;
  };
  // (rare: a run of 0xFFFF digits settles) the run's bytes behind the iteration's digits; a run that does not fit the
  // slot gives the job to the fallback
  bool too_long = false;
  auto put_run = [&](unsigned int fill, unsigned int bytes) {
    if (n + bytes + 2u > L::kDigits) {
      too_long = true;
      return;
    }
    for (unsigned int k = 0; k < bytes; k += 2u) *reinterpret_cast<unsigned short*>(dstage + n + k) = static_cast<unsigned short>(fill);
    n += bytes;
  };
  // one coder call on [lo, hi) / 2^16 for the lanes with `act` — the generic call of range_lanes.h
  // (enc_lanes_kernel::call), every case of the held-digit bookkeeping
  auto call = [&](unsigned int lo, unsigned int hi, bool act) __attribute__((always_inline)) {
    const unsigned int a = scale16(s1, lo);
    const unsigned int b = scale16(s1, hi) - 1u;
    const unsigned int bs = base + a;
    const unsigned int t1 = b - a;
    const bool carry = act && bs < a;
    const bool ren = act && (t1 >> 16) == 0u;
    const unsigned int e = bs >> 16;
    const bool solid = ren && e != 0xFFFFu;
    const bool ffff = ren && e == 0xFFFFu;
    const bool held = had != 0u;
    const unsigned int X = (hd + (carry ? 1u : 0u)) & 0xFFFFu;
    const bool emit = held && (solid || (carry && (!ffff || rn != 0u)));
    put(X, emit);
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
#if TFC_PIPE_TIMING
  unsigned long long t_blocks = 0ull, n_blocks = 0ull, n_redone = 0ull, t_redo = 0ull;
#endif
  // one hand-scheduled block on 16 call words
  auto block = [&](const unsigned int (&ww)[kPipeBlock]) __attribute__((always_inline)) {
    const unsigned int base0 = base, s10 = s1, hd0 = hd, had0 = had;
    unsigned int na = ds_off + n, had2 = had << 1;
#if TFC_PIPE_TIMING
    const unsigned long long tb0 = clock64();
#endif
    asm volatile(
        "s_mov_b64 s[56:57], exec\n\t"
        "v_mov_b32 v101, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v119, 2\n\t"
        TFC_PENC_STEP_H(W0) TFC_PENC_STEP_H(W1) TFC_PENC_STEP_H(W2) TFC_PENC_STEP_H(W3)
        TFC_PENC_STEP_H(W4) TFC_PENC_STEP_H(W5) TFC_PENC_STEP_H(W6) TFC_PENC_STEP_H(W7)
        TFC_PENC_STEP_H(W8) TFC_PENC_STEP_H(W9) TFC_PENC_STEP_H(W10) TFC_PENC_STEP_H(W11)
        TFC_PENC_STEP_H(W12) TFC_PENC_STEP_H(W13) TFC_PENC_STEP_H(W14) TFC_PENC_STEP_H(W15)
        "s_mov_b64 exec, s[56:57]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [BASE] "+v"(base), [S] "+v"(s1), [H] "+v"(hd), [HAD] "+v"(had2), [NA] "+v"(na)
        : [NOCALL] "s"(kPipeNoCall), [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu),
          [W0] "v"(ww[0]), [W1] "v"(ww[1]), [W2] "v"(ww[2]), [W3] "v"(ww[3]), [W4] "v"(ww[4]), [W5] "v"(ww[5]),
          [W6] "v"(ww[6]), [W7] "v"(ww[7]), [W8] "v"(ww[8]), [W9] "v"(ww[9]), [W10] "v"(ww[10]), [W11] "v"(ww[11]),
          [W12] "v"(ww[12]), [W13] "v"(ww[13]), [W14] "v"(ww[14]), [W15] "v"(ww[15])
        : "vcc", "memory", "s52", "s53", "s56", "s57", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v117", "v118", "v119");
#if TFC_PIPE_TIMING
    t_blocks += clock64() - tb0;
    ++n_blocks;
#endif
    // a lane that came in inside a run of 0xFFFF digits, or holds 0xFFFF behind some step of the block (it left the block
    // there: the held digit is still the one)
    if (__builtin_expect(!__any(rn != 0u || hd >= 0xFFFFu), 1)) {
      n = na - ds_off;
      had = had2 >> 1;
    } else {
#if TFC_PIPE_TIMING
      ++n_redone;
      const unsigned long long tr0 = clock64();
#endif
      // a digit 0xFFFF was shifted out (it opens a run a later carry may ripple through), or a lane came
      // in inside such a run: the block again from the saved state, call by call
      base = base0; s1 = s10; hd = hd0; had = had0;
#pragma nounroll
      for (unsigned int k = 0; k < kPipeBlock; ++k) {
        unsigned int word = ww[0];
#pragma unroll
        for (unsigned int q = 1; q < kPipeBlock; ++q) word = k == q ? ww[q] : word;
        const unsigned int hi = word >> 16;
        call(word & 0xFFFFu, hi == 0u ? 65536u : hi, word != kPipeNoCall);
      }
#if TFC_PIPE_TIMING
      t_redo += clock64() - tr0;
#endif
    }
  };
  // the call words of an iteration's first / second block out of their slot
  auto read_words = [&](unsigned int (&ww)[kPipeBlock], unsigned int slot, unsigned int half) {
    const uint4* const p = reinterpret_cast<const uint4*>(area + L::kCalls + slot * L::kCallSlot + L::kCallStride * lane + 64u * half);
#pragma unroll
    for (unsigned int c = 0; c < kPipeBlock / 4u; ++c) {
      const uint4 v = p[c];
      ww[4 * c] = v.x; ww[4 * c + 1] = v.y; ww[4 * c + 2] = v.z; ww[4 * c + 3] = v.w;
    }
  };
  // -> the slot's sequence word once it holds iteration b (or the helper has given up)
  // (the two waits are timed in TFC_PIPE_TIMING builds only — tools/chain_clock_probe.py: four clock reads per iteration
  // are ~7 cycles per row of a 177-cycle row)
  unsigned long long waited_words = 0, waited_digits = 0;
  auto wait_words = [&](unsigned int b) {
    unsigned int v;
#if TFC_PIPE_TIMING
    const unsigned long long c0 = clock64();
#endif
    while ((((v = sync[L::kSeq + b % L::kSlots]) & ~(L::kLast | L::kBail)) != b + 1u) && !(v & L::kBail)) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
#if TFC_PIPE_TIMING
    waited_words += clock64() - c0;
#endif
    return v;
  };

  unsigned int wa[kPipeBlock], wb[kPipeBlock];
  unsigned int seq = wait_words(0u);
  bool bail = (seq & L::kBail) != 0u;
  if (!bail) read_words(wa, 0u, 0u);
  // A look at a counter in LDS is a round trip this wave has nothing to cover with (~100 cycles, two per iteration of 32
  // rows): the storer's counter is read again only when the value read last does not free this iteration's slot (the
  // helpers are usually slots ahead), and the next iteration's sequence word is read in front of the block and looked at
  // behind it.
  unsigned int dig_done = 0u;
  for (unsigned int b = 0; !bail; ++b) {
    const bool last = (seq & L::kLast) != 0u;
    read_words(wb, b % L::kSlots, 1u);
    const unsigned int seq_early = last ? 0u : sync[L::kSeq + (b + 1u) % L::kSlots];
    // the digit slot of this iteration: free once the helper has stored the iteration that used it before
    if (dig_done + L::kDigSlots <= b) {
#if TFC_PIPE_TIMING
      const unsigned long long c0 = clock64();
#endif
      while ((dig_done = sync[L::kDigDone]) + L::kDigSlots <= b) __builtin_amdgcn_s_sleep(1);
#if TFC_PIPE_TIMING
      waited_digits += clock64() - c0;
#endif
    }
    const unsigned int dslot = b % L::kDigSlots;
    dstage = area + L::kDig + dslot * L::kDigSlot + L::kDigStride * lane;
    ds_off = area_off + L::kDig + dslot * L::kDigSlot + L::kDigStride * lane;
    n = 0u;
    block(wa);
    // (the block waited for `wb`: the slot has been read)
    if (lane == 0u) sync[L::kConsumed] = b + 1u;
    if (!last) {
      seq = ((seq_early & ~(L::kLast | L::kBail)) == b + 2u || (seq_early & L::kBail)) ? seq_early : wait_words(b + 1u);
      if (seq & L::kBail) { bail = true; break; }
      read_words(wa, (b + 1u) % L::kSlots, 0u);
    }
    block(wb);
    // hand the iteration's digits over: (position in the slab, bytes) next to them
    reinterpret_cast<uint2*>(area + L::kRec)[dslot * 64u + lane] = make_uint2(wpos, n);
    overflow |= (n != 0u && wpos + ((n + 15u) & ~15u) > pa.cap) ? 1u : 0u;
    wpos += n;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0u) sync[L::kDigPub] = b + 1u;
    if (__any(too_long)) {
      bail = true;
      if (lane == 0u) atomicOr(&pa.fallback[job], 1u);
      break;
    }
    if (last) break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0u) sync[L::kExit] = 1u;
  if (gi == 0u && lane == 0u) {
    g_pipe_clock[0] = clock64() - clk0;
    g_pipe_clock[1] = wall_clock64() - wall0;
    g_pipe_clock[4] = waited_words;
    g_pipe_clock[6] = waited_digits;
#if TFC_PIPE_TIMING
    g_enc_clock[0] = t_blocks; g_enc_clock[1] = n_blocks; g_enc_clock[2] = n_redone; g_enc_clock[3] = t_redo;
#endif
  }
  if (bail) return;
  if (live) {
    pa.stage_state[static_cast<size_t>(gi) * 64 + lane] = make_uint4(base, s1, (hd & 0xFFFFu) | (had << 31), rn);
    pa.stage_out[static_cast<size_t>(gi) * 64 + lane] = make_uint2(wpos, overflow);
  }
}

// The same chain as ONE wave that requests its call words and stores its digits itself: the workgroups of a small
// launch (a model step's few groups) — 64 threads and 5 KB of LDS find room on a CU that convolutions of other steps
// keep busy, where a chain workgroup with helpers (30 KB, four waves) waits for one to come free (bmshj2018 with
// steps in flight: 36.2 against 33.4 ms per step).  Runs behind the expansion on the caller's stream.
__global__ void __launch_bounds__(64) enc_chain_direct_kernel(const PipeChainJobs jobs, const PipeEncArgs pa) {
  // rows per iteration of the main loop: two hand-scheduled blocks, so that the rows requested in one iteration have a
  // whole iteration (~6 k cycles) to arrive
  constexpr unsigned int kRows = PipeEncChainLds::kRows;
  constexpr unsigned int kDigits = 2 * kRows;           // digit bytes staged per lane between two flushes (one digit per call at most)
  constexpr int kStride = lane_stride(kDigits + 8);
  __shared__ __attribute__((aligned(16))) unsigned char stage[64 * kStride];
  const unsigned int gi = blockIdx.x;
  const unsigned int job = gi / static_cast<unsigned int>(pa.groups_per_job);
  const unsigned int gidx = gi % static_cast<unsigned int>(pa.groups_per_job);
  if (__hip_atomic_load(&pa.fallback[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  const PipeChainJob& J = jobs.job[job];
  const unsigned int lane = threadIdx.x;
  const int64_t s = static_cast<int64_t>(gidx) * 64 + lane;
  const bool live = s < jobs.streams;

  uint4 st = live ? J.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
  unsigned int base = st.x, s1 = st.y, hd = st.z & 0xFFFFu, had = st.z >> 31, rn = st.w;
  lanes_pin(base, s1, hd, rn);

  const unsigned int ds_off = static_cast<unsigned int>(reinterpret_cast<size_t>(
                                  (__attribute__((address_space(3))) unsigned char*)stage)) + kStride * lane;
  unsigned char* const dstage = stage + kStride * lane;
  unsigned char* const out = J.chunk + (live ? s : 0) * static_cast<int64_t>(pa.cap);
  unsigned int wpos = 0u, n = 0u, overflow = 0u;

  auto put = [&](unsigned int d, bool on) {
    *reinterpret_cast<unsigned short*>(dstage + n) = __builtin_bswap16(static_cast<unsigned short>(d));
    n += on ? 2u : 0u;
  };
  auto flush = [&]() {
#pragma unroll
    for (unsigned int c = 0; c < kDigits / 16u; ++c) {
      if (16u * c < n) {
        uint2 v[2];
        v[0] = reinterpret_cast<const uint2*>(dstage)[2 * c];
        v[1] = reinterpret_cast<const uint2*>(dstage)[2 * c + 1];
        if (wpos + 16u * c + 16u <= pa.cap) lanes_gstore16(out + wpos + 16u * c, v[0], v[1]);
        else overflow = 1u;
      }
    }
    wpos += n;
    n = 0u;
  };
  auto put_run = [&](unsigned int fill, unsigned int bytes) {
    flush();
    for (unsigned int k = 0; k < bytes; k += 2u) {
      const unsigned short be = static_cast<unsigned short>(fill);
      if (wpos + 2u <= pa.cap) lanes_gstore_elem(reinterpret_cast<unsigned short*>(out + wpos), be);
      else overflow = 1u;
      wpos += 2u;
    }
  };
  // one coder call on [lo, hi) / 2^16 for the lanes with `act` — the generic call of range_lanes.h
  // (enc_lanes_kernel::call), every case of the held-digit bookkeeping
  auto call = [&](unsigned int lo, unsigned int hi, bool act) __attribute__((always_inline)) {
    const unsigned int a = scale16(s1, lo);
    const unsigned int b = scale16(s1, hi) - 1u;
    const unsigned int bs = base + a;
    const unsigned int t1 = b - a;
    const bool carry = act && bs < a;
    const bool ren = act && (t1 >> 16) == 0u;
    const unsigned int e = bs >> 16;
    const bool solid = ren && e != 0xFFFFu;
    const bool ffff = ren && e == 0xFFFFu;
    const bool held = had != 0u;
    const unsigned int X = (hd + (carry ? 1u : 0u)) & 0xFFFFu;
    const bool emit = held && (solid || (carry && (!ffff || rn != 0u)));
    put(X, emit);
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

  // ---- what the expansion has released so far --------------------------------------------------------
  const unsigned int nt = static_cast<unsigned int>(pa.nt);
  const unsigned int* const status = pa.status + static_cast<size_t>(gi) * nt * 64 + lane;
  const unsigned int* const done = pa.done + static_cast<size_t>(gi) * nt;
  unsigned int avail = 0u;      // rows of this lane's stream that are in memory; all of them once tn == nt
  unsigned int tn = 0u;         // tiles [0, tn) are in
  bool bail = false;
  // rows [0, need) of every stream in memory (or all the stream has)
  auto ensure = [&](unsigned int need) {
    while (tn < nt && __any(live && avail < need)) {
      const unsigned int* const flag = done + tn;
      long long t0 = 0;
      bool timing = false;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        const long long now = static_cast<long long>(wall_clock64());
        if (!timing) { t0 = now; timing = true; }
        else if (now - t0 > pa.poll_ticks) { bail = true; return; }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      avail = __hip_atomic_load(status + static_cast<size_t>(tn) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kPipeRowsMask;
      ++tn;
      // a tile that did not fit, or too many escape codes somewhere in the job: nothing of it is kept
      if (__hip_atomic_load(&pa.fallback[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bail = true; return; }
    }
  };

  // (an iteration's words of the group are 8 KB in a row, [stream][32]: this lane's are 128 bytes of them)
  const unsigned char* const mine = reinterpret_cast<const unsigned char*>(pa.calls + static_cast<size_t>(gi) * 64 * pa.rows) + 128u * lane;
  // The rows of iteration b.  Requested a whole iteration before they are used, and taken over (`w = wn`) in front of
  // the iteration's own loads and stores: hipcc turns any wait for a load into s_waitcnt vmcnt(0) while a store may be
  // in flight (gfx9 counts both on vmcnt), so the one wait of an iteration has to sit where everything in flight
  // is an iteration old.
  unsigned int w[kRows] = {}, wn[kRows] = {};
  auto load_rows = [&](unsigned int b) {
    const unsigned char* p = mine + static_cast<size_t>(b) * (64u * kRows * 4u);
#pragma unroll
    for (unsigned int c = 0; c < kRows / 4u; ++c) {
      const uint4 v = lanes_gload16(p + 16u * c);
      wn[4 * c] = v.x; wn[4 * c + 1] = v.y; wn[4 * c + 2] = v.z; wn[4 * c + 3] = v.w;
    }
  };
  // one hand-scheduled block on 16 call words
  auto block = [&](unsigned int w0, unsigned int w1, unsigned int w2, unsigned int w3, unsigned int w4, unsigned int w5,
                   unsigned int w6, unsigned int w7, unsigned int w8, unsigned int w9, unsigned int w10, unsigned int w11,
                   unsigned int w12, unsigned int w13, unsigned int w14, unsigned int w15) __attribute__((always_inline)) {
    const unsigned int base0 = base, s10 = s1, hd0 = hd, had0 = had;
    unsigned int flag = rn, na = ds_off + n;
    asm volatile(
        "s_mov_b64 s[56:57], exec\n\t"
        "v_mov_b32 v101, 0\n\tv_mov_b32 v103, 0\n\t"
        TFC_PENC_STEP(W0) TFC_PENC_STEP(W1) TFC_PENC_STEP(W2) TFC_PENC_STEP(W3)
        TFC_PENC_STEP(W4) TFC_PENC_STEP(W5) TFC_PENC_STEP(W6) TFC_PENC_STEP(W7)
        TFC_PENC_STEP(W8) TFC_PENC_STEP(W9) TFC_PENC_STEP(W10) TFC_PENC_STEP(W11)
        TFC_PENC_STEP(W12) TFC_PENC_STEP(W13) TFC_PENC_STEP(W14) TFC_PENC_STEP(W15)
        "s_mov_b64 exec, s[56:57]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [BASE] "+v"(base), [S] "+v"(s1), [H] "+v"(hd), [HAD] "+v"(had), [NA] "+v"(na), [FLAG] "+v"(flag)
        : [NOCALL] "s"(kPipeNoCall), [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x0c0c0001u),
          [W0] "v"(w0), [W1] "v"(w1), [W2] "v"(w2), [W3] "v"(w3), [W4] "v"(w4), [W5] "v"(w5),
          [W6] "v"(w6), [W7] "v"(w7), [W8] "v"(w8), [W9] "v"(w9), [W10] "v"(w10), [W11] "v"(w11),
          [W12] "v"(w12), [W13] "v"(w13), [W14] "v"(w14), [W15] "v"(w15)
        : "vcc", "memory", "s52", "s53", "s56", "s57", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v117", "v118", "v119");
    if (__builtin_expect(!__any(flag != 0u), 1)) {
      n = na - ds_off;
    } else {
      // a digit 0xFFFF was shifted out (it opens a run a later carry may ripple through), or a lane came
      // in inside such a run: the block again from the saved state, call by call
      base = base0; s1 = s10; hd = hd0; had = had0;
      const unsigned int ww[kPipeBlock] = {w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15};
#pragma nounroll
      for (unsigned int k = 0; k < kPipeBlock; ++k) {
        unsigned int word = ww[0];
#pragma unroll
        for (unsigned int q = 1; q < kPipeBlock; ++q) word = k == q ? ww[q] : word;
        const unsigned int hi = word >> 16;
        call(word & 0xFFFFu, hi == 0u ? 65536u : hi, word != kPipeNoCall);
      }
    }
  };

  ensure(kRows);
  if (!bail) load_rows(0u);
  // (waited for here: a value that may still be in flight when the loop is entered would put a wait in front of
  // the blocks in every iteration)
#pragma unroll
  for (unsigned int k = 0; k < kRows; ++k) asm volatile("" : "+v"(wn[k]));
  // (while tiles are outstanding every stream has the whole iteration's rows — ensure() — and more to come)
  for (unsigned int b = 0; !bail && (tn < nt || __any(b * kRows < avail)); ++b) {
#pragma unroll
    for (unsigned int k = 0; k < kRows; ++k) w[k] = wn[k];
    ensure((b + 2u) * kRows);
    if (bail) break;
    load_rows(b + 1u);            // (one iteration's rows are allocated behind the last stream's)
    flush();                      // the digits of the iteration before this one
    if (__any(avail < (b + 1u) * kRows)) {
      // the last rows of the shorter streams (every tile is in by now): "no call" behind a stream's last
#pragma unroll
      for (unsigned int k = 0; k < kRows; ++k) w[k] = b * kRows + k < avail ? w[k] : kPipeNoCall;
    }
    block(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]);
    block(w[16], w[17], w[18], w[19], w[20], w[21], w[22], w[23], w[24], w[25], w[26], w[27], w[28], w[29], w[30], w[31]);
  }
  flush();
  if (bail) {
    if (lane == 0) atomicOr(&pa.fallback[job], 1u);
    return;
  }
  if (live) {
    pa.stage_state[static_cast<size_t>(gi) * 64 + lane] = make_uint4(base, s1, (hd & 0xFFFFu) | (had << 31), rn);
    pa.stage_out[static_cast<size_t>(gi) * 64 + lane] = make_uint2(wpos, overflow);
  }
}


// In front of the expansion, on its stream, when the chain's workgroups take a CU each: returns once they are all
// running (or `ticks` of wall_clock64() have passed).  Launched side by side, the expansion's thirty thousand
// workgroups would fill every CU first and a chain workgroup — it needs a CU with nothing else on it — would only be
// placed when they are through.
__global__ void enc_gate_kernel(const unsigned int* started, unsigned int want, long long ticks) {
  const long long t0 = static_cast<long long>(wall_clock64());
  while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want &&
         static_cast<long long>(wall_clock64()) - t0 < ticks)
    __builtin_amdgcn_s_sleep(8);
}

// Behind the expansion and the chain: the staged successor states become the handles' — unless the job fell back
// (then the lane-per-stream kernel codes it from the untouched state).
__global__ void __launch_bounds__(64) enc_commit_kernel(const PipeChainJobs jobs, const PipeEncArgs pa) {
  const unsigned int gi = blockIdx.x;
  const unsigned int job = gi / static_cast<unsigned int>(pa.groups_per_job);
  const unsigned int gidx = gi % static_cast<unsigned int>(pa.groups_per_job);
  if (pa.fallback[job] != 0u) return;
  const int64_t s = static_cast<int64_t>(gidx) * 64 + threadIdx.x;
  if (s >= jobs.streams) return;
  const PipeChainJob& J = jobs.job[job];
  J.state[s] = pa.stage_state[static_cast<size_t>(gi) * 64 + threadIdx.x];
  const uint2 o = pa.stage_out[static_cast<size_t>(gi) * 64 + threadIdx.x];
  // (a stream that outgrew its slab kept counting without storing: its piece gets length 0, like the wave family's
  // guard, so that finalize packs nothing from behind the slab; the handle is flagged and coded again by the caller)
  J.chunk_len[s] = o.y ? 0u : o.x;
  if (o.y) atomicOr(J.overflow_flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------

// Raw-plane entry of a row in which a lane made no step (it waits for the wave's tail phase, see dec_chain_kernel):
// not the start of an element, and skipped when an escape code's bit rows are collected
constexpr unsigned int kPipeSkipRow = 0xFFFFu;
// A row's entry, 16 bits (round 5; 32 before: the raw plane was the largest intermediate of a decode launch):
//   element rows (M = 0)   the symbol (< 2^15: rows of at most 32 767 symbols take these kernels)
//   bit rows (M != 0)      0x8000 | bit — the mode counter itself is not stored: dec_parse_kernel walks a code's bit rows
//                          in order and steps the counter as the chain did (pipe_mode_step)
//   kPipeSkipRow           0xFFFF (a bit row's low bits are 0 or 1)
__device__ inline unsigned int pipe_raw_entry(unsigned int sym_or_bit, int M) { return (M != 0 ? 0x8000u : 0u) | sym_or_bit; }
// the mode counter behind a bit row (see TFC_PDEC_STEP): M < 0 unary prefix, -M - 1 zeros seen; M > 0 calls to go
__device__ inline int pipe_mode_step(int M, unsigned int bit) {
  if (M < 0) return bit ? -M : (M == -31 ? 32 : M - 1);      // the 31st zero: the prefix ends here (range_lanes.h, bit_step)
  return M - 1;
}

struct PipeDecArgs {
  unsigned short* raw;           // [group][row][lane]: every step's entry (pipe_raw_entry; M: see dec_chain_kernel)
  unsigned int* posrec;          // [group][block + 1][lane]: elements started before the block
  unsigned int* kend;            // [group]: rows written (a multiple of kPipeBlock)
  uint4* state_out;              // [group][lane]: successor state, committed by dec_parse_kernel
  const unsigned short* rowaddr; // index mode: [job][stream][element] LDS address of the element's directory entry
  unsigned int* fallback;        // [job]
  unsigned int* progress;        // [group]: rows (and their block records) the chain has released | kPipeFinal at its end
  unsigned int* tile_done;       // [group][tile of kParseRows rows]: the parse running next to the chain has taken it
  unsigned int* started;         // chain workgroups that are running (enc_gate_kernel)
  int rows;                      // row capacity of a group (multiple of kPipeBlock)
  int groups_per_job;
  int groups;                    // of the launch
  int concurrent;                // dec_parse_kernel: 1 next to the chain (tiles as they are released), 0 behind it (the rest)
  long long poll_ticks;          // wall_clock64() ticks a concurrent parse workgroup watches its group make no progress before it gives up
};
constexpr unsigned int kPipeFinal = 0x80000000u;
#ifndef TFC_PIPE_RELEASE
#define TFC_PIPE_RELEASE 64
#endif
constexpr unsigned int kPipeRelease = TFC_PIPE_RELEASE;      // blocks between two releases of the chain's rows
struct PipeDecJob { const uint8_t* blob; const long long* off; const uint4* state; };
struct PipeDecJobs {
  int64_t streams, elems;
  int blocks_per_job, n;
  PipeDecJob job[64];
};

// Index mode, stage 0: table index -> LDS address of its directory entry (16 bytes per entry, directory at
// LDS address 0), range-checked like EntropyDecodeIndex (range_coder_kernels.cc:388-393).
struct PipeRowJobs {
  int64_t per_job;               // streams * elems
  int ntab, n;
  struct { const int32_t* index; unsigned long long* first_error; } job[64];
};
__global__ void __launch_bounds__(256) dec_rows_kernel(const PipeRowJobs jobs, unsigned short* rowaddr) {
  const int k = blockIdx.y;
  const int32_t* index = jobs.job[k].index;
  unsigned short* out = rowaddr + static_cast<size_t>(k) * jobs.per_job;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x, e = min(jobs.per_job, (static_cast<int64_t>(blockIdx.x) + 1) * 1024);
       i < e; i += 256) {
    int t = tfc_gload(index + i);
    if (t < 0 || t >= jobs.ntab) {
      atomicMin(jobs.job[k].first_error, static_cast<unsigned long long>(i));
      t = 0;
    }
    out[i] = static_cast<unsigned short>(16 * t);
  }
}

// LDS of one decoder wave behind the tables' image: per lane a code-byte window and (index mode) a window of
// row addresses, two blocks' worth each (a step consumes at most one 16-bit digit and one element)
struct PipeDecLds {
  // (two blocks of 2-byte entries and the one entry a step requests ahead, in whole 16-byte loads)
  static constexpr int kWords = 2 * kPipeBlock * 2 / 8 + 2;
  static constexpr int kStride = lane_stride(8 * kWords);
  static constexpr int kCodes = 0;
  static constexpr int kRows = kCodes + 64 * kStride;
  static constexpr int kBytes = kRows + 64 * kStride;
  // behind the windows (channel mode: behind the code window): the raw entries of a block, [row][lane] dwords
  static constexpr int kStage = kPipeBlock * 128;
};

// ---- one decoder step of every lane, hand-scheduled -------------------------------------------------
// The step of range_lanes.h (TFC_LDEC_STEP: quotient estimate -> rank in the row's boundary bitmap -> exact
// bounds = verification -> successor state) on the lane's CURRENT ROW R0..R3 (cdf - 2, info, bits, cum), plus
// the per-lane mode counter M:
//   M = 0   the step decodes an element of the lane's table row; an escape symbol sets M = -1
//   M < 0   unary prefix of an Elias-gamma code, -M - 1 zeros seen: a zero decrements M, a one leaves
//           M = -M calls to go (the magnitude's lower bits, then the sign)
//   M > 0   M binary calls to go
// (the arithmetic and its place in the step: "Round 5" below.)
// A lane with M' = 0 has completed an element: its row pointer PW moves on and the next step's row is the next
// element's directory entry; with M' != 0 the next row is the built-in binary row.  Every step writes its raw entry
// (pipe_raw_entry: the symbol, or 0x8000 | bit on the binary row) to row K of the wave's staging area.  FLAG: the
// verification failed (estimate one off, ~1e-6, or damaged input) — the caller then repeats the block from its saved
// state step by step.  Fixed temporaries v104-v142; v123 = v125 = 0.
// TFC_PDEC_ABL (build switch, bits, timing experiments only — results are wrong): 1 the step stores nothing, 2 one load per
// window request instead of five, 4 no flush of the raw rows
#ifndef TFC_PDEC_ABL
#define TFC_PDEC_ABL 0
#endif
// A step's raw entry goes to the wave's staging area in LDS ([row of the block][lane], flushed by the memory phase).
#if TFC_PDEC_ABL & 1
#define TFC_PDEC_STORE(KOFF)
#define TFC_PDEC_WAIT2 "s_waitcnt lgkmcnt(1)\n\t"
#else
#define TFC_PDEC_STORE(KOFF) "ds_write_b16 %[STG], v117 offset:" #KOFF "\n\t"
#define TFC_PDEC_WAIT2 "s_waitcnt lgkmcnt(2)\n\t"
#endif
// Round 5, first pass: the schedule of a step.  The wave issues one instruction per ~4 cycles whatever it is, and the two
// LDS round trips (quotient -> bitmap word + count; symbol -> cdf entries) have to be covered by instructions that do not
// need what they return.  Round 4's step derived the mode counter from the SYMBOL, which is only known after the first
// trip: 15 of its 21 bookkeeping instructions sat in the second trip's shadow, 6 in the first's, and the wave waited
// ~80 + ~40 cycles.  But whether a step's symbol is the escape symbol (or, on the binary row, which bit it is) is a
// comparison of the QUOTIENT with one boundary of the row — q >= cdf[escape symbol] (q >= 2^(p-1) on the binary row) <=>
// rank(q) is the last symbol — and the rank structure returns exactly rank(q_est), so the comparison on q_est gives the
// bookkeeping the very symbol class the verification then checks.  The row's boundary (ESCLO) sits in the upper half of
// the directory entry's info word (dec_chain_kernel rewrites the directory of its LDS copy), the whole mode arithmetic
// needs nothing the trips return, and the step spreads it and the row select over both shadows; the verification and the
// code cursor's increment are taken in the NEXT step's first shadow, the digit's byte swap is folded into the v_perm that
// renormalises D.  The ROW of the next step is one LDS read of the directory entry at the SELECTED address (M' = 0: the
// next element's entry, else the binary row's) straight into v104-v107, where a block keeps the current row; the mode
// counter alternates between two registers (MI -> MO); and the "31 zeros in a prefix" test (damaged input) is the
// caller's, once per block: a lane that enters a block with M > -16 cannot get to -32 inside it.
// LDS operations of a step, in issue order (they complete in order): bitmap word, count | (index mode: next row
// address) digit | cdf lo, cdf hi, the raw entry's write, the next step's row.
// Round 5, second pass (tools/r05_step_pad.sh: two dummy instructions cost 8.0 cycles in ANY of the step's four parts —
// the waits are gone, a step is its instruction count at 4 cycles each), 54 -> 45 instructions:
//  * the quotient estimate is the EXACT quotient's float image: the symbol of an offset D is the rank of
//    q* = ceil((D + 1) 2^p / (S + 1)) - 1 among the row's boundaries (D >= ((S + 1) c) >> 16  <=>  c < (D + 1) 2^p / (S + 1)),
//    and floor(float((D + 1) 2^p) / float(S + 1)) differs from it across a boundary for 1.3e-6 of the (D, S) of config 2's
//    tables — (D + 1/2) 2^p / S, round 4's estimate, for 3.1e-5 (it is off by up to 2^p / 2 S, 1 / 32 at the smallest span;
//    simulated, and measured: 76 of 3 086 blocks repeated, ~9 000 cycles each = 14 cycles per row) — for one more
//    instruction (S + 1 in float);
//  * it is kept NEGATED, nq = ~q = trunc(-(D + 1) 2^p / (S + 1) - 1): its low six bits are the left shift that
//    brings the quotient's bit of the bitmap word to the top (no v_not), ~q >> 6 = ~w addresses the word and its count
//    downwards from the entry's pointers (v_mad_i32_i24 with -8 / -2; the kernel's LDS copy of the directory holds bits - 8
//    and cum - 2), and q >= ESCLO is (~q & 0xFFFF) <= 0xFFFF - ESCLO on 16-bit halves (the info word's upper half holds
//    0xFFFF - ESCLO);
//  * no clamp of the quotient: an offset at the very top of the span estimates 2^p = bit 0 of the word BEHIND the row's
//    bitmap — the next row's first word (bit 0 cleared, below) or the spare word behind the last row — and every row's
//    counts have an entry for that word (tfc_tables_create);
//  * the counts are unsigned and the bitmaps leave out cdf[0] (the kernel clears bit 0 of a row's first word in its LDS
//    copy and sets that word's count to 0), and the binary row's counts carry 0x8000: the rank there is 0x8000 | bit — the
//    raw entry as it is stored, no flag register, no or — with the row's cdf pointer 0x10000 lower;
//  * mode counter: with t = M - 1, X = M >> 31 (all ones inside a unary prefix), z = -[M = 0]:  M' = t ^ (e ? X : z)
//    (M = 0: -1 ^ 0 opens a prefix, -1 ^ -1 stays; M < 0: ~(M - 1) = -M ends the prefix; M > 0: M - 1).  t' and z' of the
//    next step come out of the v_subrev_co that also yields the "M' = 0" condition of the row select (its borrow);
//  * the verification through EXEC: v_cmpx keeps the lanes whose offset lies inside the symbol's interval, a lane that
//    fails drops out of the rest of the block (the caller restores its state anyway); the block's head sets FLAG and its
//    tail clears it for the lanes still there.
// With these the two shadows (8 and 7 instructions) are as long as the LDS trips they cover, within two instructions (two
// dummy instructions in the first shadow cost nothing, anywhere else 8 cycles): moving the row select's five instructions
// from the second shadow into the first made the step 10 cycles slower (the second trip exposed), three of them 5 cycles,
// one nothing — a step is now  quotient 11 + rank 6 + bounds 13 instructions + the two trips: 264.5 cycles (291.8 before
// this pass); a wait that waits for nothing costs ~3 cycles.
// Measured and dropped: 1 / S issued four instructions early (no gain: nothing waits for v_rcp_f32), the two bounds as
// three v_mad_u32_u16 / shift / v_mad_u32_u16 each instead of v_mad_u64_u32 + v_alignbit (+8 cycles: the 64-bit multiply
// costs one slot like everything else).
// TFC_PDEC_PAD = 1 .. 4 (build switch, tools/r05_step_pad.sh): two extra instructions in the quotient part, the first
// shadow, the second shadow, the bounds part of every step — which parts of a step are bound by instruction issue
#ifndef TFC_PDEC_PAD
#define TFC_PDEC_PAD 0
#endif
#define TFC_PDEC_PAD2 "v_mov_b32 v139, v139\n\tv_mov_b32 v139, v139\n\t"
#if TFC_PDEC_PAD == 1
#define TFC_PDEC_PAD_A TFC_PDEC_PAD2
#elif TFC_PDEC_PAD == 5
#define TFC_PDEC_PAD_A "s_waitcnt lgkmcnt(15)\n\ts_waitcnt lgkmcnt(15)\n\t"      /* (what a wait that waits for nothing costs) */
#else
#define TFC_PDEC_PAD_A
#endif
#if TFC_PDEC_PAD == 2
#define TFC_PDEC_PAD_B TFC_PDEC_PAD2
#else
#define TFC_PDEC_PAD_B
#endif
#if TFC_PDEC_PAD == 3
#define TFC_PDEC_PAD_C TFC_PDEC_PAD2
#else
#define TFC_PDEC_PAD_C
#endif
#if TFC_PDEC_PAD == 4
#define TFC_PDEC_PAD_D TFC_PDEC_PAD2
#else
#define TFC_PDEC_PAD_D
#endif
// Registers across steps: v104-v107 the row, v135 = M - 1, v136 = -[M = 0], v130 / v131 the previous step's pending
// verification, s[56:57] its "renormalised" condition (the code cursor moves on in the next step's first shadow, which has
// two slots to spare; the bounds part has none); v123 = v125 = 0.
#define TFC_PDEC_STEP(KOFF, MI, MO, AHEAD, NEXT, PWSTEP)                                    \
  "v_cvt_f32_u32 v111, %[S]\n\t"                                                          \
  "v_add_f32 v111, 1.0, v111\n\t"                                                         \
  "v_rcp_f32 v111, v111\n\t"                                                              \
  TFC_PDEC_PAD_A                                                                          \
  "v_cvt_f32_u32 v110, %[D]\n\t"                                                          \
  "v_fma_f32 v110, -v110, %[SCALE], -%[HSCALE]\n\t"                                       \
  "v_fma_f32 v110, v110, v111, -1.0\n\t"                                                  \
  "v_cvt_i32_f32 v110, v110\n\t"                                                          \
  "v_ashrrev_i32 v111, 6, v110\n\t"                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                              \
  "v_mad_i32_i24 v112, v111, -8, v106\n\t"                                                \
  "ds_read_b64 v[114:115], v112\n\t"                                                      \
  "v_mad_i32_i24 v113, v111, -2, v107\n\t"                                                \
  "ds_read_u16 v116, v113\n\t"                                                            \
  AHEAD                                                                                   \
  "v_cndmask_b32_e64 v133, 0, 2, s[56:57]\n\t"                                            \
  "v_add_u32 %[CP], %[CP], v133\n\t"                                                      \
  "ds_read_u16 v109, %[CP]\n\t"                                                           \
  "v_ashrrev_i32 v137, 31, " MI "\n\t"                                                    \
  "v_cmp_le_u32_sdwa vcc, v110, v105 src0_sel:WORD_0 src1_sel:WORD_1\n\t"                 \
  "v_cndmask_b32 v137, v136, v137, vcc\n\t"                                               \
  "v_xor_b32 " MO ", v135, v137\n\t"                                                      \
  "v_cmpx_le_u32 vcc, v130, v131\n\t"                                                     \
  TFC_PDEC_PAD_B                                                                          \
  "s_waitcnt lgkmcnt(1)\n\t"                                                              \
  "v_lshlrev_b64 v[118:119], v110, v[114:115]\n\t"                                        \
  "v_bcnt_u32_b32 v116, v118, v116\n\t"                                                   \
  "v_bcnt_u32_b32 v117, v119, v116\n\t"                                                   \
  "v_lshl_add_u32 v112, v117, 1, v104\n\t"                                                \
  "ds_read_u16 v122, v112 offset:2\n\t"                                                   \
  "ds_read_u16 v124, v112 offset:4\n\t"                                                   \
  TFC_PDEC_PAD_C                                                                          \
  TFC_PDEC_STORE(KOFF)                                                                    \
  "v_subrev_co_u32 v135, vcc, 1, " MO "\n\t"                                              \
  "v_cndmask_b32_e64 v136, 0, -1, vcc\n\t"                                                \
  "v_add_u32 v140, " #PWSTEP ", %[PW]\n\t"                                                \
  "v_cndmask_b32 %[PW], %[PW], v140, vcc\n\t"                                             \
  "v_cndmask_b32 v142, %[BINROW], " NEXT ", vcc\n\t"                                      \
  "ds_read_b128 v[104:107], v142\n\t"                                                     \
  TFC_PDEC_WAIT2                                                                          \
  TFC_PDEC_PAD_D                                                                          \
  "v_mad_u64_u32 v[126:127], s[52:53], v122, %[S], v[122:123]\n\t"                        \
  "v_mad_u64_u32 v[128:129], s[52:53], v124, %[S], v[124:125]\n\t"                        \
  "v_alignbit_b32 v126, v127, v126, 16\n\t"                                               \
  "v_alignbit_b32 v128, v129, v128, 16\n\t"                                               \
  "v_add_u32 v128, -1, v128\n\t"                                                          \
  "v_min_u32 v128, v128, %[S]\n\t"                                                        \
  "v_sub_u32 v131, v128, v126\n\t"                                                        \
  "v_cmp_gt_u32_e64 s[56:57], %[K64K], v131\n\t"                                          \
  "v_lshl_or_b32 v132, v131, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32_e64 %[S], v131, v132, s[56:57]\n\t"                                      \
  "v_sub_u32 v130, %[D], v126\n\t"                                                        \
  "v_perm_b32 v132, v130, v109, %[PERM]\n\t"                                              \
  "v_cndmask_b32_e64 %[D], v130, v132, s[56:57]\n\t"
// Round 6: the same step on the COMPACT image (tfc_tables_create, "Compact image"): the bitmaps mark every SECOND bound of
// a row, one bit per PAIR of quotient values — half the bytes.  The rank i among them says the symbol is one of three:
// the window cdf[k - 1 .. k + 2], k = 2 i + o, comes with ONE ds_read2_b32 (the directory's cdf pointer is the window of
// i = 0, windows are four bytes apart) as W0 = cdf[k - 1] | cdf[k] << 16, W1 = cdf[k + 1] | cdf[k + 2] << 16, and two
// comparisons of Q = q << (16 - p) with the two middle entries (SDWA picks the halves) slide it:
//     t0 = [Q >= cdf[k]], t1 = [Q >= cdf[k + 1]]:   (lower, upper) = t1 ? (W1.lo, W1.hi) : t0 ? (W0.hi, W1.lo) : (W0.lo, W0.hi)
// — four SDWA selects straight into the registers the bounds multiply from — and the raw entry is 2 i + t0 + t1 (two
// add-with-carry; its store moves behind the trip).  Nine instructions more than TFC_PDEC_STEP: nq >> 1 (the pair's place
// in its word), Q (first shadow), 2 + 4 + 2 behind the second trip, one LDS read less.  The interval test of the next step
// verifies the choice like every estimate.  %[NSH] = -(1 << (16 - p)).
// Measured (TFC_PIPE_TIMING builds, profiles/r06_notes.md): 320.0 cycles per row inside the blocks against 265.3 of
// TFC_PDEC_STEP — the nine instructions at the ~6 cycles a dependent instruction of this chain costs; Q formed in the
// second shadow instead: 323.0; the window as two ds_read_b32: 320.0 (one instruction more, free: the trip is not what
// is waited for); as one ds_read_b64 at its 4-byte alignment: 372.7.
#define TFC_PDEC_STEP_H(KOFF, MI, MO, AHEAD, NEXT, PWSTEP)                                  \
  "v_cvt_f32_u32 v111, %[S]\n\t"                                                          \
  "v_add_f32 v111, 1.0, v111\n\t"                                                         \
  "v_rcp_f32 v111, v111\n\t"                                                              \
  "v_cvt_f32_u32 v110, %[D]\n\t"                                                          \
  "v_fma_f32 v110, -v110, %[SCALE], -%[HSCALE]\n\t"                                       \
  "v_fma_f32 v110, v110, v111, -1.0\n\t"                                                  \
  "v_cvt_i32_f32 v110, v110\n\t"                                                          \
  "v_ashrrev_i32 v143, 1, v110\n\t"                                                       \
  "v_ashrrev_i32 v111, 6, v143\n\t"                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                              \
  "v_mad_i32_i24 v112, v111, -8, v106\n\t"                                                \
  "ds_read_b64 v[114:115], v112\n\t"                                                      \
  "v_mad_i32_i24 v113, v111, -2, v107\n\t"                                                \
  "ds_read_u16 v116, v113\n\t"                                                            \
  AHEAD                                                                                   \
  "v_cndmask_b32_e64 v133, 0, 2, s[56:57]\n\t"                                            \
  "v_add_u32 %[CP], %[CP], v133\n\t"                                                      \
  "ds_read_u16 v109, %[CP]\n\t"                                                           \
  "v_ashrrev_i32 v137, 31, " MI "\n\t"                                                    \
  "v_cmp_le_u32_sdwa vcc, v110, v105 src0_sel:WORD_0 src1_sel:WORD_1\n\t"                 \
  "v_cndmask_b32 v137, v136, v137, vcc\n\t"                                               \
  "v_xor_b32 " MO ", v135, v137\n\t"                                                      \
  "v_cmpx_le_u32 vcc, v130, v131\n\t"                                                     \
  "v_mad_i32_i24 v141, v110, %[NSH], %[NSH]\n\t"                                          \
  "s_waitcnt lgkmcnt(1)\n\t"                                                              \
  "v_lshlrev_b64 v[118:119], v143, v[114:115]\n\t"                                        \
  "v_bcnt_u32_b32 v116, v118, v116\n\t"                                                   \
  "v_bcnt_u32_b32 v117, v119, v116\n\t"                                                   \
  "v_lshl_add_u32 v112, v117, 2, v104\n\t"                                                \
  "ds_read2_b32 v[144:145], v112 offset1:1\n\t"                                           \
  "v_subrev_co_u32 v135, vcc, 1, " MO "\n\t"                                              \
  "v_cndmask_b32_e64 v136, 0, -1, vcc\n\t"                                                \
  "v_add_u32 v140, " #PWSTEP ", %[PW]\n\t"                                                \
  "v_cndmask_b32 %[PW], %[PW], v140, vcc\n\t"                                             \
  "v_cndmask_b32 v142, %[BINROW], " NEXT ", vcc\n\t"                                      \
  "ds_read_b128 v[104:107], v142\n\t"                                                     \
  "s_waitcnt lgkmcnt(1)\n\t"                                                              \
  "v_cmp_ge_u32_sdwa vcc, v141, v144 src0_sel:DWORD src1_sel:WORD_1\n\t"                  \
  "v_cndmask_b32_sdwa v122, v144, v144, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
  "v_cndmask_b32_sdwa v124, v144, v145, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t" \
  "v_addc_co_u32_e64 v117, s[58:59], v117, v117, vcc\n\t"                                 \
  "v_cmp_ge_u32_sdwa vcc, v141, v145 src0_sel:DWORD src1_sel:WORD_0\n\t"                  \
  "v_cndmask_b32_sdwa v122, v122, v145, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" \
  "v_cndmask_b32_sdwa v124, v124, v145, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
  "v_addc_co_u32_e64 v117, s[58:59], v117, v123, vcc\n\t"                                 \
  TFC_PDEC_STORE(KOFF)                                                                    \
  "v_mad_u64_u32 v[126:127], s[52:53], v122, %[S], v[122:123]\n\t"                        \
  "v_mad_u64_u32 v[128:129], s[52:53], v124, %[S], v[124:125]\n\t"                        \
  "v_alignbit_b32 v126, v127, v126, 16\n\t"                                               \
  "v_alignbit_b32 v128, v129, v128, 16\n\t"                                               \
  "v_add_u32 v128, -1, v128\n\t"                                                          \
  "v_min_u32 v128, v128, %[S]\n\t"                                                        \
  "v_sub_u32 v131, v128, v126\n\t"                                                        \
  "v_cmp_gt_u32_e64 s[56:57], %[K64K], v131\n\t"                                          \
  "v_lshl_or_b32 v132, v131, 16, %[KFFFF]\n\t"                                            \
  "v_cndmask_b32_e64 %[S], v131, v132, s[56:57]\n\t"                                      \
  "v_sub_u32 v130, %[D], v126\n\t"                                                        \
  "v_perm_b32 v132, v130, v109, %[PERM]\n\t"                                              \
  "v_cndmask_b32_e64 %[D], v130, v132, s[56:57]\n\t"
// channel mode: the next element's entry is the one behind the current (PW + 16); index mode: its address comes out
// of the lane's window of row addresses, requested in the first shadow
#define TFC_PDEC_STEP_CH(KOFF, MI, MO) TFC_PDEC_STEP(KOFF, MI, MO, "", "v140", 16)
#define TFC_PDEC_STEP_IX(KOFF, MI, MO) TFC_PDEC_STEP(KOFF, MI, MO, "ds_read_u16 v108, %[PW] offset:2\n\t", "v108", 2)
#define TFC_PDEC_STEP_PCH(KOFF, MI, MO) TFC_PDEC_STEP_H(KOFF, MI, MO, "", "v140", 16)
#define TFC_PDEC_STEP_PIX(KOFF, MI, MO) TFC_PDEC_STEP_H(KOFF, MI, MO, "ds_read_u16 v108, %[PW] offset:2\n\t", "v108", 2)
// a block: the row into v104-v107, nothing pending from a step before it (v130 <= v131, s[56:57] = 0); behind it the last
// step's pending verification and cursor increment, and the row the next step decodes from back to the caller
#define TFC_PDEC_STEP2(STEP, K0, K1) STEP(K0, "%[M]", "v138") STEP(K1, "v138", "%[M]")
#define TFC_PDEC_BLOCK_HEAD                                                               \
  "v_mov_b32 v104, %[R0]\n\tv_mov_b32 v105, %[R1]\n\tv_mov_b32 v106, %[R2]\n\tv_mov_b32 v107, %[R3]\n\t" \
  "v_mov_b32 v123, 0\n\tv_mov_b32 v125, 0\n\t"                                           \
  "v_mov_b32 v130, 0\n\tv_mov_b32 v131, 0\n\ts_mov_b64 s[56:57], 0\n\t"                   \
  "v_subrev_co_u32 v135, vcc, 1, %[M]\n\tv_cndmask_b32_e64 v136, 0, -1, vcc\n\t"           \
  "s_mov_b64 s[54:55], exec\n\tv_mov_b32 %[FLAG], 1\n\t"
#define TFC_PDEC_BLOCK_TAIL                                                               \
  "v_cndmask_b32_e64 v133, 0, 2, s[56:57]\n\t"                                            \
  "v_add_u32 %[CP], %[CP], v133\n\t"                                                      \
  "s_waitcnt lgkmcnt(0)\n\t"                                                              \
  "v_cmpx_le_u32 vcc, v130, v131\n\tv_mov_b32 %[FLAG], 0\n\ts_mov_b64 exec, s[54:55]\n\t"   \
  "v_mov_b32 %[R0], v104\n\tv_mov_b32 %[R1], v105\n\tv_mov_b32 %[R2], v106\n\tv_mov_b32 %[R3], v107\n\t"
// one step as a block of its own (its raw entry at %[STG] + 0): what a block whose verification failed is repeated with,
// step by step, so that only the step that fails again takes the generic path
#define TFC_PDEC_ONE(STEP) TFC_PDEC_BLOCK_HEAD STEP(0, "%[M]", "v138") "v_mov_b32 %[M], v138\n\t" TFC_PDEC_BLOCK_TAIL
#define TFC_PDEC_BLOCK(STEP)                                                              \
  TFC_PDEC_BLOCK_HEAD                                                                     \
  TFC_PDEC_STEP2(STEP, 0, 128) TFC_PDEC_STEP2(STEP, 256, 384) TFC_PDEC_STEP2(STEP, 512, 640)            \
  TFC_PDEC_STEP2(STEP, 768, 896) TFC_PDEC_STEP2(STEP, 1024, 1152) TFC_PDEC_STEP2(STEP, 1280, 1408)      \
  TFC_PDEC_STEP2(STEP, 1536, 1664) TFC_PDEC_STEP2(STEP, 1792, 1920)                       \
  TFC_PDEC_BLOCK_TAIL

// PAIRS: `la.image` is the COMPACT image (tfc_tables_create builds it in this kernel's final form) and the steps are
// TFC_PDEC_STEP_P; else the lane-per-stream kernels' image, patched below, and TFC_PDEC_STEP.
template <bool INDEXED, bool PAIRS>
__global__ void __launch_bounds__(512) dec_chain_kernel(const PipeDecJobs jobs, const LaneArgs la, const PipeDecArgs pa) {
  extern __shared__ unsigned char lanes_lds[];
  lanes_load_image(lanes_lds, la);
  using L = PipeDecLds;
  if constexpr (!PAIRS) {
    // This kernel's form of the image (its LDS copy only; the image on the device is shared with the lane-per-stream
    // kernels) — see "Round 5, second pass" above:
    //  * tables: bit 0 of a row's first bitmap word (cdf[0] = 0) cleared and that word's count 0 instead of -1: the counts
    //    are unsigned, rank = count + popcount as before; the binary row's counts + 0x8000 (its rank is the raw entry
    //    0x8000 | bit) and its cdf pointer 0x10000 lower;
    //  * directory: info = limit | has_escape << 15 | (0xFFFF - ESCLO) << 16, ESCLO = the escape symbol's lower bound on the
    //    tables' own scale (0xFFFF, which no quotient reaches at precision <= 15, for a row without one; 2^(p-1) for the
    //    binary row behind the repeated entries: there q >= ESCLO is the decoded bit); bits - 8 and cum - 2 (the step
    //    addresses a word from the NEGATED quotient).
    const unsigned int entries = static_cast<unsigned int>(la.ntab) + kLaneDirRepeat + 1u;
    const unsigned int nw = max(1u, (1u << la.precision) >> 6) + 1u;      // counts of a row: its words and the word behind it
    for (unsigned int i = threadIdx.x; i < entries; i += blockDim.x) {
      if (i >= static_cast<unsigned int>(la.ntab) && i + 1u != entries) continue;     // (repeated entries: tables patched already)
      const LaneRow* e = reinterpret_cast<const LaneRow*>(lanes_lds) + i;
      *reinterpret_cast<unsigned long long*>(lanes_lds + e->bits) &= ~1ull;
      *reinterpret_cast<unsigned short*>(lanes_lds + e->cum) = 0;
    }
    __syncthreads();
    {
      const LaneRow* e = reinterpret_cast<const LaneRow*>(lanes_lds) + (entries - 1u);
      for (unsigned int w = threadIdx.x; w < nw; w += blockDim.x)
        *reinterpret_cast<unsigned short*>(lanes_lds + e->cum + 2u * w) += 0x8000u;
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < entries; i += blockDim.x) {
      LaneRow* e = reinterpret_cast<LaneRow*>(lanes_lds) + i;
      const unsigned int info = e->info, limit = info & 0x7FFFu, esc = info >> 31;
      unsigned int esclo = 0xFFFFu;
      if (i + 1u == entries) esclo = 1u << (la.precision - 1);
      else if (esc) esclo = lds_u16(lanes_lds, e->cdf + 2u * limit + 2u) >> (16 - la.precision);
      e->info = limit | (esc << 15) | ((0xFFFFu - esclo) << 16);
      e->bits -= 8u;
      e->cum -= 2u;
      if (i + 1u == entries) e->cdf -= 0x10000u;
    }
    __syncthreads();
  }

  const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
  if (threadIdx.x == 0u) __hip_atomic_fetch_add(pa.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned int job = blockIdx.x / static_cast<unsigned int>(jobs.blocks_per_job);
  const unsigned int wv = (blockIdx.x % static_cast<unsigned int>(jobs.blocks_per_job)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wv >= static_cast<unsigned int>(pa.groups_per_job)) return;
  const unsigned int gi = job * static_cast<unsigned int>(pa.groups_per_job) + wv;
  const PipeDecJob& J = jobs.job[job];
  const unsigned int lane = threadIdx.x & 63u;
  const int64_t s = static_cast<int64_t>(wv) * 64 + lane;
  const bool live = s < jobs.streams;
  const unsigned int elems = live ? static_cast<unsigned int>(jobs.elems) : 0u;

  const uint4 st = live ? J.state[s] : make_uint4(0u, 0xFFFFFFFFu, 0u, 2u);
  unsigned int D = st.z - st.x;      // window - base
  unsigned int s1 = st.y;            // span - 1
  const long long o0 = live ? J.off[s] : 0;
  unsigned int len = live ? static_cast<unsigned int>(J.off[s + 1] - o0) : 0u;
  unsigned int pos_start = 2u * st.w;
  lanes_pin(D, s1, len, pos_start);

  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) unsigned char*)lanes_lds));
  const unsigned int wave_off = static_cast<unsigned int>(la.lds_image) + (threadIdx.x >> 6) * static_cast<unsigned int>(la.lds_wave);
  LaneWindow<L::kWords> cw;
  const unsigned int cw_off = wave_off + L::kCodes + L::kStride * lane;
  cw.lds = lanes_lds + cw_off;
  cw.g = J.blob + o0;
  cw.len = len;
  cw.request(pos_start);
  cw.base = pos_start;
  unsigned int cp = cw_off;          // LDS offset of the next code digit: stream position cw.base + (cp - cw_off)
  LaneWindow<L::kWords> iw;          // index mode: row addresses, 2 bytes per element
  const unsigned int iw_off = wave_off + L::kRows + L::kStride * lane;
  if (INDEXED) {
    iw.lds = lanes_lds + iw_off;
    iw.g = reinterpret_cast<const unsigned char*>(pa.rowaddr + (static_cast<size_t>(job) * jobs.streams + (live ? s : 0)) * jobs.elems);
    iw.len = elems * 2u;
    iw.request(0u);
    iw.base = 0u;
  }

  // Raw entries of a block are collected in LDS and leave as whole lines in the NEXT memory phase, in front of its
  // window requests: a store issued inside the block sat between the window's loads and the wait for them (vmcnt counts
  // both in order), so every block waited for its own sixteen stores to reach memory (~110 cycles per row outside the
  // hand-scheduled steps, tools/chain_clock_probe.py on a TFC_PIPE_TIMING build), and a store cost its step ~20 cycles.
  const unsigned int stg_off = wave_off + (INDEXED ? L::kBytes : L::kRows);
  const unsigned int stg = stg_off + 2u * lane;
  bool staged = false;               // wave-uniform: the staging area holds the rows staged_k ... + kPipeBlock
  unsigned int staged_k = 0u;

  const float scale = static_cast<float>(1u << la.precision);
  float hscale = scale;              // (the quotient estimate's numerator is (D + 1) 2^p: see the step)
  asm volatile("" : "+v"(hscale));
  const unsigned int cp_max = (1u << la.precision) - 1u;
  // compact image: the quotient on the entries' 2^16 scale, Q = q << (16 - p) = (nq + 1) * -(1 << (16 - p)) (TFC_PDEC_STEP_H)
  const unsigned int pair_sh = 16u - static_cast<unsigned int>(la.precision);
  const unsigned int pair_nsh = 0u - (1u << pair_sh);
  const unsigned int dir_end = 16u * static_cast<unsigned int>(la.ntab);
  // the built-in binary row: directory entry behind the repeated ones
  const unsigned int bin_addr = dir_end + 16u * kLaneDirRepeat;
  uint4 bin = *reinterpret_cast<const uint4*>(lanes_lds + bin_addr);
  asm volatile("" : "+v"(bin.x), "+v"(bin.y), "+v"(bin.z), "+v"(bin.w));
  unsigned int bin_addr_v = bin_addr;     // (in vector registers: v_cndmask takes one scalar operand, and that is vcc)
  asm volatile("" : "+v"(bin_addr_v));

  unsigned int pos = 0u;             // elements completed
  int M = 0;
  // pw: channel mode, LDS offset of the directory entry of element `pos`; index mode, LDS offset of its row address
  unsigned int pw = INDEXED ? iw_off : 0u;
  uint4 R = make_uint4(0u, 0u, 0u, 0u);   // the row the next step decodes from (loaded behind the first memory phase)
  bool row_loaded = false;

  unsigned short* const raw = pa.raw + static_cast<size_t>(gi) * pa.rows * 64;
  unsigned int* const posrec = pa.posrec + static_cast<size_t>(gi) * (pa.rows / kPipeBlock + 1) * 64 + lane;
  unsigned int k = 0u;               // rows written
  auto flush = [&]() {
    if (!staged) return;
#if TFC_PDEC_ABL & 4
    staged = false;                  // (timing experiment, results wrong: the raw rows stay in LDS)
    return;
#endif
    unsigned char* dst = reinterpret_cast<unsigned char*>(raw + static_cast<size_t>(staged_k) * 64) + 16u * lane;
#pragma unroll
    for (unsigned int j = 0; j < L::kStage / 1024u; ++j) {
      const uint4 v = *reinterpret_cast<const uint4*>(lanes_lds + stg_off + 1024u * j + 16u * lane);
      lanes_gstore16(dst + 1024u * j, make_uint2(v.x, v.y), make_uint2(v.z, v.w));
    }
    staged = false;
  };

  // the row of element `pos` (channel mode: pw is kept inside the directory)
  auto load_row = [&]() {
    const unsigned int a = INDEXED ? static_cast<unsigned int>(*reinterpret_cast<const unsigned short*>(lanes_lds + pw)) : pw;
    R = *reinterpret_cast<const uint4*>(lanes_lds + a);
  };

  // one generic step of the lanes with `act`: any mode, any exception
  auto gstep = [&](bool act, unsigned int row, bool to_stage = false) {
    unsigned int entry = 0u;
    if (act) {
      const unsigned int dig = __builtin_bswap16(*reinterpret_cast<const unsigned short*>(lanes_lds + cp));
      if (M == 0) {
        // ---- symbol first: quotient estimate -> rank among the row's boundaries (range_lanes.h) ----
        const float fq = (static_cast<float>(D) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(s1)) * scale;
        const unsigned int q = min(static_cast<unsigned int>(fq), cp_max);
        if constexpr (PAIRS) {
          // compact image: rank i among every second bound, at pair resolution -> the window's first candidate; the
          // exact test below then steps the symbol as on the full image (entry e of the window of rank 0 is at
          // R.x + 2 e: the original cdf entries, symbol s = entry s + 1 of that window minus the row's o - 1 ... the
          // raw entry counts from the window: 2 i + t0 + t1)
          const unsigned int j = q >> 1, w = j >> 6;
          const unsigned long long word = *reinterpret_cast<const unsigned long long*>(lanes_lds + R.z + 8u + 8u * w);
          const unsigned int cum = *reinterpret_cast<const unsigned short*>(lanes_lds + R.w + 2u + 2u * w);
          const unsigned long long below = ~0ull >> (63u - (j & 63u));
          const unsigned int i = cum + static_cast<unsigned int>(__popcll(word & below));
          const unsigned int Q = q << pair_sh;
          // u: entry index inside the row's window space (entry u = lower bound, u + 1 = upper bound)
          unsigned int u = 2u * i;
          u += Q >= lds_u16(lanes_lds, R.x + 2u * u + 2u) ? 1u : 0u;
          u += (u & 1u) != 0u && Q >= lds_u16(lanes_lds, R.x + 4u * i + 4u) ? 1u : 0u;
          unsigned int lo = lds_u16(lanes_lds, R.x + 2u * u), hi = lds_u16(lanes_lds, R.x + 2u * u + 2u);
          unsigned int A = scale16(s1, lo);
          unsigned int b = scale16(s1, hi) - 1u;
          b = hi == 0u ? s1 : b;
          // (u of symbol 0 and of the last symbol: the row's o - 1 is not in the image — found by the bounds themselves:
          // no step below an entry whose lower bound is 0 with a zero in front of it... the ends are where hi wraps or
          // lo is the pad's zero: stepping stops when the entry read is not a bound of the row)
          for (int fix = 0; fix < 4; ++fix) {
            if (D - A > b - A) {
              if (D < A) u = u > 0u ? u - 1u : 0u;
              else if (hi != 0u) u = u + 1u;
              lo = lds_u16(lanes_lds, R.x + 2u * u);
              hi = lds_u16(lanes_lds, R.x + 2u * u + 2u);
              A = scale16(s1, lo);
              b = scale16(s1, hi) - 1u;
              b = hi == 0u ? s1 : b;
            }
          }
          D -= A;
          s1 = b - A;
          const bool ren = (s1 >> 16) == 0u;
          D = ren ? (D << 16) | dig : D;
          s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
          cp += ren ? 2u : 0u;
          entry = u;
          // the escape symbol is the one whose lower bound is ESCLO (info's upper half holds 0xFFFF - ESCLO)
          M = ((R.y >> 15) & 1u) != 0u && lo == ((0xFFFFu - (R.y >> 16)) << pair_sh) ? -1 : 0;
        } else {
        const unsigned int w = q >> 6;
        // (this kernel's LDS copy: entry pointers 8 / 2 bytes low, counts without the "- 1", bitmaps without cdf[0])
        const unsigned long long word = *reinterpret_cast<const unsigned long long*>(lanes_lds + R.z + 8u + 8u * w);
        const unsigned int cum = *reinterpret_cast<const unsigned short*>(lanes_lds + R.w + 2u + 2u * w);
        const unsigned long long below = ~0ull >> (63u - (q & 63u));
        unsigned int sym = cum + static_cast<unsigned int>(__popcll(word & below));
        unsigned int lo = lds_u16(lanes_lds, R.x + 2u * sym + 2u);
        unsigned int hi = lds_u16(lanes_lds, R.x + 2u * sym + 4u);
        unsigned int A = scale16(s1, lo);
        unsigned int b = scale16(s1, hi) - 1u;
        b = hi == 0u ? s1 : b;
        const unsigned int nsym = (R.y & 0x7FFFu) + ((R.y >> 15) & 1u);
        for (int fix = 0; fix < 4; ++fix) {
          if (D - A > b - A) {
            if (D < A) sym = sym > 0u ? sym - 1u : 0u;
            else sym = sym + 1u < nsym ? sym + 1u : nsym - 1u;
            lo = lds_u16(lanes_lds, R.x + 2u * sym + 2u);
            hi = lds_u16(lanes_lds, R.x + 2u * sym + 4u);
            A = scale16(s1, lo);
            b = scale16(s1, hi) - 1u;
            b = hi == 0u ? s1 : b;
          }
        }
        D -= A;
        s1 = b - A;
        const bool ren = (s1 >> 16) == 0u;
        D = ren ? (D << 16) | dig : D;
        s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
        cp += ren ? 2u : 0u;
        entry = pipe_raw_entry(sym, 0);
        M = ((R.y >> 15) & 1u) != 0u && sym == (R.y & 0x7FFFu) ? -1 : 0;
        }
      } else {
        // ---- one bit of an Elias-gamma code (range_coder_kernels.cc:449-471): the uniform binary cdf at
        // precision 1 needs no table ------------------------------------------------------------------
        const unsigned int half = scale16(s1, 32768u);
        const unsigned int bit = D >= half ? 1u : 0u;
        const unsigned int A = bit ? half : 0u;
        const unsigned int b = bit ? s1 : half - 1u;
        D -= A;
        s1 = b - A;
        const bool ren = (s1 >> 16) == 0u;
        D = ren ? (D << 16) | dig : D;
        s1 = ren ? (s1 << 16) | 0xFFFFu : s1;
        cp += ren ? 2u : 0u;
        // (compact image: the fast step's entry of a bit row is 0x8000 + t0 + t1 = 0x8001 + bit)
        entry = PAIRS ? 0x8001u + bit : pipe_raw_entry(bit, M);
        if (M < 0) {
          if (bit) M = -M;
          else M = M == -31 ? 32 : M - 1;      // the 31st zero: the prefix ends here (range_lanes.h, bit_step)
        } else {
          --M;
        }
      }
      if (M == 0) {
        ++pos;
        pw += INDEXED ? 2u : 16u;
        if (!INDEXED && pw == dir_end) pw = 0u;
        load_row();
      } else {
        R = bin;
      }
      // (a block repeated inside the steady-state loop keeps its rows in the staging area, like the steps it replaces)
      if (to_stage) *reinterpret_cast<unsigned short*>(lanes_lds + stg + 128u * (row - k)) = static_cast<unsigned short>(entry);
      else raw[static_cast<size_t>(row) * 64 + lane] = static_cast<unsigned short>(entry);
    }
  };

  // Two phases.  While any lane has a whole block of elements left, the lanes that do take blocks (hand-scheduled
  // steps; a block never runs past a lane's last element) and a lane that does not — its stream is nearly done, it met
  // more escape codes than its neighbours and fell behind, or fewer and is ahead — sits them out, marking its rows
  // kPipeSkipRow.  Then the tail: every lane's last elements (fewer than a block each, plus their escape bits) with
  // the generic steps, all lanes together — so that the wave does not drop to the generic steps for as long as its
  // lanes are spread out, only for one short pass at the end.
#if TFC_PIPE_TIMING
  // (measurement aid, tools/chain_clock_probe.py: cycles inside the hand-scheduled blocks / their number)
  unsigned long long t_asm = 0ull, n_asm = 0ull, t_commit = 0ull, t_rest = 0ull;
#endif
  // memory phase of a block: park the windows requested at the previous phase, send the previous block's raw rows off,
  // request the windows from the current positions
  auto memory_phase = [&]() __attribute__((always_inline)) {
#if TFC_PIPE_TIMING
    const unsigned long long tm0 = clock64();
#endif
    const unsigned int cpos = cw.base + (cp - cw_off);
    cw.commit();
    cp = cw_off + (cpos - cw.base);
    if (INDEXED) {
      iw.commit();
      pw = iw_off + (2u * pos - iw.base);
    }
#if TFC_PIPE_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tm1 = clock64();
    t_commit += tm1 - tm0;
#endif
    flush();
    if (k != 0u && (k / kPipeBlock) % kPipeRelease == 0u) {
      // the rows so far (and their block records) to the parse running next to this kernel: a release costs the wave a
      // wait for its stores, so it is rare
      posrec[static_cast<size_t>(k / kPipeBlock) * 64] = pos + (M != 0 ? 1u : 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0u) __hip_atomic_store(&pa.progress[gi], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    cw.request(cpos);
    if (INDEXED) iw.request(2u * pos);
    // elements STARTED before row k (an escape code in progress has its first row behind us)
    posrec[static_cast<size_t>(k / kPipeBlock) * 64] = pos + (M != 0 ? 1u : 0u);
#if TFC_PIPE_TIMING
    t_rest += clock64() - tm1;
#endif
  };
  // kPipeBlock hand-scheduled steps of the lanes in EXEC
  auto fast_block = [&](unsigned int& flag) __attribute__((always_inline)) {
#define TFC_PDEC_OPERANDS                                                                                              \
        : [D] "+v"(D), [S] "+v"(s1), [CP] "+v"(cp), [M] "+v"(M), [PW] "+v"(pw), [FLAG] "+v"(flag),                    \
          [R0] "+v"(R.x), [R1] "+v"(R.y), [R2] "+v"(R.z), [R3] "+v"(R.w)                                              \
        : [STG] "v"(stg), [BINROW] "v"(bin_addr_v), [SCALE] "s"(scale), [HSCALE] "v"(hscale),        \
          [K64K] "s"(0x10000u), [KFFFF] "s"(0xFFFFu), [PERM] "s"(0x05040001u), [NSH] "s"(pair_nsh) \
        : "vcc", "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
          "v114", "v115", "v116", "v117", "v118", "v119", "v122", "v123", "v124", "v125", "v126", "v127", "v128",      \
          "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145"
#if TFC_PIPE_TIMING
    const unsigned long long ta = clock64();
#endif
    if constexpr (INDEXED && PAIRS) asm volatile(TFC_PDEC_BLOCK(TFC_PDEC_STEP_PIX) TFC_PDEC_OPERANDS);
    else if constexpr (PAIRS) asm volatile(TFC_PDEC_BLOCK(TFC_PDEC_STEP_PCH) TFC_PDEC_OPERANDS);
    else if constexpr (INDEXED) asm volatile(TFC_PDEC_BLOCK(TFC_PDEC_STEP_IX) TFC_PDEC_OPERANDS);
    else asm volatile(TFC_PDEC_BLOCK(TFC_PDEC_STEP_CH) TFC_PDEC_OPERANDS);
#if TFC_PIPE_TIMING
    t_asm += clock64() - ta;
    ++n_asm;
#endif
  };
  // one hand-scheduled step, its raw entry to row `slot` of the staging area
  auto fast_step = [&](unsigned int& flag, unsigned int slot) __attribute__((always_inline)) {
    const unsigned int stg_row = stg + 128u * slot;
    {
      const unsigned int stg = stg_row;       // (the operand list names `stg`)
      if constexpr (INDEXED && PAIRS) asm volatile(TFC_PDEC_ONE(TFC_PDEC_STEP_PIX) TFC_PDEC_OPERANDS);
      else if constexpr (PAIRS) asm volatile(TFC_PDEC_ONE(TFC_PDEC_STEP_PCH) TFC_PDEC_OPERANDS);
      else if constexpr (INDEXED) asm volatile(TFC_PDEC_ONE(TFC_PDEC_STEP_IX) TFC_PDEC_OPERANDS);
      else asm volatile(TFC_PDEC_ONE(TFC_PDEC_STEP_CH) TFC_PDEC_OPERANDS);
    }
#undef TFC_PDEC_OPERANDS
  };
  bool gave_up = false;
  while (__any(pos < elems)) {
    // Steady state — every stream of the wave has a whole block of elements left — as a loop of its own: a memory
    // phase, a hand-scheduled block, three wave-wide tests.  (Round 4 ran these blocks through the general body below,
    // whose tests, state copies and branches for lanes that sit out, finish an escape code or take generic steps cost
    // ~1 900 cycles per block of ~4 800 — tools/chain_clock_probe.py on a TFC_PIPE_TIMING build.)  Anything else — the
    // first block, a lane near its end, a failed verification, a plane that runs out — leaves it for the general body;
    // a block whose verification fails is repeated on the spot, step by step.
    if (lds0 == 0u && row_loaded && (INDEXED || static_cast<unsigned int>(la.ntab) >= kPipeBlock)) {
      // (M > -16: a lane's unary prefix cannot reach its 31st zero — where the reference stops counting — inside a block)
      while (__all(!live || (pos + kPipeBlock <= elems && M > -16)) && k + kPipeBlock <= static_cast<unsigned int>(pa.rows)) {
        memory_phase();
        const unsigned int D0 = D, s10 = s1, cp0 = cp, pw0 = pw;
        const int M0 = M;
        const uint4 R0 = R;
        unsigned int flag = 0u;
        // (every lane, also those of a last group that have no stream — they decode zeros from their own windows and
        // are ignored: under `if (live)` the compiler copied the block's nine operands twice)
        fast_block(flag);
        if (__builtin_expect(__any(live && flag != 0u), 0)) {
          // A verification failed somewhere in the block (the quotient estimate one off: ~1e-5 of the symbols, 2.5 % of the
          // blocks): again from the saved state, one hand-scheduled step at a time, and only the step that fails again
          // takes the generic path (16 generic steps are ~20 000 cycles; this is ~7 000).
          D = D0; s1 = s10; cp = cp0; pw = pw0; M = M0; R = R0;
#pragma nounroll
          for (unsigned int i = 0; i < kPipeBlock; ++i) {
            const unsigned int D1 = D, s11 = s1, cp1 = cp, pw1 = pw;
            const int M1 = M;
            const uint4 R1 = R;
            unsigned int f1 = 0u;
            if (live) fast_step(f1, i);
            if (__any(live && f1 != 0u)) {
              D = D1; s1 = s11; cp = cp1; pw = pw1; M = M1; R = R1;
              gstep(live, k + i, true);
            } else {
              pos += (pw - pw1) / (INDEXED ? 2u : 16u);
              if (!INDEXED) pw -= pw >= dir_end ? dir_end : 0u;
            }
          }
          staged = true;
          staged_k = k;
          k += kPipeBlock;
          continue;
        }
        pos += (pw - pw0) / (INDEXED ? 2u : 16u);
        if (!INDEXED) pw -= pw >= dir_end ? dir_end : 0u;      // (a block is at most one turn of the directory: ntab >= kPipeBlock)
        staged = true;
        staged_k = k;
        k += kPipeBlock;
      }
      if (!__any(pos < elems)) break;
    }
    if (k + kPipeBlock > static_cast<unsigned int>(pa.rows)) {
      gave_up = true;                // more rows than planned for (escape codes far beyond the tables' tail mass)
      break;
    }
    memory_phase();
    if (!row_loaded) {
      if (M == 0) load_row(); else R = bin;
      row_loaded = true;
    }
    const bool whole = pos + kPipeBlock <= elems;          // a whole block of elements left
    const bool blocks = __any(whole);                      // first phase
    if (blocks && __any(!whole && pos < elems && M != 0)) {
      // A lane that is about to sit out is inside an escape code: it finishes the code first (generic steps, everybody
      // else sits these rows out), so that a code's bit rows stay together in the raw plane.
      if (pos < elems) {
#pragma unroll
        for (unsigned int i = 0; i < kPipeBlock; ++i) raw[static_cast<size_t>(k + i) * 64 + lane] = static_cast<unsigned short>(kPipeSkipRow);
      }
#pragma nounroll
      for (unsigned int i = 0; i < kPipeBlock; ++i) gstep(!whole && pos < elems && M != 0, k + i);
      k += kPipeBlock;
      continue;
    }
    const bool busy = blocks ? whole : pos < elems;
    const bool sitting = blocks && !whole && pos < elems;      // this lane sits the block out
    if (__builtin_expect(lds0 == 0u && blocks && !__any(busy && M <= -16), 1)) {
      if (__any(sitting)) {
        if (sitting) {
#pragma unroll
          for (unsigned int i = 0; i < kPipeBlock; ++i) *reinterpret_cast<unsigned short*>(lanes_lds + stg + 128u * i) = static_cast<unsigned short>(kPipeSkipRow);
        }
      }
      const unsigned int D0 = D, s10 = s1, cp0 = cp, pw0 = pw;
      const int M0 = M;
      const uint4 R0 = R;
      unsigned int flag = 0u;
      if (busy) {
        fast_block(flag);
      }
      if (__builtin_expect(!__any(flag != 0u), 1)) {
        if (busy) {
          pos += (pw - pw0) / (INDEXED ? 2u : 16u);
        }
        // (fewer tables than rows in a block: the directory cursor wraps more than once)
        if (!INDEXED)
          while (__any(busy && pw >= dir_end)) pw -= pw >= dir_end ? dir_end : 0u;
        staged = true;
        staged_k = k;
        k += kPipeBlock;
        continue;
      }
      D = D0; s1 = s10; cp = cp0; pw = pw0; M = M0; R = R0;      // an exception somewhere in the wave: the generic steps
    }
    if (sitting) {
      // (the generic steps store their rows themselves: so do the lanes that sit them out)
#pragma unroll
      for (unsigned int i = 0; i < kPipeBlock; ++i) raw[static_cast<size_t>(k + i) * 64 + lane] = static_cast<unsigned short>(kPipeSkipRow);
    }
#pragma nounroll
    for (unsigned int i = 0; i < kPipeBlock; ++i) gstep(busy && pos < elems, k + i);
    k += kPipeBlock;
  }
  flush();
  posrec[static_cast<size_t>(k / kPipeBlock) * 64] = pos;
  if (gave_up) {
    if (lane == 0) {
      atomicOr(&pa.fallback[job], 1u);
      __hip_atomic_store(&pa.progress[gi], kPipeFinal, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (lane == 0) pa.kend[gi] = k;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) __hip_atomic_store(&pa.progress[gi], k | kPipeFinal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (gi == 0 && lane == 0) {
    g_pipe_clock[2] = clock64() - clk0;
    g_pipe_clock[3] = wall_clock64() - wall0;
#if TFC_PIPE_TIMING
    g_pipe_clock[5] = t_asm | (t_commit << 32);
    g_pipe_clock[7] = n_asm | (static_cast<unsigned long long>(k) << 16) | (t_rest << 32);
#endif
  }
  if (live) {
    // back to the (base, span - 1, window, digits pulled) form shared with the other kernels
    const unsigned int cpos = cw.base + (cp - cw_off);
    const unsigned char* srcp = J.blob + o0;
    unsigned int window = 0u;
    for (int i = -4; i < 0; ++i) {
      const long long q = static_cast<long long>(cpos) + i;
      window = (window << 8) | ((q >= 0 && q < static_cast<long long>(len)) ? srcp[q] : 0u);
    }
    pa.state_out[static_cast<size_t>(gi) * 64 + lane] = make_uint4(window - D, s1, window, cpos >> 1);
  }
}

// ---- stage 2: raw rows -> elements -------------------------------------------------------------------
// A workgroup takes kParseRows rows of one group (64 streams): rows in (coalesced) -> LDS -> a wave walks one
// stream's column 64 rows at a time; the rows that start an element (M = 0) are numbered by a ballot prefix from
// the block records of the chain, an escape symbol collects its value from the bit rows behind it, and the
// elements of a stream leave in order (coalesced along the stream).
constexpr int kParseRows = 128;
constexpr unsigned int kParseMargin = 96;     // rows behind a tile the parse next to the chain waits for

constexpr int kParseEscTables = 2048;     // escape symbols of up to this many tables are staged in LDS

// PAIRS: the chain ran on the compact image — an element row's entry is the symbol minus the row's adjust[t] (o - 1,
// tfc_tables_create), a bit row's entry 0x8001 + bit.
struct PipePairMap { const int* adjust; };
template <bool INDEXED, bool PAIRS, typename Dst>
__global__ void __launch_bounds__(256) dec_parse_kernel(const DecLaneJobs<Dst> jobs, const PipeDecArgs pa, const DecRow* dir, int ntab,
                                                        const PipePairMap pm) {
  // (16-bit entries at a pitch of 33 dwords: a wave's column walk — lane i reads row i — meets every bank twice, what 64
  // lanes cost anyway; 17 KB + the rows' escape symbols, so that two of these workgroups fit a CU beside a chain
  // workgroup of two waves on the compact image — 64 batches of BASELINE config 2 in one launch: the 32-bit buffer of
  // round 5, 33 KB, left room for one, and 5 ms of the parse ran behind the chain)
  constexpr int kPitch = 66;
  __shared__ unsigned short buf[kParseRows * kPitch];
  // escape symbol of the row, or -1, in the low half; compact image: the row's adjust + 2 above it
  __shared__ unsigned int escsym[kParseEscTables];
  __shared__ unsigned int incomplete;
  const unsigned int tiles = static_cast<unsigned int>(pa.rows) / kParseRows + 1u;
  // (tile-major: next to the chain, the tiles it releases first are dispatched first)
  const unsigned int gi = blockIdx.x % static_cast<unsigned int>(pa.groups), kt = blockIdx.x / static_cast<unsigned int>(pa.groups);
  const unsigned int job = gi / static_cast<unsigned int>(pa.groups_per_job);
  const unsigned int wv = gi % static_cast<unsigned int>(pa.groups_per_job);
  const DecLaneJob<Dst>& J = jobs.job[job];
  const unsigned int tid = threadIdx.x, lane = tid & 63u;
  const unsigned int k0 = kt * kParseRows;
  unsigned int* const tile_done = pa.tile_done + static_cast<size_t>(gi) * tiles + kt;
  unsigned int kend;               // rows that may be read
  bool final = true;               // ... are all the chain wrote
  if (pa.concurrent) {
    // Next to the chain: this tile once the chain has released its rows and a margin behind them (the bit rows of an
    // escape code that starts in the tile).  Rows that do not arrive in time, or a code that runs past the margin:
    // the tile is left to the pass behind the chain.
    const unsigned int need = k0 + kParseRows + kParseMargin;
    unsigned int* const abandon = pa.started + 1;      // a workgroup saw its group stand still: the chain is not running next to us
    // ONE thread watches the flags and the whole workgroup takes its verdict (round 5).  Every thread used to poll for
    // itself: a workgroup that started while another one was raising `abandon` (or whose threads read `progress` on both
    // sides of a release at their time-out) went on with SOME of its threads, whose missing neighbours then never loaded
    // their columns of the tile — wrong elements for 16-64 streams of one tile, seen once a launch had more groups than
    // the chip has CUs (two rounds of chain workgroups: the first pass is abandoned after 2 ms).
    unsigned int v = 0u;
    if (tid == 0u) {
      unsigned int seen = ~0u, go = 1u;
      long long t0 = 0;
      if (__hip_atomic_load(abandon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) go = 0u;
      while (go) {
        v = __hip_atomic_load(&pa.progress[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v & kPipeFinal) || v >= need) break;
        const long long now = static_cast<long long>(wall_clock64());
        if (v != seen) { seen = v; t0 = now; }
        else if (now - t0 > pa.poll_ticks) {
          __hip_atomic_store(abandon, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          go = 0u;
          break;
        }
        __builtin_amdgcn_s_sleep(32);
      }
      incomplete = go ? v : 0xFFFFFFFFu;     // (the verdict travels in `incomplete`, which the tile's parse clears below)
    }
    __syncthreads();
    v = incomplete;
    __syncthreads();
    if (v == 0xFFFFFFFFu) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (__hip_atomic_load(&pa.fallback[job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    final = (v & kPipeFinal) != 0u;
    kend = v & ~kPipeFinal;
    if (tid == 0u) incomplete = 0u;
  } else {
    if (pa.fallback[job] != 0u) return;
    if (kt == 0u && tid < 64u) {
      // the chain's successor states become the handle's (nothing is committed when the job fell back)
      const int64_t s = static_cast<int64_t>(wv) * 64 + tid;
      if (s < jobs.streams) J.state[s] = pa.state_out[static_cast<size_t>(gi) * 64 + tid];
    }
    if (*tile_done != 0u) return;
    kend = pa.kend[gi];
  }
  if (k0 >= kend) return;
  const unsigned short* const raw = pa.raw + static_cast<size_t>(gi) * pa.rows * 64;
  const unsigned int nrows = min(static_cast<unsigned int>(kParseRows), kend - k0);
  for (unsigned int r = tid >> 6; r < nrows; r += 4u) buf[r * kPitch + lane] = raw[static_cast<size_t>(k0 + r) * 64 + lane];
  const bool esc_lds = ntab <= kParseEscTables;
  if (esc_lds)
    for (int i = tid; i < ntab; i += 256) {
      escsym[i] = (static_cast<unsigned int>(dir[i].w) & 0xFFFFu) | (PAIRS ? static_cast<unsigned int>(pm.adjust[i] + 2) << 16 : 0u);
    }
  __syncthreads();
  const unsigned int bitshift = PAIRS ? 1u : 0u;
  const unsigned int* const posrec = pa.posrec + static_cast<size_t>(gi) * (pa.rows / kPipeBlock + 1) * 64;
  const Dst dst = J.dst;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned int untab = static_cast<unsigned int>(ntab);
  // (the loop exists twice — escape symbols in LDS, or read from the directory: as a run-time condition inside it the
  // directory's load met the LDS path in front of the element's store, where hipcc then waits with s_waitcnt vmcnt(0) —
  // also for the stores of the iteration before; and the bit rows behind the tile are read through a global-address-space
  // pointer: a FLAT load anywhere in the loop makes every wait of the loop a wait for everything.  Round 6.)
  auto walk = [&](auto in_lds) __attribute__((always_inline)) {
  constexpr bool esc_in_lds = decltype(in_lds)::value;
  for (unsigned int l = tid >> 6; l < 64u; l += 4u) {
    const int64_t s = static_cast<int64_t>(wv) * 64 + l;
    if (s >= jobs.streams) continue;
    unsigned int p = posrec[static_cast<size_t>(k0 / kPipeBlock) * 64 + l];
    const unsigned int p1 = posrec[static_cast<size_t>((k0 + nrows) / kPipeBlock) * 64 + l];
    const int64_t base = s * jobs.elems;
    unsigned int pmod = p % untab;                 // channel mode: table of element p
    for (unsigned int c = 0; c < nrows; c += 64u) {
      const unsigned int i = c + lane;
      const unsigned int e = i < nrows ? buf[i * kPitch + l] : kPipeSkipRow;
      const bool start = (e & 0x8000u) == 0u;
      const unsigned long long mask = __ballot(start);
      const unsigned int before = static_cast<unsigned int>(__popcll(mask & lt));
      const unsigned int ord = p + before;
      const unsigned int total = static_cast<unsigned int>(__popcll(mask));
      if (start && ord < p1) {
        const int64_t at = base + ord;
        int t;
        if (INDEXED) {
          t = tfc_gload(J.index + at);
          t = (t < 0 || t >= ntab) ? 0 : t;
        } else {
          unsigned int m = pmod + before;          // < ntab + 64
          if (untab >= 64u) m -= m >= untab ? untab : 0u;
          else m %= untab;
          t = static_cast<int>(m);
        }
        int v = static_cast<int>(e);
        const unsigned int ew = esc_in_lds ? escsym[t] : 0u;
        if constexpr (PAIRS) v += esc_in_lds ? static_cast<int>(ew >> 16) - 2 : pm.adjust[t];
        // escape symbol of the row, or -1
        const int es = esc_in_lds ? static_cast<int>(static_cast<short>(ew & 0xFFFFu)) : dir[t].w;
        if (v == es) {
          // the rows behind an escape symbol carry its Elias-gamma code: M < 0 the unary prefix (-M - 1 zeros
          // before the row), M > 0 calls to go (M = 1: the sign; bit M - 2 of the magnitude otherwise)
          unsigned int val = 0u;
          bool neg = false, closed = false;
          int m = -1;                             // the mode counter in front of the row (the escape symbol left -1)
          for (unsigned int q = k0 + i + 1u; q < kend; ++q) {
            const unsigned int x = q - k0 < nrows ? buf[(q - k0) * kPitch + l]
                                                  : *reinterpret_cast<const TFC_AS1 unsigned short*>((const TFC_AS1 void*)(raw + static_cast<size_t>(q) * 64 + l));
            const unsigned int bit = (x >> bitshift) & 1u;
            if (x == kPipeSkipRow) continue;      // the lane sat this row out
            if (m < 0) {
              if (bit) val = 1u << (-m - 1);
              else if (m == -31) val = 1u << 31;
            } else if (m > 1) {
              val |= bit << (m - 2);
            } else {
              neg = bit != 0u;
              closed = true;
              break;
            }
            m = pipe_mode_step(m, bit);
          }
          if (!closed && !final) incomplete = 1u;      // the code's last rows are not released yet: the pass behind the chain
          v = neg ? -static_cast<int>(val) : static_cast<int>(val) + es - 1;
        }
        dst.store(at, t, v);
      }
      p += total;
      pmod = (pmod + total) % untab;
    }
  }
  };
  if (esc_lds) walk(std::true_type{}); else walk(std::false_type{});
  if (pa.concurrent) {
    __syncthreads();
    if (tid == 0u && incomplete == 0u) *tile_done = 1u;     // (read by the pass behind the chain: another kernel)
  }
}

}  // namespace tfc
