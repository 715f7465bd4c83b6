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

// span - 1 is kept unnormalised-aware: after a renormalisation span = (t + 1) << 16 exactly, so
// the next bound (span * hi) >> 16 is just t * hi + hi with NO shift.  Keeping t and the shift
// amount (instead of the shifted value) takes the compare + select that would build
// (t << 16) | 0xFFFF out of the per-symbol dependency chain: the multiply starts from t as soon
// as it is read from the winning lane, and the shift amount (decided by one scalar bit test)
// is only needed after the multiply.
struct FastDecState {   // wave-uniform
  unsigned int D;       // window - base
  unsigned int t;       // span - 1 = sh ? t : (t << 16) | 0xFFFF
  unsigned int sh;      // 16: t is span - 1 itself; 0: the span was just renormalised
  unsigned int pos;     // digits consumed since the window register was loaded
};

__device__ inline unsigned int span_minus1(const FastDecState& st) {
  return st.sh ? st.t : ((st.t << 16) | 0xFFFFu);
}
__device__ inline void set_span_minus1(FastDecState& st, unsigned int s) {
  st.t = s;
  st.sh = 16u;
}

struct DecWindow {
  const uint8_t* src;
  long long len;
  unsigned int wbase;   // digit index held by lane 0 of reg
  int reg;              // digits wbase .. wbase + 63
  // digits wbase + 64 .. wbase + 127, requested one batch ahead and kept as RAW bytes: anything
  // computed from a load result makes the compiler wait for the load on the spot, so the bytes are
  // only combined (window_next) when the following batch stitches its window.
  unsigned int next_hi, next_lo;
  bool next_hi_ok, next_lo_ok;
};

// Branch-free on purpose: with `b < len ? src[b] : 0` the compiler puts each byte load in its own
// block and waits for it at the join.  Addresses are clamped into the stream instead and the
// value is masked afterwards (in window_next / fast_window_fetch).
__device__ inline void fast_window_request(DecWindow& w, unsigned int first, int lane) {
  const long long b = 2ll * (static_cast<long long>(first) + lane);
  const long long last = w.len - 1;                       // w.src is readable on [0, max(len, 1))
  w.next_hi = w.src[b < last ? b : (last < 0 ? 0 : last)];
  w.next_lo = w.src[b + 1 < last ? b + 1 : (last < 0 ? 0 : last)];
  w.next_hi_ok = b < w.len;
  w.next_lo_ok = b + 1 < w.len;
}
__device__ inline int window_next(const DecWindow& w) {
  return static_cast<int>(((w.next_hi_ok ? w.next_hi : 0u) << 8) | (w.next_lo_ok ? w.next_lo : 0u));
}

// Synchronous (re)load of both registers at wbase.
__device__ inline void fast_window_load(DecWindow& w, int lane) {
  fast_window_request(w, w.wbase, lane);
  w.reg = window_next(w);
  fast_window_request(w, w.wbase + 64u, lane);
}

// Advances the window by `shift` (0..64) consumed digits without waiting for HBM: the new
// register is stitched from reg/next with two ds_bpermute, and the following 64 digits are
// requested now, to be used a whole batch later.
__device__ inline void fast_window_advance(DecWindow& w, unsigned int shift, int lane) {
  const int idx = (lane + static_cast<int>(shift)) & 63;
  const int from_reg = __builtin_amdgcn_ds_bpermute(idx << 2, w.reg);
  const int from_next = __builtin_amdgcn_ds_bpermute(idx << 2, window_next(w));
  const bool in_reg = lane + static_cast<int>(shift) < 64;
  // shift == 64 moves `next` into place unchanged (idx == lane)
  w.wbase += shift;
  w.reg = in_reg ? from_reg : from_next;
  fast_window_request(w, w.wbase + 64u, lane);
}

// ---- candidate-per-lane selection --------------------------------------------------------
// One wave per SIMD.  Measured on MI355X (tools/ubench/dec_step.hip, profiles/r01_f_dec_step.txt):
//   * every instruction occupies ~4.1 cycles of the wave whatever its type, s_nop wait states
//     included, so the chain is written for instruction COUNT;
//   * an SALU instruction reading an SGPR that a VALU instruction wrote (v_cmp mask, v_readlane)
//     costs +16 cycles that other VALU work cannot hide, and v_readlane whose LANE SELECT was
//     written by SALU costs +20.  The textbook "ballot, s_ff1, v_readlane" selection pays both.
// Hence: the first candidate that contains the offset is found through EXEC — v_cmpx makes the
// hit lanes the active ones, v_readfirstlane reads the winner's successor state (and its lane
// id), one s_mov restores EXEC — and NOTHING coming out of it is ever touched by SALU
// arithmetic: the digit position, the renormalisation shift and the fine-stage table address
// are all carried per lane and read the same way.
//   * B = (span * hi) >> 16 with span = s + 1, as ONE v_mad_u64_u32 (s * hi + hi) + v_alignbit.
//     The addend is made opaque so that LLVM does not refactor it into (s + 1) * hi, which
//     needs a 33-bit multiplicand and two multiplies.
//   * A = B of the previous lane: v_mov_b32_dpp wave_shr:1 bound_ctrl:1 (lane 0 reads 0),
//     written by hand — LLVM's DPP combine folds the builtin into
//     `v_subrev_u32_dpp ... wave_shr:1 bound_ctrl:1`, which returned wrong lanes (ROCm 7.2;
//     caught by the K3/K4/K6 vectors).  Two instructions sit between the write of B and the
//     DPP read (the wait states it needs).
//   * b = B - 1 is formed explicitly: for s = 2^32 - 1 and hi = 2^16 the bound B = 2^32 wraps
//     to 0, and only D <= B - 1 / span' - 1 = (B - 1) - A survive that (K3 pins this).
// The semantics of the hand-written instructions are checked on the device by
// tools/ubench/asm_semantics.hip.
__device__ inline unsigned long long scale_product(unsigned int s, unsigned int hi) {
  unsigned int add = hi;
  asm volatile("" : "+v"(add));
  return static_cast<unsigned long long>(s) * hi + add;
}

struct Bounds { unsigned int b, A, t1, posv; };   // per lane; posv = digit position, broadcast

// a0 = lower offset of lane 0's candidate (0 when ZERO_A0: a whole row).
template <bool ZERO_A0>
__device__ inline Bounds bounds_step(const FastDecState& st, unsigned int hi, unsigned int a0) {
  const unsigned long long P = scale_product(st.t, hi);
  const unsigned int plo = static_cast<unsigned int>(P), phi = static_cast<unsigned int>(P >> 32);
  Bounds o;
  unsigned int B;
  if (ZERO_A0) {
    asm("v_alignbit_b32 %0, %5, %6, %8\n\t"
        "v_add_u32 %1, -1, %0\n\t"
        "v_mov_b32 %4, %7\n\t"
        "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32 %3, %1, %2"
        : "=&v"(B), "=&v"(o.b), "=&v"(o.A), "=&v"(o.t1), "=&v"(o.posv)
        : "v"(phi), "v"(plo), "s"(st.pos), "s"(st.sh));
  } else {
    asm("v_alignbit_b32 %0, %5, %6, %8\n\t"
        "v_add_u32 %1, -1, %0\n\t"
        "v_mov_b32 %4, %7\n\t"
        "v_mov_b32 %2, %9\n\t"
        "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_sub_u32 %3, %1, %2"
        : "=&v"(B), "=&v"(o.b), "=&v"(o.A), "=&v"(o.t1), "=&v"(o.posv)
        : "v"(phi), "v"(plo), "s"(st.pos), "s"(st.sh), "s"(a0));
  }
  return o;
}

// Digit at the current position (lane select = pos mod 64 in hardware; written by
// v_readfirstlane at least five instructions earlier, see select_step).
__device__ inline unsigned int window_digit(const DecWindow& w, const FastDecState& st) {
  unsigned int dig;
  asm("v_readlane_b32 %0, %1, %2" : "=s"(dig) : "v"(w.reg), "s"(st.pos));
  return dig;
}

// Moves the state to the first candidate containing the offset and returns `tag` of that lane
// (its symbol).  No candidate (damaged input only): EXEC is empty and lane 0 is read.
template <bool ZERO_A0>
__device__ inline int select_step(FastDecState& st, unsigned int hi, unsigned int a0, unsigned int dig,
                                  int tag) {
  const Bounds o = bounds_step<ZERO_A0>(st, hi, a0);
  const unsigned int Dn = st.D - o.A;
  const bool ren = o.t1 < 65536u;
  const unsigned int D2 = ren ? ((Dn << 16) | dig) : Dn;
  const unsigned int P2 = o.posv + (ren ? 1u : 0u);
  const unsigned int SH = ren ? 0u : 16u;
  int L;
  // Output order = safety margin for "VALU writes SGPR -> use" wait states, which the compiler
  // cannot see inside asm: pos (v_readlane lane select: 4) first, t and D (plain operands: 2) last.
  asm volatile("v_cmpx_le_u32 vcc, %5, %6\n\t"
               "s_nop 4\n\t"      // wait states before v_readfirstlane sees the new EXEC (see below)
               "v_readfirstlane_b32 %0, %7\n\t"
               "v_readfirstlane_b32 %1, %8\n\t"
               "v_readfirstlane_b32 %2, %9\n\t"
               "v_readfirstlane_b32 %3, %10\n\t"
               "v_readfirstlane_b32 %4, %11\n\t"
               "s_mov_b64 exec, -1"
               : "=&s"(st.pos), "=&s"(L), "=&s"(st.sh), "=&s"(st.t), "=&s"(st.D)
               : "s"(st.D), "v"(o.b), "v"(P2), "v"(tag), "v"(SH), "v"(o.t1), "v"(D2)
               : "vcc");
  return L;
}

// ---- the narrow-row step, hand-scheduled -----------------------------------------------
// 27 instruction slots per symbol.  The distances the hardware needs — 2 slots between a VALU
// write of an SGPR and its use as an operand, 4 before its use as a v_readlane lane select, 2
// before a DPP read of a fresh VGPR — are covered by useful instructions.  Between v_cmpx and
// the first v_readfirstlane that must see the new EXEC the hardware does NOT interlock:
// 3 slots pass tools/ubench/cmpx_hazard.hip, but the decoder itself needed 4 (found with the
// K4/K6 vectors), so 5 are kept: the read of the row two symbols ahead, the output write of the
// PREVIOUS symbol, the LDS wait (all ignore EXEC) and one s_nop 1.  Temporaries are fixed registers (v40-v55) so that
// the upper bounds can be double-buffered in two 64-bit pairs whose high halves stay zero — the
// v_mad_u64_u32 addend — without a copy; a run of steps is therefore ONE asm statement.
//   v[40:41] / v[42:43]  current / next row's upper bounds (alternating), high halves 0
//   v[44:45] product   v46 B   v47 b   v48 A   v49 t1   v50 pos   v51 Dn, D2   v52 tmp
//   v53 pos'   v54 shift'   v55 LDS address
// N2 = (symbol index + 2) & 63, CUR/NXT = 40/42 or 42/40, OUTPREV = the deferred output write.
#define TFC_DEC_STEP(N2, CUR, NXT, OUTPREV)                                                   \
  "v_lshl_add_u32 v55, %[sx], 2, %[lane4]\n\t"                                                \
  "ds_read_b32 v" #NXT ", v55\n\t"                                                            \
  "v_readlane_b32 %[dig], %[wreg], %[pos]\n\t"                                                \
  "v_mad_u64_u32 v[44:45], vcc, v" #CUR ", %[t], v[" #CUR ":" TFC_STR(TFC_INC(CUR)) "]\n\t"    \
  "v_alignbit_b32 v46, v45, v44, %[sh]\n\t"                                                   \
  "v_add_u32 v47, -1, v46\n\t"                                                                \
  "v_mov_b32 v50, %[pos]\n\t"                                                                 \
  "v_mov_b32_dpp v48, v46 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
  "v_sub_u32 v49, v47, v48\n\t"                                                               \
  "v_sub_u32 v51, %[D], v48\n\t"                                                              \
  "v_cmp_gt_u32 vcc, %[k64], v49\n\t"                                                         \
  "v_lshl_or_b32 v52, v51, 16, %[dig]\n\t"                                                    \
  "v_cndmask_b32 v51, v51, v52, vcc\n\t"                                                      \
  "v_cndmask_b32 v54, %[c16], %[zero], vcc\n\t"                                                 \
  "v_addc_co_u32 v53, vcc, 0, v50, vcc\n\t"                                                   \
  "v_cmpx_le_u32 vcc, %[D], v47\n\t"                                                          \
  "v_readlane_b32 %[sx], %[rowx], " #N2 "\n\t"                                                \
  OUTPREV                                                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
  "s_nop 1\n\t"                                                                               \
  "v_readfirstlane_b32 %[pos], v53\n\t"                                                       \
  "v_readfirstlane_b32 %[L], %[lanev]\n\t"                                                    \
  "v_readfirstlane_b32 %[sh], v54\n\t"                                                        \
  "v_readfirstlane_b32 %[t], v49\n\t"                                                         \
  "v_readfirstlane_b32 %[D], v51\n\t"                                                         \
  "s_mov_b64 exec, -1\n\t"
#define TFC_STR2(x) #x
#define TFC_STR(x) TFC_STR2(x)
#define TFC_INC(x) TFC_INC_##x
#define TFC_INC_40 41
#define TFC_INC_42 43
#define TFC_OUT(N) "v_writelane_b32 %[out], %[L], " #N "\n\t"
#define TFC_NOOUT "s_nop 0\n\t"
// eight steps for symbols a..h (i, j = the two after); FIRST = output filler of the first step
#define TFC_DEC_STEP8(FIRST, a, b, c, d, e, f, g, h, i, j)                                     \
  TFC_DEC_STEP(c, 40, 42, FIRST) TFC_DEC_STEP(d, 42, 40, TFC_OUT(a))                           \
  TFC_DEC_STEP(e, 40, 42, TFC_OUT(b)) TFC_DEC_STEP(f, 42, 40, TFC_OUT(c))                      \
  TFC_DEC_STEP(g, 40, 42, TFC_OUT(d)) TFC_DEC_STEP(h, 42, 40, TFC_OUT(e))                      \
  TFC_DEC_STEP(i, 40, 42, TFC_OUT(f)) TFC_DEC_STEP(j, 42, 40, TFC_OUT(g))
#define TFC_DEC_STEP64 \
  TFC_DEC_STEP8(TFC_NOOUT, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9) \
  TFC_DEC_STEP8(TFC_OUT(7), 8, 9, 10, 11, 12, 13, 14, 15, 16, 17) \
  TFC_DEC_STEP8(TFC_OUT(15), 16, 17, 18, 19, 20, 21, 22, 23, 24, 25) \
  TFC_DEC_STEP8(TFC_OUT(23), 24, 25, 26, 27, 28, 29, 30, 31, 32, 33) \
  TFC_DEC_STEP8(TFC_OUT(31), 32, 33, 34, 35, 36, 37, 38, 39, 40, 41) \
  TFC_DEC_STEP8(TFC_OUT(39), 40, 41, 42, 43, 44, 45, 46, 47, 48, 49) \
  TFC_DEC_STEP8(TFC_OUT(47), 48, 49, 50, 51, 52, 53, 54, 55, 56, 57) \
  TFC_DEC_STEP8(TFC_OUT(55), 56, 57, 58, 59, 60, 61, 62, 63, 0, 1)
// ---- the two-stage step (a batch that contains rows wider than 64 symbols) ---------------
// Stage 1 over the row's 64 pivots finds the chunk: its lower offset (B of the previous pivot)
// and its first symbol c0 = lane * chunk are read through EXEC; stage 2 is the narrow step on
// the chunk's entries tab[first + c0 + lane], whose LDS read is the one dependent load of the
// step (~60 cycles, partly covered).  v56 c0 candidates, v57/v58 LDS addresses, v[60:61] stage-2
// bounds (high half 0), v62 symbol of each stage-2 lane.  N = symbol index.
#define TFC_DEC_WSTEP(N, N2, CUR, NXT, OUTPREV)                                               \
  "v_lshl_add_u32 v55, %[sx], 2, %[lane4]\n\t"                                                \
  "ds_read_b32 v" #NXT ", v55\n\t"                                                            \
  "v_readlane_b32 %[dig], %[wreg], %[pos]\n\t"                                                \
  "v_mad_u64_u32 v[44:45], vcc, v" #CUR ", %[t], v[" #CUR ":" TFC_STR(TFC_INC(CUR)) "]\n\t"    \
  "v_readlane_b32 %[chunk], %[chunkv], " #N "\n\t"                                            \
  "v_readlane_b32 %[first], %[firstv], " #N "\n\t"                                            \
  "v_alignbit_b32 v46, v45, v44, %[sh]\n\t"                                                   \
  "v_add_u32 v47, -1, v46\n\t"                                                                \
  "v_mul_u32_u24 v56, %[chunk], %[lanev]\n\t"                                                 \
  "v_mov_b32_dpp v48, v46 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
  "v_lshl_add_u32 v58, %[first], 2, %[lane4]\n\t"                                             \
  "v_cmpx_le_u32 vcc, %[D], v47\n\t"                                                          \
  "v_readlane_b32 %[sx], %[rowx], " #N2 "\n\t"                                                \
  OUTPREV                                                                                     \
  "s_nop 2\n\t"                                                                               \
  "v_readfirstlane_b32 %[c0], v56\n\t"                                                        \
  "v_readfirstlane_b32 %[a0], v48\n\t"                                                        \
  "s_mov_b64 exec, -1\n\t"                                                                    \
  "v_lshl_add_u32 v57, %[c0], 2, v58\n\t"                                                     \
  "ds_read_b32 v60, v57\n\t"                                                                  \
  "v_add_u32 v62, %[c0], %[lanev]\n\t"                                                        \
  "v_mov_b32 v50, %[pos]\n\t"                                                                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
  "v_mad_u64_u32 v[44:45], vcc, v60, %[t], v[60:61]\n\t"                                      \
  "v_alignbit_b32 v46, v45, v44, %[sh]\n\t"                                                   \
  "v_add_u32 v47, -1, v46\n\t"                                                                \
  "v_mov_b32 v48, %[a0]\n\t"                                                                  \
  "v_mov_b32_dpp v48, v46 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                          \
  "v_sub_u32 v49, v47, v48\n\t"                                                               \
  "v_sub_u32 v51, %[D], v48\n\t"                                                              \
  "v_cmp_gt_u32 vcc, %[k64], v49\n\t"                                                         \
  "v_lshl_or_b32 v52, v51, 16, %[dig]\n\t"                                                    \
  "v_cndmask_b32 v51, v51, v52, vcc\n\t"                                                      \
  "v_cndmask_b32 v54, %[c16], %[zero], vcc\n\t"                                               \
  "v_addc_co_u32 v53, vcc, 0, v50, vcc\n\t"                                                   \
  "v_cmpx_le_u32 vcc, %[D], v47\n\t"                                                          \
  "s_nop 4\n\t"                                                                               \
  "v_readfirstlane_b32 %[pos], v53\n\t"                                                       \
  "v_readfirstlane_b32 %[L], v62\n\t"                                                         \
  "v_readfirstlane_b32 %[sh], v54\n\t"                                                        \
  "v_readfirstlane_b32 %[t], v49\n\t"                                                         \
  "v_readfirstlane_b32 %[D], v51\n\t"                                                         \
  "s_mov_b64 exec, -1\n\t"
#define TFC_DEC_WSTEP8(FIRST, a, b, c, d, e, f, g, h, i, j)                                    \
  TFC_DEC_WSTEP(a, c, 40, 42, FIRST) TFC_DEC_WSTEP(b, d, 42, 40, TFC_OUT(a))                   \
  TFC_DEC_WSTEP(c, e, 40, 42, TFC_OUT(b)) TFC_DEC_WSTEP(d, f, 42, 40, TFC_OUT(c))              \
  TFC_DEC_WSTEP(e, g, 40, 42, TFC_OUT(d)) TFC_DEC_WSTEP(f, h, 42, 40, TFC_OUT(e))              \
  TFC_DEC_WSTEP(g, i, 40, 42, TFC_OUT(f)) TFC_DEC_WSTEP(h, j, 42, 40, TFC_OUT(g))
#define TFC_DEC_WSTEP64 \
  TFC_DEC_WSTEP8(TFC_NOOUT, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9) \
  TFC_DEC_WSTEP8(TFC_OUT(7), 8, 9, 10, 11, 12, 13, 14, 15, 16, 17) \
  TFC_DEC_WSTEP8(TFC_OUT(15), 16, 17, 18, 19, 20, 21, 22, 23, 24, 25) \
  TFC_DEC_WSTEP8(TFC_OUT(23), 24, 25, 26, 27, 28, 29, 30, 31, 32, 33) \
  TFC_DEC_WSTEP8(TFC_OUT(31), 32, 33, 34, 35, 36, 37, 38, 39, 40, 41) \
  TFC_DEC_WSTEP8(TFC_OUT(39), 40, 41, 42, 43, 44, 45, 46, 47, 48, 49) \
  TFC_DEC_WSTEP8(TFC_OUT(47), 48, 49, 50, 51, 52, 53, 54, 55, 56, 57) \
  TFC_DEC_WSTEP8(TFC_OUT(55), 56, 57, 58, 59, 60, 61, 62, 63, 0, 1)
// ---- mixed batches: the step of every symbol chosen by its own row ------------------------------
// A batch that contains ANY row wider than 64 symbols used to run all 64 symbols through the two-stage
// step (52 slots against 27): with bmshj2018's scale tables ~7 % of the symbols have wide rows but 99 % of
// the batches contain one, so the whole stream decoded at the wide rate.  Here bit N of the scalar mask
// `wm` (ballot of "row N is wide") picks the step: narrow rows fall through an untaken branch (~12
// cycles) into the 27-slot step; wide rows jump to an out-of-line copy of the two-stage step behind the
// batch and come back.  Both steps keep the same register protocol (v40/v42 stage-1 bounds of the current /
// next symbol, sx = next-but-one row), so they can follow each other in any order.
#define TFC_DEC_MSTEP_IN(N, N2, CUR, NXT, OUTPREV)                                             \
  "s_bitcmp1_b64 %[wm], " #N "\n\t"                                                           \
  "s_cbranch_scc1 .Ltfcw%=_" #N "\n\t"                                                        \
  TFC_DEC_STEP(N2, CUR, NXT, OUTPREV)                                                         \
  ".Ltfcb%=_" #N ":\n\t"
#define TFC_DEC_MSTEP_OUT(N, N2, CUR, NXT, OUTPREV)                                            \
  ".Ltfcw%=_" #N ":\n\t"                                                                      \
  TFC_DEC_WSTEP(N, N2, CUR, NXT, OUTPREV)                                                     \
  "s_branch .Ltfcb%=_" #N "\n\t"
#define TFC_DEC_MSTEP8(M, FIRST, a, b, c, d, e, f, g, h, i, j)                                 \
  M(a, c, 40, 42, FIRST) M(b, d, 42, 40, TFC_OUT(a))                                           \
  M(c, e, 40, 42, TFC_OUT(b)) M(d, f, 42, 40, TFC_OUT(c))                                      \
  M(e, g, 40, 42, TFC_OUT(d)) M(f, h, 42, 40, TFC_OUT(e))                                      \
  M(g, i, 40, 42, TFC_OUT(f)) M(h, j, 42, 40, TFC_OUT(g))
#define TFC_DEC_MSTEP64(M) \
  TFC_DEC_MSTEP8(M, TFC_NOOUT, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9) \
  TFC_DEC_MSTEP8(M, TFC_OUT(7), 8, 9, 10, 11, 12, 13, 14, 15, 16, 17) \
  TFC_DEC_MSTEP8(M, TFC_OUT(15), 16, 17, 18, 19, 20, 21, 22, 23, 24, 25) \
  TFC_DEC_MSTEP8(M, TFC_OUT(23), 24, 25, 26, 27, 28, 29, 30, 31, 32, 33) \
  TFC_DEC_MSTEP8(M, TFC_OUT(31), 32, 33, 34, 35, 36, 37, 38, 39, 40, 41) \
  TFC_DEC_MSTEP8(M, TFC_OUT(39), 40, 41, 42, 43, 44, 45, 46, 47, 48, 49) \
  TFC_DEC_MSTEP8(M, TFC_OUT(47), 48, 49, 50, 51, 52, 53, 54, 55, 56, 57) \
  TFC_DEC_MSTEP8(M, TFC_OUT(55), 56, 57, 58, 59, 60, 61, 62, 63, 0, 1)
// a run = prologue, inline steps, epilogue, jump over the out-of-line steps
#define TFC_DEC_MRUN(FIRSTROW, LAST, IN, OUT) \
  TFC_DEC_WPROLOGUE(FIRSTROW) IN TFC_DEC_EPILOGUE(LAST) "\n\ts_branch .Ltfce%=\n\t" OUT ".Ltfce%=:"
// ---- re-entrant checked runs: a batch in eight blocks of eight symbols inside ONE asm statement -------
// Used once a stream has met an escape code.  Each block saves the state it starts from (four s_mov, the
// current bounds register and its own number) and, after its eighth symbol, compares the block's eight
// decoded symbols with their rows' escape symbols; a hit leaves the statement at once — the caller
// decodes that block from the saved state with the checked loop (which reads the Elias-gamma bits) and
// re-enters the statement at the following block through the dispatch at its top.  Against separate
// 8-symbol asm statements (the previous form: ~75 cycles per symbol for prologues, re-declaring the
// state uniform, the switch and a cold row pipeline per block) this costs ~6 cycles per symbol, and
// nothing behind the block with the escape is decoded in vain.
// STEPS = the eight steps of block K (first one with TFC_NOOUT), L7 = 8 K + 7, SH = 8 K.
#define TFC_DEC_RBLOCK(K, STEPS, L7, SH)                                                      \
  ".Ltfcr%=_" #K ":\n\t"                                                                      \
  "s_mov_b32 %[sD], %[D]\n\t"                                                                 \
  "s_mov_b32 %[sT], %[t]\n\t"                                                                 \
  "s_mov_b32 %[sSh], %[sh]\n\t"                                                               \
  "s_mov_b32 %[sPos], %[pos]\n\t"                                                             \
  "v_mov_b32 %[sHi], v40\n\t"                                                                 \
  "s_mov_b32 %[blk], " #K "\n\t"                                                              \
  STEPS                                                                                       \
  "v_writelane_b32 %[out], %[L], " #L7 "\n\t"                                                 \
  "v_cmp_eq_u32 vcc, %[out], %[escv]\n\t"                                                     \
  "s_lshr_b64 vcc, vcc, " #SH "\n\t"                                                          \
  "s_and_b32 vcc_lo, vcc_lo, 0xff\n\t"                                                        \
  "s_cmp_lg_u32 vcc_lo, 0\n\t"                                                                \
  "s_cbranch_scc1 .Ltfcx%=\n\t"
#define TFC_DEC_RDISPATCH                                                                     \
  "s_cmp_eq_u32 %[entry], 1\n\ts_cbranch_scc1 .Ltfcr%=_1\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 2\n\ts_cbranch_scc1 .Ltfcr%=_2\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 3\n\ts_cbranch_scc1 .Ltfcr%=_3\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 4\n\ts_cbranch_scc1 .Ltfcr%=_4\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 5\n\ts_cbranch_scc1 .Ltfcr%=_5\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 6\n\ts_cbranch_scc1 .Ltfcr%=_6\n\t"                                 \
  "s_cmp_eq_u32 %[entry], 7\n\ts_cbranch_scc1 .Ltfcr%=_7\n\t"
// prologue with the row to read ahead given at run time (frow = 8 entry + 1)
#define TFC_DEC_RPROLOGUE \
  "v_readlane_b32 %[sx], %[rowx], %[frow]\n\tv_mov_b32 v40, %[hi]\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v61, 0\n\t"
// the eight blocks with step macro family S8(FIRST, a .. j)
#define TFC_DEC_RBLOCKS(S8)                                                                   \
  TFC_DEC_RBLOCK(0, S8(TFC_NOOUT, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9), 7, 0)                          \
  TFC_DEC_RBLOCK(1, S8(TFC_NOOUT, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17), 15, 8)                 \
  TFC_DEC_RBLOCK(2, S8(TFC_NOOUT, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25), 23, 16)              \
  TFC_DEC_RBLOCK(3, S8(TFC_NOOUT, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33), 31, 24)              \
  TFC_DEC_RBLOCK(4, S8(TFC_NOOUT, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41), 39, 32)              \
  TFC_DEC_RBLOCK(5, S8(TFC_NOOUT, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49), 47, 40)              \
  TFC_DEC_RBLOCK(6, S8(TFC_NOOUT, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57), 55, 48)              \
  TFC_DEC_RBLOCK(7, S8(TFC_NOOUT, 56, 57, 58, 59, 60, 61, 62, 63, 0, 1), 63, 56)
#define TFC_DEC_MIN8(FIRST, a, b, c, d, e, f, g, h, i, j) TFC_DEC_MSTEP8(TFC_DEC_MSTEP_IN, FIRST, a, b, c, d, e, f, g, h, i, j)
#define TFC_DEC_MOUT8(FIRST, a, b, c, d, e, f, g, h, i, j) TFC_DEC_MSTEP8(TFC_DEC_MSTEP_OUT, FIRST, a, b, c, d, e, f, g, h, i, j)
// the out-of-line two-stage steps of all eight blocks (mixed variant only)
#define TFC_DEC_ROUTS                                                                         \
  TFC_DEC_MOUT8(TFC_NOOUT, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9)                                       \
  TFC_DEC_MOUT8(TFC_NOOUT, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17)                               \
  TFC_DEC_MOUT8(TFC_NOOUT, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25)                             \
  TFC_DEC_MOUT8(TFC_NOOUT, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33)                             \
  TFC_DEC_MOUT8(TFC_NOOUT, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41)                             \
  TFC_DEC_MOUT8(TFC_NOOUT, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49)                             \
  TFC_DEC_MOUT8(TFC_NOOUT, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57)                             \
  TFC_DEC_MOUT8(TFC_NOOUT, 56, 57, 58, 59, 60, 61, 62, 63, 0, 1)
// narrow rows only / any mix of rows
#define TFC_DEC_RRUN_NARROW \
  TFC_DEC_RPROLOGUE TFC_DEC_RDISPATCH TFC_DEC_RBLOCKS(TFC_DEC_STEP8) \
  "s_mov_b32 %[blk], 8\n\t.Ltfcx%=:\n\tv_mov_b32 %[hi], v40"
#define TFC_DEC_RRUN_MIXED \
  TFC_DEC_RPROLOGUE TFC_DEC_RDISPATCH TFC_DEC_RBLOCKS(TFC_DEC_MIN8) \
  "s_mov_b32 %[blk], 8\n\t.Ltfcx%=:\n\tv_mov_b32 %[hi], v40\n\ts_branch .Ltfce%=\n\t" TFC_DEC_ROUTS ".Ltfce%=:"
#define TFC_DEC_ROPERANDS(P_st, P_hi, P_out, P_rowx, P_wreg, P_lane4, P_lanev, P_c0, P_c16, P_sx, P_dig, P_L, \
                          P_chunkv, P_firstv, P_chunk, P_first, P_a0, P_cc, P_wm, P_escv, P_entry, P_frow,    \
                          P_sD, P_sT, P_sSh, P_sPos, P_sHi, P_blk)                                            \
  : [t] "+s"(P_st.t), [D] "+s"(P_st.D), [pos] "+s"(P_st.pos), [sh] "+s"(P_st.sh), [out] "+v"(P_out),      \
    [hi] "+v"(P_hi), [sx] "=&s"(P_sx), [dig] "=&s"(P_dig), [L] "=&s"(P_L), [chunk] "=&s"(P_chunk),        \
    [first] "=&s"(P_first), [a0] "=&s"(P_a0), [c0] "=&s"(P_cc), [sD] "=&s"(P_sD), [sT] "=&s"(P_sT),       \
    [sSh] "=&s"(P_sSh), [sPos] "=&s"(P_sPos), [sHi] "=&v"(P_sHi), [blk] "=&s"(P_blk)                      \
  : [rowx] "v"(P_rowx), [wreg] "v"(P_wreg), [lane4] "v"(P_lane4), [lanev] "v"(P_lanev), [zero] "v"(P_c0), \
    [c16] "v"(P_c16), [k64] "s"(65536u), [chunkv] "v"(P_chunkv), [firstv] "v"(P_firstv), [wm] "s"(P_wm),  \
    [escv] "v"(P_escv), [entry] "s"(P_entry), [frow] "s"(P_frow)                                          \
  : "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",  \
    "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v60", "v61", "v62"
#define TFC_DEC_WPROLOGUE(FIRSTROW) TFC_DEC_PROLOGUE(FIRSTROW) "v_mov_b32 v61, 0\n\t"
#define TFC_DEC_WOPERANDS(P_st, P_hi, P_out, P_rowx, P_wreg, P_lane4, P_lanev, P_c0, P_c16, P_sx, P_dig, P_L, \
                          P_chunkv, P_firstv, P_chunk, P_first, P_a0, P_cc, P_wm)                         \
  : [t] "+s"(P_st.t), [D] "+s"(P_st.D), [pos] "+s"(P_st.pos), [sh] "+s"(P_st.sh), [out] "+v"(P_out),      \
    [hi] "+v"(P_hi), [sx] "=&s"(P_sx), [dig] "=&s"(P_dig), [L] "=&s"(P_L), [chunk] "=&s"(P_chunk),        \
    [first] "=&s"(P_first), [a0] "=&s"(P_a0), [c0] "=&s"(P_cc)                                            \
  : [rowx] "v"(P_rowx), [wreg] "v"(P_wreg), [lane4] "v"(P_lane4), [lanev] "v"(P_lanev), [zero] "v"(P_c0), \
    [c16] "v"(P_c16), [k64] "s"(65536u), [chunkv] "v"(P_chunkv), [firstv] "v"(P_firstv), [wm] "s"(P_wm)   \
  : "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",  \
    "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v60", "v61", "v62"
// FIRSTROW = index of the run's second symbol (its row is read ahead of the first step)
#define TFC_DEC_PROLOGUE(FIRSTROW) \
  "v_readlane_b32 %[sx], %[rowx], " #FIRSTROW "\n\tv_mov_b32 v40, %[hi]\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v43, 0\n\t"
#define TFC_DEC_EPILOGUE(LAST) "v_writelane_b32 %[out], %[L], " #LAST "\n\tv_mov_b32 %[hi], v40"
#define TFC_DEC_OPERANDS(P_st, P_hi, P_out, P_rowx, P_wreg, P_lane4, P_lanev, P_c0, P_c16, P_sx, P_dig, P_L) \
  : [t] "+s"(P_st.t), [D] "+s"(P_st.D), [pos] "+s"(P_st.pos), [sh] "+s"(P_st.sh), [out] "+v"(P_out),      \
    [hi] "+v"(P_hi), [sx] "=&s"(P_sx), [dig] "=&s"(P_dig), [L] "=&s"(P_L)                                 \
  : [rowx] "v"(P_rowx), [wreg] "v"(P_wreg), [lane4] "v"(P_lane4), [lanev] "v"(P_lanev), [zero] "v"(P_c0),   \
    [c16] "v"(P_c16), [k64] "s"(65536u)                                                                   \
  : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",         \
    "v51", "v52", "v53", "v54", "v55"

// Coarse step over pivots: finds the chunk, no state update.  Everything the fine stage needs
// from the chosen pivot lane is read through EXEC: the chunk's lower offset (B of the previous
// pivot, 0 for chunk 0), and two per-lane values the caller prepared (table address of the
// chunk's first entry, first symbol of the chunk).
__device__ inline void pivot_step(const FastDecState& st, unsigned int pivot, int v0, int v1,
                                  unsigned int* a0, int* r0, int* r1) {
  const unsigned long long P = scale_product(st.t, pivot);
  const unsigned int plo = static_cast<unsigned int>(P), phi = static_cast<unsigned int>(P >> 32);
  unsigned int B, b, prev;
  asm volatile("v_alignbit_b32 %0, %6, %7, %9\n\t"
               "v_add_u32 %1, -1, %0\n\t"
               "s_nop 0\n\t"
               "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_cmpx_le_u32 vcc, %8, %1\n\t"
               "s_nop 4\n\t"
               "v_readfirstlane_b32 %3, %2\n\t"
               "v_readfirstlane_b32 %4, %10\n\t"
               "v_readfirstlane_b32 %5, %11\n\t"
               "s_mov_b64 exec, -1"
               : "=&v"(B), "=&v"(b), "=&v"(prev), "=&s"(*a0), "=&s"(*r0), "=&s"(*r1)
               : "v"(phi), "v"(plo), "s"(st.D), "s"(st.sh), "v"(v0), "v"(v1)
               : "vcc");
}

// Binary digit with the uniform cdf {0,1,2} at precision 1
// (range_coder_kernels.cc:449-471 via DecodeLinearly), on the offset state.
__device__ inline int fast_bit(FastDecState& st, const DecWindow& w) {
  const unsigned long long span = static_cast<unsigned long long>(span_minus1(st)) + 1;
  const unsigned long long target = (static_cast<unsigned long long>(st.D) + 1) << 1;
  const unsigned int bit = target <= span ? 0u : 1u;
  const unsigned int A = static_cast<unsigned int>((span * bit) >> 1);
  const unsigned int b = static_cast<unsigned int>(((span * (bit + 1)) >> 1) - 1);
  st.D -= A;
  unsigned int s = b - A;
  if ((s >> 16) == 0) {
    const unsigned int dig = window_digit(w, st);
    s = (s << 16) | 0xFFFFu;
    st.D = (st.D << 16) | dig;
    ++st.pos;
  }
  set_span_minus1(st, s);
  return static_cast<int>(bit);
}

// LLVM's uniformity analysis marks the results of an asm statement with read-write ("+s")
// operands divergent; the state is re-declared uniform after each run (a handful of
// instructions per 64 symbols).
__device__ inline void reassert_uniform(FastDecState& st) {
  st.t = __builtin_amdgcn_readfirstlane(st.t);
  st.D = __builtin_amdgcn_readfirstlane(st.D);
  st.pos = __builtin_amdgcn_readfirstlane(st.pos);
  st.sh = __builtin_amdgcn_readfirstlane(st.sh);
}

#ifdef TFC_PHASE_TIMING
__device__ unsigned long long g_dec_phase[4];
#endif

template <typename Dst>
__global__ void dec_fast_kernel(DecParams p, Dst dst) {
  extern __shared__ int32_t lds[];
  if (p.job_guard != nullptr && *p.job_guard == 0u) return;     // (wave-uniform, in front of the table copy)
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
  if (p.only_flagged != nullptr && p.only_flagged[s] == 0) return;   // second pass of throughput mode

  const uint4 st0 = p.state[s];
  FastDecState st;
  set_span_minus1(st, __builtin_amdgcn_readfirstlane(st0.y));
  st.D = __builtin_amdgcn_readfirstlane(st0.z) - __builtin_amdgcn_readfirstlane(st0.x);
  st.pos = 0;
  DecWindow w;
  const long long o0 = p.off[s];
  w.len = p.off[s + 1] - o0;
  // an empty stream reads (and discards) one byte of the offsets array instead of blob[o0]
  w.src = w.len > 0 ? p.blob + o0 : reinterpret_cast<const uint8_t*>(p.off);
  w.wbase = __builtin_amdgcn_readfirstlane(st0.w);
  const int ntab = p.tab.ntab;
  int ch0 = 0;
  // Batches left in "blocks mode": a stream that met an escape code decodes its next kBlocksAfterEscape
  // batches as eight checked 8-symbol blocks straight away instead of speculating on the whole batch first.
  // Escapes cluster per stream, and at the tables' own tail mass (2^-8 of the symbols) a third of the
  // batches contain one: the speculative pass would be wasted on most of them (one check per block costs
  // ~5 cycles per symbol, a wasted pass ~190).
  const int kBlocksAfterEscape = p.blocks_after_escape;
  int blocks_mode = 0;
  // index mode: the table indexes of the NEXT batch are requested a batch ahead (one HBM round trip per
  // 64 symbols is ~50 cycles per symbol on this chain when it is waited for on the spot)
  int t_ahead = 0;
  if (p.index && p.elems > 0) {
    const int64_t j = lane;
    t_ahead = j < p.elems ? p.index[s * p.elems + j] : 0;
  }

#ifdef TFC_PHASE_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
    // ---- vector phase: row of every symbol of the batch ---------------------
#ifdef TFC_PHASE_TIMING
    const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
    const int64_t j = j0 + lane;
    const bool valid = j < p.elems;
    int t = 0;
    if (p.index) {
      t = t_ahead;
      const int64_t jn = j + 64;
      t_ahead = jn < p.elems ? p.index[s * p.elems + jn] : 0;      // consumed by the next batch
    }
    if (valid) {
      if (p.index) {
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
    const unsigned long long widemask = __ballot(valid && (row.z >> 16) > 1);    // bit n: symbol n's row is wide
    const bool anywide = widemask != 0;

    if (j0 == 0) fast_window_load(w, lane); else fast_window_advance(w, st.pos, lane);
    st.pos = 0;
    int outv = 0;
    // operands of the hand-scheduled step: LDS byte address of tab[lane], and 0 / 16 in VGPRs
    // (opaque, so that they stay registers)
    unsigned int lane4 = static_cast<unsigned int>(reinterpret_cast<size_t>(
                             (__attribute__((address_space(3))) int32_t*)tab)) + 4u * lane;
    unsigned int vzero = 0u, vsixteen = 16u;
    asm volatile("" : "+v"(lane4), "+v"(vzero), "+v"(vsixteen));
    // per-symbol row constants as vectors, so that the per-symbol code only READS lanes of them
    // (no scalar arithmetic on values that came out of a vector instruction, see the header)
    const int chunkv = row.z >> 16;        // symbols per pivot (1: narrow row)
    const int first1v = row.y + 1;         // table index of the row's first upper bound
    // stage-1 bounds (the row itself, or its pivots) are fetched one symbol ahead
    unsigned int hi_cur = static_cast<unsigned int>(tab[__builtin_amdgcn_readlane(row.x, 0) + lane]);

    auto fetch_hi = [&](int n) {
      return static_cast<unsigned int>(tab[__builtin_amdgcn_readlane(row.x, n & 63) + lane]);
    };
    // One symbol of a narrow row (<= 64 symbols): candidates = the row itself.
    auto narrow_step = [&](int n) {
      const unsigned int hi_next = fetch_hi(n + 1);
      const unsigned int dig = window_digit(w, st);
      const int sym = select_step<true>(st, hi_cur, 0u, dig, lane);
      hi_cur = hi_next;
      return sym;
    };
    // One symbol by the two-stage route: 64 pivots -> chunk -> the chunk's entries.  A narrow
    // row goes through it as chunk size 1 (its "pivots" are the row), so a batch that contains
    // any wide row can run one branch-free code path.
    auto wide_step = [&](int n) {
      const unsigned int hi_next = fetch_hi(n + 1);
      const unsigned int dig = window_digit(w, st);
      const int chunk = __builtin_amdgcn_readlane(chunkv, n);
      const int first1 = __builtin_amdgcn_readlane(first1v, n);
      const int sym0v = __mul24(lane, chunk);
      int idxv = first1 + lane;            // + chunk start below: kept a VECTOR add on purpose
      asm volatile("" : "+v"(idxv));
      unsigned int a0;
      int cstart, cstart2;
      pivot_step(st, hi_cur, sym0v, sym0v, &a0, &cstart, &cstart2);
      const unsigned int hi2 = static_cast<unsigned int>(tab[idxv + cstart]);
      const int sym = select_step<false>(st, hi2, a0, dig, cstart2 + lane);
      hi_cur = hi_next;
      return sym;
    };

    // ---- checked loop: escapes and partial batches ----------------------------
    auto checked = [&](int n0, int n1) {
      for (int n = n0; n < n1; ++n) {
        // the window register holds 64 digits: a symbol takes at most one, the escape code behind it
        // at most five more (61 binary calls)
        if (st.pos >= 56u) {
          fast_window_advance(w, st.pos, lane);
          st.pos = 0;
        }
        const int escsym = __builtin_amdgcn_readlane(row.w, n);
        int sym = ((widemask >> n) & 1ull) ? wide_step(n) : narrow_step(n);
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
      }
    };

#ifdef TFC_PHASE_TIMING
    const unsigned long long tb1 = __builtin_readcyclecounter();
    if (s == 0 && lane == 0) g_dec_phase[0] += tb1 - tb0;
#endif
    if (cnt == 64) {
      // Level 1: the whole batch speculatively, fully unrolled (immediate lane indices, no
      // branch).  Level 2, only if an escape symbol turned up: blocks of 8 symbols with a
      // check after each block; a block containing an escape is replayed from its saved
      // start state by the checked loop (level 3), which decodes the Elias-gamma bits.
      // A stream that met an escape in the previous batch skips level 1 (escapes cluster: with 1 % of
      // the symbols escaping, 47 % of the batches contain one and the speculative run is wasted).
      const FastDecState saved64 = st;
      const unsigned int hi_saved64 = hi_cur;
      bool blocks = blocks_mode > 0;
      if (blocks_mode > 0) --blocks_mode;
      if (blocks) {
        // straight to level 2
      } else if (!anywide) {
        unsigned int sx, dg;
        int L;
        reassert_uniform(st);
        asm volatile(TFC_DEC_PROLOGUE(1) TFC_DEC_STEP64 TFC_DEC_EPILOGUE(63)
                     TFC_DEC_OPERANDS(st, hi_cur, outv, row.x, w.reg, lane4, lane, vzero, vsixteen, sx, dg, L));
        reassert_uniform(st);
        outv &= 63;   // no-op for symbols of narrow rows; keeps damaged input inside the row range
      } else if (__popcll(widemask) >= 24) {
        // mostly wide rows (one wide prior for all channels, or the upper channels of the bench tables): the
        // branch-free two-stage run for the whole batch — a narrow row goes through it as chunk size 1 —
        // beats a taken branch out of line and back per wide symbol once a third of the batch is wide
        unsigned int sx, dg, ck, fs, a0, cc;
        int L;
        reassert_uniform(st);
        asm volatile(TFC_DEC_WPROLOGUE(1) TFC_DEC_WSTEP64 TFC_DEC_EPILOGUE(63)
                     TFC_DEC_WOPERANDS(st, hi_cur, outv, row.x, w.reg, lane4, lane, vzero, vsixteen, sx, dg, L,
                                       chunkv, first1v, ck, fs, a0, cc, widemask));
        reassert_uniform(st);
      } else {
        unsigned int sx, dg, ck, fs, a0, cc;
        int L;
        reassert_uniform(st);
        asm volatile(TFC_DEC_MRUN(1, 63, TFC_DEC_MSTEP64(TFC_DEC_MSTEP_IN), TFC_DEC_MSTEP64(TFC_DEC_MSTEP_OUT))
                     TFC_DEC_WOPERANDS(st, hi_cur, outv, row.x, w.reg, lane4, lane, vzero, vsixteen, sx, dg, L,
                                       chunkv, first1v, ck, fs, a0, cc, widemask));
        reassert_uniform(st);
      }
      if (!blocks && __ballot(outv == row.w) != 0) {
        st = saved64;
        hi_cur = hi_saved64;
        blocks = true;
      }
      if (blocks) {
        // the batch as eight checked blocks inside one re-entrant asm run (TFC_DEC_RBLOCK): out at the first
        // block whose symbols contain an escape, that block again from its saved state in the checked loop,
        // back in at the next block
        int entry = 0;
        while (entry < 8) {
          // the rest of the batch takes at most one digit per symbol (a further escape leaves the run again)
          if (st.pos + static_cast<unsigned int>(64 - 8 * entry) > 64u) {
            fast_window_advance(w, st.pos, lane);
            st.pos = 0;
          }
          unsigned int sx, dg, ck, fs, a0, cc, sD, sT, sSh, sPos, sHi;
          int L, blk;
          const int frow = 8 * entry + 1;
          reassert_uniform(st);
          if (!anywide) {
            asm volatile(TFC_DEC_RRUN_NARROW
                         TFC_DEC_ROPERANDS(st, hi_cur, outv, row.x, w.reg, lane4, lane, vzero, vsixteen, sx, dg, L,
                                           chunkv, first1v, ck, fs, a0, cc, widemask, row.w, entry, frow,
                                           sD, sT, sSh, sPos, sHi, blk));
          } else {
            asm volatile(TFC_DEC_RRUN_MIXED
                         TFC_DEC_ROPERANDS(st, hi_cur, outv, row.x, w.reg, lane4, lane, vzero, vsixteen, sx, dg, L,
                                           chunkv, first1v, ck, fs, a0, cc, widemask, row.w, entry, frow,
                                           sD, sT, sSh, sPos, sHi, blk));
          }
          reassert_uniform(st);
          blk = __builtin_amdgcn_readfirstlane(blk);
          if (blk >= 8) break;
          // block `blk` decoded an escape symbol: again from where it started, bit by bit
          st.D = __builtin_amdgcn_readfirstlane(sD);
          st.t = __builtin_amdgcn_readfirstlane(sT);
          st.sh = __builtin_amdgcn_readfirstlane(sSh);
          st.pos = __builtin_amdgcn_readfirstlane(sPos);
          hi_cur = sHi;
          checked(8 * blk, 8 * blk + 8);
          blocks_mode = kBlocksAfterEscape;
          entry = blk + 1;
        }
      }
    } else {
      checked(0, cnt);
    }
#ifdef TFC_PHASE_TIMING
    const unsigned long long tb2 = __builtin_readcyclecounter();
    if (s == 0 && lane == 0) g_dec_phase[1] += tb2 - tb1;
#endif
    if (valid) dst.store(s * p.elems + j, t, outv);
#ifdef TFC_PHASE_TIMING
    if (s == 0 && lane == 0) g_dec_phase[2] += __builtin_readcyclecounter() - tb2;
#endif
  }
#ifdef TFC_PHASE_TIMING
  if (s == 0 && lane == 0)
    printf("dec stream 0: total %llu cycles for %lld symbols: batch prologue %llu, steps %llu, store %llu\n",
           __builtin_readcyclecounter() - t_begin, (long long)p.elems, g_dec_phase[0], g_dec_phase[1], g_dec_phase[2]);
#endif

  w.wbase += st.pos;
  if (lane == 0) {
    // back to the (base, span-1, window, digits pulled) form the other kernels use
    const long long b = 2ll * w.wbase;
    unsigned int window = 0;
    for (int i = -4; i < 0; ++i) {
      const long long q = b + i;
      window = (window << 8) | ((q >= 0 && q < w.len) ? w.src[q] : 0u);
    }
    p.state[s] = make_uint4(window - st.D, span_minus1(st), window, w.wbase);
  }
}

}  // namespace tfc
