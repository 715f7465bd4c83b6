// Cycle accounting of the fast decoder's per-symbol step (narrow rows) on one wave.
// The instruction sequence is the one hipcc emits for select_step<true> + prefetch + output
// (range_decoder_fast.h); groups are removed one at a time to see what each costs.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/dec_step.hip -o /tmp/dec_step && /tmp/dec_step
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(x) REP8(x) REP8(x) REP8(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define G_OUT   "v_writelane_b32 v11, s20, 5\n"
#define G_DSEL  "v_cndmask_b32_e64 v15, v15, v16, s[28:29]\n"
#define G_DIG   "v_readlane_b32 s25, v12, s23\n"
#define G_DRD   "v_readlane_b32 s22, v15, s20\n"
#define G_PREF  "v_readlane_b32 s30, v13, 7\n s_nop 1\n v_lshl_add_u32 v17, s30, 2, v14\n ds_read_b32 v18, v17\n s_waitcnt lgkmcnt(1)\n"
#define G_MAD   "v_mov_b32 v20, v10\n s_nop 0\n v_mad_u64_u32 v[22:23], s[32:33], v10, s21, v[20:21]\n"
#define G_BND   "v_alignbit_b32 v20, v23, v22, s24\n v_add_u32 v24, -1, v20\n v_cmp_le_u32_e64 s[26:27], s22, v24\n"
#define G_DPP   "v_mov_b32_dpp v25, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_u32 v26, v24, v25\n"
#define G_FF1   "s_ff1_i32_b64 s20, s[26:27]\n"
#define G_DN    "v_sub_u32 v15, s22, v25\n v_lshl_or_b32 v16, v15, 16, s25\n"
#define G_REN   "v_cmp_gt_u32_e64 s[28:29], s34, v26\n"
#define G_TRD   "v_readlane_b32 s21, v26, s20\n"
#define G_POS   "s_bitcmp1_b64 s[28:29], s20\n s_cselect_b32 s24, 0, 16\n s_addc_u32 s23, s23, 0\n"

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v20","v21","v22","v23","v24","v25","v26", \
             "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s32","s33","s34","vcc","scc","memory"

#define KERNEL(name, BODY)                                                              \
  __global__ void name(unsigned long long* out, unsigned int seed) {                    \
    __shared__ unsigned int buf[2048];                                                  \
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = (i * 2654435761u + seed) >> 16; \
    __syncthreads();                                                                    \
    asm volatile("v_mov_b32 v10, 0x8000\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0x1234\n"   \
                 "v_and_b32 v13, 63, %0\n v_lshlrev_b32 v14, 2, v13\n v_mov_b32 v15, 5\n v_mov_b32 v16, 9\n" \
                 "v_mov_b32 v21, 0\n v_mov_b32 v18, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 70000\n" \
                 "s_mov_b32 s20, 3\n s_mov_b32 s21, 0x12345\n s_mov_b32 s22, 77\n s_mov_b32 s23, 0\n" \
                 "s_mov_b32 s24, 16\n s_mov_b32 s25, 0\n s_mov_b64 s[28:29], 0\n s_mov_b32 s34, 0x10000\n" \
                 :: "v"(threadIdx.x) : CLOB);                                           \
    unsigned long long t0 = __builtin_readcyclecounter();                               \
    for (int i = 0; i < 64; ++i) asm volatile(REP64(BODY) ::: CLOB);                    \
    unsigned long long t1 = __builtin_readcyclecounter();                               \
    unsigned int r;                                                                      \
    asm volatile("v_add_u32 %0, v11, v18\n v_add_u32 %0, %0, v26" : "=v"(r) :: CLOB);    \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r + buf[5]; }                    \
  }

KERNEL(k_full,    G_OUT G_DSEL G_DIG G_DRD G_PREF G_MAD G_BND G_DPP G_FF1 G_DN G_REN G_TRD G_POS)
KERNEL(k_nopref,  G_OUT G_DSEL G_DIG G_DRD        G_MAD G_BND G_DPP G_FF1 G_DN G_REN G_TRD G_POS)
KERNEL(k_noout,         G_DSEL G_DIG G_DRD G_PREF G_MAD G_BND G_DPP G_FF1 G_DN G_REN G_TRD G_POS)
KERNEL(k_nopos,   G_OUT G_DSEL G_DIG G_DRD G_PREF G_MAD G_BND G_DPP G_FF1 G_DN G_REN G_TRD)
KERNEL(k_nod,     G_OUT                    G_PREF G_MAD G_BND G_DPP G_FF1       G_REN G_TRD G_POS)
KERNEL(k_core,                                    G_MAD G_BND G_DPP G_FF1             G_TRD)
KERNEL(k_core_nodpp,                              G_MAD G_BND       G_FF1             G_TRD)
KERNEL(k_core_noff1,                              G_MAD G_BND G_DPP                   "v_readlane_b32 s21, v26, 5\n")

KERNEL(k_pref_only, G_PREF)
KERNEL(k_pref_nonop, "v_readlane_b32 s30, v13, 7\n v_lshl_add_u32 v17, s30, 2, v14\n ds_read_b32 v18, v17\n s_waitcnt lgkmcnt(1)\n")
KERNEL(k_pref_nowait, "v_readlane_b32 s30, v13, 7\n s_nop 1\n v_lshl_add_u32 v17, s30, 2, v14\n ds_read_b32 v18, v17\n")
KERNEL(k_pref_nods, "v_readlane_b32 s30, v13, 7\n s_nop 1\n v_lshl_add_u32 v17, s30, 2, v14\n")
KERNEL(k_rl_imm, "v_readlane_b32 s30, v13, 7\n")
KERNEL(k_rl_sreg, "v_readlane_b32 s25, v12, s23\n")
KERNEL(k_rl_then_valu, "v_readlane_b32 s30, v13, 7\n v_add_u32 v17, s30, v14\n")
KERNEL(k_rl_nop_valu, "v_readlane_b32 s30, v13, 7\n s_nop 1\n v_add_u32 v17, s30, v14\n")
KERNEL(k_dpath, G_DSEL G_DIG G_DRD G_DN)
KERNEL(k_cmp_cnd, "v_cmp_gt_u32_e64 s[28:29], s34, v26\n v_cndmask_b32_e64 v15, v15, v16, s[28:29]\n")
KERNEL(k_cmp_x_cnd, "v_cmp_gt_u32_e64 s[28:29], s34, v26\n v_add_u32 v17, v14, v14\n v_add_u32 v16, v14, v14\n v_cndmask_b32_e64 v15, v15, v16, s[28:29]\n")
KERNEL(k_ff1_rl, "s_ff1_i32_b64 s20, s[26:27]\n v_readlane_b32 s21, v26, s20\n")
KERNEL(k_salu_rl, "s_add_u32 s20, s20, 1\n v_readlane_b32 s21, v26, s20\n")
KERNEL(k_salu_valu, "s_add_u32 s20, s20, 1\n v_add_u32 v17, s20, v14\n")
KERNEL(k_salu_wl, "s_add_u32 s20, s20, 1\n v_writelane_b32 v11, s20, 5\n")
KERNEL(k_cmp_ff1, "v_cmp_le_u32_e64 s[26:27], s22, v24\n s_ff1_i32_b64 s20, s[26:27]\n")
KERNEL(k_cmp_x_ff1, "v_cmp_le_u32_e64 s[26:27], s22, v24\n v_add_u32 v17, v14, v14\n v_add_u32 v16, v14, v14\n v_add_u32 v15, v14, v14\n s_ff1_i32_b64 s20, s[26:27]\n")
KERNEL(k_rl_salu, "v_readlane_b32 s21, v26, 3\n s_add_u32 s20, s21, 1\n")
KERNEL(k_rl_x_salu, "v_readlane_b32 s21, v26, 3\n v_add_u32 v17, v14, v14\n v_add_u32 v16, v14, v14\n s_add_u32 s20, s21, 1\n")
KERNEL(k_bitcmp, "s_bitcmp1_b64 s[28:29], s20\n s_cselect_b32 s24, 0, 16\n s_addc_u32 s23, s23, 0\n")
KERNEL(k_mad, "v_mad_u64_u32 v[22:23], s[32:33], v10, s21, v[20:21]\n")
KERNEL(k_mad_align, "v_mad_u64_u32 v[22:23], s[32:33], v10, s21, v[20:21]\n v_alignbit_b32 v20, v23, v22, s24\n")
KERNEL(k_valu4, "v_add_u32 v17, v14, v14\n v_add_u32 v16, v14, v14\n v_add_u32 v15, v14, v14\n v_add_u32 v24, v14, v14\n")

// ---- candidate redesign: first hit through EXEC (v_cmpx + v_readfirstlane), no SALU on the chain
//   v27 = lane id, v28 = 0, v29 = 16;  s35 = L, s36 = sh
#define N_MAD   "v_mov_b32 v20, v10\n v_mad_u64_u32 v[22:23], s[32:33], v10, s21, v[20:21]\n"
#define N_BND   "v_alignbit_b32 v20, v23, v22, s36\n v_add_u32 v24, -1, v20\n"
#define N_PREF  "v_readlane_b32 s30, v13, 7\n"
#define N_DPP   "v_mov_b32_dpp v25, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_u32 v26, v24, v25\n v_sub_u32 v15, s22, v25\n"
#define N_PREF2 "v_lshl_add_u32 v17, s30, 2, v14\n ds_read_b32 v18, v17\n"
#define N_REN   "v_cmp_gt_u32 vcc, s34, v26\n v_lshl_or_b32 v16, v15, 16, s25\n v_cndmask_b32 v15, v15, v16, vcc\n v_mov_b32 v32, s23\n v_addc_co_u32 v19, vcc, 0, v32, vcc\n"
#define N_SH    "v_cmp_gt_u32 vcc, s34, v26\n v_cndmask_b32 v30, v29, v28, vcc\n"
#define N_SEL   "v_cmpx_le_u32 vcc, s22, v24\n v_readfirstlane_b32 s21, v26\n v_readfirstlane_b32 s22, v15\n v_readfirstlane_b32 s23, v19\n v_readfirstlane_b32 s36, v30\n"
#define N_OUTL  "v_readfirstlane_b32 s35, v27\n s_mov_b64 exec, -1\n v_writelane_b32 v11, s35, 5\n"
#define N_OUTA  "ds_min_u32 v31, v27 offset:20\n s_mov_b64 exec, -1\n"
#define N_DIG   "v_readlane_b32 s25, v12, s23\n s_waitcnt lgkmcnt(1)\n"
#define NCLOB CLOB, "v19", "v27", "v28", "v29", "v30", "v31", "v32", "s35", "s36"
#define NKERNEL(name, BODY)                                                             \
  __global__ void name(unsigned long long* out, unsigned int seed) {                    \
    __shared__ unsigned int buf[2048];                                                  \
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = (i * 2654435761u + seed) >> 16; \
    __syncthreads();                                                                    \
    asm volatile("v_mov_b32 v10, 0x8000\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0x1234\n"   \
                 "v_and_b32 v13, 63, %0\n v_lshlrev_b32 v14, 2, v13\n v_mov_b32 v15, 5\n v_mov_b32 v16, 9\n" \
                 "v_mov_b32 v21, 0\n v_mov_b32 v18, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 70000\n" \
                 "v_mov_b32 v27, v13\n v_mov_b32 v28, 0\n v_mov_b32 v29, 16\n v_mov_b32 v30, 16\n v_mov_b32 v31, 0\n v_mov_b32 v19, 0\n" \
                 "s_mov_b32 s20, 3\n s_mov_b32 s21, 0x12345\n s_mov_b32 s22, 77\n s_mov_b32 s23, 0\n" \
                 "s_mov_b32 s36, 16\n s_mov_b32 s25, 0\n s_mov_b32 s34, 0x10000\n s_mov_b32 s35, 0\n" \
                 :: "v"(threadIdx.x) : NCLOB);                                          \
    unsigned long long t0 = __builtin_readcyclecounter();                               \
    for (int i = 0; i < 64; ++i) asm volatile(REP64(BODY) ::: NCLOB);                   \
    unsigned long long t1 = __builtin_readcyclecounter();                               \
    unsigned int r;                                                                      \
    asm volatile("v_add_u32 %0, v11, v18\n v_add_u32 %0, %0, v26" : "=v"(r) :: NCLOB);  \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r + buf[5]; }                    \
  }
NKERNEL(n_full,  N_MAD N_BND N_PREF N_DPP N_PREF2 N_REN N_SH N_SEL N_OUTL N_DIG)
NKERNEL(n_atomic, N_MAD N_BND N_PREF N_DPP N_PREF2 N_REN N_SH N_SEL N_OUTA N_DIG)
NKERNEL(n_nopref, N_MAD N_BND        N_DPP         N_REN N_SH N_SEL N_OUTL "v_readlane_b32 s25, v12, s23\n")
NKERNEL(n_sel_only, N_SEL "s_mov_b64 exec, -1\n")
NKERNEL(n_cmpx_rfl1, "v_cmpx_le_u32 vcc, s22, v24\n v_readfirstlane_b32 s21, v26\n s_mov_b64 exec, -1\n")
NKERNEL(n_cmpx_rfl1_use, "v_cmpx_le_u32 vcc, s22, v24\n v_readfirstlane_b32 s21, v26\n s_mov_b64 exec, -1\n v_add_u32 v24, s21, v24\n")
NKERNEL(n_rfl_rl, "v_readfirstlane_b32 s23, v19\n v_readlane_b32 s25, v12, s23\n")
NKERNEL(n_rfl_4_rl, "v_readfirstlane_b32 s23, v19\n v_add_u32 v17, v14, v14\n v_add_u32 v16, v14, v14\n v_add_u32 v15, v14, v14\n v_add_u32 v30, v14, v14\n v_readlane_b32 s25, v12, s23\n")

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  struct { const char* n; void (*k)(unsigned long long*, unsigned int); } ks[] = {
      {"full step (25 instr)", k_full}, {"- prefetch group", k_nopref}, {"- output writelane", k_noout},
      {"- pos/shift scalar trio", k_nopos}, {"- D path (cndmask, 2 readlane, sub, lshl_or)", k_nod},
      {"core chain: mad, bounds, dpp, ff1, readlane t", k_core}, {"core without dpp+sub", k_core_nodpp},
      {"core without ff1 (fixed lane)", k_core_noff1},
      {"prefetch group alone (5 slots)", k_pref_only}, {"  without s_nop 1", k_pref_nonop}, {"  without s_waitcnt", k_pref_nowait},
      {"  without ds_read+waitcnt", k_pref_nods},
      {"v_readlane imm lane", k_rl_imm}, {"v_readlane sgpr lane", k_rl_sreg},
      {"v_readlane -> v_add (sgpr operand)", k_rl_then_valu}, {"v_readlane, s_nop 1, v_add", k_rl_nop_valu},
      {"D path alone (5 instr)", k_dpath},
      {"v_cmp_e64 -> v_cndmask_e64", k_cmp_cnd}, {"v_cmp, 2 valu, v_cndmask", k_cmp_x_cnd},
      {"s_ff1 -> v_readlane(lane sel)", k_ff1_rl}, {"s_add -> v_readlane(lane sel)", k_salu_rl},
      {"s_add -> v_add(sgpr operand)", k_salu_valu}, {"s_add -> v_writelane", k_salu_wl},
      {"v_cmp_e64 -> s_ff1", k_cmp_ff1}, {"v_cmp, 3 valu, s_ff1", k_cmp_x_ff1},
      {"v_readlane -> s_add", k_rl_salu}, {"v_readlane, 2 valu, s_add", k_rl_x_salu},
      {"s_bitcmp1, s_cselect, s_addc", k_bitcmp},
      {"NEW step, L via readfirstlane + writelane", n_full}, {"NEW step, L via ds_min atomic", n_atomic},
      {"NEW step without prefetch/waitcnt", n_nopref}, {"cmpx + 4 rfl + exec restore", n_sel_only},
      {"cmpx, rfl, exec restore", n_cmpx_rfl1}, {"cmpx, rfl, exec restore, v_add(use)", n_cmpx_rfl1_use},
      {"rfl -> v_readlane lane sel (no nops!)", n_rfl_rl}, {"rfl, 4 valu, v_readlane lane sel", n_rfl_4_rl},
      {"v_mad_u64_u32 (indep)", k_mad}, {"v_mad_u64 -> v_alignbit", k_mad_align}, {"4 indep v_add", k_valu4}};
  for (auto& e : ks) {
    unsigned long long h[2];
    for (int r = 0; r < 2; ++r) {
      hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, d, 12345u);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    }
    printf("%-52s %7.2f counter ticks / step\n", e.n, (double)h[0] / 4096.0);
  }
  return 0;
}
