// Single-wave dependent-chain latency probes for gfx950 (design input for the
// range coder: its per-symbol cost is a chain of ~10-30 such instructions).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/chain.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// Each probe: 64 copies of BODY per loop iteration, ITER iterations, one wave.
#define PROBE(name, BODY, PER)                                                   \
  __global__ void name(unsigned long long* out, unsigned int seed) {             \
    unsigned int a = seed + threadIdx.x, b = seed * 3 + 1, c = 7, d = 11;        \
    unsigned int s0 = __builtin_amdgcn_readfirstlane(seed) | 0x10001u, s1 = 12345u, s2 = 0, s3 = 0;      \
    unsigned long long t0 = __builtin_readcyclecounter();                        \
    for (int i = 0; i < 64; ++i) {                                               \
      asm volatile(REP64(BODY)                                                   \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) \
                   :: "vcc", "scc", "memory");                                   \
    }                                                                            \
    unsigned long long t1 = __builtin_readcyclecounter();                        \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c + d + s0 + s1 + s2 + s3; out[2] = PER; } \
  }

// s_memtime-based counter ticks at shader clock? we also time with wall clock from host.
PROBE(p_v_add_dep,      "v_add_u32 %0, %0, %1\n", 1)
PROBE(p_v_add_indep,    "v_add_u32 %0, %1, %1\n v_add_u32 %2, %3, %3\n", 2)
PROBE(p_v_mad64_dep,    "v_mad_u64_u32 v[10:11], vcc, %0, %1, 0\n v_mov_b32 %0, v11\n", 2)
PROBE(p_v_mad64_only,   "v_mad_u64_u32 v[10:11], vcc, %0, %1, v[10:11]\n", 1)
PROBE(p_v_mulhi_dep,    "v_mul_hi_u32 %0, %0, %1\n", 1)
PROBE(p_v_mullo_dep,    "v_mul_lo_u32 %0, %0, %1\n", 1)
PROBE(p_v_mul24_dep,    "v_mul_u32_u24 %0, %0, %1\n", 1)
PROBE(p_v_alignbit_dep, "v_alignbit_b32 %0, %0, %1, 16\n", 1)
PROBE(p_v_cndmask_dep,  "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n", 2)
PROBE(p_v_dpp_shr,      "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n", 1)
PROBE(p_v_add_dpp,      "s_nop 1\n v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n", 1)
PROBE(p_s_add_dep,      "s_add_u32 %4, %4, %5\n", 1)
PROBE(p_s_mul_dep,      "s_mul_i32 %4, %4, %5\n", 1)
PROBE(p_s_mulhi_dep,    "s_mul_hi_u32 %4, %4, %5\n", 1)
PROBE(p_s_lshr64_dep,   "s_lshr_b64 s[20:21], s[20:21], 1\n", 1)
PROBE(p_s_indep,        "s_add_u32 %4, %5, %5\n s_add_u32 %6, %5, 3\n", 2)
PROBE(p_rfl_roundtrip,  "v_readfirstlane_b32 %4, %0\n v_add_u32 %0, %4, %0\n", 2)
PROBE(p_readlane_rt,    "v_readlane_b32 %4, %0, 5\n v_add_u32 %0, %4, %0\n", 2)
PROBE(p_cmp_ff1,        "v_cmp_gt_u32 vcc, %0, %1\n s_ff1_i32_b64 %4, vcc\n v_add_u32 %0, %4, %0\n", 3)
PROBE(p_cmpx_rfl,       "v_cmpx_gt_u32 exec, %1, %0\n v_readfirstlane_b32 %4, %0\n s_mov_b64 exec, -1\n v_add_u32 %0, %4, %0\n", 4)
PROBE(p_s_branch_nt,    "s_cmp_eq_u32 %4, 0\n s_cbranch_scc1 1f\n s_add_u32 %4, %4, 1\n1:\n", 3)
PROBE(p_writelane,      "s_mov_b32 m0, %5\n v_writelane_b32 %0, %4, m0\n", 2)
PROBE(p_v_lshlor_dep,   "v_lshl_or_b32 %0, %0, 16, %1\n", 1)
PROBE(p_v_sub_dep,      "v_sub_u32 %0, %1, %0\n", 1)
PROBE(p_mix_valu_salu,  "v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n", 2)

__global__ void p_lds_lat(unsigned long long* out, unsigned int seed) {
  __shared__ unsigned int buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = (i * 17 + seed) & 1023;
  __syncthreads();
  unsigned int a = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 4096; ++i) a = buf[a];
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a; out[2] = 1; }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  struct P { const char* n; void (*k)(unsigned long long*, unsigned int); int n_body; };
  std::vector<P> ps = {
    {"v_add dep", p_v_add_dep, 4096}, {"v_add x2 indep", p_v_add_indep, 4096},
    {"v_mad_u64_u32 + mov (dep)", p_v_mad64_dep, 4096}, {"v_mad_u64_u32 acc-dep", p_v_mad64_only, 4096},
    {"v_mul_hi_u32 dep", p_v_mulhi_dep, 4096}, {"v_mul_lo_u32 dep", p_v_mullo_dep, 4096},
    {"v_mul_u32_u24 dep", p_v_mul24_dep, 4096}, {"v_alignbit dep", p_v_alignbit_dep, 4096},
    {"v_cmp+v_cndmask dep", p_v_cndmask_dep, 4096}, {"v_mov_dpp wave_shr dep", p_v_dpp_shr, 4096},
    {"s_nop1+v_add_dpp wave_shr dep", p_v_add_dpp, 4096},
    {"s_add dep", p_s_add_dep, 4096}, {"s_mul_i32 dep", p_s_mul_dep, 4096},
    {"s_mul_hi_u32 dep", p_s_mulhi_dep, 4096}, {"s_lshr_b64 dep", p_s_lshr64_dep, 4096},
    {"s_add x2 indep", p_s_indep, 4096},
    {"readfirstlane->v_add roundtrip", p_rfl_roundtrip, 4096}, {"readlane->v_add roundtrip", p_readlane_rt, 4096},
    {"v_cmp->s_ff1->v_add", p_cmp_ff1, 4096}, {"v_cmpx->rfl->exec restore->v_add", p_cmpx_rfl, 4096},
    {"s_cmp+branch(not taken)+s_add", p_s_branch_nt, 4096}, {"s_mov m0 + v_writelane", p_writelane, 4096},
    {"v_lshl_or dep", p_v_lshlor_dep, 4096}, {"v_sub dep", p_v_sub_dep, 4096},
    {"v_add + s_add interleaved (2 chains)", p_mix_valu_salu, 4096},
    {"LDS dependent read (ds_read_b32 chain)", p_lds_lat, 4096},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& p : ps) {
    unsigned long long h[3];
    hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, d, 12345u);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-42s counter/body %7.2f   (kernel %.1f us => %.2f ns/body)\n", p.n,
           (double)h[0] / p.n_body, ms * 1e3, ms * 1e6 / p.n_body);
  }
  return 0;
}
