// How many wait states does v_readfirstlane need after a v_cmpx that rewrote EXEC (gfx950)?
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/cmpx_hazard.hip -o /tmp/cmpx && /tmp/cmpx
#include <hip/hip_runtime.h>
#include <cstdio>
#define VARIANT(name, FILL)                                                          \
  __global__ void name(int* out, const int* in) {                                    \
    unsigned int B = in[threadIdx.x];                                                \
    unsigned int thr = __builtin_amdgcn_readfirstlane(in[20]);                       \
    int first_val, first_lane, lanev = threadIdx.x;                                  \
    asm volatile("v_cmpx_le_u32 vcc, %3, %4\n\t" FILL                                \
                 "v_readfirstlane_b32 %0, %4\n\t"                                    \
                 "v_readfirstlane_b32 %1, %2\n\t"                                    \
                 "s_mov_b64 exec, -1"                                                \
                 : "=&s"(first_val), "=&s"(first_lane), "+v"(lanev) : "s"(thr), "v"(B) : "vcc", "s20", "s21", "s22"); \
    out[threadIdx.x] = first_lane;                                                   \
  }
VARIANT(k0, "")
VARIANT(k1, "s_nop 0\n\t")
VARIANT(k2, "s_nop 1\n\t")
VARIANT(k3, "s_nop 2\n\t")
VARIANT(k4, "s_nop 3\n\t")
VARIANT(k5, "s_nop 4\n\t")
VARIANT(kr, "v_readlane_b32 s20, %2, 3\n\t")
VARIANT(kr2, "v_readlane_b32 s20, %2, 3\n\t v_readlane_b32 s21, %2, 4\n\t")
VARIANT(kmix, "v_readlane_b32 s20, %2, 3\n\t v_writelane_b32 %2, s20, 63\n\t s_waitcnt lgkmcnt(0)\n\t")
VARIANT(kmix2, "v_readlane_b32 s20, %2, 3\n\t s_nop 0\n\t s_waitcnt lgkmcnt(0)\n\t")
VARIANT(kr3, "v_readlane_b32 s20, %2, 3\n\t v_readlane_b32 s21, %2, 4\n\t v_readlane_b32 s22, %2, 5\n\t")
VARIANT(ksm, "s_mov_b32 s20, 3\n\t")
VARIANT(ksm2, "s_mov_b32 s20, 3\n\t s_mov_b32 s21, 3\n\t")
int main() {
  int h[64], *din, *dout;
  for (int i = 0; i < 64; ++i) h[i] = 1000 + i;
  hipMalloc(&din, 256); hipMalloc(&dout, 256);
  hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
  struct { const char* n; void (*k)(int*, const int*); } ks[] = {{"no filler", k0}, {"s_nop 0", k1}, {"s_nop 1", k2},
      {"s_nop 2", k3}, {"s_nop 3", k4}, {"s_nop 4", k5}, {"1 v_readlane", kr}, {"2 v_readlane", kr2},
      {"readlane, writelane, waitcnt", kmix}, {"readlane, nop, waitcnt", kmix2}, {"3 v_readlane", kr3}, {"1 s_mov", ksm}, {"2 s_mov", ksm2}};
  for (auto& e : ks) {
    int o[64];
    hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, dout, din);
    hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    printf("%-14s first active lane read: %d (want 20)\n", e.n, o[0]);
  }
  return 0;
}
