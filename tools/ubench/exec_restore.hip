// Does a VALU instruction right after `s_mov_b64 exec, -1` (following a v_cmpx-narrowed EXEC)
// run on all lanes?  Build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/exec_restore.hip -o /tmp/er && /tmp/er
#include <hip/hip_runtime.h>
#include <cstdio>
#define VARIANT(name, FILL)                                                          \
  __global__ void name(int* out, const int* in) {                                    \
    unsigned int B = in[threadIdx.x];                                                \
    unsigned int thr = __builtin_amdgcn_readfirstlane(in[20]);                       \
    int first_lane, lanev = threadIdx.x, mark = -1;                                  \
    asm volatile("v_cmpx_le_u32 vcc, %3, %4\n\t s_nop 2\n\t"                         \
                 "v_readfirstlane_b32 %0, %2\n\t"                                    \
                 "s_mov_b64 exec, -1\n\t" FILL                                       \
                 "v_mov_b32 %1, %2"                                                  \
                 : "=&s"(first_lane), "+v"(mark) : "v"(lanev), "s"(thr), "v"(B) : "vcc"); \
    out[threadIdx.x] = mark;                                                         \
  }
VARIANT(k0, "")
VARIANT(k1, "s_nop 0\n\t")
VARIANT(k2, "s_nop 1\n\t")
VARIANT(k3, "s_nop 2\n\t")
int main() {
  int h[64], *din, *dout;
  for (int i = 0; i < 64; ++i) h[i] = 1000 + i;
  hipMalloc(&din, 256); hipMalloc(&dout, 256);
  hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
  struct { const char* n; void (*k)(int*, const int*); } ks[] = {{"no filler", k0}, {"s_nop 0", k1}, {"s_nop 1", k2}, {"s_nop 2", k3}};
  for (auto& e : ks) {
    int o[64], bad = 0;
    hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, dout, din);
    hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) bad += o[i] != i;
    printf("%-12s lanes not written after exec restore: %d\n", e.n, bad);
  }
  return 0;
}
