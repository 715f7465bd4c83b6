// Device check: what ds_read_b64_tr_b16 returns.  LDS holds u16 values lds[i] = i; lane l reads at byte address
// addr(l) (three patterns) and the four 16-bit values it receives are printed per lane.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read_probe.hip -o /tmp/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned int* out, int pattern) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = static_cast<unsigned short>(i);
  __syncthreads();
  const int l = threadIdx.x;
  unsigned int addr;
  if (pattern == 0) addr = l * 8;                                   // lane l: elements 4 l .. 4 l + 3
  else if (pattern == 1) addr = (l & 15) * 64 + (l >> 4) * 8;        // 16 rows of 32 elements (64 B), lane group picks the column quad
  else addr = (l & 3) * 64 + ((l >> 2) & 3) * 8 + (l >> 4) * 512;    // quads: 4 rows x 4-element chunks
  addr += static_cast<unsigned int>(reinterpret_cast<size_t>(lds) & 0xFFFF);
  unsigned int lo, hi;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  lo = v.x; hi = v.y;
  out[2 * l] = lo;
  out[2 * l + 1] = hi;
}

int main() {
  unsigned int* d;
  hipMalloc(&d, 64 * 2 * 4);
  std::vector<unsigned int> h(128);
  for (int pattern = 0; pattern < 3; ++pattern) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern);
    hipMemcpy(h.data(), d, 128 * 4, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d: %4u %4u %4u %4u\n", l, h[2 * l] & 0xFFFF, h[2 * l] >> 16, h[2 * l + 1] & 0xFFFF, h[2 * l + 1] >> 16);
  }
  return 0;
}
