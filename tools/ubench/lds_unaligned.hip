// Device check (round 5): does ds_read_b32 return the right bytes at a 2-byte-aligned LDS address on gfx950
// (ROCm 7.2 default memory configuration)?  hipcc --offload-arch=gfx950 -O2 lds_unaligned.hip -o lds_unaligned && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  __shared__ unsigned short tab[256];
  for (int i = threadIdx.x; i < 256; i += 64) tab[i] = static_cast<unsigned short>(i * 257 + 1);
  __syncthreads();
  const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned short*)tab)) + 2u * threadIdx.x;
  unsigned v, w0, w1;
  asm volatile("ds_read_b32 %0, %3\n\tds_read_u16 %1, %3\n\tds_read_u16 %2, %3 offset:2\n\ts_waitcnt lgkmcnt(0)"
               : "=v"(v), "=v"(w0), "=v"(w1) : "v"(addr) : "memory");
  out[threadIdx.x] = v == (w0 | (w1 << 16)) ? 1u : 0u;
  out[64 + threadIdx.x] = v;
}
int main() {
  unsigned* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int ok = 0; for (int i = 0; i < 64; ++i) ok += h[i];
  printf("ds_read_b32 at 2-byte alignment: %d of 64 lanes equal the two ds_read_u16 (lane 1: %08x)\n", ok, h[65]);
  return 0;
}
