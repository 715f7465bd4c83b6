// Copy bandwidth of a [pixels, 192] bf16 tensor (the GDN C3 tensor) for the per-instruction access
// patterns a GDN kernel could use.  One wave copies tiles of 32 pixels x 384 bytes with 12 loads and
// 12 stores of 16 bytes per lane; only the lane -> address mapping differs:
//   rows32x32B : lane (pixel l%32, half l/32): 32 rows x 32 bytes per instruction   (shipped kernel)
//   rows16x64B : lane (pixel l%16, quarter l/16): 16 rows x 64 bytes                (16x16x32 MFMA layout)
//   rows8x128B : 8 rows x 128 bytes
//   linear     : 1 KB contiguous per instruction
// and mixed load/store patterns.  Build: hipcc --offload-arch=gfx950 -O3 -o copy_patterns copy_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kRowBytes = 384;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// byte offset inside a 32-pixel tile (12288 bytes) of lane `l`'s 16 bytes for instruction `s` (0..11)
template <int P> __device__ inline int offset(int l, int s) {
  if (P == 0) return (l & 31) * kRowBytes + 32 * s + 16 * (l >> 5);                  // 32 rows x 32 B
  if (P == 1) {                                                                       // 16 rows x 64 B
    const int sub = s / 6, k = s % 6;                                                 // two 16-pixel sub-tiles
    return (16 * sub + (l & 15)) * kRowBytes + 64 * k + 16 * (l >> 4);
  }
  if (P == 2) {                                                                       // 8 rows x 128 B
    const int sub = s / 3, k = s % 3;
    return (8 * sub + (l & 7)) * kRowBytes + 128 * k + 16 * (l >> 3);
  }
  return 1024 * s + 16 * l;                                                           // linear
}

template <int PL, int PS, int NT = 0>
__global__ void __launch_bounds__(512) copy_kernel(const unsigned char* x, unsigned char* y, long long tiles) {
  const int lane = threadIdx.x & 63;
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = static_cast<long long>(gridDim.x) * (blockDim.x >> 6);
  for (long long t = wave; t < tiles; t += nwaves) {
    const unsigned char* src = x + t * 32 * kRowBytes;
    unsigned char* dst = y + t * 32 * kRowBytes;
    u32x4 v[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const u32x4* q = reinterpret_cast<const u32x4*>(src + offset<PL>(lane, s));
      v[s] = (NT & 1) ? __builtin_nontemporal_load(q) : *q;
    }
    if (PL != PS) {
      // different mapping on the way out: the data is not the same permutation, but the traffic is
#pragma unroll
      for (int s = 0; s < 12; ++s) v[s] += 1u;
    }
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      u32x4* q = reinterpret_cast<u32x4*>(dst + offset<PS>(lane, s));
      if (NT & 2) __builtin_nontemporal_store(v[s], q); else *q = v[s];
    }
  }
}

template <int PL, int PS, int NT = 0>
void run(const char* name, const unsigned char* x, unsigned char* y, long long pixels, int blocks) {
  const long long tiles = pixels / 32;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<PL, PS, NT>), dim3(blocks), dim3(512), 0, 0, x, y, tiles);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_kernel<PL, PS, NT>), dim3(blocks), dim3(512), 0, 0, x, y, tiles);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps;
  printf("%-28s blocks %5d  %7.1f us  %7.1f GB/s\n", name, blocks, us, 2.0 * pixels * kRowBytes / us / 1e3);
}

int main() {
  const long long pixels = 256LL * 32 * 32;
  unsigned char *x, *y;
  hipMalloc(&x, pixels * kRowBytes);
  hipMalloc(&y, pixels * kRowBytes);
  hipMemset(x, 1, pixels * kRowBytes);
  for (int blocks : {256, 512}) {
    run<0, 0>("rows32x32B -> rows32x32B", x, y, pixels, blocks);
    run<0, 0, 1>("rows32x32B nt loads", x, y, pixels, blocks);
    run<0, 0, 2>("rows32x32B nt stores", x, y, pixels, blocks);
    run<0, 0, 3>("rows32x32B nt both", x, y, pixels, blocks);
    run<3, 3, 3>("linear nt both", x, y, pixels, blocks);
    run<1, 1>("rows16x64B -> rows16x64B", x, y, pixels, blocks);
    run<2, 2>("rows8x128B -> rows8x128B", x, y, pixels, blocks);
    run<3, 3>("linear -> linear", x, y, pixels, blocks);
    run<3, 0>("linear -> rows32x32B", x, y, pixels, blocks);
    run<0, 3>("rows32x32B -> linear", x, y, pixels, blocks);
    run<3, 1>("linear -> rows16x64B", x, y, pixels, blocks);
    run<1, 3>("rows16x64B -> linear", x, y, pixels, blocks);
  }
  return 0;
}
