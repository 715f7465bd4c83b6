// Semantics check (on the GPU box) of the hand-picked instructions the fast decoder uses:
//   ds_read_addtid_b32 (address = M0[15:0] + offset + 4 * lane), v_subb_co_u32 with an SGPR-pair
//   borrow-in, v_cmpx + v_readfirstlane (first active lane), v_mov_b32_dpp wave_shr:1 bound_ctrl:1 (lane 0 reads 0), s_ff1_i32_b64 of 0 (-1).
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/asm_semantics.hip -o /tmp/asm_sem && /tmp/asm_sem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out, const int* in, int base_bytes) {
  extern __shared__ int lds[];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  int m = __builtin_amdgcn_readfirstlane(base_bytes);
  int v;
  asm volatile("s_mov_b32 m0, %1\n s_nop 0\n ds_read_addtid_b32 %0 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "s"(m) : "memory");
  unsigned int B = in[threadIdx.x] * 3 + 7, A = in[threadIdx.x];
  unsigned int t1;
  unsigned long long ones = ~0ull;
  asm volatile("v_subb_co_u32 %0, vcc, %1, %2, %3" : "=v"(t1) : "v"(B), "v"(A), "s"(ones) : "vcc");
  unsigned int sh;
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(sh) : "v"(B));
  unsigned long long zero = __ballot(B == 0xFFFFFFFFu);
  int L;
  asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(L) : "s"(zero));
  // first hit through EXEC: lanes with B >= thr become active, v_readfirstlane reads the lowest
  unsigned int thr = __builtin_amdgcn_readfirstlane(in[20]) * 3 + 7;   // = B of lane 20
  int first_val, first_lane, lanev = threadIdx.x;
  asm volatile("v_cmpx_le_u32 vcc, %2, %3\n\t"
               "v_readfirstlane_b32 %0, %3\n\t"
               "v_readfirstlane_b32 %1, %4\n\t"
               "s_mov_b64 exec, -1"
               : "=&s"(first_val), "=&s"(first_lane) : "s"(thr), "v"(B), "v"(lanev) : "vcc");
  out[256 + threadIdx.x] = first_val;
  out[320 + threadIdx.x] = first_lane;
  out[threadIdx.x] = v;
  out[64 + threadIdx.x] = t1;
  out[128 + threadIdx.x] = sh;
  out[192 + threadIdx.x] = L;
}
int main() {
  std::vector<int> in(256), out(384);
  for (int i = 0; i < 256; ++i) in[i] = 1000 + i;
  int *din, *dout;
  hipMalloc(&din, 1024); hipMalloc(&dout, 1536);
  hipMemcpy(din, in.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dout, din, 40);
  hipMemcpy(out.data(), dout, 1536, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int want_v = in[(40 + 8) / 4 + l];
    const unsigned B = in[l] * 3 + 7, A = in[l];
    const unsigned want_t = B - A - 1;
    const unsigned want_sh = l == 0 ? 0u : (unsigned)(in[l - 1] * 3 + 7);
    if (out[l] != want_v) { if (!bad++) printf("addtid lane %d: %d want %d\n", l, out[l], want_v); }
    if ((unsigned)out[64 + l] != want_t) { if (!bad++) printf("subb lane %d: %u want %u\n", l, out[64 + l], want_t); }
    if ((unsigned)out[128 + l] != want_sh) { if (!bad++) printf("dpp lane %d: %u want %u\n", l, out[128 + l], want_sh); }
    if (out[256 + l] != in[20] * 3 + 7 || out[320 + l] != 20) { if (!bad++) printf("cmpx/rfl lane %d: %d %d\n", l, out[256 + l], out[320 + l]); }
    if (out[192 + l] != -1) { if (!bad++) printf("ff1 lane %d: %d want -1\n", l, out[192 + l]); }
  }
  printf(bad ? "FAILED (%d)\n" : "asm semantics ok\n", bad);
  return bad != 0;
}
