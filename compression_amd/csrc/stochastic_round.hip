// StochasticRound for gfx950 — the reference's CPU kernel
// (cc/kernels/quantization_kernels.cc:47-96) draws ONE xoshiro256+ number per element from a single
// generator, in flat element order.  The same numbers in parallel:
//
//   * the generator's state update is linear over GF(2), a 256x256 bit matrix T; element i uses the
//     state T^i s0.  A lane owns kLane consecutive elements and a wave 64 lanes' worth, so wave w starts
//     from T^(w * 64 * kLane) s0 and lane l from a further (T^kLane)^l;
//   * a matrix-vector product over GF(2) is done by the whole wave: lane l holds rows l, l+64, l+128,
//     l+192 (4 x 256 bits), ANDs them with the (wave-uniform) state, takes the parity, and the four
//     ballots ARE the four words of the new state.  ~70 instructions per product;
//   * wave start: the products T^(2^k * 64 * kLane) for the set bits k of w (matrices computed once per
//     device on the host by repeated squaring); lane starts: 63 chained products with T^kLane, each lane
//     keeping the state of its own turn;
//   * each lane then runs the generator serially, 64 draws at a time into LDS (row = lane), and the wave
//     consumes the tile four rows at a time so that the tensor traffic is coalesced (row r = 64
//     consecutive elements of lane r's segment = 16 lanes x 4 elements); the tile's inputs are requested
//     before the draws are generated, so their latency is covered by the generator.
//
// Per element: 4 (bf16 / f16: 2) bytes in, 4 out.
#include <chrono>
#include <map>
#include <mutex>
#include <random>
#include <vector>

#include <hip/hip_fp16.h>

#include "common.h"

namespace tfc {
namespace {

constexpr int kLaneLog2 = 8;
constexpr int kLane = 1 << kLaneLog2;          // elements per lane
constexpr int kWaveLog2 = kLaneLog2 + 6;       // elements per wave = 64 lanes
constexpr int kJumps = 40 - kWaveLog2;         // wave index bits covered (n < 2^40 elements)
constexpr int kTile = 64;                      // draws per lane per LDS tile
constexpr int kPitch = kTile + 1;              // LDS row pitch in words (conflict-free both ways)

struct State { uint64_t w[4]; };
struct BitMatrix { uint64_t row[256][4]; };    // row r, bit j (word j/64, bit j%64): new bit r <- old bit j

// ---- host: T and its powers ---------------------------------------------------------------------
// Same update as quantization_kernels.cc:35-45 (xoshiro256+), on a state given by value.
State advance(State s) {
  const uint64_t t = s.w[1] << 17;
  s.w[2] ^= s.w[0];
  s.w[3] ^= s.w[1];
  s.w[1] ^= s.w[2];
  s.w[0] ^= s.w[3];
  s.w[2] ^= t;
  s.w[3] = (s.w[3] << 45) | (s.w[3] >> 19);
  return s;
}

struct Columns {  // column form: col[j] = image of unit vector j
  State col[256];
  State apply(const State& v) const {
    State y{};
    for (int j = 0; j < 256; ++j)
      if ((v.w[j >> 6] >> (j & 63)) & 1)
        for (int q = 0; q < 4; ++q) y.w[q] ^= col[j].w[q];
    return y;
  }
  Columns squared() const {
    Columns out;
    for (int j = 0; j < 256; ++j) out.col[j] = apply(col[j]);
    return out;
  }
  void rows(BitMatrix* m) const {
    for (int r = 0; r < 256; ++r)
      for (int q = 0; q < 4; ++q) m->row[r][q] = 0;
    for (int j = 0; j < 256; ++j)
      for (int r = 0; r < 256; ++r)
        if ((col[j].w[r >> 6] >> (r & 63)) & 1) m->row[r][j >> 6] |= 1ull << (j & 63);
  }
};

// Device table: [0] = T^kLane, [1 + k] = T^(2^k * 64 * kLane).
struct JumpTable {
  BitMatrix* dev = nullptr;
};

int jump_table(const BitMatrix** out) {
  static std::mutex mu;
  static std::map<int, JumpTable> per_device;
  static std::vector<BitMatrix> host;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  TFC_HIP(hipGetDevice(&dev));
  auto it = per_device.find(dev);
  if (it == per_device.end()) {
    if (host.empty()) {
      host.resize(1 + kJumps);
      Columns m;
      for (int j = 0; j < 256; ++j) {
        State e{};
        e.w[j >> 6] = 1ull << (j & 63);
        m.col[j] = advance(e);
      }
      for (int e = 0; e < kWaveLog2 + kJumps; ++e) {  // m = T^(2^e)
        if (e == kLaneLog2) m.rows(&host[0]);
        if (e >= kWaveLog2) m.rows(&host[1 + e - kWaveLog2]);
        m = m.squared();
      }
    }
    JumpTable t;
    TFC_HIP(hipMalloc(reinterpret_cast<void**>(&t.dev), host.size() * sizeof(BitMatrix)));
    TFC_HIP(hipMemcpy(t.dev, host.data(), host.size() * sizeof(BitMatrix), hipMemcpyHostToDevice));
    it = per_device.emplace(dev, t).first;
  }
  *out = it->second.dev;
  return 0;
}

// ---- device ---------------------------------------------------------------------------------------
struct MyRows { uint64_t r[4][4]; };  // rows lane, lane+64, lane+128, lane+192

__device__ inline MyRows load_rows(const BitMatrix* m, int lane) {
  MyRows out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(m->row[q * 64 + lane]);
    const ulonglong2 a = p[0], b = p[1];
    out.r[q][0] = a.x; out.r[q][1] = a.y; out.r[q][2] = b.x; out.r[q][3] = b.y;
  }
  return out;
}

// Wave-uniform state in, wave-uniform state out.
__device__ inline State product(const MyRows& m, const State& s) {
  State out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint64_t x = (m.r[q][0] & s.w[0]) ^ (m.r[q][1] & s.w[1]) ^ (m.r[q][2] & s.w[2]) ^ (m.r[q][3] & s.w[3]);
    const uint32_t fold = static_cast<uint32_t>(x) ^ static_cast<uint32_t>(x >> 32);
    out.w[q] = __ballot(__popc(fold) & 1);
  }
  return out;
}

// Four consecutive elements as one load; `widen` is exact for the 16-bit types.
template <int DT> struct Elem;
template <> struct Elem<0> {
  using T = float;
  using Vec = float4;
  static __device__ float widen(T v) { return v; }
  static __device__ void unpack(const Vec& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
};
template <> struct Elem<1> {
  using T = uint16_t;
  using Vec = uint2;
  static __device__ float widen(T v) { return __uint_as_float(static_cast<uint32_t>(v) << 16); }
  static __device__ void unpack(const Vec& v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  }
};
template <> struct Elem<2> {
  using T = __half;
  using Vec = uint2;
  static __device__ float widen(T v) { return __half2float(v); }
  static __device__ void unpack(const Vec& v, float (&f)[4]) {
    f[0] = __half2float(__ushort_as_half(static_cast<unsigned short>(v.x)));
    f[1] = __half2float(__ushort_as_half(static_cast<unsigned short>(v.x >> 16)));
    f[2] = __half2float(__ushort_as_half(static_cast<unsigned short>(v.y)));
    f[3] = __half2float(__ushort_as_half(static_cast<unsigned short>(v.y >> 16)));
  }
};

struct RoundParams {
  const void* x;
  int32_t* out;
  int64_t n;
  float step;
  State s0;
  const BitMatrix* jumps;
};

template <int DT>
__global__ __launch_bounds__(64) void stochastic_round_kernel(RoundParams p) {
  using E = Elem<DT>;
  __shared__ uint32_t tile[64 * kPitch];
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x;
  const int64_t base = wave << kWaveLog2;
  const int64_t left = p.n - base;
  const int lanes_used = left >= (64 << kLaneLog2) ? 64 : static_cast<int>((left + kLane - 1) >> kLaneLog2);

  // wave start: T^(wave * 64 * kLane) s0
  State s = p.s0;
  for (int k = 0; (wave >> k) != 0; ++k) {
    if ((wave >> k) & 1) s = product(load_rows(p.jumps + 1 + k, lane), s);
  }
  // lane starts: lane l keeps (T^kLane)^l s
  State mine = s;
  {
    const MyRows step_rows = load_rows(p.jumps, lane);
    for (int l = 1; l < lanes_used; ++l) {
      s = product(step_rows, s);
      if (lane == l) mine = s;
    }
  }

  // A tile = kTile draws of each lane = 64 rows (one per lane segment) of kTile consecutive elements.
  // It is consumed 4 rows at a time: 16 lanes x 4 elements per row, so a row is one 256-byte access.
  const typename E::T* x = static_cast<const typename E::T*>(p.x);
  const float step = p.step;
  const int row0 = lane >> 4, col = (lane & 15) * 4;
  for (int sub = 0; sub < kLane / kTile; ++sub) {
    const int64_t tile_base = base + sub * kTile;  // + r * kLane for row r
    if (tile_base >= p.n) break;                   // uniform: nothing of any row is left
    // the tile's inputs are requested first; the generator below runs while they arrive
    typename E::Vec xin[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t i = tile_base + (static_cast<int64_t>(it * 4 + row0) << kLaneLog2) + col;
      if (i + 4 <= p.n) xin[it] = *reinterpret_cast<const typename E::Vec*>(x + i);
    }
#pragma unroll 8
    for (int j = 0; j < kTile; ++j) {
      tile[lane * kPitch + j] = static_cast<uint32_t>((mine.w[0] + mine.w[3]) >> 40);
      const uint64_t t = mine.w[1] << 17;
      mine.w[2] ^= mine.w[0];
      mine.w[3] ^= mine.w[1];
      mine.w[1] ^= mine.w[2];
      mine.w[0] ^= mine.w[3];
      mine.w[2] ^= t;
      mine.w[3] = (mine.w[3] << 45) | (mine.w[3] >> 19);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + row0;
      const int64_t i = tile_base + (static_cast<int64_t>(r) << kLaneLog2) + col;
      const uint32_t* draws = &tile[r * kPitch + col];
      auto round_one = [&](float v, uint32_t d) {
        const float number = __fdiv_rn(v, step);
        const float integral = floorf(number);
        const float draw = static_cast<float>(d) * 0x1.0p-24f;
        return static_cast<int32_t>(integral) + (draw < number - integral ? 1 : 0);
      };
      if (i + 4 <= p.n) {
        float f[4];
        E::unpack(xin[it], f);
        int4 o;
        o.x = round_one(f[0], draws[0]); o.y = round_one(f[1], draws[1]);
        o.z = round_one(f[2], draws[2]); o.w = round_one(f[3], draws[3]);
        *reinterpret_cast<int4*>(p.out + i) = o;
      } else {
        for (int k = 0; k < 4 && i + k < p.n; ++k) p.out[i + k] = round_one(E::widen(x[i + k]), draws[k]);
      }
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace tfc

extern "C" int tfc_stochastic_round(const void* inputs, int dtype, int64_t n, float step_size,
                                    const int32_t* seed, int64_t seed_len, int32_t* outputs, void* stream) {
  using namespace tfc;
  if (dtype < 0 || dtype > 2)
    return fail("tfc_stochastic_round: dtype must be 0 (float32), 1 (bfloat16) or 2 (float16)");
  if (n < 0 || seed_len < 0) return fail("tfc_stochastic_round: negative size");
  if (n >= (1ll << 40)) return fail("tfc_stochastic_round: at most 2^40 elements");
  if (n == 0) return 0;
  RoundParams p{};
  p.x = inputs; p.out = outputs; p.n = n; p.step = step_size;
  uint32_t words[8];
  if (seed_len > 0) {
    std::seed_seq seq(seed, seed + seed_len);
    seq.generate(words, words + 8);
  } else {  // quantization_kernels.cc:75-81: best-effort seeding from the clock
    const uint64_t now = std::chrono::high_resolution_clock::now().time_since_epoch().count();
    std::seed_seq seq{static_cast<uint32_t>(now), static_cast<uint32_t>(now >> 32)};
    seq.generate(words, words + 8);
  }
  for (int i = 0; i < 4; ++i) p.s0.w[i] = words[2 * i] | (static_cast<uint64_t>(words[2 * i + 1]) << 32);
  if (int rc = jump_table(&p.jumps)) return rc;
  const int64_t waves = (n + (1ll << kWaveLog2) - 1) >> kWaveLog2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(stochastic_round_kernel<0>, dim3(waves), dim3(64), 0, st, p); break;
    case 1: hipLaunchKernelGGL(stochastic_round_kernel<1>, dim3(waves), dim3(64), 0, st, p); break;
    default: hipLaunchKernelGGL(stochastic_round_kernel<2>, dim3(waves), dim3(64), 0, st, p); break;
  }
  TFC_HIP(hipGetLastError());
  return 0;
}
