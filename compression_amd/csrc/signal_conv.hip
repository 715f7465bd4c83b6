// SignalConv2D for gfx950: 2-D, non-separable, `same_zeros`, NHWC
// (python/layers/signal_conv.py:663-690 `_correlate_down_explicit` and
//  :778-847 `_up_convolve_transpose_explicit`, extra_pad_end=True).
//
// ONE implicit-GEMM kernel serves both directions:
//   down (analysis):  y[q] = sum_u x[q*sd + u - k/2] * w[u]            (cross-correlation)
//   up (synthesis):   y[q*s + phi] = sum_i x[i] * w[phi + (q-i)*s + k/2]  (transposed conv)
// The upsampled convolution is decomposed into its s*s output phases; each phase
// is a small stride-1 correlation over the LOW-resolution input, and the phases
// become extra output columns (N' = s*s*Cout) that the epilogue scatters back
// (depth-to-space).  No zero-insertion, no wasted multiplies on inserted zeros.
//
// GEMM orientation is the same as in gdn.hip: D^T = W'^T * patches^T, i.e. the
// MFMA A operand is the packed weight tile (rows = output columns) staged in
// LDS, the B operand is 8 consecutive input channels of ONE pixel per lane —
// a plain 16-byte load from the NHWC tensor, zero-filled outside the image.
// A lane therefore ends up with groups of 4 consecutive output channels of one
// pixel, which leave as 8/16-byte stores after bias (+ ReLU).
//
// bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  f32: v_mfma_f32_32x32x2_f32
// (exact fp32 FMA chains; 1/16 of the bf16 rate) for the parity path.
// Cin must be a multiple of 16, or <= 4 (image input: the host packs it to a
// zero-bordered 4-channel buffer and whole kernel rows become contiguous K runs).
// Roofline: MFMA-bound, 2*M*K*N FLOP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/tfc_hip.h"
#include "common.h"
#include "gdn_params.h"

namespace tfc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kConvWaves = 4;          // pixel tiles (32 px) per workgroup
#ifndef TFC_CONV_MAX_TILES
#define TFC_CONV_MAX_TILES 6
#endif
constexpr int kMaxTiles = TFC_CONV_MAX_TILES;   // 32-column tiles per column group (6: 192 columns)
constexpr int kMaxGroups = 16;         // column groups that can carry their own tap sub-rectangle

struct ConvGeom {
  // input
  long long N;
  int H, W, Cin;          // Cin as seen by the kernel (4 for packed image input)
  int Hp, Wp;             // packed image input: padded extents (else H, W)
  // low-resolution output grid the GEMM rows run over
  int OHq, OWq;
  int sd;                 // input step per output row/col
  int Uy, Ux;             // taps of the equivalent correlation
  int py0, px0;           // zero padding before
  // columns
  int Cout, su;           // real output channels, depth-to-space factor
  int cols;               // su*su*Cout
  int groups;             // column groups of `tiles` 32-wide tiles
  int tiles;
  // K
  int ksteps;             // K steps of 16
  int kchunk;             // K steps staged in LDS at a time
  int small_cin;          // 1: packed-image mode (K runs along kernel rows)
  int kw4;                // small_cin: K steps per kernel row
  int activation;         // 0 none, 1 relu
  int out_f32;            // bf16 kernels: write the fp32 accumulators (the tap products of conv_up_small_cout)
  // output
  int OH, OW;
  // Compact K (second-generation bf16 kernel, transposed convolution whose column groups are whole output
  // phases): a phase only has the taps t = phi + d*s + k/2 inside the kernel, e.g. 3x3, 3x2, 2x3, 2x2 of the
  // 3x3 taps of a 5x5 stride-2 kernel, so a group's K loop (and its packed weights) runs over the
  // sub-rectangle [ty0, ty1) x [tx0, tx1) of taps only: 25 instead of 36 tap blocks in that example.
  int compact;
  int ty0[kMaxGroups], ty1[kMaxGroups], tx0[kMaxGroups], tx1[kMaxGroups];
  // Third-generation kernel: K runs channel block by channel block, a group's taps inside each
  // (K step = cbi * taps + tap); with `compact` tap rectangles for every group.
  int cbmajor;
  // GDN / IGDN as the layer's activation (third-generation kernel: a workgroup holds all channels of its pixels):
  // 0 none, 1 y / (beta + gamma^T |y|), 2 y * (beta + gamma^T |y|); the prepared bfloat16 image of tfc_gdn_params
  int gdn;
  const void* gdn_image;
  int xcd;                  // 1: the third-generation kernel's workgroups take their blocks in XCD order (xcd_order)
  int nt_out;               // third generation: the output's whole-line stores non-temporal (an output beyond the caches)
  // first / second generation: pixel -> (image, row, column) with the divisions as multiplications where the launch has
  // fewer than 2^31 low-resolution pixels (pix32; fast_div by OWq, OHq)
  unsigned int owq_mul, owq_sh, ohq_mul, ohq_sh;
  int pix32;
};

// Third-generation kernel: a workgroup computes an 8 x 32 block of low-resolution output pixels of ONE image
// from an input patch staged in LDS one 16-channel block at a time.
struct Conv3Geom {
  int BXn, BYn;            // blocks per image row / column
  int PW, PWh;             // patch columns; columns per x-parity plane (ceil(PW / sd))
  int lg;                  // log2(sd), sd in {1, 2}
  int granules;            // 16-byte granules of a patch: PH * sd * 2 * PWh, order [row][x parity][h][x / sd]
  int pixels;              // PH * PW
  int gcount;              // column groups this launch covers ...
  int glist[kMaxGroups];   // ... and which
  int ostage;              // byte offset in LDS of the epilogue's staging area (4 waves x 8 KB), see the kernel
  int obias;               // ... and of the bias (Cout floats, parked by the prologue)
  int oimage;              // GDN as the activation: where gamma's fragment image lives in LDS
  unsigned int bx_mul, bx_sh, by_mul, by_sh, gc_mul, gc_sh, pw_mul, pw_sh;   // fast_div by BXn, BYn, gcount, PW
};

// n / d for 0 <= n < 2^31 as a multiplication: mul = ceil(2^(31 + s) / d), s = ceil(log2 d), n / d = (n * mul) >> (31 + s)
// exactly (mul * d - 2^(31 + s) < d <= 2^s: the error term is below 1 / d); mul = 0: d = 1.
inline void fast_div_setup(unsigned int d, unsigned int* mul, unsigned int* sh) {
  if (d <= 1) { *mul = 0; *sh = 0; return; }
  unsigned int s = 0;
  while ((1ull << s) < d) ++s;
  *mul = static_cast<unsigned int>(((1ull << (31 + s)) + d - 1) / d);
  *sh = s - 1;
}
__device__ inline unsigned int fast_div(unsigned int n, unsigned int mul, unsigned int sh) {
  return mul ? __umulhi(n, mul) >> sh : n;
}

template <typename T> struct ConvTraits;
template <> struct ConvTraits<__bf16> {
  static constexpr int kFragBytes = 16;        // A fragment bytes per lane per K step per tile
};
template <> struct ConvTraits<float> {
  static constexpr int kFragBytes = 32;
};

// ---------------------------------------------------------------------------
// Weight packing: A fragments in the order the main kernel reads them:
//   packed[((group * ksteps + ks) * tiles + t) * 64 + lane]  (16 B bf16 / 32 B f32)
// lane (i = lane & 31, h = lane >> 5), element e (0..7): K offset 8h + e of step ks,
// output column group*tiles*32 + 32 t + i.
// ---------------------------------------------------------------------------
// Workgroup -> work item, XCD-aware.  The dispatcher hands consecutive workgroup ids to the 8 XCDs in turn, each with
// its own L2: with work item = workgroup id, neighbouring blocks of an image — which read the same halo rows — run on
// different XCDs and each fetches its own copy.  xcd_order gives XCD x the x-th contiguous eighth of the items instead
// (a bijection for any count), so the blocks an XCD runs at one time are neighbours.  `on` = 0: the identity.
// Measured at the C4 shapes (same box, alternating): the third-generation kernel's transposed layers 8.05-8.14 -> 7.92-7.96
// ms (192x128 input) and 0.64 -> 0.61 ms (48x32); the second-generation stride-2 layers 6.79 -> 6.96 ms: only the third
// generation uses it.
constexpr unsigned int kXcds = 8;
__device__ inline unsigned int xcd_order(unsigned int id, unsigned int count, int on) {
  if (!on || count < 2 * kXcds) return id;
  const unsigned int x = id % kXcds, k = id / kXcds;
  const unsigned int per = count / kXcds, extra = count % kXcds;
  return x * per + (x < extra ? x : extra) + k;
}

// (TFC_CONV_XCD=0 in the environment: workgroup id = work item, for same-box comparisons)
inline int xcd_blocks() {
  static const int on = [] { const char* e = std::getenv("TFC_CONV_XCD"); return !(e && e[0] == '0'); }();
  return on;
}

struct PackGeom {
  int kh, kw, Cin_real, Cout, su, up;
  int Uy, Ux, dmax_y, dmax_x;
};

__device__ inline float packed_weight(const float* w, const PackGeom& g, const ConvGeom& c, int ks,
                                      int koff, int col) {
  if (col >= c.cols) return 0.f;
  int uy, ux, ci;
  if (c.small_cin) {
    uy = ks / c.kw4;
    const int o = (ks % c.kw4) * 16 + koff;
    ux = o >> 2;
    ci = o & 3;
    if (ux >= c.Ux || ci >= g.Cin_real) return 0.f;
  } else if (c.cbmajor) {
    const int grp = col / (c.tiles * 32);
    const int wx = c.tx1[grp] - c.tx0[grp];
    const int nt = (c.ty1[grp] - c.ty0[grp]) * wx;
    const int cbi = ks / nt, tap = ks % nt;
    if (cbi >= c.Cin / 16) return 0.f;
    uy = c.ty0[grp] + tap / wx;
    ux = c.tx0[grp] + tap % wx;
    ci = cbi * 16 + koff;
  } else if (c.compact) {
    const int grp = col / (c.tiles * 32);
    const int cb = c.Cin / 16;
    const int tap = ks / cb;
    const int wx = c.tx1[grp] - c.tx0[grp];
    uy = c.ty0[grp] + tap / wx;
    ux = c.tx0[grp] + tap % wx;
    if (uy >= c.ty1[grp]) return 0.f;
    ci = (ks % cb) * 16 + koff;
  } else {
    const int cb = c.Cin / 16;
    const int tap = ks / cb;
    uy = tap / c.Ux;
    ux = tap % c.Ux;
    ci = (ks % cb) * 16 + koff;
  }
  const int co = col % g.Cout;
  int ty, tx;
  if (g.up) {
    const int phase = col / g.Cout;
    const int phy = phase / g.su, phx = phase % g.su;
    // tap u reads input i = q + u - dmax, i.e. d = q - i = dmax - u; kernel index t = phi + d*s + k/2
    ty = phy + (g.dmax_y - uy) * g.su + g.kh / 2;
    tx = phx + (g.dmax_x - ux) * g.su + g.kw / 2;
    if (ty < 0 || ty >= g.kh || tx < 0 || tx >= g.kw) return 0.f;
  } else {
    ty = uy;
    tx = ux;
  }
  return w[((static_cast<long long>(ty) * g.kw + tx) * g.Cin_real + ci) * g.Cout + co];
}

// ---------------------------------------------------------------------------
// Packed weights of an inference layer, kept between calls.  Every kernel here reads the layer's float32 HWIO kernel as
// fragments in its own order, packed by a small kernel in front of it — 60-90 us each, four to nine a model step (0.35 ms
// of bls2017's 7.4 ms, profiles/r04_bls2017_stats.md).  A caller that knows the weights do not change between calls says
// so with tfc_conv2d_weights_key (a number that names this VALUE of the weights; include/tfc_hip.h): the fragments of
// (key, packing site, geometry) are then packed once and kept until tfc_conv2d_drop_weights(key).  Without a key —
// training, or weights that are tensors computed per call — every call packs, as before.
// The entry is made on the first caller's stream; another stream waits for its event (a completed event costs nothing).
// ---------------------------------------------------------------------------
thread_local unsigned long long t_next_weights_key = 0;      // set by tfc_conv2d_weights_key, taken by the next conv call
thread_local unsigned long long t_weights_key = 0;           // of the call in progress on this thread
struct WeightsCache {
  struct Key {
    unsigned long long key;
    int site, dev;
    long long dims[16];
    bool operator<(const Key& o) const {
      if (key != o.key) return key < o.key;
      if (site != o.site) return site < o.site;
      if (dev != o.dev) return dev < o.dev;
      return std::lexicographical_compare(dims, dims + 16, o.dims, o.dims + 16);
    }
  };
  struct Entry {
    DevBuf buf;
    hipEvent_t ready = nullptr;
    hipStream_t made_on = nullptr;
    std::vector<hipStream_t> users;          // other streams whose kernels have read the fragments (a handful)
  };
  std::mutex mu;
  std::map<Key, Entry> entries;
  static WeightsCache& get() {
    static WeightsCache* c = new WeightsCache;               // leaked: entries may outlive static destruction order
    return *c;
  }
};

// The packed weights of this call: *p = `bytes` of fragments, written by pack(p) on `st` — now into `local` (no key), or
// once into the cache.  site: which packing (the kernels' orders differ); dims: whatever the packing depends on.
template <typename Pack>
int packed_weights(int site, std::initializer_list<long long> dims, size_t bytes, hipStream_t st, DevBuf& local, void** p,
                   Pack&& pack) {
  const unsigned long long key = t_weights_key;
  if (!key) {
    TFC_HIP(local.alloc(bytes, st));
    *p = local.p;
    return pack(local.p);
  }
  WeightsCache& c = WeightsCache::get();
  WeightsCache::Key k{};
  k.key = key; k.site = site;
  (void)hipGetDevice(&k.dev);
  int i = 0;
  for (long long d : dims) k.dims[i++] = d;
  k.dims[15] = static_cast<long long>(bytes);
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.entries.find(k);
  if (it == c.entries.end()) {
    WeightsCache::Entry e;
    TFC_HIP(e.buf.alloc(bytes, st));
    const int rc = pack(e.buf.p);
    if (rc) return rc;
    TFC_HIP(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
    TFC_HIP(hipEventRecord(e.ready, st));
    e.made_on = st;
    it = c.entries.emplace(k, std::move(e)).first;
  } else if (it->second.made_on != st) {
    TFC_HIP(hipStreamWaitEvent(st, it->second.ready, 0));
    auto& users = it->second.users;
    if (std::find(users.begin(), users.end(), st) == users.end()) users.push_back(st);
  }
  *p = it->second.buf.p;
  return 0;
}

template <typename T>
__global__ void conv_pack_kernel(const float* w, PackGeom g, ConvGeom c, void* packed) {
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  const long long total = static_cast<long long>(c.groups) * c.ksteps * c.tiles * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  long long rest = idx >> 6;
  const int t = rest % c.tiles; rest /= c.tiles;
  const int ks = rest % c.ksteps;
  const int group = static_cast<int>(rest / c.ksteps);
  const int i = lane & 31, h = lane >> 5;
  const int col = (group * c.tiles + t) * 32 + i;
  if constexpr (std::is_same<T, __bf16>::value) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = static_cast<__bf16>(packed_weight(w, g, c, ks, 8 * h + e, col));
    static_cast<bf16x8*>(packed)[idx] = v;
  } else {
    // f32: K offset of MFMA step u (0..7), half h is 8h + u -> store u-major
    float* dst = static_cast<float*>(packed) + idx * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = packed_weight(w, g, c, ks, 8 * h + e, col);
  }
}

// Image input (Cin <= 4): zero-bordered 4-channel copy so that a kernel row is one
// contiguous K run.  xp[n][y + py0][x + px0][c] = x[n][y][x][c].
template <typename T>
__global__ void conv_pad_image_kernel(const T* x, ConvGeom c, int cin_real, T* xp) {
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  const long long total = c.N * c.Hp * c.Wp;
  if (idx >= total) return;
  const int xw = idx % c.Wp;
  const int yh = (idx / c.Wp) % c.Hp;
  const long long n = idx / (static_cast<long long>(c.Wp) * c.Hp);
  const int sy = yh - c.py0, sx = xw - c.px0;
  T v[4] = {T(0.f), T(0.f), T(0.f), T(0.f)};
  if (sy >= 0 && sy < c.H && sx >= 0 && sx < c.W)
    for (int ch = 0; ch < cin_real; ++ch) v[ch] = x[((n * c.H + sy) * c.W + sx) * cin_real + ch];
  for (int ch = 0; ch < 4; ++ch) xp[idx * 4 + ch] = v[ch];
}

// ---------------------------------------------------------------------------
// Main kernel
// ---------------------------------------------------------------------------
template <typename T, int TILES>
__global__ void __launch_bounds__(64 * kConvWaves) conv_kernel(const T* x, const void* packed,
                                                               const float* bias, T* y, ConvGeom c) {
  constexpr bool BF = std::is_same<T, __bf16>::value;
  constexpr int FB = ConvTraits<T>::kFragBytes;
  extern __shared__ unsigned char smem[];          // kchunk * TILES * 64 fragments

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int h = lane >> 5;
  const int group = blockIdx.x % c.groups;         // column groups innermost: neighbours share inputs
  const long long pblock = blockIdx.x / c.groups;
  const long long M = c.N * c.OHq * c.OWq;
  const long long m = (pblock * kConvWaves + wid) * 32 + (lane & 31);
  const bool live = m < M;
  const long long mm = live ? m : M - 1;
  int qx, qy;
  long long n;
  if (c.pix32) {          // (a 64-bit division is ~150 instructions in front of the first request)
    const unsigned int m32 = static_cast<unsigned int>(mm);
    const unsigned int r1 = fast_div(m32, c.owq_mul, c.owq_sh);
    qx = static_cast<int>(m32 - r1 * static_cast<unsigned int>(c.OWq));
    const unsigned int n1 = fast_div(r1, c.ohq_mul, c.ohq_sh);
    qy = static_cast<int>(r1 - n1 * static_cast<unsigned int>(c.OHq));
    n = n1;
  } else {
    qx = mm % c.OWq;
    qy = (mm / c.OWq) % c.OHq;
    n = mm / (static_cast<long long>(c.OWq) * c.OHq);
  }

  f32x16 acc[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned char* wsrc = static_cast<const unsigned char*>(packed) +
                              static_cast<size_t>(group) * c.ksteps * TILES * 64 * FB;
  const int cb = c.small_cin ? 1 : c.Cin / 16;

  for (int k0 = 0; k0 < c.ksteps; k0 += c.kchunk) {
    const int kn = min(c.kchunk, c.ksteps - k0);
    __syncthreads();
    {  // stage the chunk's weight fragments: contiguous 16-byte copies
      const u32x4* src = reinterpret_cast<const u32x4*>(wsrc + static_cast<size_t>(k0) * TILES * 64 * FB);
      u32x4* dst = reinterpret_cast<u32x4*>(smem);
      const int n16 = kn * TILES * 64 * FB / 16;
      for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    for (int kk = 0; kk < kn; ++kk) {
      const int ks = k0 + kk;
      // ---- B fragment: 8 input values of this lane's pixel at K offset 8h ----
      const T* src;
      bool ok = live;
      if (c.small_cin) {
        const int uy = ks / c.kw4;
        const int seg = ks % c.kw4;
        // zero-bordered image: always in bounds (the buffer carries a right margin)
        src = x + ((n * c.Hp + (qy * c.sd + uy)) * c.Wp + qx * c.sd) * 4 + seg * 16 + 8 * h;
      } else {
        const int tap = ks / cb;
        const int uy = tap / c.Ux, ux = tap % c.Ux;
        const int iy = qy * c.sd + uy - c.py0, ix = qx * c.sd + ux - c.px0;
        ok = ok && iy >= 0 && iy < c.H && ix >= 0 && ix < c.W;
        src = x + ((n * c.H + (ok ? iy : 0)) * c.W + (ok ? ix : 0)) * c.Cin + (ks % cb) * 16 + 8 * h;
      }
      if constexpr (BF) {
        u32x4 v = ok ? *reinterpret_cast<const u32x4*>(src) : u32x4{0u, 0u, 0u, 0u};
        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
        const bf16x8* a = reinterpret_cast<const bf16x8*>(smem) + (kk * TILES) * 64 + lane;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t * 64], bfrag, acc[t], 0, 0, 0);
      } else {
        f32x4 v0 = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = ok ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* a = reinterpret_cast<const f32x4*>(smem) + ((kk * TILES) * 64 + lane) * 2;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const f32x4 a0 = a[t * 128], a1 = a[t * 128 + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], v0[e], acc[t], 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], v1[e], acc[t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: acc[t][4q + r] = column group_base + 32t + 8q + 4h + r of pixel m ----
#ifdef TFC_CONV_NOSTORE
  if (!live || !bias || bias[0] != 12345.f) return;
#else
  if (!live) return;
#endif
  const int colbase = group * TILES * 32;
  if constexpr (BF) {
    if (c.su == 1 && (c.Cout & 7) == 0) {
      // 16-byte stores: lanes l and l + 32 (one pixel) trade halves of a pair of column groups, see the
      // second-generation kernel's epilogue
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int col0 = colbase + 32 * t + 16 * qp;
          if (col0 >= c.cols) continue;
          u32x4 o;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int colq = col0 + 8 * half + 4 * h;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float b = (bias && colq + r < c.cols) ? bias[colq + r] : 0.f;
              v[r] = acc[t][4 * (2 * qp + half) + r] + b;
              if (c.activation == 1) v[r] = fmaxf(v[r], 0.f);
            }
            o[2 * half] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
            o[2 * half + 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
          const int colh = col0 + 8 * h;
          if (colh < c.cols)
            *reinterpret_cast<u32x4*>(y + mm * c.Cout + colh) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      return;
    }
  }
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col0 = colbase + 32 * t + 8 * q + 4 * h;
      if (col0 >= c.cols) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = col0 + r;
        float b = 0.f;
        if (bias && col < c.cols) b = bias[col % c.Cout];
        v[r] = acc[t][4 * q + r] + b;
        if (c.activation == 1) v[r] = fmaxf(v[r], 0.f);
      }
      if (c.su == 1 && (c.Cout & 3) == 0) {
        T* dst = y + mm * c.Cout + col0;
        if constexpr (BF) {
          u32x2 o;
          o.x = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          o.y = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          *reinterpret_cast<u32x2*>(dst) = o;
        } else {
          *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = col0 + r;
          if (col >= c.cols) continue;
          const int co = col % c.Cout, phase = col / c.Cout;
          const int oy = qy * c.su + phase / c.su, ox = qx * c.su + phase % c.su;
          y[((n * c.OH + oy) * c.OW + ox) * c.Cout + co] = static_cast<T>(v[r]);
        }
      }
    }
}

// ---------------------------------------------------------------------------
// bf16 main kernel, second generation.  The first kernel above (still used for f32) reads its
// six A fragments from LDS for every 6 MFMAs, which saturates LDS (4 waves x 6 KB per K step),
// and waits for each K step's B load.  Here
//   * a wave owns MT pixel tiles (MT * 32 pixels) x TILES column tiles, so an A fragment read
//     feeds MT MFMAs (MT = 2: 12 accumulators = 192 AGPRs, one wave per SIMD);
//   * A fragments are double-buffered in registers (the next K step's reads fly under this
//     step's MFMAs) and B fragments are fetched PF K steps ahead into a register ring;
//   * weight chunks are double-buffered in LDS: the next chunk travels global -> registers while
//     the current one is consumed, registers -> LDS at the chunk boundary (one barrier).
// ---------------------------------------------------------------------------
#ifndef TFC_CONV_PF
#define TFC_CONV_PF 4
#endif
#ifndef TFC_CONV_CHUNK
#define TFC_CONV_CHUNK 4
#endif
#ifndef TFC_CONV_INTERLEAVE
#define TFC_CONV_INTERLEAVE 1
#endif
#ifndef TFC_CONV_NO_FASTK
#define TFC_CONV_NO_FASTK 0
#endif
constexpr int kPF = TFC_CONV_PF;         // B fragments in flight per pixel tile
constexpr int kChunk2 = TFC_CONV_CHUNK;  // K steps per LDS weight buffer (multiple of kPF)

#ifndef TFC_CONV_WGS
#define TFC_CONV_WGS 1
#endif
// FASTK (the host sets it when the channel blocks of a tap are a whole number of weight chunks, Cin % 64 == 0,
// and B fragments are requested exactly one chunk ahead): all K steps of a chunk then belong to ONE tap, so
// the tap's coordinates, the bounds test and the per-lane pointer (the zero page for lanes outside the image)
// are set up once per chunk and the K steps' loads are that pointer plus an immediate offset — no address
// arithmetic, no tap walk and no selects inside the K step.
template <int TILES, int MT, bool FASTK, bool OUTF32 = false>
__global__ void __launch_bounds__(256, TFC_CONV_WGS) conv_bf16_kernel(const __bf16* x, const void* packed,
                                                        const float* bias, __bf16* y, ConvGeom c) {
  static_assert(!FASTK || kPF == kChunk2, "FASTK requests the B fragments of the next chunk during this one");
  extern __shared__ unsigned char smem[];          // 2 x kChunk2 * TILES * 64 fragments of 16 B
  constexpr int CHUNK_FRAGS = kChunk2 * TILES * 64;
  constexpr int STAGE = (CHUNK_FRAGS + 255) / 256;  // 16-byte pieces each thread moves per chunk
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int h = lane >> 5;
  // (workgroups in XCD order, xcd_order, were measured on this kernel: 6.79 -> 6.96 ms for the 384x256 stride-2 layer)
  const int group = blockIdx.x % c.groups;
  const long long pblock = blockIdx.x / c.groups;
  const long long M = c.N * c.OHq * c.OWq;

  long long nn[MT];
  int qy[MT], qx[MT];
  bool live[MT];
  long long mm[MT];
  // per lane, fixed for the whole K loop: element offset of (n, qy*sd - py0, qx*sd - px0, 8h) and
  // the two coordinates the taps are added to (kept as int: the bounds test is two unsigned compares)
  long long base_off[MT];
  int iy0[MT], ix0[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    const long long m = ((pblock * 4 + wid) * MT + p) * 32 + (lane & 31);
    live[p] = m < M;
    mm[p] = live[p] ? m : M - 1;
    if (c.pix32) {          // (a 64-bit division is ~150 instructions in front of the first request)
      const unsigned int m32 = static_cast<unsigned int>(mm[p]);
      const unsigned int r1 = fast_div(m32, c.owq_mul, c.owq_sh);
      qx[p] = static_cast<int>(m32 - r1 * static_cast<unsigned int>(c.OWq));
      const unsigned int n1 = fast_div(r1, c.ohq_mul, c.ohq_sh);
      qy[p] = static_cast<int>(r1 - n1 * static_cast<unsigned int>(c.OHq));
      nn[p] = n1;
    } else {
      qx[p] = static_cast<int>(mm[p] % c.OWq);
      qy[p] = static_cast<int>((mm[p] / c.OWq) % c.OHq);
      nn[p] = mm[p] / (static_cast<long long>(c.OWq) * c.OHq);
    }
    iy0[p] = qy[p] * c.sd - c.py0;
    ix0[p] = qx[p] * c.sd - c.px0;
    base_off[p] = ((nn[p] * c.H + iy0[p]) * c.W + ix0[p]) * c.Cin + 8 * h;
  }
  const int cb = c.Cin / 16;
  // 16 zero bytes behind the packed weights: where lanes outside the image (or the tensor) read
  const __bf16* zeros = reinterpret_cast<const __bf16*>(
      static_cast<const unsigned char*>(packed) + static_cast<size_t>(c.groups) * c.ksteps * TILES * 64 * 16);

  // Position of a K step inside the (tap row, tap column, 16-channel block) order, advanced
  // incrementally: the per-step divisions by runtime Cin / kernel width were the single largest
  // block of non-MFMA instructions in the loop.
  struct KPos { int uy, ux, cbi; long long off; };      // off = (uy * W + ux) * Cin + 16 * cbi
  // taps of this column group: all of them, or the sub-rectangle of its output phase
  const int uy0 = c.compact ? c.ty0[group] : 0, uy1 = c.compact ? c.ty1[group] : c.Uy;
  const int ux0 = c.compact ? c.tx0[group] : 0, ux1 = c.compact ? c.tx1[group] : c.Ux;
  const int ksteps = (uy1 - uy0) * (ux1 - ux0) * cb;
  const long long row_step = static_cast<long long>(c.W - (ux1 - ux0)) * c.Cin + 16;   // last tap of a row -> next row
  // (selects, not branches: the unrolled K steps of a chunk stay ONE basic block, which is what lets the
  // scheduler put their address arithmetic, LDS reads and loads between the MFMAs)
  auto advance = [&](KPos& k) {
    const int cbi1 = k.cbi + 1;
    const bool wc = cbi1 == cb;                      // next channel block, or the next pixel of the row: contiguous
    const int uxn = k.ux + (wc ? 1 : 0);
    const bool wx = uxn == ux1;                      // last tap of a row -> next row
    k.cbi = wc ? 0 : cbi1;
    k.ux = wx ? ux0 : uxn;
    k.uy += wx ? 1 : 0;
    k.off += wx ? row_step : 16;
  };
  // B fragment of pixel tile p at K position k: 8 input values at K offset 8h, zero outside the image
  auto bload = [&](const KPos& k, int p) -> u32x4 {
    // (& on purpose: && is control flow, an EXEC-masked block per term)
    const bool ok = live[p] & (static_cast<unsigned int>(iy0[p] + k.uy) < static_cast<unsigned int>(c.H)) &
                    (static_cast<unsigned int>(ix0[p] + k.ux) < static_cast<unsigned int>(c.W)) & (k.uy < uy1);
    // address as an integer select (a pointer select became an EXEC-masked block per load)
    unsigned long long a_in = reinterpret_cast<unsigned long long>(x) +
                              2ull * static_cast<unsigned long long>(base_off[p] + k.off);
    asm volatile("" : "+v"(a_in));     // computed for every lane: the select below stays two v_cndmask
#ifdef TFC_CONV_NOBLOAD     // experiment: the K loop without its input traffic
    const unsigned long long a = ok && k.uy > 1000 ? a_in : reinterpret_cast<unsigned long long>(zeros);
#else
    const unsigned long long a = ok ? a_in : reinterpret_cast<unsigned long long>(zeros);
#endif
    return *reinterpret_cast<__attribute__((address_space(1))) const u32x4*>(a);     // a global, not a flat, load
  };

  f32x16 acc[MT][TILES];
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;

  const u32x4* wsrc = reinterpret_cast<const u32x4*>(
      static_cast<const unsigned char*>(packed) + static_cast<size_t>(group) * c.ksteps * TILES * 64 * 16);
  const int nchunks = (ksteps + kChunk2 - 1) / kChunk2;
  const long long wtotal = static_cast<long long>(ksteps) * TILES * 64;     // fragments of this group in use

  u32x4 stage[STAGE];
  // Unconditional loads from a clamped index (fragments past the chunk / the group's K steps are never
  // read by an MFMA that counts: their B operand is zero).  A load under a condition is a path WITHOUT
  // the load for the compiler's s_waitcnt bookkeeping: it then allows fewer loads in flight at the first
  // MFMA of a chunk than really are, and the wave waits there for the weights it has just requested
  // (SQ_WAIT_ANY was 30 % of the wave cycles of the 192 -> 192 layers).
  auto wfetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < STAGE; ++i) {
      const long long f = static_cast<long long>(chunk) * CHUNK_FRAGS + i * 256 + threadIdx.x;
      stage[i] = wsrc[f < wtotal ? f : wtotal - 1];
    }
  };
  auto wstore = [&](int buf) {
    u32x4* dst = reinterpret_cast<u32x4*>(smem) + buf * CHUNK_FRAGS;
#pragma unroll
    for (int i = 0; i < STAGE; ++i)
      if (i * 256 + static_cast<int>(threadIdx.x) < CHUNK_FRAGS) dst[i * 256 + threadIdx.x] = stage[i];
  };

  // prologue: chunk 0 into buffer 0, first PF B fragments
  wfetch(0);
  wstore(0);
  u32x4 bq[kPF][MT];
  KPos kpre{uy0, ux0, 0, (static_cast<long long>(uy0) * c.W + ux0) * c.Cin};   // the next B fragment to request
  // FASTK: per-lane source of the chunk being requested (K step kk of it at + 32 kk bytes), and its tap
  unsigned long long bsrc[MT];
  int fuy = uy0, fux = ux0, fcbi = 0;
  auto chunk_sources = [&]() __attribute__((always_inline)) {
    const long long off = (static_cast<long long>(fuy) * c.W + fux) * c.Cin + 16 * fcbi;
#pragma unroll
    for (int p = 0; p < MT; ++p) {
      const bool ok = live[p] & (static_cast<unsigned int>(iy0[p] + fuy) < static_cast<unsigned int>(c.H)) &
                      (static_cast<unsigned int>(ix0[p] + fux) < static_cast<unsigned int>(c.W)) & (fuy < uy1);
      unsigned long long a_in = reinterpret_cast<unsigned long long>(x) +
                                2ull * static_cast<unsigned long long>(base_off[p] + off);
      asm volatile("" : "+v"(a_in));
      bsrc[p] = ok ? a_in : reinterpret_cast<unsigned long long>(zeros);
    }
    // on to the next chunk's position: kChunk2 channel blocks further, by selects
    const int cbn = fcbi + kChunk2;
    const bool wc = cbn == cb;
    const int uxn = fux + (wc ? 1 : 0);
    const bool wx = uxn == ux1;
    fcbi = wc ? 0 : cbn;
    fux = wx ? ux0 : uxn;
    fuy += wx ? 1 : 0;
  };
  auto fast_load = [&](int kk, int p) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<__attribute__((address_space(1))) const u32x4*>(bsrc[p] + 32ull * kk);
  };
  if constexpr (FASTK) {
    chunk_sources();
#pragma unroll
    for (int k = 0; k < kPF; ++k)
#pragma unroll
      for (int p = 0; p < MT; ++p) bq[k][p] = fast_load(k, p);
  } else {
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
#pragma unroll
      for (int p = 0; p < MT; ++p) bq[k][p] = bload(kpre, p);
      advance(kpre);
    }
  }
  __syncthreads();

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
#ifndef TFC_CONV_BARE       // (experiment: MFMAs only -- no weight staging, no barrier, with NOAREAD / NOBLOAD)
    wfetch(chunk + 1);          // past the last chunk: the clamped fragment, not used
#endif
    if constexpr (FASTK) chunk_sources();     // of chunk + 1, whose fragments this chunk's K steps request
    const bf16x8* abase = reinterpret_cast<const bf16x8*>(smem) + buf * CHUNK_FRAGS + lane;
    bf16x8 af[2][TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) af[0][t] = abase[t * 64];
#pragma unroll
    for (int kk = 0; kk < kChunk2; ++kk) {
#ifndef TFC_CONV_NOAREAD    // (experiment: without the A fragment reads of the steps after a chunk's first)
      if (kk + 1 < kChunk2) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) af[(kk + 1) & 1][t] = abase[((kk + 1) * TILES + t) * 64];
      }
#else
#pragma unroll
      for (int t = 0; t < TILES; ++t) af[(kk + 1) & 1][t] = af[kk & 1][t];
#endif
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int p = 0; p < MT; ++p)
          acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              af[kk & 1][t], __builtin_bit_cast(bf16x8, bq[kk % kPF][p]), acc[p][t], 0, 0, 0);
      // refill the ring slot with K step ks + PF (past the last tap row: zeros); after the MFMAs
      // that read it so that no copy of the slot is needed
#ifndef TFC_CONV_BARE
      if constexpr (FASTK) {
#pragma unroll
        for (int p = 0; p < MT; ++p) bq[kk % kPF][p] = fast_load(kk, p);
      } else {
#pragma unroll
        for (int p = 0; p < MT; ++p) bq[kk % kPF][p] = bload(kpre, p);
        advance(kpre);
      }
#endif
#if TFC_CONV_INTERLEAVE
      // One wave per SIMD: an MFMA occupies the matrix pipe for 32 cycles = 8 issue slots, about five other
      // instructions fit in its shadow, and this K step has ~4.5 of them per MFMA (address arithmetic of
      // the two B loads, the six A reads of the next step) -- but only if they sit BETWEEN the MFMAs;
      // left to itself the scheduler issues the 12 MFMAs in a row and everything else behind them.
#pragma unroll
      for (int i = 0; i < TILES * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
        if (i < TILES) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one A fragment read
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                    // two VALU
        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);                    // three SALU
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef TFC_CONV_BARE
    if (chunk + 1 < nchunks) wstore(buf ^ 1);
    __syncthreads();
#endif
  }

  // ---- epilogue: acc[p][t][4q + r] = column group_base + 32t + 8q + 4h + r of pixel mm[p] ----
  const int colbase = group * TILES * 32;
  const bool vec4 = (c.Cout & 3) == 0;       // 4 consecutive columns = 4 channels of one phase
  if ((c.Cout & 7) == 0 && !OUTF32) {
    // 8 consecutive columns = 8 channels of one pixel: lanes l and l + 32 (the same pixel, h = 0 / 1) trade
    // halves of a pair of column groups (v_permlane32_swap, as in gdn_common.h) and each stores 16 bytes,
    // 32 contiguous bytes per pixel and instruction.  With 8-byte stores the output traffic of the layers
    // that write the most (the image-side analysis layer, the last 192 -> 192 synthesis layer) took 2.6 and
    // 3.3 ms of their 4.4 and 14.8 ms at the C4 shapes.
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int col0 = colbase + 32 * t + 16 * qp;          // the pair covers columns col0 .. col0 + 15
        if (col0 >= c.cols) continue;                          // wave-uniform
        const int co = col0 % c.Cout, phase = col0 / c.Cout;
        f32x4 be = f32x4{0.f, 0.f, 0.f, 0.f}, bo = be;
        if (bias) {
          be = *reinterpret_cast<const f32x4*>(bias + co + 4 * h);
          if (col0 + 8 < c.cols) bo = *reinterpret_cast<const f32x4*>(bias + (col0 + 8) % c.Cout + 4 * h);
        }
#pragma unroll
        for (int p = 0; p < MT; ++p) {
          u32x4 o;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const f32x4& b4 = half ? bo : be;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[p][t][4 * (2 * qp + half) + r] + b4[r];
              if (c.activation == 1) v[r] = fmaxf(v[r], 0.f);
            }
            o[2 * half] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
            o[2 * half + 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
          // lane h stores columns col0 + 8h .. + 7 (h = 1: the odd group, possibly another phase / past the end)
          const int colh = col0 + 8 * h;
          if (!live[p] || colh >= c.cols) continue;
          const int coh = colh % c.Cout, ph = colh / c.Cout;
          const int oy = qy[p] * c.su + ph / c.su, ox = qx[p] * c.su + ph % c.su;
          *reinterpret_cast<u32x4*>(y + ((nn[p] * c.OH + oy) * c.OW + ox) * c.Cout + coh) =
              u32x4{s0[0], s1[0], s0[1], s1[1]};
          (void)phase;
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // one column tile at a time (register pressure)
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col0 = colbase + 32 * t + 8 * q + 4 * h;
      const int co = col0 % c.Cout, phase = col0 / c.Cout;
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias && col0 < c.cols) {
        if (vec4) {
          b4 = *reinterpret_cast<const f32x4*>(bias + co);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) b4[r] = col0 + r < c.cols ? bias[(col0 + r) % c.Cout] : 0.f;
        }
      }
#pragma unroll
      for (int p = 0; p < MT; ++p) {
#ifdef TFC_CONV_NOSTORE      // experiment: the kernel without its output traffic (stores only for an impossible bias)
        if (!live[p] || col0 >= c.cols || b4[0] != 12345.f) continue;
#else
        if (!live[p] || col0 >= c.cols) continue;
#endif
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[p][t][4 * q + r] + b4[r];
          if (c.activation == 1) v[r] = fmaxf(v[r], 0.f);
        }
        if (OUTF32 && vec4) {
          const int oy = qy[p] * c.su + phase / c.su, ox = qx[p] * c.su + phase % c.su;
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + ((nn[p] * c.OH + oy) * c.OW + ox) * c.Cout + co) =
              f32x4{v[0], v[1], v[2], v[3]};
        } else if (vec4) {
          // su == 1: phase 0, (oy, ox) = (qy, qx); else depth-to-space of the column's phase
          const int oy = qy[p] * c.su + phase / c.su, ox = qx[p] * c.su + phase % c.su;
          u32x2 o;
          o.x = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          o.y = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          *reinterpret_cast<u32x2*>(y + ((nn[p] * c.OH + oy) * c.OW + ox) * c.Cout + co) = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = col0 + r;
            if (col >= c.cols) continue;
            const int co1 = col % c.Cout, ph1 = col / c.Cout;
            const int oy = qy[p] * c.su + ph1 / c.su, ox = qx[p] * c.su + ph1 % c.su;
            y[((nn[p] * c.OH + oy) * c.OW + ox) * c.Cout + co1] = static_cast<__bf16>(v[r]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // one column tile at a time (register pressure)
  }
}

// ---------------------------------------------------------------------------
// bf16 main kernel, third generation (5x5 / 3x3 layers between wide feature maps: Cin % 16 == 0,
// Cout = 128 or 192 per group).  What the second generation spends beside its MFMAs is the B operand: a
// 16-byte gather per lane, tap and K step straight from the NHWC tensor (every input value requested kh*kw/sd^2
// times), each with its bounds test, address select and tap walk.  Here a workgroup owns an 8 x 32 block of
// low-resolution output pixels of one image and
//   * stages the input PATCH of that block in LDS, one 16-channel block at a time, double-buffered: every input
//     value is requested once per workgroup (+ the halo), bounds are tested once per granule and channel block
//     by the loader, and out-of-image positions are zeros in LDS;
//   * runs K channel block by channel block, the taps inside: a tap is a wave-uniform LDS offset, so a B
//     fragment is one ds_read_b128 at (per-lane base) + (scalar tap offset) — no per-lane arithmetic;
//   * lays the patch out as [row][x parity][h][x / sd] granules of 16 bytes (h = which 8 of the 16 channels):
//     the 32 pixels of a tile (one output row segment, input stride sd) are consecutive granules of one
//     parity / h plane, i.e. all 64 banks once per 16-lane group of the ds_read_b128;
//   * keeps of the second generation: weights as packed A fragments in a double-buffered LDS chunk (here CH K steps =
//     CH taps of one channel block, CH | taps), A / B fragments double-buffered in registers, 2 pixel tiles x TILES
//     column tiles per wave.
// NPG = 16-byte patch pieces per thread (granules / 256, rounded up).
// Round 6 (measured from the inside with the timing build below; DESIGN.md §3, profiles/r06_notes.md):
//   * the patch loader reads a pixel's 32 bytes of the channel block with a lane PAIR (32 lines per request instead of 64);
//   * the weight chunks go global -> LDS by buffer_load ... lds (WDMA), not through registers;
//   * every staging instruction sits behind an MFMA of its own (slots, below);
//   * the output leaves as whole 128-byte lines through a wave-private LDS area, non-temporal for big outputs;
//   * GDN / IGDN as the activation: builds of their own (GDNK), no scratch.
//   * one ITEM per workgroup: a block and one of the launch's column groups (round 6; before: a workgroup took a
//     block's groups, or every W-th block, one after the other with the next item's first requests under the last K
//     steps — per item the same time, tools/conv3_clock_probe.py: what the prologue saved the longer K loop and the
//     bookkeeping between items took, and it kept ~90 registers of next-item state alive through the epilogue);
//   * the K steps of a channel block are unrolled into ONE basic block (template NCH chunks x CH K steps): taps are
//     compile-time indexes into a table of scalar offsets, and the staging of a chunk boundary sits between the MFMAs.
// TFC_CONV3_EXP (build switch, timing experiments only — results are wrong): 1 no barriers, 2 no weight staging,
// 4 no patch staging, 32 no output stores (tools/conv3_variants.sh).
// ---------------------------------------------------------------------------
#ifndef TFC_CONV3_EXP
#define TFC_CONV3_EXP 0
#endif
#if TFC_CONV3_EXP & 64
// timing builds only (tools/conv3_clock_probe.py): per workgroup (the first kConv3ClockWgs of a launch) the 100 MHz
// clock at entry, behind the prologue's barrier, at the end of the K loop, with the stores issued, and with them
// acknowledged; [5] = the CU (XCC, SE, CU id) it ran on
constexpr int kConv3ClockWgs = 16384;
__device__ unsigned long long g_conv3_clocks[kConv3ClockWgs * 8];
// ... and core-clock cycles its first wave waited in the K loop: [ch] for the weight chunk stored at the start of chunk ch of
// a channel block (ch < 5), [5] at the barriers, [6] the K loop, [7] for the LDS reads in front of the MFMAs of a K step
__device__ unsigned long long g_conv3_waits[kConv3ClockWgs * 8];
// (timing build: the cycles a staging instruction takes to ISSUE — the wave issues in order, the matrix pipe has nothing
// to start while it does — summed per kind in kwait[kind]: 0 weight requests, 1 patch requests, 2 / 3 their LDS writes)
#define TFC_CONV3_ISSUE(kind, stmt)                       \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    const long long i0__ = __builtin_readcyclecounter();  \
    stmt;                                                 \
    const long long i1__ = __builtin_readcyclecounter();  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    kwait[kind] += i1__ - i0__;                           \
  } while (0)
#define TFC_CONV3_CLOCK(slot)                                                                              \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < kConv3ClockWgs) g_conv3_clocks[blockIdx.x * 8 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define TFC_CONV3_ISSUE(kind, stmt) stmt
#define TFC_CONV3_CLOCK(slot) do {} while (0)
#endif
// workgroup barrier that orders LDS traffic only (a __syncthreads also waits for the global loads in flight)
#define TFC_LDS_BARRIER()                                                \
  do {                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      \
    __builtin_amdgcn_s_barrier();                                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");      \
  } while (0)
// OUTF32 (round 6): the float32 accumulators + bias leave as float32 (the float32 layers that run as six bfloat16
// planes, conv_split_x_kernel).
template <int TILES, int CH, int NCH, int NPG, int GDNK = 0, bool OUTF32 = false>
__global__ void __launch_bounds__(256, 1) conv3_bf16_kernel(const __bf16* x, const void* packed,
                                                            const float* bias, __bf16* y, ConvGeom c,
                                                            Conv3Geom d) {
  extern __shared__ unsigned char smem[];          // 2 patches of PATCH_BYTES | 2 weight chunks of STAGE * 4 KB
  // GDNK: 0 none, 1 GDN (y / norm), 2 IGDN (y * norm) as the activation.  A build each — ONE straight-line epilogue: with
  // the two as branches of one build the compiler hoisted what they share (every y word split into its two floats, 192
  // registers) in front of the branch, and the epilogue ran out of scratch (12 us per item instead of 5)
  constexpr bool GDN = GDNK != 0;
  // MT = 2 pixel tiles per wave: four waves, one per SIMD.  (Round 6, measured and not kept: MT = 1 with EIGHT waves, two
  // per SIMD at <= 220 registers, so that one wave's MFMAs run while the other issues its staging instructions — the
  // stride-2 5x5 layer took 5.68 ms against 5.64: the second wave of a SIMD does not fill those gaps, profiles/r06_notes.md)
  constexpr int MT = 2;
  constexpr int NTHR = 256;
  constexpr unsigned int TPIECE = NTHR * 16u;                // bytes of one 16-byte piece per thread
  constexpr int CHUNK_FRAGS = CH * TILES * 64;
  constexpr int STAGE = (CHUNK_FRAGS + NTHR - 1) / NTHR;    // 16-byte pieces per thread and weight chunk
  // a patch buffer: the granules + room for the (unread) granules of threads past the patch's last pixel
  constexpr unsigned int PATCH_BYTES = NPG * 4096u + (NPG > 4 ? 2048u : 0u);
  constexpr int NPT = NPG;
  constexpr unsigned int WBUF_BYTES = STAGE * TPIECE;
  static_assert(NPG % 2 == 0, "two pieces per patch pixel");
  // WDMA (round 6): the weight chunks travel global -> LDS without passing the wave's registers (buffer_load ... lds: a
  // wave's piece is 1 KB contiguous on both sides; no ds_write_b128, 37 cycles of issue each, and 32 registers less).
  // Short chunks (3 / 4 K steps): chunk c + 2 is requested in the LAST K step of chunk c — behind the barrier that ended the
  // reads of chunk c's own buffer, which it goes to — and has to have landed at chunk c + 1's barrier: CH K steps.  The
  // 5-K-step chunks: chunk c + 1 in the FIRST K step of chunk c (CH - 1 K steps; measured against the last K step of the chunk
  // before on the stride-2 5x5 layer: 5.11 / 5.18 ms, through registers 5.22; the transposed layer's three launches 5.68
  // against 5.75).  TFC_CONV3_WDMA = 0: through registers (requested a chunk earlier).
#ifndef TFC_CONV3_WDMA
#define TFC_CONV3_WDMA 1
#endif
  constexpr bool WDMA = TFC_CONV3_WDMA != 0;
  constexpr bool WEARLY = CH == 5;            // (see above) request in K step 0 of the chunk before, else K step CH - 1 of two before
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6, h = lane >> 5, l = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int lg = d.lg, sd = 1 << lg, PWh = d.PWh;
  const int cb = c.Cin / 16;
  unsigned char* wl = smem + 2 * PATCH_BYTES;

  // ---- the workgroup's ITEM: (block u / gcount, group glist[u % gcount]) for workgroup u in XCD order (the groups
  // of a block, and neighbouring blocks, meet in one L2) ----
  const unsigned int u = xcd_order(blockIdx.x, gridDim.x, c.xcd);
  TFC_CONV3_CLOCK(0);
#if TFC_CONV3_EXP & 64
  if (threadIdx.x == 0 && blockIdx.x < kConv3ClockWgs) {
    unsigned int hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_conv3_clocks[blockIdx.x * 8 + 5] = (static_cast<unsigned long long>(xcc & 0xF) << 32) | hwid;
  }
#endif

  struct Item {                  // wave-uniform
    long long n;
    int qx0, qy0, group;
    int uy0, uy1, ux0, ux1;
  };
  // (divisions as multiplications, fast_div: a 64-bit division is a ~150-instruction loop in front of the first request)
  auto item_at = [&](unsigned int t) -> Item {
    Item it;
    const unsigned int blk = fast_div(t, d.gc_mul, d.gc_sh);
    it.group = d.glist[t - blk * d.gcount];
    const unsigned int row = fast_div(blk, d.bx_mul, d.bx_sh);          // n * BYn + by
    const unsigned int img = fast_div(row, d.by_mul, d.by_sh);
    it.qx0 = static_cast<int>(blk - row * d.BXn) * 32;
    it.qy0 = static_cast<int>(row - img * d.BYn) * 8;
    it.n = img;
    it.uy0 = c.ty0[it.group]; it.uy1 = c.ty1[it.group]; it.ux0 = c.tx0[it.group]; it.ux1 = c.tx1[it.group];
    return it;
  };

  // ---- patch loader.  Piece j of a thread = half tid & 1 (8 of the 16 channels) of patch pixel j * 128 + tid / 2: a
  // lane PAIR reads the 32 contiguous bytes a pixel has of the channel block, an instruction 32 pixels.  (Round 6;
  // before, a thread read both halves of pixel j * 256 + tid with two instructions of 64 pixels each: the CU's address
  // unit takes ~4 cycles per 128-byte line an instruction touches — 250 cycles measured for one of those, 13 for a
  // request of 1 KB contiguous, tools/conv3_clock_probe.py — and with 64 lines each the patch requests of the four waves
  // kept it busy 390 cycles of a K step's 650: half the lines per instruction, the same number of instructions.)
  // The granule a piece goes to is the same for every item, the place it comes from (poff) is per item.  Buffer loads:
  // a pixel outside the image has an offset outside the image's buffer and reads as zeros ----
  unsigned int pdst[NPT];                 // granule of (pixel, h = tid & 1)
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int q = j * (NTHR / 2) + (tid >> 1);
    const int py = fast_div(q, d.pw_mul, d.pw_sh), px = q - py * d.PW;
    // (pixels past the patch: a granule behind it, inside the padded buffer, that nobody reads)
    pdst[j] = q < d.pixels ? static_cast<unsigned int>(((((py << lg) + (px & (sd - 1))) * 2 + (tid & 1)) * PWh + (px >> lg)) * 16)
                           : PATCH_BYTES - 16u * PWh - 16u;
  }
  auto patch_offsets = [&](int qx0, int qy0, unsigned int* poff) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int q = j * (NTHR / 2) + (tid >> 1);
      const int py = fast_div(q, d.pw_mul, d.pw_sh), px = q - py * d.PW;
      const int iy = qy0 * sd - c.py0 + py, ix = qx0 * sd - c.px0 + px;
      const bool ok = (q < d.pixels) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(c.H)) &
                      (static_cast<unsigned int>(ix) < static_cast<unsigned int>(c.W));
      poff[j] = ok ? static_cast<unsigned int>((iy * c.W + ix) * c.Cin * 2 + 16 * (tid & 1)) : 0x80000000u;
    }
  };
  auto image_rsrc = [&](long long n) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + n * c.H * c.W * c.Cin), 0,
                                             c.H * c.W * c.Cin * 2, 0x00020000);
  };
  u32x4 pst[NPT];
  auto pfetch = [&](__amdgpu_buffer_rsrc_t xr, const unsigned int* poff, int cbi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) pst[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, poff[j], cbi * 32, 0);
  };
  auto pstore = [&](int buf) __attribute__((always_inline)) {
    unsigned char* dst = smem + buf * PATCH_BYTES;
#pragma unroll
    for (int j = 0; j < NPT; ++j) *reinterpret_cast<u32x4*>(dst + pdst[j]) = pst[j];
  };

  // ---- weights: the packed A fragments of a group, chunk by chunk; the request runs two chunks ahead of the K loop
  // (chunks past the group's end are outside the buffer: zeros, no traffic) ----
  auto weight_rsrc = [&](const Item& it) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(static_cast<const unsigned char*>(packed)) +
            static_cast<size_t>(it.group) * c.ksteps * TILES * 64 * 16,
        0, NCH * cb * (CHUNK_FRAGS * 16), 0x00020000);
  };
  u32x4 stage[STAGE];
  auto wfetch = [&](__amdgpu_buffer_rsrc_t r, int chunk) __attribute__((always_inline)) {
    const unsigned int v0 = static_cast<unsigned int>(chunk) * (CHUNK_FRAGS * 16u) + tid * 16u;
#pragma unroll
    for (int i = 0; i < STAGE; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(r, v0 + i * TPIECE, 0, 0);
  };
  auto wstore = [&](int buf) __attribute__((always_inline)) {
    u32x4* dst = reinterpret_cast<u32x4*>(wl + buf * WBUF_BYTES) + tid;
#pragma unroll
    for (int i = 0; i < STAGE; ++i) dst[i * NTHR] = stage[i];
  };

  // ---- B: per-lane granule of (tile p, tap (0, 0)); a tap adds a wave-uniform offset ----
  unsigned int lb[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    const int row = MT * wid + p;
    lb[p] = static_cast<unsigned int>(((row * sd * sd * 2 + h) * PWh + l) * 16);
  }
  auto tap_offset = [&](int uy, int ux) -> unsigned int {
    return static_cast<unsigned int>(((((uy << lg) + (ux & (sd - 1))) * 2) * PWh + (ux >> lg)) * 16);
  };

  f32x16 acc[MT][TILES];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < MT; ++p)
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;
  };
  zero_acc();

  // ---- epilogue of an item: acc[p][t][4q + r] = column group_base + 32t + 8q + 4h + r of pixel
  // (qy0 + MT wid + p, qx0 + l); 16-byte stores as in the second generation (Cout % 8 == 0) ----
  // ---- GDN / IGDN as the activation (GDN): the block's accumulators are all TILES * 32 channels of its pixels, in
  // the register layout the GDN kernel's B fragments have (gdn_common.h: K step s of a lane = channels 16 s + 4 h +
  // {0..3} and + 8), so |y| goes into the gamma contraction straight from the accumulators: 4 TILES^2 MFMAs per wave
  // against the K loop's thousands.  gamma's A fragments (the prepared image, 72 KB at 192 channels) are staged in LDS
  // over the two weight buffers — from L2 a K step of them is ~0.7 us away and 0.1 us of MFMAs (measured: +50 % on the
  // layer).  y is rounded to bfloat16 first: the same values the
  // unfused pair (convolution, then the GDN kernel on its output) works on, contracted in the same order ----
  constexpr int GDN_IMAGE_BYTES = TILES * 2 * TILES * 64 * 16 + TILES * 32 * 4;                   // fragments, beta
  constexpr int GDN_PIECES = (GDN_IMAGE_BYTES + static_cast<int>(TPIECE) - 1) / static_cast<int>(TPIECE);   // 16-byte pieces per thread
  // GRES: the transposed layers' builds (small patch, short weight chunks) have the LDS to keep the image for the whole
  // item — copied by the prologue beside the first patch and weight chunk, no barriers or copy in the stage: their items
  // are 4-9 taps long (17-40 us of K loop), the copy + its two barriers were ~3 us of each
  constexpr bool GRES = GDN && NPG == 4;
  const __amdgpu_buffer_rsrc_t gimage_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(GDN ? c.gdn_image : nullptr), 0, GDN ? GDN_IMAGE_BYTES : 0, 0x00020000);
  // y = convolution + bias as the bfloat16 tensor would hold it, packed (K step s of the contraction is xb[p][s] with the
  // sign bits cleared); the epilogue multiplies it with the norm's power as it packs the output — the 192 results of a
  // lane never exist at once (as a loop of their own they sat in registers beside y: 50 of them went to scratch, and
  // the reloads, each waited for, were 13 us of every item, tools/conv3_clock_probe.py)
  unsigned int xb[GDN ? MT : 1][GDN ? 2 * TILES : 1][4];      // (single words: four-register tuples of them made the allocator spill)
  auto gdn_stage = [&]() __attribute__((always_inline)) {
    constexpr int KT = TILES, KS = 2 * TILES;
    unsigned char* const gl = smem + d.oimage;
    const float* const bias_s = reinterpret_cast<const float*>(smem + d.obias);      // (zeros without a bias)
    constexpr int R0 = (GDN_PIECES + 1) / 2;
    u32x4 g1[GRES ? 1 : GDN_PIECES - R0];
    if constexpr (!GRES) {
      // the image -> LDS over the weight buffers, in two rounds of requests (registers); only the image's bytes are
      // written (the bias sits behind it)
      TFC_LDS_BARRIER();                  // every wave is through with the weight buffers
      {
        u32x4 g0[R0];
#pragma unroll
        for (int i = 0; i < R0; ++i) g0[i] = __builtin_amdgcn_raw_buffer_load_b128(gimage_rsrc, tid * 16u, i * TPIECE, 0);
#pragma unroll
        for (int i = 0; i < R0; ++i) *reinterpret_cast<u32x4*>(gl + i * TPIECE + tid * 16) = g0[i];
      }
#pragma unroll
      for (int i = R0; i < GDN_PIECES; ++i) g1[i - R0] = __builtin_amdgcn_raw_buffer_load_b128(gimage_rsrc, tid * 16u, i * TPIECE, 0);
    }
    // (under the second round of the image) y, packed; the accumulators are then free to take the norm
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int t0 = s >> 1, q0 = 2 * (s & 1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + 32 * t0 + 8 * (q0 + half) + 4 * h);
        const int e = 4 * (q0 + half);
#pragma unroll
        for (int p = 0; p < MT; ++p) {
          xb[p][s][2 * half] = __builtin_bit_cast(unsigned int, __builtin_convertvector(
                                   f32x2{acc[p][t0][e] + b4[0], acc[p][t0][e + 1] + b4[1]}, bf16x2));
          xb[p][s][2 * half + 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(
                                       f32x2{acc[p][t0][e + 2] + b4[2], acc[p][t0][e + 3] + b4[3]}, bf16x2));
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // (16 channels at a time: registers)
    }
    if constexpr (!GRES) {
#pragma unroll
      for (int i = R0; i < GDN_PIECES; ++i)
        if (i * TPIECE + tid * 16 < GDN_IMAGE_BYTES) *reinterpret_cast<u32x4*>(gl + i * TPIECE + tid * 16) = g1[i - R0];
    }
    zero_acc();
    if constexpr (!GRES) TFC_LDS_BARRIER();
    TFC_CONV3_CLOCK(6);
    const bf16x8* const afr = reinterpret_cast<const bf16x8*>(gl) + lane;
    // gamma's fragments in the order they are used, (s, t) = (f / KT, f % KT), through a ring of four (three reads ahead:
    // a fragment serves two MFMAs, 64 cycles, an LDS read is ~150 away) — a whole K step of them ahead took 48 registers
    bf16x8 ring[4];
    auto frag_at = [&](int f) -> bf16x8 { return afr[((f % KT) * KS + f / KT) * 64]; };
#pragma unroll
    for (int f = 0; f < 3; ++f) ring[f] = frag_at(f);
    bf16x8 bfrag[MT];
#pragma unroll
    for (int f = 0; f < KS * KT; ++f) {
      const int s = f / KT, t = f % KT;
      if (f + 3 < KS * KT) ring[(f + 3) & 3] = frag_at(f + 3);
      if (t == 0) {
#pragma unroll
        for (int p = 0; p < MT; ++p)
          bfrag[p] = __builtin_bit_cast(bf16x8, u32x4{xb[p][s][0] & 0x7FFF7FFFu, xb[p][s][1] & 0x7FFF7FFFu,
                                                       xb[p][s][2] & 0x7FFF7FFFu, xb[p][s][3] & 0x7FFF7FFFu});
      }
#pragma unroll
      for (int p = 0; p < MT; ++p)
        acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f & 3], bfrag[p], acc[p][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    TFC_CONV3_CLOCK(7);
  };
  auto epilogue = [&](const Item& it, const int pb_last) __attribute__((always_inline)) {
    if constexpr (GDN) gdn_stage();
    // Output through LDS, a wave for itself.  The accumulators of a lane are 4 (+ 4 of its partner half) consecutive
    // channels of ONE pixel: stored straight from them, an instruction is 32 pixels x 32 bytes — 32 partial lines, ~50
    // cycles each in the CU's store path, 9-10 us per item (tools/conv3_clock_probe.py).  Instead one 128-byte line per
    // pixel at a time (bfloat16: two column tiles, float32: one) goes to the wave's 8 KB of the staging area as
    // [pixel][8 granules], granule g at g ^ (pixel / 2 & 7) (16 lanes of a ds_write_b128 / ds_read_b128: all banks
    // once), and leaves as 8 pixels x 128 bytes per instruction: whole lines.  The bias comes from LDS (parked there by
    // the prologue, zeros without one): a global load here would wait, in vmcnt order, for every store issued before
    // it.  Straight-line code: pixels outside the map get a buffer offset outside the image (the store is dropped).
    static_assert(TILES % 2 == 0, "two column tiles per output line");
    static_assert(!(GDN && OUTF32), "float32 output: no fused GDN");
    // (d.ostage < 0: no room of its own — the patch buffer of the LAST channel block instead: every read of it was
    // complete at the K loop's last barrier, the loop's last K step reads ahead into the other buffers only)
    // (d.ostage == -2, builds with the GDN image resident: the weight buffers — what the last K step reads ahead from
    // them is never used)
    unsigned char* const ost = smem + (d.ostage >= 0 ? d.ostage : d.ostage == -2 ? static_cast<int>(2 * PATCH_BYTES)
                                                                                 : pb_last * static_cast<int>(PATCH_BYTES)) + wid * (MT * 4096);
    // what is added to an accumulator: the bias; with GDN the norm's beta (behind gamma's fragments in the image)
    const float* const bias_s = reinterpret_cast<const float*>(smem + (GDN ? d.oimage + TILES * 2 * TILES * 1024 : d.obias));
    const int phy = it.group / c.su, phx = it.group % c.su;       // a group = one output phase, all its Cout channels
    constexpr int ROUNDS = OUTF32 ? TILES : TILES / 2;
    constexpr int ESZ = OUTF32 ? 4 : 2;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(y) + it.n * c.OH * c.OW * c.Cout * ESZ, 0, c.OH * c.OW * c.Cout * ESZ, 0x00020000);
    auto wave_sync = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto slot = [&](int pix, int g) -> unsigned char* { return ost + pix * 128 + ((g ^ ((pix >> 1) & 7)) << 4); };
    // where this lane's 8 read-back granules go: granule j = pixel 8 j + lane / 8 = row p = j / 4 of the wave's two,
    // column 8 (j & 3) + lane / 8 — 16 bytes at 16 (lane & 7) of its line.  Per row a lane offset (outside the image
    // for a row outside the map), the column step and the line of the round as the scalar offset
    unsigned int yrow[MT];
    const int qxl = it.qx0 + (lane >> 3);
#pragma unroll
    for (int p = 0; p < MT; ++p) {
      const int qy = it.qy0 + MT * wid + p;
      yrow[p] = qy < c.OHq ? static_cast<unsigned int>(((qy * c.su + phy) * c.OW + qxl * c.su + phx) * c.Cout * ESZ + 16 * (lane & 7))
                           : 0x80000000u;
    }
    const int xroom = c.OWq - qxl;                       // column 8 k of the lane is inside the map iff 8 k < xroom
    const unsigned int xstep = static_cast<unsigned int>(8 * c.su * c.Cout * ESZ);
    auto rounds = [&](auto relu_tag, auto inv_tag) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_tag)::value, INV = decltype(inv_tag)::value;
      auto bias4 = [&](int ch) -> f32x4 { return *reinterpret_cast<const f32x4*>(bias_s + ch + 4 * h); };
      // output value of accumulator element 4 q + r of (p, t): + bias; with GDN y / norm (IGDN: y * norm), norm = the
      // accumulator + beta, y = the packed word of the same channel
      auto elem = [&](int p, int t, int q, int r, const f32x4& b4) -> float {
        float v = acc[p][t][4 * q + r] + b4[r];
        if constexpr (GDN) {
          const unsigned int word = xb[p][2 * t + (q >> 1)][2 * (q & 1) + (r >> 1)];
          const float yv = __uint_as_float((r & 1) ? (word & 0xFFFF0000u) : (word << 16));
          v = yv * (INV ? v : __builtin_amdgcn_rcpf(v));
        }
        if (RELU) v = fmaxf(v, 0.f);
        return v;
      };
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        if constexpr (OUTF32) {
          const int t = rd;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = bias4(32 * t + 8 * q);
#pragma unroll
            for (int p = 0; p < MT; ++p) {
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = elem(p, t, q, r, b4);
              *reinterpret_cast<f32x4*>(slot(32 * p + l, 2 * q + h)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);        // 8 channels at a time (registers)
          }
        } else {
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int t = 2 * rd + t2;
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const f32x4 be = bias4(32 * t + 16 * qp), bo = bias4(32 * t + 16 * qp + 8);
#pragma unroll
              for (int p = 0; p < MT; ++p) {
                u32x4 o;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                  const f32x4& b4 = half ? bo : be;      // channels 32 t + 16 qp + 8 half + 4 h + {0 .. 3}
                  float v[4];
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] = elem(p, t, 2 * qp + half, r, b4);
                  o[2 * half] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
                  o[2 * half + 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
                }
                // the halves trade their inner words: this lane then holds channels 32 t + 16 qp + 8 h + {0 .. 7}
                const auto s0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
                *reinterpret_cast<u32x4*>(slot(32 * p + l, 4 * t2 + 2 * qp + h)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
              }
              __builtin_amdgcn_sched_barrier(0);      // 16 channels at a time (registers)
            }
          }
        }
        wave_sync();
#pragma unroll
        for (int p = 0; p < MT; ++p) {          // a row's four granules at a time (registers)
          u32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4*>(slot(32 * p + 8 * k + (lane >> 3), lane & 7));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#if TFC_CONV3_EXP & 32
            if (v[k].x != 0x12345u) continue;     // (no stores)
#endif
            // (non-temporal — cache policy 2 — where the output cannot stay in the caches: the lines do not displace the
            // patches and weights other workgroups are about to read)
            if (c.nt_out) __builtin_amdgcn_raw_buffer_store_b128(v[k], yr, 8 * k < xroom ? yrow[p] : 0x80000000u, 128 * rd + k * xstep, 2);
            else __builtin_amdgcn_raw_buffer_store_b128(v[k], yr, 8 * k < xroom ? yrow[p] : 0x80000000u, 128 * rd + k * xstep, 0);
          }
        }
        wave_sync();
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if constexpr (GDN) {
      rounds(F{}, std::integral_constant<bool, GDNK == 2>{});       // (no other activation beside the GDN)
    } else {
      if (c.activation == 1) rounds(T{}, F{}); else rounds(F{}, F{});
    }
  };

  // Schedule of a chunk c (CH K steps, weights in LDS buffer c & 1; the slots of every kind of staging: at channel_block):
  //   the next chunks' weights   WDMA: chunk c + 1 requested into buffer (c + 1) & 1 in chunk c's first K step (5-K-step
  //                chunks) or chunk c + 2 into buffer c & 1 in its last (shorter chunks) — either way into a buffer whose
  //                last readers finished before a barrier, and waited for (vmcnt) in front of the barrier that publishes it.
  //                Through registers (TFC_CONV3_WDMA = 0): requested a chunk earlier, registers -> LDS in the first K step
  //   first chunk of a channel block: the next channel block's patch requested
  //   K step kk    reads the fragments of K step kk + 1 under its MFMAs
  //   end of kk = CH - 2   (last chunk of a channel block: the patch registers -> the other patch buffer;) wait for
  //                this wave's LDS traffic, barrier
  //   kk = CH - 1  the fragments it reads ahead are those of chunk c + 1's first K step: from the buffers published
  //                before the barrier, so no K step ever waits for a read it has just issued behind a barrier
  // (no __syncthreads: its fence would also wait for the global prefetches in flight).
  constexpr int NT = CH * NCH;            // taps of a group = K steps of a channel block
  // (the first requests go out before the rest of the bookkeeping: it runs under their latency)
  const Item cur = item_at(u);
  unsigned int poff[NPT];
  const __amdgpu_buffer_rsrc_t wr = weight_rsrc(cur);
  wfetch(wr, 0);
  patch_offsets(cur.qx0, cur.qy0, poff);
  const __amdgpu_buffer_rsrc_t xr = image_rsrc(cur.n);
  pfetch(xr, poff, 0);
  // the taps' patch offsets (wave-uniform, one SGPR each: the K steps below are unrolled over a whole channel block)
  unsigned int toff[NT];
  auto tap_table = [&](const Item& it) __attribute__((always_inline)) {
    int uy = it.uy0, ux = it.ux0;           // tap k = (uy0 + k / wx, ux0 + k % wx), walked
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      toff[k] = tap_offset(uy, ux);
      if (++ux == it.ux1) { ux = it.ux0; ++uy; }
    }
  };
  tap_table(cur);
  if (tid < TILES * 8)           // the bias, zeros without one (the epilogue's; with GDN: added before its contraction)
    *reinterpret_cast<f32x4*>(smem + d.obias + tid * 16) =
        bias ? *reinterpret_cast<const f32x4*>(bias + tid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (GRES) {          // gamma's fragment image, for the whole item
    u32x4 gi[GDN_PIECES];
#pragma unroll
    for (int i = 0; i < GDN_PIECES; ++i) gi[i] = __builtin_amdgcn_raw_buffer_load_b128(gimage_rsrc, tid * 16u, i * TPIECE, 0);
#pragma unroll
    for (int i = 0; i < GDN_PIECES; ++i)
      if (i * TPIECE + tid * 16 < GDN_IMAGE_BYTES) *reinterpret_cast<u32x4*>(smem + d.oimage + i * TPIECE + tid * 16) = gi[i];
  }
  pstore(0);
  wstore(0);
  if constexpr (WDMA && WEARLY) {
    // (chunk 1: requested by chunk 0's first K step)
  } else if constexpr (WDMA) {   // chunk 1 -> the second buffer, on its way while chunk 0 is worked on
#pragma unroll
    for (int pc = 0; pc < STAGE; ++pc)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wr, (__attribute__((address_space(3))) void*)(wl + WBUF_BYTES + wave_u * 1024 + pc * TPIECE), 16,
          CHUNK_FRAGS * 16u + tid * 16u + pc * TPIECE, 0, 0, 0);
  } else {
    wfetch(wr, 1);
  }
  TFC_LDS_BARRIER();
  TFC_CONV3_CLOCK(1);

  bf16x8 af[2][TILES];
  u32x4 bq[2][MT];
#pragma unroll
  for (int t = 0; t < TILES; ++t) af[0][t] = (reinterpret_cast<const bf16x8*>(wl) + lane)[t * 64];
#pragma unroll
  for (int p = 0; p < MT; ++p) bq[0][p] = *reinterpret_cast<const u32x4*>(smem + lb[p] + toff[0]);
  int gchunk = 0;                         // chunks so far: weight buffer gchunk & 1
  int pcb = 0;                            // channel blocks so far: patch buffer pcb & 1

  // ONE CHANNEL BLOCK: NT K steps in NCH weight chunks, fully unrolled — one basic block, so that the staging of a
  // chunk boundary (registers -> LDS, the next requests) and of the patch sits between the MFMAs of the K steps
  // around it, and a tap is a compile-time index into the offset table.  Schedule of chunk c (weights in LDS buffer
  // c & 1):
  //   first K step   chunk c + 1 (requested during chunk c - 1) registers -> LDS buffer (c + 1) & 1, whose last readers
  //                  finished before the barrier of chunk c - 1; request chunk c + 2; first chunk of the block: request
  //                  the next channel block's patch (the next ITEM's first, behind an item's last)
  //   K step kk      reads the fragments of K step kk + 1 under its MFMAs
  //   end of kk = CH - 2   (last chunk: the patch registers -> the other patch buffer;) LDS barrier
  //   kk = CH - 1    the fragments it reads ahead are those of chunk c + 1's first K step, from the buffers published
  //                  before the barrier: no K step waits for a read it has just issued behind a barrier
  // Fragment register sets alternate per K step; a block of an odd number of K steps ends with a copy so that every
  // block starts from set 0.
#if TFC_CONV3_EXP & 64
  long long kwait[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // Where the staging sits inside a chunk (round 6).  A K step is TILES * MT SLOTS — one MFMA each (32 cycles of the
  // matrix pipe), the order pinned slot by slot (sched_barrier) — and every staging instruction of the wave gets a slot of
  // its own: as bursts in front of a chunk's first MFMA (eight 13-cycle ds_write_b128 with a vmcnt wait each, then 8 + NPG
  // buffer loads; the compiler's schedule, sched_group_barrier masks or not) the matrix pipe stood still for them — one
  // wave per SIMD, nobody else to issue — and the builds without them (TFC_CONV3_EXP 2 / 4) were 12 % / 18 % faster.
  //   every K step     slots 0 .. MT - 1: the next K step's B fragments, MT .. MT + TILES - 1: its A fragments
  //   K step 0         slots 0 .. STAGE - 1: weight chunk c + 1, registers -> LDS
  //   K step 1         slots 0 .. STAGE - 1: weight chunk c + 2 requested
  //   K step PFK of a block's first chunk: the next channel block's patch requested (none behind the last block: a
  //                    descriptor of no records — no traffic, zeros — instead of a branch in the block)
  //   K step CH - 2 of its last chunk: the patch registers -> the other patch buffer; then the LDS barrier
  constexpr int SLOTS = TILES * MT;
  constexpr int PFK = NCH == 1 ? 0 : (CH > 2 ? 2 : CH - 1);           // (one chunk per block: as early as possible)
  constexpr int PF0 = NCH == 1 ? STAGE : 0;                           // its first slot
  constexpr int PS0 = (CH - 2 == 0 || CH - 2 == 1) ? STAGE : 0;       // the patch store's first slot (behind the weights' of that step)
  static_assert(CH >= 3, "K steps 0, 1 and CH - 2 of a chunk carry its staging");
  // (piece pc of a kind whose first slot is F sits in slot (F + pc) % SLOTS: the 128-channel builds have 8 slots for 10
  // patch pieces)
  auto channel_block = [&](const int cbi, const int chunk0, const int wpar, const int pb)
                           __attribute__((always_inline)) {
    const bool more = cbi + 1 < cb;
    const unsigned char* pbase = smem + pb * PATCH_BYTES;
    const unsigned char* pnext = smem + (pb ^ 1) * PATCH_BYTES;
    unsigned char* const pdstb = smem + (pb ^ 1) * PATCH_BYTES;
    const __amdgpu_buffer_rsrc_t xrn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(x + cur.n * c.H * c.W * c.Cin), 0, more ? c.H * c.W * c.Cin * 2 : 0, 0x00020000);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int buf = wpar ^ (ch & 1);
      const bf16x8* abase = reinterpret_cast<const bf16x8*>(wl + buf * WBUF_BYTES) + lane;
      const bf16x8* anext = reinterpret_cast<const bf16x8*>(wl + (buf ^ 1) * WBUF_BYTES) + lane;
      u32x4* const wdst = reinterpret_cast<u32x4*>(wl + (buf ^ 1) * WBUF_BYTES) + tid;
      unsigned char* const wdma = wl + (WEARLY ? buf ^ 1 : buf) * WBUF_BYTES + wave_u * 1024;      // this wave's KB of a piece (+ lane * 16: the hardware)
      const unsigned int wv0 = static_cast<unsigned int>(chunk0 + ch + 2) * (CHUNK_FRAGS * 16u) + tid * 16u;   // (past the last chunk: outside the buffer)
#if TFC_CONV3_EXP & 64
      {   // (timing build: the wait the first store below begins with, by hand)
        const long long w0 = clock64();
        if (NCH > 1 && ch == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");      // (behind it: the patch gather)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        kwait[4] += clock64() - w0;
      }
#endif
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int k = ch * CH + kk;
        const int cur_set = k & 1, nxt_set = cur_set ^ 1;
        const bool last_kk = kk + 1 == CH, block_end = ch + 1 == NCH;
        // the next K step's tap (the item's last step reads ahead for nothing)
        const unsigned int tn = last_kk ? toff[!block_end && k + 1 < NT ? k + 1 : 0] : toff[k + 1 < NT ? k + 1 : 0];
        const unsigned char* const bsrc = last_kk && block_end ? pnext : pbase;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
          const int t = i / MT, p = i % MT;
          acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              af[cur_set][t], __builtin_bit_cast(bf16x8, bq[cur_set][p]), acc[p][t], 0, 0, 0);
#pragma unroll
          for (int f = 0; f < MT + TILES; ++f) {          // fragment f of the next K step: B first, in slot f % SLOTS
            if (f % SLOTS != i) continue;
            if (f < MT) bq[nxt_set][f] = *reinterpret_cast<const u32x4*>(bsrc + lb[f] + tn);
            else af[nxt_set][f - MT] = last_kk ? anext[(f - MT) * 64] : abase[((kk + 1) * TILES + (f - MT)) * 64];
          }
#if !(TFC_CONV3_EXP & 2)
          if constexpr (WDMA) {
            // (a request past the item's last chunk is outside the weights' buffer: no traffic, zeros written; the one the
            // last K step makes is waited for behind the loop.  Under a condition instead, the branch cost 4 % of the layer)
            if (kk == (WEARLY ? 0 : CH - 1)) {
#pragma unroll
              for (int pc = 0; pc < STAGE; ++pc)
                if (pc % SLOTS == i)
                  TFC_CONV3_ISSUE(0, __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                         wr, (__attribute__((address_space(3))) void*)(wdma + pc * TPIECE), 16,
                                         wv0 - (WEARLY ? CHUNK_FRAGS * 16u : 0u) + pc * TPIECE, 0, 0, 0));
            }
          } else {
            if (kk == 0) {
#pragma unroll
              for (int pc = 0; pc < STAGE; ++pc)
                if (pc % SLOTS == i) TFC_CONV3_ISSUE(2, wdst[pc * NTHR] = stage[pc]);
            }
            if (kk == 1) {
#pragma unroll
              for (int pc = 0; pc < STAGE; ++pc)
                if (pc % SLOTS == i) TFC_CONV3_ISSUE(0, stage[pc] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv0 + pc * TPIECE, 0, 0));
            }
          }
#endif
#if !(TFC_CONV3_EXP & 4)
          if (ch == 0 && kk == PFK) {
#pragma unroll
            for (int pc = 0; pc < NPT; ++pc)
              if ((PF0 + pc) % SLOTS == i)
                TFC_CONV3_ISSUE(1, pst[pc] = __builtin_amdgcn_raw_buffer_load_b128(xrn, poff[pc], (cbi + 1) * 32, 0));
          }
          // requested in the block's first chunk, stored in its last: the gather has the whole block to arrive
          if (block_end && kk == CH - 2) {
#pragma unroll
            for (int pc = 0; pc < NPT; ++pc)
              if ((PS0 + pc) % SLOTS == i)
                TFC_CONV3_ISSUE(3, *reinterpret_cast<u32x4*>(pdstb + pdst[pc]) = pst[pc]);
          }
#endif
#if TFC_CONV3_EXP & 64
          if (i == SLOTS - 1) TFC_CONV3_ISSUE(7, (void)0);        // (calibration: the pair of clock reads by itself)
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kk == CH - 2) {
          if constexpr (WDMA) {       // this wave's pieces of the next chunk have landed (behind them in the queue: a patch gather
                                      // requested in this chunk)
            if (ch == 0 && PFK <= CH - 2 && (!WEARLY || PFK > 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
#if !(TFC_CONV3_EXP & 1)
#if TFC_CONV3_EXP & 64
          const long long b0 = clock64();
          TFC_LDS_BARRIER();
          kwait[5] += clock64() - b0;
#else
          TFC_LDS_BARRIER();
#endif
#endif
        }
      }
    }
    if constexpr (NT & 1) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) af[0][t] = af[1][t];
#pragma unroll
      for (int p = 0; p < MT; ++p) bq[0][p] = bq[1][p];
    }
  };
#if TFC_CONV3_EXP & 64
  const long long k0 = clock64();
#endif
  for (int cbi = 0; cbi < cb; ++cbi) {
    channel_block(cbi, cbi * NCH, gchunk & 1, pcb & 1);
    gchunk += NCH;
    ++pcb;
  }
#if TFC_CONV3_EXP & 64
  kwait[6] = clock64() - k0;
  if (threadIdx.x == 0 && blockIdx.x < kConv3ClockWgs)
    for (int i = 0; i < 8; ++i) g_conv3_waits[blockIdx.x * 8 + i] = kwait[i];
#endif
  if constexpr (WDMA && !WEARLY) {
    // the last K step's request (zeros for a chunk past the item's last) has landed before the epilogue takes LDS over —
    // with gamma's image resident its staging area IS the weight buffers, and another wave's piece may lie in this wave's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (GRES) TFC_LDS_BARRIER();
  }
  TFC_CONV3_CLOCK(2);
  epilogue(cur, (pcb - 1) & 1);
#if TFC_CONV3_EXP & 64
  TFC_CONV3_CLOCK(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFC_CONV3_CLOCK(4);
#endif
}

// Host side of the third-generation kernel: 0 = launched, -1 = not this shape (the caller goes on with the second
// generation), > 0 = error.
// TFC_CONV_GEN: 2 the second generation everywhere; 3 (default) the third on the transposed 5x5 layers of wide maps;
// 4 the third wherever it is built.  Measured on C4 (profiles/r03_notes.md): a step's convolutions take 31.0 instead of
// 34.0 ms alone on the chip, and with 8 steps in flight the step 40.8 instead of 43.0 ms (before the coder's workgroups
// were packed four waves to a CU it was the other way round, 48.8 against 47.0: the third generation's workgroups hold
// 152 KB of LDS and found even fewer CUs free of coder waves).  Read per call: tests compare the generations in one
// process.
#if TFC_CONV3_EXP & 64
extern "C" int tfc_debug_conv3_clocks(unsigned long long* out, int wgs) {
  if (wgs > kConv3ClockWgs) wgs = kConv3ClockWgs;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv3_clocks), sizeof(unsigned long long) * 8 * wgs) != hipSuccess) return -1;
  return wgs;
}
extern "C" int tfc_debug_conv3_waits(unsigned long long* out, int wgs) {
  if (wgs > kConv3ClockWgs) wgs = kConv3ClockWgs;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv3_waits), sizeof(unsigned long long) * 8 * wgs) != hipSuccess) return -1;
  return wgs;
}
#endif

int conv3_gen() {
  const char* e = std::getenv("TFC_CONV_GEN");
  return e ? std::atoi(e) : 3;
}

int run_conv3(const __bf16* x, const float* w, const float* bias, __bf16* y, ConvGeom c, PackGeom g,
              hipStream_t st) {
  if (conv3_gen() < 3 || c.small_cin || (c.out_f32 && c.gdn) || c.Cin % 16 || (c.sd != 1 && c.sd != 2)) return -1;
  if (c.Cout != 128 && c.Cout != 192) return -1;
  // Where it is used (measured, tools/conv3_check.py and profiles/r03_notes.md): the transposed 5x5 layers, where it is
  // 18-25 % ahead of the second generation.  On the stride-2 analysis layers it is level with it (TFC_CONV_GEN=4 runs
  // it there): their patch is 4x the pixels per block and its 32-byte gathers re-fetch every 128-byte line of the input
  // once per channel block.  The rule looks at the map only, never at the batch: an image's result must not depend on
  // the batch it is coded in.
  // (small stride-2 maps — bls2017's 64x64 -> 32x32 at 512 images: 1.37 -> 1.26 ms — do not have that problem)
  const bool small_down = !g.up && c.sd == 2 && c.OWq * c.OHq <= 4096;
  // Since its workgroups take their blocks in XCD order (xcd_order: neighbouring blocks' patches meet in one L2) it is
  // ahead on the wide stride-2 maps too: 6.87-6.98 -> 6.32-6.37 ms at 384x256, 1.73-1.76 -> 1.57-1.61 at 192x128 (same
  // box, alternating; TFC_CONV_DOWN3=0 keeps those on the second generation).
  static const bool down3 = [] { const char* e = std::getenv("TFC_CONV_DOWN3"); return !(e && e[0] == '0'); }();
  const bool wide_down = down3 && !g.up && c.sd == 2 && c.xcd;
  if (conv3_gen() < 4 && !(g.up && c.su == 2) && !small_down && !wide_down) return -1;
  {
    const int bxn = (c.OWq + 31) / 32, byn = (c.OHq + 7) / 8;
    if (static_cast<double>(c.OWq) * c.OHq < 0.85 * (bxn * 32.0 * byn * 8.0)) return -1;   // blocks mostly outside the map
    if (bxn * byn < 4 && conv3_gen() < 4) return -1;                                       // a block or two per image
  }
  c.tiles = c.Cout / 32;
  c.groups = c.su * c.su;                              // one output phase per group
  if (c.groups > kMaxGroups) return -1;
  c.compact = 1;
  c.cbmajor = 1;
  const int cb = c.Cin / 16;
  int most = 0;
  for (int grp = 0; grp < c.groups; ++grp) {
    int y0 = 0, y1 = c.Uy, x0 = 0, x1 = c.Ux;
    if (g.up) {
      const int phy = grp / g.su, phx = grp % g.su;
      y0 = c.Uy; y1 = 0; x0 = c.Ux; x1 = 0;
      for (int u = 0; u < c.Uy; ++u) {
        const int t = phy + (g.dmax_y - u) * g.su + g.kh / 2;
        if (t >= 0 && t < g.kh) { y0 = std::min(y0, u); y1 = std::max(y1, u + 1); }
      }
      for (int u = 0; u < c.Ux; ++u) {
        const int t = phx + (g.dmax_x - u) * g.su + g.kw / 2;
        if (t >= 0 && t < g.kw) { x0 = std::min(x0, u); x1 = std::max(x1, u + 1); }
      }
      if (y1 <= y0 || x1 <= x0) return -1;
    }
    c.ty0[grp] = y0; c.ty1[grp] = y1; c.tx0[grp] = x0; c.tx1[grp] = x1;
    most = std::max(most, (y1 - y0) * (x1 - x0));
  }
  c.ksteps = most * cb;
  Conv3Geom d{};
  d.lg = c.sd == 2 ? 1 : 0;
  d.BXn = (c.OWq + 31) / 32;
  d.BYn = (c.OHq + 7) / 8;
  const int PH = 7 * c.sd + c.Uy;
  d.PW = 31 * c.sd + c.Ux;
  d.PWh = (d.PW + c.sd - 1) / c.sd;
  d.granules = PH * c.sd * 2 * d.PWh;
  d.pixels = PH * d.PW;
  fast_div_setup(d.BXn, &d.bx_mul, &d.bx_sh);
  fast_div_setup(d.BYn, &d.by_mul, &d.by_sh);
  fast_div_setup(d.PW, &d.pw_mul, &d.pw_sh);
  const int npg = 2 * ((d.pixels + 255) / 256);        // 16-byte pieces per thread: two per patch pixel
  const int npgt = npg <= 4 ? 4 : 10;                  // the built loader widths
  const size_t patch_bytes = static_cast<size_t>(npgt) * 4096 + (npgt > 4 ? 2048 : 0);
  if (npg > npgt || static_cast<size_t>(d.granules) * 16 + 16 * d.PWh + 16 > patch_bytes) return -1;
  // a launch per tap count: the kernel is built for 25 taps (5 chunks of 5 K steps; the big patch loader), and 9
  // (3 x 3), 6 (2 x 3), 4 (1 x 4) taps with the small one
  int nts[kMaxGroups];
  for (int grp = 0; grp < c.groups; ++grp) {
    nts[grp] = (c.ty1[grp] - c.ty0[grp]) * (c.tx1[grp] - c.tx0[grp]);
    const bool built = (nts[grp] == 25 && npgt == 10) || ((nts[grp] == 9 || nts[grp] == 6 || nts[grp] == 4) && npgt == 4);
    if (!built) return -1;
    if (cb * (nts[grp] == 25 ? 5 : nts[grp] == 9 ? 3 : nts[grp] == 6 ? 2 : 1) < 2) return -1;   // (weights are requested two chunks ahead)
  }
  DevBuf packed_local;
  DevView packed;
  const long long frags = static_cast<long long>(c.groups) * c.ksteps * c.tiles * 64;
  {
    const int rc = packed_weights(1, {g.kh, g.kw, g.Cin_real, g.Cout, g.su, g.up, g.Uy, g.Ux, g.dmax_y, g.dmax_x, c.groups, c.ksteps,
                                      c.tiles, c.compact * 4 + c.cbmajor * 2 + c.small_cin, c.kw4},
                                  static_cast<size_t>(frags) * 16 + 64, st, packed_local, &packed.p, [&](void* dst) {
      SlowCall slow("conv_pack_kernel launch", __FILE__, __LINE__);
      hipLaunchKernelGGL((conv_pack_kernel<__bf16>), dim3(static_cast<unsigned>(ceil_div(frags, 256))), dim3(256), 0, st,
                         w, g, c, dst);
      return 0;
    });
    if (rc) return rc;
  }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  KernelTimer timer("conv2d", st);
  const int tap_counts[4] = {25, 9, 6, 4};
  for (int ntv : tap_counts) {
    d.gcount = 0;
    for (int grp = 0; grp < c.groups; ++grp)
      if (nts[grp] == ntv) d.glist[d.gcount++] = grp;
    if (!d.gcount) continue;
    fast_div_setup(d.gcount, &d.gc_mul, &d.gc_sh);
    const int chv = ntv == 25 ? 5 : ntv == 4 ? 4 : 3;
    // LDS: two patch buffers | two weight chunks | with GDN as the activation gamma's fragment image + beta — over the
    // weight buffers (copied in by the GDN stage) in the big-patch builds, behind them for the whole item in the
    // small-patch ones | the bias | the epilogue's staging area (4 waves x 8 KB) where there is room; else it takes the
    // weight buffers (image resident: nothing of them is needed after the K loop) or the patch buffer the last channel
    // block has left (the kernel's comments)
    const size_t wbufs = 2 * static_cast<size_t>((chv * c.tiles * 64 + 255) / 256) * 4096;
    const size_t image_bytes = static_cast<size_t>(c.tiles) * 2 * c.tiles * 64 * 16 + static_cast<size_t>(c.tiles) * 32 * 4;
    const bool resident = c.gdn && npgt == 4;
    size_t lds_all = 2 * patch_bytes;
    d.oimage = static_cast<int>(lds_all);
    if (resident) {
      lds_all += wbufs;
      d.oimage = static_cast<int>(lds_all);
      lds_all += (image_bytes + 1023) / 1024 * 1024;
    } else {
      lds_all += std::max(wbufs, c.gdn ? image_bytes : size_t{0});
    }
    d.obias = static_cast<int>(lds_all);
    lds_all += 1024;
    if (lds_all + 32768 <= 160 * 1024) {
      d.ostage = static_cast<int>(lds_all);
      lds_all += 32768;
    } else if (resident && wbufs >= 32768) {
      d.ostage = -2;
    } else if (patch_bytes >= 32768) {
      d.ostage = -1;
    } else {
      return -1;
    }
    if (lds_all > 160 * 1024) return -1;
    // One workgroup per block and group.  (A grid of one workgroup per CU, each taking every W-th block, is the same
    // speed alone on the chip — round 6, with the cheaper epilogue: 2 % / 6 % ahead on the stride-2 / transposed layer —
    // but keeps the kernels of other steps in flight out of its CUs: C4 51.6 instead of 47.6 ms per step,
    // profiles/r03_notes.md.)
    {
      static const int nt_env = [] { const char* e = std::getenv("TFC_CONV_NT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
      const long long out_bytes = c.N * c.OH * c.OW * c.Cout * (c.out_f32 ? 4 : 2);
      c.nt_out = nt_env >= 0 ? nt_env : (out_bytes > (128ll << 20) ? 1 : 0);
    }
    const long long nblk = c.N * d.BXn * d.BYn * d.gcount;
    if (nblk >= (1ll << 31)) return fail("tfc_conv2d: problem too large for one launch");
    const dim3 grid(static_cast<unsigned>(nblk));
#define TFC_CONV3_LAUNCH_G(NT, CHV, NCHV, NPGV, G, F32)                                                    \
    do {                                                                                                   \
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bf16_kernel<NT, CHV, NCHV, NPGV, G, F32>),  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_all)));     \
      hipLaunchKernelGGL((conv3_bf16_kernel<NT, CHV, NCHV, NPGV, G, F32>), grid, dim3(256), lds_all, st, x, packed.p, bias, y, c, d); \
    } while (0)
#define TFC_CONV3_LAUNCH(NT, CHV, NCHV, NPGV)                                                              \
    do {                                                                                                   \
      if (c.out_f32) TFC_CONV3_LAUNCH_G(NT, CHV, NCHV, NPGV, 0, true);                                     \
      else if (c.gdn == 2) TFC_CONV3_LAUNCH_G(NT, CHV, NCHV, NPGV, 2, false);                              \
      else if (c.gdn) TFC_CONV3_LAUNCH_G(NT, CHV, NCHV, NPGV, 1, false);                                   \
      else TFC_CONV3_LAUNCH_G(NT, CHV, NCHV, NPGV, 0, false);                                              \
    } while (0)
#define TFC_CONV3_TAPS(NT)                                                   \
    do {                                                                     \
      if (ntv == 25) TFC_CONV3_LAUNCH(NT, 5, 5, 10);                         \
      else if (ntv == 9) TFC_CONV3_LAUNCH(NT, 3, 3, 4);                      \
      else if (ntv == 6) TFC_CONV3_LAUNCH(NT, 3, 2, 4);                      \
      else TFC_CONV3_LAUNCH(NT, 4, 1, 4);                                    \
    } while (0)
    if (c.tiles == 6) TFC_CONV3_TAPS(6); else TFC_CONV3_TAPS(4);
#undef TFC_CONV3_TAPS
#undef TFC_CONV3_LAUNCH
#undef TFC_CONV3_LAUNCH_G
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int run_conv(const void* x, const float* w, const float* bias, void* y, ConvGeom c, PackGeom g,
             hipStream_t st, int* gdn_fused = nullptr) {
  constexpr int FB = ConvTraits<T>::kFragBytes;
  if constexpr (std::is_same<T, __bf16>::value) {
    const int rc = run_conv3(static_cast<const __bf16*>(x), w, bias, static_cast<__bf16*>(y), c, g, st);
    if (rc == 0 && c.gdn && gdn_fused) *gdn_fused = 1;
    if (rc >= 0) return rc;
  }
  c.gdn = 0;                          // (the other kernels leave the activation to the caller)
  const int tiles_total = (c.cols + 31) / 32;
  // image-side layers (first kernel): 3 column tiles per group — half the accumulators, twice the waves per
  // CU; measured 3.39 -> 2.98 ms on the 5x5 3 -> 192 /2 layer of bmshj2018 at 128 x 768x512 (for the 192 -> 192
  // layers of the second kernel the same split costs 45 %: every B gather is then issued twice)
  c.tiles = std::min(tiles_total, (c.small_cin && std::is_same<T, __bf16>::value) ? std::min(3, kMaxTiles) : kMaxTiles);
  c.groups = (tiles_total + c.tiles - 1) / c.tiles;
  // 64 KiB of LDS per workgroup (two workgroups per CU)
  c.kchunk = std::max(1, std::min(c.ksteps, (64 * 1024) / (c.tiles * 64 * FB)));
  const size_t lds = static_cast<size_t>(c.kchunk) * c.tiles * 64 * FB;

  c.compact = 0;
  if (std::is_same<T, __bf16>::value && !c.small_cin && g.up && c.groups <= kMaxGroups &&
      c.Cout % (c.tiles * 32) == 0) {
    // every column group lies inside one output phase: its taps are those with a kernel index in range
    c.compact = 1;
    int most = 0;
    for (int grp = 0; grp < c.groups; ++grp) {
      const int phase = grp * c.tiles * 32 / c.Cout;
      const int phy = phase / g.su, phx = phase % g.su;
      int y0 = c.Uy, y1 = 0, x0 = c.Ux, x1 = 0;
      for (int u = 0; u < c.Uy; ++u) {
        const int t = phy + (g.dmax_y - u) * g.su + g.kh / 2;
        if (t >= 0 && t < g.kh) { y0 = std::min(y0, u); y1 = std::max(y1, u + 1); }
      }
      for (int u = 0; u < c.Ux; ++u) {
        const int t = phx + (g.dmax_x - u) * g.su + g.kw / 2;
        if (t >= 0 && t < g.kw) { x0 = std::min(x0, u); x1 = std::max(x1, u + 1); }
      }
      if (y1 <= y0 || x1 <= x0) { y0 = 0; y1 = 1; x0 = 0; x1 = 1; }   // a phase without taps: one (zero) block
      c.ty0[grp] = y0; c.ty1[grp] = y1; c.tx0[grp] = x0; c.tx1[grp] = x1;
      most = std::max(most, (y1 - y0) * (x1 - x0));
    }
    c.ksteps = most * (c.Cin / 16);          // packed K steps per group (the stride of the packed buffer)
  }
  DevBuf packed_local, padded;
  DevView packed;
  const long long frags = static_cast<long long>(c.groups) * c.ksteps * c.tiles * 64;
  // + a zero page behind the fragments: what lanes outside the image read (16 bytes; the per-chunk pointers of
  // the FASTK kernel read it at + 32 (kChunk2 - 1))
  {
    const int rc = packed_weights(2, {g.kh, g.kw, g.Cin_real, g.Cout, g.su, g.up, g.Uy, g.Ux, g.dmax_y, g.dmax_x, c.groups,
                                      c.ksteps, c.tiles, c.compact * 4 + c.cbmajor * 2 + c.small_cin + 8 * static_cast<int>(sizeof(T)),
                                      c.kw4},
                                  static_cast<size_t>(frags) * FB + 32 * kChunk2, st, packed_local, &packed.p, [&](void* dst) {
      TFC_HIP(hipMemsetAsync(static_cast<unsigned char*>(dst) + static_cast<size_t>(frags) * FB, 0, 32 * kChunk2, st));
      SlowCall slow("conv_pack_kernel launch", __FILE__, __LINE__);
      hipLaunchKernelGGL((conv_pack_kernel<T>), dim3(static_cast<unsigned>(ceil_div(frags, 256))),
                         dim3(256), 0, st, w, g, c, dst);
      return 0;
    });
    if (rc) return rc;
  }
  const T* xin = static_cast<const T*>(x);
  if (c.small_cin) {
    // right margin: a K run may read up to kw4*16 values past the last window start
    const size_t elems = (static_cast<size_t>(c.N) * c.Hp * c.Wp + c.kw4 * 16 + 16) * 4;
    TFC_HIP(padded.alloc(elems * sizeof(T), st));
    TFC_HIP(hipMemsetAsync(padded.p, 0, elems * sizeof(T), st));
    const long long total = c.N * c.Hp * c.Wp;
    hipLaunchKernelGGL((conv_pad_image_kernel<T>), dim3(static_cast<unsigned>(ceil_div(total, 256))),
                       dim3(256), 0, st, xin, c, g.Cin_real, padded.as<T>());
    xin = padded.as<T>();
  }
  const long long M = c.N * c.OHq * c.OWq;
  if constexpr (std::is_same<T, __bf16>::value) if (!c.small_cin) {
    // second-generation kernel (not for the image layer: its K is only a few dozen steps and the
    // first kernel, which stages all of them at once, is faster there); 2 pixel tiles per wave
    // when that still fills the chip
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int mt = ceil_div(M, 256) * c.groups >= 2 * cus ? 2 : 1;
    const long long pb = ceil_div(M, 128 * mt);
    if (pb * c.groups >= (1ll << 31)) return fail("tfc_conv2d: problem too large for one launch");
    const dim3 grid2(static_cast<unsigned>(pb * c.groups));
    const size_t lds2 = static_cast<size_t>(2) * kChunk2 * c.tiles * 64 * 16;
    KernelTimer timer("conv2d", st);
#define TFC_CONV2_LAUNCH(NT, MTV, FK)                                                              \
    do {                                                                                           \
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<NT, MTV, FK>),   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds2))); \
      hipLaunchKernelGGL((conv_bf16_kernel<NT, MTV, FK>), grid2, dim3(256), lds2, st,              \
                         static_cast<const __bf16*>(static_cast<const void*>(xin)), packed.p, bias,  \
                         static_cast<__bf16*>(y), c);                                              \
    } while (0)
    // one tap per weight chunk (see the kernel): channel blocks per tap a multiple of the chunk
    const bool fastk = kPF == kChunk2 && (c.Cin / 16) % kChunk2 == 0 && !TFC_CONV_NO_FASTK;
    // (the fp32-output variant is a compile-time property: as a run-time flag in the epilogue it cost the
    // 6-tile kernel 832 bytes of scratch per lane, accumulators spilled inside the K loop)
    // (round 6: with the one-tap-per-chunk K loop too, for the float32 layers that run as six bfloat16 planes — built for
    // the 128- and 192-column groups, the widths of the models' layers)
#define TFC_CONV2_LAUNCH_F32K(NT, MTV, FK)                                                         \
    do {                                                                                           \
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<NT, MTV, FK, true>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds2))); \
      hipLaunchKernelGGL((conv_bf16_kernel<NT, MTV, FK, true>), grid2, dim3(256), lds2, st,        \
                         static_cast<const __bf16*>(static_cast<const void*>(xin)), packed.p, bias,  \
                         static_cast<__bf16*>(y), c);                                              \
    } while (0)
#define TFC_CONV2_LAUNCH_F32(NT, MTV)                                                              \
    do {                                                                                           \
      if constexpr (NT == 6 || NT == 4) {                                                          \
        if (fastk) TFC_CONV2_LAUNCH_F32K(NT, MTV, true); else TFC_CONV2_LAUNCH_F32K(NT, MTV, false); \
      } else {                                                                                     \
        TFC_CONV2_LAUNCH_F32K(NT, MTV, false);                                                     \
      }                                                                                            \
    } while (0)
#define TFC_CONV2_CASE(NT)                                                                         \
    case NT:                                                                                       \
      if (c.out_f32) { if (mt == 2) TFC_CONV2_LAUNCH_F32(NT, 2); else TFC_CONV2_LAUNCH_F32(NT, 1); } \
      else if (mt == 2) { if (fastk) TFC_CONV2_LAUNCH(NT, 2, true); else TFC_CONV2_LAUNCH(NT, 2, false); } \
      else { if (fastk) TFC_CONV2_LAUNCH(NT, 1, true); else TFC_CONV2_LAUNCH(NT, 1, false); }      \
      break
    switch (c.tiles) {
      TFC_CONV2_CASE(1);
      TFC_CONV2_CASE(2);
      TFC_CONV2_CASE(3);
      TFC_CONV2_CASE(4);
      TFC_CONV2_CASE(5);
      default:
        TFC_CONV2_CASE(6);
    }
#undef TFC_CONV2_CASE
#undef TFC_CONV2_LAUNCH
#undef TFC_CONV2_LAUNCH_F32
#undef TFC_CONV2_LAUNCH_F32K
    TFC_HIP(hipGetLastError());
    return 0;
  }
  const long long pblocks = ceil_div(M, 32 * kConvWaves);
  if (pblocks * c.groups >= (1ll << 31)) return fail("tfc_conv2d: problem too large for one launch");
  const dim3 grid(static_cast<unsigned>(pblocks * c.groups));
  KernelTimer timer("conv2d", st);
#define TFC_CONV_CASE(NT)                                                                        \
  case NT:                                                                                       \
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, NT>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
    hipLaunchKernelGGL((conv_kernel<T, NT>), grid, dim3(64 * kConvWaves), lds, st, xin, packed.p, \
                       bias, static_cast<T*>(y), c);                                             \
    break
  switch (c.tiles) {
    TFC_CONV_CASE(1);
    TFC_CONV_CASE(2);
    TFC_CONV_CASE(3);
    TFC_CONV_CASE(4);
    TFC_CONV_CASE(5);
    default:
      TFC_CONV_CASE(6);
  }
#undef TFC_CONV_CASE
  TFC_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Transposed convolution into a few channels (the last synthesis layer, C -> 3), bf16: as an implicit
// GEMM over output pixels it gathers every input pixel once per low-resolution tap (9 times for a 5x5
// stride-2 kernel) for 12 useful of 32 MFMA columns, and those gathers are all its time (6.8 ms at the
// C4 shape, 1.9 ms without them).  Instead: ONE 1x1 product per INPUT pixel with all kh*kw taps as output
// columns, z[i][(ty, tx, c)] = sum_ci x[i][ci] w[ty][tx][ci][c] (the same kernel, 12 K steps per pixel
// tile instead of 108), then every output pixel sums the <= ceil(k/s)^2 entries that land on it:
// o = i*s + t - k/2 (signal_conv.py:778-847; the alignment test_identity_kernel_alignment pins).
// z holds the products' fp32 accumulators (a bf16 z would round once per tap: with cancelling taps the error
// is relative to the taps, not to the output), 4 columns per tap so that a tap's channels are one aligned
// 16-byte read; the output is rounded once, like the implicit GEMM's.
// ---------------------------------------------------------------------------
__global__ void conv_up_weights_kernel(const float* w, int kh, int kw, int cin, int cout, float* w1) {
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  const long long total = static_cast<long long>(cin) * kh * kw * 4;
  if (idx >= total) return;
  const int c = idx & 3;
  const int tap = static_cast<int>((idx >> 2) % (kh * kw));
  const int ci = static_cast<int>(idx / (4ll * kh * kw));
  w1[idx] = c < cout ? w[(static_cast<long long>(tap) * cin + ci) * cout + c] : 0.f;     // [ci][tap][4]
}

__global__ void __launch_bounds__(256) conv_up_gather_kernel(const float* z, const float* bias, __bf16* y,
                                                             long long N, int H, int W, int kh, int kw, int s,
                                                             int cout, int activation) {
  const int OH = H * s, OW = W * s;
  const long long o = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (o >= N * OH * OW) return;
  const int ox = static_cast<int>(o % OW);
  const int oy = static_cast<int>((o / OW) % OH);
  const long long n = o / (static_cast<long long>(OW) * OH);
  const int zc = kh * kw * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // o = i*s + t - k/2  =>  t = (o + k/2) mod s, + s, ... ;  i = (o + k/2 - t) / s
  for (int ty = (oy + kh / 2) % s; ty < kh; ty += s) {
    const int iy = (oy + kh / 2 - ty) / s;
    if (iy < 0 || iy >= H) continue;
    for (int tx = (ox + kw / 2) % s; tx < kw; tx += s) {
      const int ix = (ox + kw / 2 - tx) / s;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(z + ((n * H + iy) * W + ix) * zc + (ty * kw + tx) * 4);
      acc[0] += v[0];
      acc[1] += v[1];
      acc[2] += v[2];
      acc[3] += v[3];
    }
  }
  for (int c = 0; c < cout; ++c) {
    float v = acc[c] + (bias ? bias[c] : 0.f);
    if (activation == 1) v = fmaxf(v, 0.f);
    y[o * cout + c] = static_cast<__bf16>(v);
  }
}

// ---------------------------------------------------------------------------
// Image-side analysis layer (Cin <= 4 -> 128 / 192 channels, stride 2 or 4, bf16): all of its time is the output
// (4.8 GB at the C4 shape) — K is a few dozen steps.  A workgroup takes an 8 x 32 block of output pixels, stages the
// image patch under it in LDS as rows of interleaved (x, c) values (19 x 67 pixels x 3 channels = 8 KB for 5x5 / 2),
// and runs K kernel row by kernel row: the kw * Cin values of a row are CONTIGUOUS in the patch, so a row is
// ceil(kw * Cin / 16) K steps (one for 5x5 x 3, against two with the 4-channel padding of the first kernel), and a
// B fragment is 16 bytes at (row, 2-byte-aligned column): four ds_read_b32.  Two workgroups per CU (LDS 40 KB, 256
// registers per wave): one's output stores drain under the other's MFMAs.
// ---------------------------------------------------------------------------
struct ImageConvGeom {
  long long N;
  int H, W, Cin, Cout;
  int kh, kw, sd;
  int py0, px0;
  int OH, OW;
  int BXn, BYn;
  int PH, RL, RS;           // patch rows; values per row in use; row stride (values)
  int ksr;                  // K steps per kernel row
  int activation;
  int pairs;                // 1: patch loaded two values at a time
  int items;                // blocks per workgroup
};

__global__ void conv_image_weights_kernel(const float* w, ImageConvGeom g, int tiles, bf16x8* packed) {
  // packed[((ty * ksr + ks) * tiles + t) * 64 + lane], lane (i, h): row values 16 ks + 8h .. + 7, output column 32t + i
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = g.kh * g.ksr * tiles * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  int r = idx >> 6;
  const int t = r % tiles; r /= tiles;
  const int ks = r % g.ksr, ty = r / g.ksr;
  const int co = 32 * t + (lane & 31), h = lane >> 5;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * ks + 8 * h + e;
    const int tx = k / g.Cin, c = k % g.Cin;
    v[e] = static_cast<__bf16>(tx < g.kw && co < g.Cout
                                   ? w[((static_cast<long long>(ty) * g.kw + tx) * g.Cin + c) * g.Cout + co] : 0.f);
  }
  packed[idx] = v;
}

template <int TILES>
__global__ void __launch_bounds__(256, 2) conv_image_kernel(const __bf16* x, const bf16x8* wpk, const float* bias,
                                                            __bf16* y, ImageConvGeom g) {
  extern __shared__ unsigned char smem[];            // patch: PH rows of RS values | weight fragments
  constexpr int MT = 2;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l = lane & 31;
  unsigned short* patch = reinterpret_cast<unsigned short*>(smem);
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem + static_cast<size_t>(g.PH) * g.RS * 2);
  const int nfr = g.kh * g.ksr * TILES * 64;
  for (int i = tid; i < nfr; i += 256) wl[i] = wpk[i];
  // the epilogue's staging area, one per wave, behind the fragments
  constexpr int ROW = 144;                             // bytes of a pixel's 64 channels in the staging area (+ 16: banks)
  unsigned char* const stg = reinterpret_cast<unsigned char*>(wl + nfr) + wid * (32 * ROW);
  // A workgroup takes g.items consecutive blocks: the weight fragments (30 KB at 5x5 x 3 -> 192, as much as a third of
  // a block's output) are staged once for all of them
  const long long nblk = g.N * g.BYn * g.BXn;
  const long long item_end = std::min(nblk, (static_cast<long long>(blockIdx.x) + 1) * g.items);
  // (requesting the next item's patch in front of this item's epilogue was measured: 1.83 -> 2.12 ms — the wait for
  // those loads is a wait for every store issued behind them)
  for (long long item = static_cast<long long>(blockIdx.x) * g.items; item < item_end; ++item) {
  long long b = item;
  const int bx = static_cast<int>(b % g.BXn); b /= g.BXn;
  const int by = static_cast<int>(b % g.BYn);
  const long long n = b / g.BYn;
  const int qx0 = bx * 32, qy0 = by * 8;
  // patch: value j of row py = x[n][iy][ix0 + j / Cin][j % Cin]: contiguous in the image row
  const unsigned short* xn = reinterpret_cast<const unsigned short*>(x) + n * g.H * g.W * g.Cin;
  const int ix0 = qx0 * g.sd - g.px0;
  constexpr int PB = 8;                                // loads in flight per thread
  if (g.pairs) {
    // two values per load: every image row and every patch row start on a 4-byte boundary (host check).  A pair that
    // straddles the image's edge is split by value, not by address: both halves are tested
    const int pvals = g.PH * g.RS / 2;
    unsigned int* patch2 = reinterpret_cast<unsigned int*>(patch);
    const int rs2 = g.RS / 2;
    for (int i0 = tid; i0 < pvals; i0 += 256 * PB) {
      unsigned int v[PB];
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const int i = i0 + 256 * k;
        const int py = i / rs2, j = 2 * (i - py * rs2);
        const int iy = qy0 * g.sd - g.py0 + py;
        const int ixa = ix0 + j / g.Cin, ixb = ix0 + (j + 1) / g.Cin;
        const bool row = (i < pvals) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(g.H));
        const bool oka = row & (j < g.RL) & (static_cast<unsigned int>(ixa) < static_cast<unsigned int>(g.W));
        const bool okb = row & (j + 1 < g.RL) & (static_cast<unsigned int>(ixb) < static_cast<unsigned int>(g.W));
        // the pair lies inside the image's allocation whenever one half is inside the image, except at the very first
        // and last value of the tensor: those pairs are read half by half
        const long long a = (static_cast<long long>(iy) * g.W + ix0) * g.Cin + j;
        const long long last = static_cast<long long>(g.H) * g.W * g.Cin - 2;
        unsigned int t = 0;
        if (oka | okb) {
          if (a >= 0 && a <= last) t = *reinterpret_cast<const unsigned int*>(xn + a);
          else t = (oka ? xn[a] : 0u) | (okb ? static_cast<unsigned int>(xn[a + 1]) << 16 : 0u);
        }
        v[k] = (oka ? t & 0xFFFFu : 0u) | (okb ? t & 0xFFFF0000u : 0u);
      }
#pragma unroll
      for (int k = 0; k < PB; ++k)
        if (i0 + 256 * k < pvals) patch2[i0 + 256 * k] = v[k];
    }
  } else {
    const int pvals = g.PH * g.RS;
    for (int i0 = tid; i0 < pvals; i0 += 256 * PB) {
      unsigned short v[PB];
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const int i = i0 + 256 * k;
        const int py = i / g.RS, j = i - py * g.RS;
        const int iy = qy0 * g.sd - g.py0 + py;
        const int ix = ix0 + j / g.Cin;                  // (j >= RL: pixels nobody multiplies with a weight)
        const bool ok = (i < pvals) & (j < g.RL) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(g.H)) &
                        (static_cast<unsigned int>(ix) < static_cast<unsigned int>(g.W));
        // clamped address, value selected afterwards: the loads of a batch are issued together
        const long long a = ok ? (static_cast<long long>(iy) * g.W + ix0) * g.Cin + j : 0;
        const unsigned short t = xn[a];
        v[k] = ok ? t : static_cast<unsigned short>(0);
      }
#pragma unroll
      for (int k = 0; k < PB; ++k)
        if (i0 + 256 * k < pvals) patch[i0 + 256 * k] = v[k];
    }
  }
  __syncthreads();
  f32x16 acc[MT][TILES];
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;
  // per-lane patch position of (tile p, kernel row 0, K step 0), in values
  int lb[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) lb[p] = ((2 * wid + p) * g.sd) * g.RS + l * g.sd * g.Cin + 8 * h;
  const int steps = g.kh * g.ksr;
  for (int st = 0; st < steps; ++st) {
    const int ty = st / g.ksr, ks = st - ty * g.ksr;
    const int o = ty * g.RS + 16 * ks;
    u32x4 bq[MT];
#pragma unroll
    for (int p = 0; p < MT; ++p) {
      const unsigned int* src = reinterpret_cast<const unsigned int*>(patch + lb[p] + o);     // 4-byte aligned
      bq[p] = u32x4{src[0], src[1], src[2], src[3]};
    }
    const bf16x8* abase = wl + st * TILES * 64 + lane;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const bf16x8 a = abase[t * 64];
#pragma unroll
      for (int p = 0; p < MT; ++p)
        acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bq[p]), acc[p][t], 0, 0, 0);
    }
  }
  // ---- epilogue: acc[p][t][4q + r] = column 32t + 8q + 4h + r of pixel (qy0 + 2 wid + p, qx0 + l).  The layer's time
  // is its output, so the stores are laid out for the memory system: a wave parks 64 channels of its 32 pixels in LDS
  // and takes them back eight lanes to a pixel, so that a
  // store instruction writes eight whole 128-byte lines — straight from the accumulators a lane holds 8 channels of a
  // pixel and an instruction touches 32 lines, 32 bytes of each ----
  __syncthreads();                                     // every wave is through with the patch: the next item may stage its own
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    const int qy = qy0 + 2 * wid + p;
#pragma unroll
    for (int cnk = 0; cnk < TILES / 2; ++cnk) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * cnk + tt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * q + 4 * h);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[p][t][4 * q + r] + b4[r];
            if (g.activation == 1) v[r] = fmaxf(v[r], 0.f);
          }
          uint2 o;
          o.x = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          o.y = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          *reinterpret_cast<uint2*>(stg + l * ROW + (32 * tt + 8 * q + 4 * h) * 2) = o;
        }
      }
      // (a wave's own staging area: its LDS accesses execute in order, no barrier)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = 8 * j + (lane >> 3), piece = lane & 7;
        const u32x4 o = *reinterpret_cast<const u32x4*>(stg + px * ROW + piece * 16);
        const int qx = qx0 + px;
        const int col = 64 * cnk + 8 * piece;
        if (qy < g.OH && qx < g.OW && col < g.Cout)
          *reinterpret_cast<u32x4*>(y + ((n * g.OH + qy) * g.OW + qx) * g.Cout + col) = o;
      }
    }
  }
  }       // items
}

// 0 = launched, -1 = not this shape, > 0 = error
int run_conv_image(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t h, int64_t wd,
                   int64_t cin, int64_t cout, int kh, int kw, int stride, int activation, hipStream_t st) {
  if (cin > 4 || (cout != 128 && cout != 192) || (stride != 2 && stride != 4) || (stride * cin) % 2) return -1;
  ImageConvGeom g{};
  g.N = n; g.H = static_cast<int>(h); g.W = static_cast<int>(wd); g.Cin = static_cast<int>(cin); g.Cout = static_cast<int>(cout);
  g.kh = kh; g.kw = kw; g.sd = stride; g.py0 = kh / 2; g.px0 = kw / 2; g.activation = activation;
  g.OH = static_cast<int>((h + stride - 1) / stride); g.OW = static_cast<int>((wd + stride - 1) / stride);
  g.BXn = (g.OW + 31) / 32; g.BYn = (g.OH + 7) / 8;
  g.PH = 7 * stride + kh;
  g.ksr = (kw * g.Cin + 15) / 16;
  g.RL = (31 * stride + kw) * g.Cin;
  g.RS = ((31 * stride * g.Cin + 16 * g.ksr + 8) + 7) & ~7;       // the last lane's last K step stays inside its row
  const int tiles = static_cast<int>(cout / 32);
  // (patch | weight fragments | the epilogue's staging area: 4 waves x 32 pixels x 144 bytes)
  const size_t lds = static_cast<size_t>(g.PH) * g.RS * 2 + static_cast<size_t>(kh) * g.ksr * tiles * 64 * 16 + 4 * 32 * 144;
  g.items = 8;
  if (lds > 160 * 1024 || n * g.BXn * g.BYn >= (1ll << 31)) return -1;
  // image rows and patch rows on 4-byte boundaries (x itself is: torch allocations are 256-byte aligned, and a slice
  // of a batch starts at a whole image)
  g.pairs = (g.W * g.Cin) % 2 == 0 && (g.px0 * g.Cin) % 2 == 0 && (g.H * g.W * g.Cin) % 2 == 0 &&
            reinterpret_cast<uintptr_t>(x) % 4 == 0;
  if (g.OW < 24) return -1;
  DevBuf wpk_local;
  DevView wpk;
  const int frags = kh * g.ksr * tiles * 64;
  {
    const int rc = packed_weights(3, {kh, kw, cin, cout, tiles, g.ksr}, static_cast<size_t>(frags) * 16, st, wpk_local, &wpk.p,
                                  [&](void* dst) {
      hipLaunchKernelGGL(conv_image_weights_kernel, dim3((frags + 255) / 256), dim3(256), 0, st, w, g, tiles,
                         static_cast<bf16x8*>(dst));
      return 0;
    });
    if (rc) return rc;
  }
  KernelTimer timer("conv2d", st);
  const dim3 grid(static_cast<unsigned>(ceil_div(n * g.BXn * g.BYn, g.items)));
  if (tiles == 6) {
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_image_kernel<6>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((conv_image_kernel<6>), grid, dim3(256), lds, st, static_cast<const __bf16*>(x), wpk.as<bf16x8>(),
                       bias, static_cast<__bf16*>(y), g);
  } else {
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_image_kernel<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((conv_image_kernel<4>), grid, dim3(256), lds, st, static_cast<const __bf16*>(x), wpk.as<bf16x8>(),
                       bias, static_cast<__bf16*>(y), g);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Image-side analysis layer, B fragments straight from the image, optionally WITH its GDN (bmshj2018's / ms2020's first
// layer: SignalConv2D(192, 5x5, /2, activation=GDN); bls2017's: 9x9, /4, its GDN as a kernel of its own — the 108 KB of
// 9x9 weight fragments and gamma's 74 KB do not share the LDS).  Cin = 3, Cout = 192, output rows of whole 32-pixel
// tiles.
// conv_image_kernel stages an image patch in LDS with 4-byte loads (0.79 ms for the 0.8 GB of bls2017's first layer at
// 512 x 256x256), and writes 4.8 GB at the C4 shape that the GDN kernel reads and writes again.  Here a wave takes 32
// consecutive pixels of an output row at a time and computes their 32 x Cout convolution outputs with KH * KSR K steps
// of MFMAs whose B fragments are 16 bytes of an image row each (the KW * Cin values of a kernel row are contiguous
// there; KSR = ceil(KW * Cin / 16) K steps per row) — which leaves them in the accumulators in exactly the layout the
// GDN contraction wants as ITS B fragments (gdn_common.h): contract |y| with gamma (fragments in LDS, as in the GDN
// kernel), divide, and store the result eight lanes to a pixel (whole 128-byte lines) through a wave-private staging
// area.  HBM traffic: the image in, the activations out, once.
// Same values as conv_image_kernel followed by the GDN kernel: the bias and beta are added to the finished float32 sums
// as there, the convolution's output is rounded to bfloat16 before the GDN as there.  (Carrying the bias in the spare
// sixteenth value of a kernel row's K step and beta as the contraction's initial accumulator was measured: 1.43 against
// 1.53 ms for the 5x5 layer, and two float32 additions that move inside the sums — 32 of 590 000 values differ, some by
// two bfloat16 units.  Interleaving a channel pair's epilogue with the next pair's MFMAs inside the wave: no change.)
// Persistent: a workgroup (8 waves) stages gamma's image and the convolution's fragments once and its waves walk the
// tiles (scalar arithmetic); image fragments live in a ring of PF registers sets, refilled — with this tile's later K
// steps, then the next tile's first ones — as soon as their MFMAs have read them.
// alpha = epsilon = 1, no rectification, not inverse.
// ---------------------------------------------------------------------------
template <int TILES, int KH, int KW, int SD, bool GDN>
__global__ void __launch_bounds__(512, 1) conv_image_direct_kernel(const __bf16* x, const bf16x8* wpk, const float* bias,
                                                                   const void* gimage, __bf16* y, ImageConvGeom g) {
  extern __shared__ unsigned char smem[];            // [gamma fragments | beta] | bias | conv fragments | staging (per wave)
  constexpr int KT = TILES, KS = 2 * TILES, C = 32 * TILES;
  constexpr int GFR = GDN ? KT * KS * 64 : 0, GBYTES = GDN ? GFR * 16 + C * 4 : 0;
  constexpr int ROW = 144;
  constexpr int CIN = 3, KSR = (KW * CIN + 15) / 16, NK = KH * KSR, PX0 = KW / 2, PY0 = KH / 2, PF = NK < 6 ? NK : 6;
  // zero padding left and right of an image row touches one lane each: (l, h) = (0, *) of a row's first tile, whose
  // fragment starts PX0 pixels left of the image, and (31, *) of its last tile
  static_assert(PX0 % 2 == 0 && (SD + PX0) % 2 == 0, "edge fix-ups move whole dwords");
  static_assert(SD >= PX0 && 16 * KSR <= CIN * (2 * SD + PX0), "only lanes 0 and 31 reach over the row's ends");
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l = lane & 31;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16x8* const ga = reinterpret_cast<const bf16x8*>(smem) + lane;
  const float* const beta_s = reinterpret_cast<const float*>(smem + GFR * 16);
  float* const bias_s = reinterpret_cast<float*>(smem + GBYTES);
  bf16x8* const wl = reinterpret_cast<bf16x8*>(smem + GBYTES + C * 4);
  constexpr int NFR = NK * TILES * 64;
  unsigned char* const stg = reinterpret_cast<unsigned char*>(wl + NFR) + wid * (32 * ROW);
  {
    if constexpr (GDN) {
      const u32x4* src = static_cast<const u32x4*>(gimage);
      u32x4* dst = reinterpret_cast<u32x4*>(smem);
      for (int i = tid; i < GFR + C / 4; i += 512) dst[i] = src[i];
    }
    for (int i = tid; i < NFR; i += 512) wl[i] = wpk[i];
    if (tid < C) bias_s[tid] = bias ? bias[tid] : 0.f;
  }
  __syncthreads();
  // (host: the tile count and the image's row count fit 32 bits)
  const unsigned int tpr = static_cast<unsigned int>(g.OW) / 32u;                  // tiles per output row
  const unsigned int ntiles = static_cast<unsigned int>(g.N) * g.OH * tpr;
  const unsigned int nwaves = gridDim.x * 8u, wave = blockIdx.x * 8u + wid;
  if (wave >= ntiles) return;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x), 0, static_cast<unsigned int>(g.N * g.H * g.W * CIN * 2), 0x00020000);
  // Fragment (K step ks = kernel row ks / KSR, part kk = ks % KSR) of lane (l, h): values 16 kk + 8 h ... + 7 of the
  // image row, counted from PX0 pixels left of pixel SD * (32 tx + l).  A row outside the image is read at an offset
  // outside the buffer (zeros).  Left of the image lie the first values of lane (0, h) in a row's first tile: that lane
  // reads DL dwords further right and the fix-up at the MFMA moves the fragment up by as much; right of it the last
  // values of lane (31, h) in the row's last tile: DT dwords further left (no load leaves the tensor), moved down.
  constexpr unsigned int kOutside = 0xFFFFFFF0u;
  auto clamp4 = [](int v) constexpr { return v < 0 ? 0 : v > 4 ? 4 : v; };
  struct Pos {
    unsigned int tx, n, oy;
    bool valid;
  };
  auto pos_of = [&](unsigned int tile) __attribute__((always_inline)) {
    Pos p;
    p.valid = tile < ntiles;
    p.tx = tile % tpr;
    const unsigned int r = tile / tpr;
    p.n = r / static_cast<unsigned int>(g.OH);
    p.oy = r - p.n * g.OH;
    return p;
  };
  auto columns = [&](const Pos& p, unsigned int (&cb)[KSR]) __attribute__((always_inline)) {
    const bool first = p.tx == 0 && l == 0, last = p.tx == tpr - 1 && l == 31;
#pragma unroll
    for (int kk = 0; kk < KSR; ++kk) {
      const int dla = clamp4((PX0 * CIN - 16 * kk + 1) / 2), dlb = clamp4((PX0 * CIN - 16 * kk - 8 + 1) / 2);
      const int dta = clamp4((16 * kk + 8 - CIN * (SD + PX0) + 1) / 2), dtb = clamp4((16 * kk + 16 - CIN * (SD + PX0) + 1) / 2);
      const int dl = h ? dlb : dla, dt = h ? dtb : dta;
      const int col = ((static_cast<int>(p.tx) * 32 + l) * SD - PX0) * CIN + 16 * kk + 8 * h;
      const bool none = (first && dl == 4) || (last && dt == 4);
      cb[kk] = none ? kOutside : static_cast<unsigned int>((col + (first ? 2 * dl : 0) - (last ? 2 * dt : 0)) * 2);
    }
  };
  auto fragment = [&](const Pos& p, const unsigned int (&cb)[KSR], int ks) __attribute__((always_inline)) {
    const int iy = static_cast<int>(p.oy) * SD - PY0 + ks / KSR;
    const bool rowok = p.valid && static_cast<unsigned int>(iy) < static_cast<unsigned int>(g.H);
    const unsigned int rowbytes = (p.n * g.H + iy) * static_cast<unsigned int>(g.W) * (CIN * 2u);   // < 2^32 (host)
    return __builtin_amdgcn_raw_buffer_load_b128(xr, rowok ? cb[ks % KSR] : kOutside, rowok ? rowbytes : 0u, 0);
  };
  auto up = [](u32x4 v, int d) __attribute__((always_inline)) {
    return d == 0 ? v : d == 1 ? u32x4{0u, v[0], v[1], v[2]} : d == 2 ? u32x4{0u, 0u, v[0], v[1]}
         : d == 3 ? u32x4{0u, 0u, 0u, v[0]} : u32x4{0u, 0u, 0u, 0u};
  };
  auto down = [](u32x4 v, int d) __attribute__((always_inline)) {
    return d == 0 ? v : d == 1 ? u32x4{v[1], v[2], v[3], 0u} : d == 2 ? u32x4{v[2], v[3], 0u, 0u}
         : d == 3 ? u32x4{v[3], 0u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
  };
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Pos cur = pos_of(wave);
  unsigned int cb[KSR];
  columns(cur, cb);
  u32x4 bq[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) bq[k] = fragment(cur, cb, k);
  for (unsigned int tile = wave;; tile += nwaves) {
    const Pos nxt = pos_of(tile + nwaves);
    unsigned int ncb[KSR];
    columns(nxt, ncb);
    const bool first = cur.tx == 0 && l == 0, last = cur.tx == tpr - 1 && l == 31;
    __bf16* const yrow = y + ((static_cast<long long>(cur.n) * g.OH + cur.oy) * g.OW + cur.tx * 32) * C;
    // ---- convolution: the weights' fragments one K step ahead of their MFMAs ----
    f32x16 acc[KT];
    bf16x8 aw[2][KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) aw[0][t] = wl[t * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int kk = ks % KSR, slot = ks % PF;
      if (ks + 1 < NK) {
#pragma unroll
        for (int t = 0; t < KT; ++t) aw[(ks + 1) & 1][t] = wl[((ks + 1) * TILES + t) * 64 + lane];
      }
      u32x4 v = bq[slot];
      {
        const int dla = clamp4((PX0 * CIN - 16 * kk + 1) / 2), dlb = clamp4((PX0 * CIN - 16 * kk - 8 + 1) / 2);
        const int dta = clamp4((16 * kk + 8 - CIN * (SD + PX0) + 1) / 2), dtb = clamp4((16 * kk + 16 - CIN * (SD + PX0) + 1) / 2);
        if (dla || dlb) v = first ? (h ? up(v, dlb) : up(v, dla)) : v;
        if (dta || dtb) v = last ? (h ? down(v, dtb) : down(v, dta)) : v;
      }
      const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
      for (int t = 0; t < KT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[ks & 1][t], bfrag, ks ? acc[t] : zero, 0, 0, 0);
      // the slot's next fragment: a later K step of this tile, then the next tile's first ones
      const int kn = ks + PF;
      bq[slot] = kn < NK ? fragment(cur, cb, kn) : fragment(nxt, ncb, kn - NK);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (GDN) {
      // y = convolution + bias as the bfloat16 tensor would hold it: acc[t][4q + r] = channel 32 t + 8 q + 4 h + r of
      // pixel l, and K step s of the contraction = channels 16 s + 4 h + {0..3} and + 8
      u32x4 xb[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int t0 = s >> 1, q0 = 2 * (s & 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + 32 * t0 + 8 * (q0 + half) + 4 * h);
          const int e = 4 * (q0 + half);
          xb[s][2 * half] = __builtin_bit_cast(unsigned int, __builtin_convertvector(
                                f32x2{acc[t0][e], acc[t0][e + 1]} + f32x2{b4[0], b4[1]}, bf16x2));
          xb[s][2 * half + 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(
                                    f32x2{acc[t0][e + 2], acc[t0][e + 3]} + f32x2{b4[2], b4[3]}, bf16x2));
        }
      }
      // gamma's fragments one K step ahead, each register set refilled as soon as its MFMA has read it (a second
      // set does not fit: 96 accumulators + 48 of y + the image fragments in flight)
      bf16x8 af[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) af[t] = ga[(t * KS) * 64];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, xb[s] & 0x7FFF7FFFu);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bfrag, s ? acc[t] : zero, 0, 0, 0);
          if (s + 1 < KS) af[t] = ga[(t * KS + s + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // y / (beta + gamma^T |y|), 64 channels at a time through the staging area, then eight lanes to a pixel
#pragma unroll
      for (int cnk = 0; cnk < TILES / 2; ++cnk) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * cnk + tt;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int s = 2 * t + (q >> 1), half = q & 1;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + 32 * t + 8 * q + 4 * h);
            f32x2 v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const unsigned int word = xb[s][2 * half + k];
              const f32x2 yv = {__uint_as_float(word << 16), __uint_as_float(word & 0xFFFF0000u)};
              const f32x2 nn = f32x2{acc[t][4 * q + 2 * k], acc[t][4 * q + 2 * k + 1]} + f32x2{b4[2 * k], b4[2 * k + 1]};
              const f32x2 rn = {__builtin_amdgcn_rcpf(nn[0]), __builtin_amdgcn_rcpf(nn[1])};
              v[k] = yv * rn;
            }
            uint2 o;
            o.x = __builtin_bit_cast(unsigned int, __builtin_convertvector(v[0], bf16x2));
            o.y = __builtin_bit_cast(unsigned int, __builtin_convertvector(v[1], bf16x2));
            *reinterpret_cast<uint2*>(stg + l * ROW + (32 * tt + 8 * q + 4 * h) * 2) = o;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int px = 8 * j + (lane >> 3), piece = lane & 7;
          const u32x4 o = *reinterpret_cast<const u32x4*>(stg + px * ROW + piece * 16);
          // (non-temporal stores measured level here: 1.56 ms either way at 128 x 768x512 — the layer writes at 3.1 TB/s where a
          // fill of its output runs at 6.9: it is bound by instruction issue, profiles/r04_notes.md)
          *reinterpret_cast<u32x4*>(yrow + px * C + 64 * cnk + 8 * piece) = o;
        }
      }
    } else {
      // bias, activation, round; 64 channels at a time through the staging area, then eight lanes to a pixel
#pragma unroll
      for (int cnk = 0; cnk < TILES / 2; ++cnk) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * cnk + tt;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + 32 * t + 8 * q + 4 * h);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[k] = acc[t][4 * q + k] + b4[k];
              if (g.activation == 1) v[k] = fmaxf(v[k], 0.f);
            }
            uint2 o;
            o.x = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
            o.y = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
            *reinterpret_cast<uint2*>(stg + l * ROW + (32 * tt + 8 * q + 4 * h) * 2) = o;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int px = 8 * j + (lane >> 3), piece = lane & 7;
          const u32x4 o = *reinterpret_cast<const u32x4*>(stg + px * ROW + piece * 16);
          // (non-temporal stores measured level here: 1.56 ms either way at 128 x 768x512 — the layer writes at 3.1 TB/s where a
          // fill of its output runs at 6.9: it is bound by instruction issue, profiles/r04_notes.md)
          *reinterpret_cast<u32x4*>(yrow + px * C + 64 * cnk + 8 * piece) = o;
        }
      }
    }
    if (!nxt.valid) break;
    cur = nxt;
#pragma unroll
    for (int kk = 0; kk < KSR; ++kk) cb[kk] = ncb[kk];
  }
}

// 0 = launched, -1 = not this shape, > 0 = error.  gdn != null: with the GDN (5x5 / 2 only).
int run_conv_image_direct(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t h, int64_t wd,
                          int64_t cin, int64_t cout, int kh, int kw, int stride, int activation, const tfc_gdn_params* gdn,
                          hipStream_t st) {
  if (cin != 3 || cout != 192 || kh != kw) return -1;
  const bool k5 = kh == 5 && stride == 2, k9 = kh == 9 && stride == 4;
  if (!k5 && !k9) return -1;
  if (gdn && (!k5 || gdn->channels != cout || gdn->dtype != 1 || activation != 0)) return -1;
  ImageConvGeom g{};
  g.N = n; g.H = static_cast<int>(h); g.W = static_cast<int>(wd); g.Cin = static_cast<int>(cin); g.Cout = static_cast<int>(cout);
  g.kh = kh; g.kw = kw; g.sd = stride; g.py0 = kh / 2; g.px0 = kw / 2; g.activation = activation;
  g.OH = static_cast<int>((h + stride - 1) / stride); g.OW = static_cast<int>((wd + stride - 1) / stride);
  g.ksr = (kw * g.Cin + 15) / 16;
  const int nk = kh * g.ksr;
  // whole tiles, the row's last pixel a whole stride from its end, dword-aligned rows
  if (g.OW % 32 != 0 || g.W % stride != 0 || reinterpret_cast<uintptr_t>(x) % 4 != 0) return -1;
  // one buffer resource over the tensor, 32-bit tile and row arithmetic in the kernel
  if (static_cast<double>(n) * h * wd * cin * 2 >= 4294967280.0 || static_cast<double>(n) * g.OH * (g.OW / 32) >= 2147483648.0)
    return -1;
  constexpr int tiles = 6;
  const size_t lds = (gdn ? static_cast<size_t>(tiles) * 2 * tiles * 64 * 16 + tiles * 32 * 4 : 0) + tiles * 32 * 4 +
                     static_cast<size_t>(nk) * tiles * 64 * 16 + 8 * 32 * 144;
  if (lds > 160 * 1024) return -1;
  DevBuf wpk_local;
  DevView wpk;
  const int frags = nk * tiles * 64;
  {
    const int rc = packed_weights(3, {kh, kw, cin, cout, tiles, g.ksr}, static_cast<size_t>(frags) * 16, st, wpk_local, &wpk.p,
                                  [&](void* dst) {
      hipLaunchKernelGGL(conv_image_weights_kernel, dim3((frags + 255) / 256), dim3(256), 0, st, w, g, tiles,
                         static_cast<bf16x8*>(dst));
      return 0;
    });
    if (rc) return rc;
  }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long ntiles = n * g.OH * (g.OW / 32);
  const unsigned grid = static_cast<unsigned>(std::min<long long>(cus, ceil_div(ntiles, 8)));
  KernelTimer timer("conv2d", st);
#define TFC_IMAGE_DIRECT_LAUNCH(...)                                                                              \
  do {                                                                                                           \
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_image_direct_kernel<__VA_ARGS__>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));             \
    hipLaunchKernelGGL((conv_image_direct_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, st,                   \
                       static_cast<const __bf16*>(x), wpk.as<bf16x8>(), bias, gdn ? gdn->image.p : nullptr,       \
                       static_cast<__bf16*>(y), g);                                                              \
  } while (0)
  if (k5 && gdn) TFC_IMAGE_DIRECT_LAUNCH(6, 5, 5, 2, true);
  else if (k5) TFC_IMAGE_DIRECT_LAUNCH(6, 5, 5, 2, false);
  else TFC_IMAGE_DIRECT_LAUNCH(6, 9, 9, 4, false);
#undef TFC_IMAGE_DIRECT_LAUNCH
  TFC_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// The same, fused: the tap products of a block never leave the CU.  A workgroup takes an 8 x 32 block of INPUT pixels
// plus the halo the block's output pixels reach into ((8 + Uy - 1) x (32 + Ux - 1) patch pixels, 340 for a 5x5
// stride-2 kernel), computes z[patch pixel][(tap, c)] for them with the MFMA (fp32, into LDS: <= 80 columns, row
// stride 84 floats so that the 16-byte accumulator writes of 8 lanes cover all banks), and then every thread sums, for
// 4 consecutive output pixels of the block's (8s) x (32s) output tile, the <= ceil(k/s)^2 products that land on each.
// HBM traffic: the input once (+ 33 % halo, mostly L2 hits) and the output, instead of + 2 x 400 bytes of z per input
// pixel (5 GB each way at the C4 shape).  K and tap order as in the unfused pair.
// ---------------------------------------------------------------------------
struct UpFusedGeom {
  long long N;
  int H, W, Cin, Cout;
  int kh, kw, s;
  int dmax_y, dmax_x;       // halo before: patch origin = block origin - dmax
  int PH, PW;               // patch rows, columns
  int BXn, BYn;
  int NC;                   // kh * kw * Cout product columns
  int activation;
};
constexpr int kUpZStride = 84;      // floats per patch pixel in LDS
constexpr int kUpColTiles = 3;      // 96 >= NC columns

__global__ void conv_up_fused_weights_kernel(const float* w, int kh, int kw, int cin, int cout, bf16x8* packed) {
  // A fragments: packed[(ks * kUpColTiles + t) * 64 + lane], lane (i, h): K offsets 8h .. 8h + 7 of step ks,
  // column 32t + i = ((ty * kw + tx) * cout + c)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int cb = cin / 16;
  const int total = cb * kUpColTiles * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  int r = idx >> 6;
  const int t = r % kUpColTiles; r /= kUpColTiles;
  const int ks = r % cb;
  const int col = 32 * t + (lane & 31), h = lane >> 5;
  const int jy = col / (kw * cout), rem = col % (kw * cout);
  const int tx = rem / cout, c = rem % cout;
  const int ty = jy;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ks * 16 + 8 * h + e;
    v[e] = static_cast<__bf16>(ty < kh ? w[((static_cast<long long>(ty) * kw + tx) * cin + ci) * cout + c] : 0.f);
  }
  packed[idx] = v;
}

constexpr int kUpThreads = 512;     // 8 waves: the block's 11 pixel tiles and 1024 output pixels over more waves
// KH, KW, S, COUT > 0: the gather is unrolled for that kernel (5, 5, 2, 3: bmshj2018's last layer); 0: any.
template <int KH, int KW, int S, int COUT>
__global__ void __launch_bounds__(kUpThreads) conv_up_fused_kernel(const __bf16* x, const bf16x8* wpk,
                                                                    const float* bias, __bf16* y, UpFusedGeom g) {
  extern __shared__ unsigned char smem[];            // z: tiles * 32 rows of kUpZStride floats | weight fragments
  constexpr int WAVES = kUpThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l = lane & 31;
  long long b = blockIdx.x;
  const int bx = static_cast<int>(b % g.BXn); b /= g.BXn;
  const int by = static_cast<int>(b % g.BYn);
  const long long n = b / g.BYn;
  const int npix = g.PH * g.PW;
  const int tiles = (npix + 31) / 32;
  float* z = reinterpret_cast<float*>(smem);
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem + static_cast<size_t>(tiles) * 32 * kUpZStride * 4);
  const int cb = g.Cin / 16;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x + n * g.H * g.W * g.Cin), 0, g.H * g.W * g.Cin * 2, 0x00020000);
  // this wave's first pixel tile: its first four K steps are requested before the weights are staged
  auto tile_offset = [&](int tile) -> unsigned int {
    const int pi = tile * 32 + l;
    const int py = pi / g.PW, px = pi - py * g.PW;
    const int iy = by * 8 - g.dmax_y + py, ix = bx * 32 - g.dmax_x + px;
    const bool ok = (pi < npix) & (static_cast<unsigned int>(iy) < static_cast<unsigned int>(g.H)) &
                    (static_cast<unsigned int>(ix) < static_cast<unsigned int>(g.W));
    return ok ? static_cast<unsigned int>(((iy * g.W + ix) * g.Cin + 8 * h) * 2) : 0x80000000u;
  };
  constexpr int PF = 4;                              // K steps in flight per pixel tile
  const int kh = KH ? KH : g.kh, kw = KW ? KW : g.kw, s = S ? S : g.s, cout = COUT ? COUT : g.Cout;
  const int OH = g.H * s, OW = g.W * s;
  const int tw = 32 * s, th = 8 * s;
  float bc[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias)
    for (int c = 0; c < cout; ++c) bc[c] = bias[c];
  u32x4 bq[PF];
  unsigned int off = tile_offset(wid);
#pragma unroll
  for (int k = 0; k < PF; ++k) bq[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, (k < cb ? k : cb - 1) * 32, 0);
  {
    const bf16x8* src = wpk;
    for (int i = tid; i < cb * kUpColTiles * 64; i += kUpThreads) wl[i] = src[i];
  }
  __syncthreads();
  // ---- z of the patch, a pixel tile (32 patch pixels) per wave at a time ----
  for (int tile = wid; tile < tiles; tile += WAVES) {
    const int pi = tile * 32 + l;
    f32x16 acc[kUpColTiles];
#pragma unroll
    for (int t = 0; t < kUpColTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned int offn = tile_offset(tile + WAVES < tiles ? tile + WAVES : tile);
    for (int ks = 0; ks < cb; ks += PF) {            // (Cin % 64 == 0: PF K steps per turn, a ring of PF fragments)
#pragma unroll
      for (int k = 0; k < PF; ++k) {
#pragma unroll
        for (int t = 0; t < kUpColTiles; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[((ks + k) * kUpColTiles + t) * 64 + lane],
                                                          __builtin_bit_cast(bf16x8, bq[k]), acc[t], 0, 0, 0);
        // refill the slot: K step ks + k + PF of this tile, or the next tile's first ones
        const int kn = ks + k + PF;
        bq[k] = kn < cb ? __builtin_amdgcn_raw_buffer_load_b128(xr, off, kn * 32, 0)
                        : __builtin_amdgcn_raw_buffer_load_b128(xr, offn, (kn - cb) * 32, 0);
      }
    }
    off = offn;
    // acc[t][4q + r] = column 32t + 8q + 4h + r of patch pixel pi; columns < 80 go to LDS
    float* zrow = z + static_cast<size_t>(pi) * kUpZStride;
#pragma unroll
    for (int t = 0; t < kUpColTiles; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (32 * t + 8 * q >= kUpZStride) continue;
        if (32 * t + 8 * q + 4 * h + 4 > kUpZStride) continue;      // (columns 84 .. 87 of the h = 1 lanes: no room)
        *reinterpret_cast<f32x4*>(zrow + 32 * t + 8 * q + 4 * h) =
            f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
      }
  }
  __syncthreads();
  // ---- gather: thread -> 4 consecutive output pixels of the (8 s) x (32 s) output tile ----
  for (int grp = tid; grp < th * (tw / 4); grp += kUpThreads) {
    const int ry = grp / (tw / 4), rx = (grp % (tw / 4)) * 4;
    const int oy = by * th + ry;
    if (oy >= OH) continue;
    // o = i*s + t - k/2  =>  t = (o + k/2) mod s, + s, ... ;  i = (o + k/2 - t) / s
    const int ty0 = (oy + kh / 2) % s;
    const int py0 = (oy + kh / 2 - ty0) / s - (by * 8 - g.dmax_y);       // patch row of tap ty0; ty0 + s: one row up
    unsigned short outv[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = bx * tw + rx + k;
      const int tx0 = (ox + kw / 2) % s;
      const int px0 = (ox + kw / 2 - tx0) / s - (bx * 32 - g.dmax_x);
      float a[4] = {bc[0], bc[1], bc[2], bc[3]};
#pragma unroll
      for (int jy = 0; jy < (KH ? (KH + S - 1) / S : 8); ++jy) {
        const int ty = ty0 + jy * s;
        if (ty >= kh) break;
#pragma unroll
        for (int jx = 0; jx < (KW ? (KW + S - 1) / S : 8); ++jx) {
          const int tx = tx0 + jx * s;
          if (tx >= kw) break;
          const float* zp = z + static_cast<size_t>((py0 - jy) * g.PW + (px0 - jx)) * kUpZStride +
                            (ty * kw + tx) * cout;
#pragma unroll
          for (int c = 0; c < (COUT ? COUT : 4); ++c)
            if (c < cout) a[c] += zp[c];
        }
      }
#pragma unroll
      for (int c = 0; c < (COUT ? COUT : 4); ++c) {
        float v = a[c];
        if (g.activation == 1) v = fmaxf(v, 0.f);
        outv[k * 4 + c] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
      }
    }
    __bf16* dst = y + ((n * OH + oy) * OW + bx * tw + rx) * cout;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (bx * tw + rx + k >= OW) continue;
#pragma unroll
      for (int c = 0; c < (COUT ? COUT : 4); ++c)
        if (c < cout) dst[k * cout + c] = __builtin_bit_cast(__bf16, outv[k * 4 + c]);
    }
  }
}

// ---------------------------------------------------------------------------
// Transposed convolution into three channels (the synthesis transforms' last layer: bls2017's 9x9 x4, bmshj2018's 5x5
// x2, 192 -> 3) as a 3x3 convolution over the INPUT grid with S * S * 3 output columns, one per (output phase, channel):
//   y[S I + ry, S J + rx, c] = sum over dy, dx in {-1, 0, 1}, ci of x[I + dy, J + dx, ci] * w[t(ry, dy), t(rx, dx), ci, c]
// with t(r, d) = (r + K/2) mod S + S ((r + K/2) div S - d), taps outside [0, K) contributing nothing (from
// o = S i + t - K/2, the transposed convolution behind signal_conv.py:778-847's same_zeros padding).  The sums run in
// the MFMA's K dimension (v_mfma_f32_16x16x32_bf16: 16 columns x 16 pixels x 32 channels), so no tap products pass
// through memory or LDS (conv_up_fused_kernel parks 80 float32 products per input pixel in LDS and gathers; a 9x9
// kernel's 243 do not fit and that layer ran as an implicit GEMM over output pixels: 1.12 ms at 512 x 64x64).
// Columns: 16 per MFMA — S = 4: the twelve (rx, c) of one output row phase ry (+ 4 unused), four groups; S = 2: eight
// per ry (six used), one group.  Lane (n, g) of an accumulator then holds four consecutive values of output pixel
// row S I + ry at byte 2 (3 S (J0 + n) + 4 g): a wave's stores cover whole contiguous runs of the output row.
// A workgroup (8 waves) takes 8 input rows x 32 pixels, a wave one row (two 16-pixel halves); per 32-channel chunk the
// block's input tile (10 x 34 pixels, 96-byte pixel stride: conflict-free ds_read_b128 for the b128 lane groups) and
// the chunk's weight fragments are staged in LDS, the next chunk's global loads in flight under the MFMAs.  Two
// workgroups per CU.  For S = 4 the row phases ry >= 1 have no tap at dy = -1: those MFMAs are not issued.
// ---------------------------------------------------------------------------
struct UpPhaseGeom {
  long long N;
  int H, W, Cin;
  int activation;
  int nchunk;               // Cin / 32
};
constexpr int kPhaseRows = 8;                          // input rows per workgroup
constexpr int kPhaseTilePix = (kPhaseRows + 2) * 34;

template <int K, int S>
struct UpPhase {
  static constexpr int GROUPS = S == 4 ? 4 : 1;        // MFMA column groups
  // fragments per chunk, in the order the kernel walks them: dy, dx, group (skipping the groups without a tap)
  static constexpr int FRAGS = S == 4 ? 3 * 1 + 2 * 3 * 4 : 9;
  // 32-channel chunks staged at a time, and the tile's pixel stride in LDS (bytes; 64 of data per chunk): S = 2 takes two
  // chunks = one whole 128-byte line of a pixel per stage (a half line per stage was measured: 1.22 against 1.18 ms at
  // 128 x 256x384 — the other half has left the L2 by the next stage); S = 4 has 27 KB of fragments per chunk and
  // stays with one.  Both strides keep ds_read_b128 conflict-free for the b128 lane groups (16 n + g -> n PXS + 16 g).
  static constexpr int CPS = S == 4 ? 1 : 2;
  static constexpr int PXS = CPS == 1 ? 96 : 160;
  __host__ __device__ static constexpr int tap(int r, int d) { return (r + K / 2) % S + S * ((r + K / 2) / S - d); }
  __host__ __device__ static constexpr bool group_has(int grp, int dy) {
    return S != 4 || (tap(grp, dy) >= 0 && tap(grp, dy) < K);
  }
};

template <int K, int S>
__global__ void conv_up_phase_weights_kernel(const float* w, int cin, int cout, bf16x8* packed) {
  // packed[(chunk * FRAGS + slot) * 64 + lane], lane (m, g): column m of the slot's group, channels 32 chunk + 8 g ... + 7
  using P = UpPhase<K, S>;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (cin / 32) * P::FRAGS * 64;
  if (idx >= total) return;
  const int lane = idx & 63, m = lane & 15, gq = lane >> 4;
  int slot = (idx >> 6) % P::FRAGS;
  const int chunk = (idx >> 6) / P::FRAGS;
  int dy = -1, dx = -1, grp = 0;
  {
    int s = 0;
    bool found = false;
    for (int a = -1; a <= 1 && !found; ++a)
      for (int b = -1; b <= 1 && !found; ++b)
        for (int gr = 0; gr < P::GROUPS && !found; ++gr) {
          if (!P::group_has(gr, a)) continue;
          if (s == slot) { dy = a; dx = b; grp = gr; found = true; }
          ++s;
        }
  }
  // column m of the group -> (ry, rx, c)
  int ry, rest;
  if (S == 4) { ry = grp; rest = m; } else { ry = m >> 3; rest = m & 7; }
  const int rx = rest / 3, c = rest % 3;
  const bool col_ok = rest < 3 * S && c < cout;
  const int ty = P::tap(ry, dy), tx = col_ok ? P::tap(rx, dx) : -1;
  const bool ok = col_ok && ty >= 0 && ty < K && tx >= 0 && tx < K;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = 32 * chunk + 8 * gq + e;
    v[e] = static_cast<__bf16>(ok ? w[((static_cast<long long>(ty) * K + tx) * cin + ci) * cout + c] : 0.f);
  }
  packed[idx] = v;
}

template <int K, int S>
__global__ void __launch_bounds__(512, 2) conv_up_phase_kernel(const __bf16* x, const bf16x8* wpk, const float* bias,
                                                               __bf16* y, UpPhaseGeom g) {
  using P = UpPhase<K, S>;
  constexpr int GROUPS = P::GROUPS, FRAGS = P::FRAGS, HALVES = 2, CPS = P::CPS, PXS = P::PXS;
  extern __shared__ unsigned char smem[];            // input tile (CPS chunks) | weight fragments (CPS chunks)
  unsigned char* const xt = smem;
  bf16x8* const wt = reinterpret_cast<bf16x8*>(smem + kPhaseTilePix * PXS);
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bxn = (g.W + 31) / 32, byn = (g.H + kPhaseRows - 1) / kPhaseRows;
  long long b = blockIdx.x;                          // (blocks in XCD order, xcd_order: measured level, 1.17-1.22 ms)
  const int bx = static_cast<int>(b % bxn); b /= bxn;
  const int by = static_cast<int>(b % byn);
  const long long img = b / byn;
  const int I0 = by * kPhaseRows, J0 = bx * 32;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x + img * g.H * g.W * g.Cin), 0, static_cast<unsigned int>(g.H * g.W * g.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16x8*>(wpk), 0, static_cast<unsigned int>(g.nchunk * FRAGS * 1024), 0x00020000);
  // staging: 16-byte pieces, 4 CPS to a tile pixel t (row t / 34, column t % 34 of the tile, origin (I0 - 1, J0 - 1))
  constexpr int PPP = 4 * CPS;                         // pieces per pixel
  constexpr int XP = (kPhaseTilePix * PPP + 511) / 512, WP = (CPS * FRAGS * 64 + 511) / 512;
  unsigned int xoff[XP];
#pragma unroll
  for (int r = 0; r < XP; ++r) {
    const int piece = tid + 512 * r, t = piece / PPP, q = piece % PPP;
    const int iy = I0 - 1 + t / 34, ix = J0 - 1 + t % 34;
    const bool ok = t < kPhaseTilePix && static_cast<unsigned int>(iy) < static_cast<unsigned int>(g.H) &&
                    static_cast<unsigned int>(ix) < static_cast<unsigned int>(g.W);
    xoff[r] = ok ? static_cast<unsigned int>(((iy * g.W + ix) * g.Cin) * 2 + q * 16) : 0x80000000u;
  }
  u32x4 xs[XP], ws[WP];
  auto request = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < XP; ++r) xs[r] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[r], stage * (64 * CPS), 0);
#pragma unroll
    for (int r = 0; r < WP; ++r) {
      const int i = tid + 512 * r;
      ws[r] = __builtin_amdgcn_raw_buffer_load_b128(wr, i < CPS * FRAGS * 64 ? i * 16u : 0x80000000u,
                                                    stage * (CPS * FRAGS * 1024), 0);
    }
  };
  auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < XP; ++r) {
      const int piece = tid + 512 * r;
      if (piece < kPhaseTilePix * PPP)
        *reinterpret_cast<u32x4*>(xt + (piece / PPP) * PXS + (piece % PPP) * 16) = xs[r];
    }
#pragma unroll
    for (int r = 0; r < WP; ++r) {
      const int i = tid + 512 * r;
      if (i < CPS * FRAGS * 64) reinterpret_cast<u32x4*>(wt)[i] = ws[r];
    }
  };
  f32x4 acc[GROUPS][HALVES];
#pragma unroll
  for (int gr = 0; gr < GROUPS; ++gr)
#pragma unroll
    for (int p = 0; p < HALVES; ++p) acc[gr][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's B fragment of tap (0, 0), half 0: tile pixel (wid + 1, n + 1), channels 8 gq ... + 7 of the chunk
  const unsigned char* const bbase = xt + ((wid + 1) * 34 + n + 1) * PXS + 16 * gq;
  const int stages = g.nchunk / CPS;
  request(0);
  for (int stage = 0; stage < stages; ++stage) {
    __syncthreads();                                 // every wave is through with the previous stage
    park();
    __syncthreads();
    if (stage + 1 < stages) request(stage + 1);
#pragma unroll
    for (int sub = 0; sub < CPS; ++sub) {
      int slot = sub * FRAGS;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          bf16x8 bf[HALVES];
#pragma unroll
          for (int p = 0; p < HALVES; ++p)
            bf[p] = *reinterpret_cast<const bf16x8*>(bbase + (dy * 34 + dx + 16 * p) * PXS + 64 * sub);
#pragma unroll
          for (int gr = 0; gr < GROUPS; ++gr) {
            if (!P::group_has(gr, dy)) continue;
            const bf16x8 a = wt[slot * 64 + lane];
            ++slot;
#pragma unroll
            for (int p = 0; p < HALVES; ++p)
              acc[gr][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[p], acc[gr][p], 0, 0, 0);
          }
        }
    }
  }
  // ---- epilogue: acc[gr][p][i] = column 4 gq + i of the group, input pixel (I0 + wid, J0 + 16 p + n) ----
  const int I = I0 + wid;
  if (I >= g.H) return;
  const int OW = g.W * S;
  // the values of this lane: S = 4: (rx, c) = 4 gq + i of 12 (gq = 3: none); S = 2: ry = gq >> 1, 4 (gq & 1) + i of 6
  const int first = S == 4 ? 4 * gq : 4 * (gq & 1);
  const int count = S == 4 ? (gq < 3 ? 4 : 0) : ((gq & 1) ? 2 : 4);
  float bc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bc[i] = bias ? bias[(first + i) % 3] : 0.f;
#pragma unroll
  for (int gr = 0; gr < GROUPS; ++gr) {
    const int ry = S == 4 ? gr : (gq >> 1);
    __bf16* const yrow = y + ((img * g.H * S + static_cast<long long>(I) * S + ry) * OW) * 3;
#pragma unroll
    for (int p = 0; p < HALVES; ++p) {
      const int J = J0 + 16 * p + n;
      if (J >= g.W || count == 0) continue;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[gr][p][i] + bc[i];
        if (g.activation == 1) v[i] = fmaxf(v[i], 0.f);
      }
      unsigned int* dst = reinterpret_cast<unsigned int*>(yrow + static_cast<long long>(J) * (3 * S) + first);
      dst[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
      if (count == 4) dst[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
    }
  }
}

// 0 = launched, -1 = not this shape, > 0 = error
int run_conv_up_phase(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t h, int64_t wd,
                      int64_t cin, int64_t cout, int kh, int kw, int stride, int activation, hipStream_t st) {
  static const bool off = [] { const char* e = std::getenv("TFC_CONV_UP_PHASE"); return e && e[0] == '0'; }();
  if (off || cout != 3 || cin % 32 || kh != kw) return -1;
  const bool k5 = kh == 5 && stride == 2, k9 = kh == 9 && stride == 4;
  if (!k5 && !k9) return -1;
  if (static_cast<double>(h) * wd * cin * 2 >= 2147483648.0) return -1;          // one buffer resource per image
  // (rows of 3 S bfloat16 per input pixel: 4-byte stores need S even — both are — and a 4-byte aligned y)
  if (reinterpret_cast<uintptr_t>(y) % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return -1;
  UpPhaseGeom g{};
  g.N = n; g.H = static_cast<int>(h); g.W = static_cast<int>(wd); g.Cin = static_cast<int>(cin);
  g.activation = activation; g.nchunk = static_cast<int>(cin / 32);
  const long long blocks = n * ((g.H + kPhaseRows - 1) / kPhaseRows) * ((g.W + 31) / 32);
  if (blocks >= (1ll << 31)) return -1;
  const int frags_per_chunk = k9 ? UpPhase<9, 4>::FRAGS : UpPhase<5, 2>::FRAGS;
  const int frags = g.nchunk * frags_per_chunk * 64;
  const int cps = k9 ? UpPhase<9, 4>::CPS : UpPhase<5, 2>::CPS, pxs = k9 ? UpPhase<9, 4>::PXS : UpPhase<5, 2>::PXS;
  if (g.nchunk % cps) return -1;
  const size_t lds = static_cast<size_t>(kPhaseTilePix) * pxs + static_cast<size_t>(cps) * frags_per_chunk * 1024;
  DevBuf wpk_local;
  DevView wpk;
  {
    const int rc = packed_weights(5, {kh, kw, cin, cout, stride}, static_cast<size_t>(frags) * 16, st, wpk_local, &wpk.p,
                                  [&](void* dst) {
      if (k9)
        hipLaunchKernelGGL((conv_up_phase_weights_kernel<9, 4>), dim3((frags + 255) / 256), dim3(256), 0, st, w,
                           static_cast<int>(cin), static_cast<int>(cout), static_cast<bf16x8*>(dst));
      else
        hipLaunchKernelGGL((conv_up_phase_weights_kernel<5, 2>), dim3((frags + 255) / 256), dim3(256), 0, st, w,
                           static_cast<int>(cin), static_cast<int>(cout), static_cast<bf16x8*>(dst));
      return 0;
    });
    if (rc) return rc;
  }
  KernelTimer timer("conv2d", st);
  if (k9) {
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up_phase_kernel<9, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((conv_up_phase_kernel<9, 4>), dim3(static_cast<unsigned>(blocks)), dim3(512), lds, st,
                       static_cast<const __bf16*>(x), wpk.as<bf16x8>(), bias, static_cast<__bf16*>(y), g);
  } else {
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up_phase_kernel<5, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((conv_up_phase_kernel<5, 2>), dim3(static_cast<unsigned>(blocks)), dim3(512), lds, st,
                       static_cast<const __bf16*>(x), wpk.as<bf16x8>(), bias, static_cast<__bf16*>(y), g);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

}  // namespace tfc
extern "C" int tfc_conv2d_drop_weights(uint64_t key);
namespace tfc {
int conv_entry(const void* x, const void* w, const float* bias, void* y, int dtype, int64_t n,
               int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh, int kw, int stride,
               int activation, int up, void* stream, bool out_f32 = false, const tfc_gdn_params* gdn = nullptr,
               int gdn_inverse = 0, int* gdn_fused = nullptr);
// ---------------------------------------------------------------------------
// float32 layers on the bfloat16 matrix cores (round 6).  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bfloat16 rate, and
// the float32 model steps are 90 % convolutions.  A float32 value is the sum of three bfloat16 values to its last bit
// (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8 significant bits), and the product of two such sums is
//     a b = a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1 + (terms below 2^-25 |a b|),
// every retained product exact in float32 and accumulated in float32 by the MFMA: the convolution of float32 tensors is
// ONE bfloat16 convolution with six times the input channels, [x1 | x1 | x1 | x2 | x2 | x3] against the kernel planes
// [w1 | w2 | w3 | w1 | w2 | w1], with float32 output — 6/16 of the float32 MFMA time on kernels that are also further
// along (second generation: 16-byte gathers 4 K steps ahead, packed weights through LDS).  The result differs from a
// float32 FMA chain by float32 rounding noise (a few 1e-7 relative; tests/test_signal_conv_gpu.py holds both to the float32
// definition).  TFC_CONV_F32=native keeps the float32 MFMA kernel; layers with <= 4 channels on one side keep it anyway.
// ---------------------------------------------------------------------------
__device__ inline void split3(float a, __bf16* p1, __bf16* p2, __bf16* p3) {
  const __bf16 a1 = static_cast<__bf16>(a);
  const float r1 = a - static_cast<float>(a1);
  const __bf16 a2 = static_cast<__bf16>(r1);
  const float r2 = r1 - static_cast<float>(a2);
  *p1 = a1; *p2 = a2; *p3 = static_cast<__bf16>(r2);
}
// x float32 [pixels, C] -> xs bfloat16 [pixels, 6 C]; a thread takes 8 channels of a pixel (C % 8 == 0)
__global__ void __launch_bounds__(256) conv_split_x_kernel(const float* x, long long pixels, int C, __bf16* xs) {
  const int per = C / 8;
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= pixels * per) return;
  const long long pix = i / per;
  const int c0 = static_cast<int>(i - pix * per) * 8;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(x + pix * C + c0), hi = *reinterpret_cast<const f32x4*>(x + pix * C + c0 + 4);
  bf16x8 p1, p2, p3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 a, b, c;
    split3(e < 4 ? lo[e] : hi[e - 4], &a, &b, &c);
    p1[e] = a; p2[e] = b; p3[e] = c;
  }
  __bf16* row = xs + pix * 6 * C + c0;
  *reinterpret_cast<bf16x8*>(row) = p1;
  *reinterpret_cast<bf16x8*>(row + C) = p1;
  *reinterpret_cast<bf16x8*>(row + 2 * C) = p1;
  *reinterpret_cast<bf16x8*>(row + 3 * C) = p2;
  *reinterpret_cast<bf16x8*>(row + 4 * C) = p2;
  *reinterpret_cast<bf16x8*>(row + 5 * C) = p3;
}
// w float32 [taps, C, Cout] -> w6 float32 [taps, 6 C, Cout] holding the planes [w1 | w2 | w3 | w1 | w2 | w1] as float32
// values (each exactly a bfloat16: the packing kernels' rounding leaves them as they are)
__global__ void __launch_bounds__(256) conv_split_w_kernel(const float* w, long long taps, int C, int Cout, float* w6) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= taps * C * Cout) return;
  const int co = static_cast<int>(i % Cout);
  const long long tc = i / Cout;
  const int c = static_cast<int>(tc % C);
  const long long t = tc / C;
  __bf16 a, b, d;
  split3(w[i], &a, &b, &d);
  const float p[3] = {static_cast<float>(a), static_cast<float>(b), static_cast<float>(d)};
  const int plane[6] = {0, 1, 2, 0, 1, 0};
  float* base = w6 + (t * 6 * C + c) * Cout + co;
#pragma unroll
  for (int q = 0; q < 6; ++q) base[static_cast<long long>(q) * C * Cout] = p[plane[q]];
}

// (read per call: the tests compare the two in one process)
inline bool conv_f32_split_enabled() {
  const char* e = std::getenv("TFC_CONV_F32");
  return !(e && std::strcmp(e, "native") == 0);
}


// Fused variant of the transposed convolution into few channels: 0 = launched, -1 = not this shape, > 0 = error.
int run_conv_up_fused(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t h, int64_t wd,
                      int64_t cin, int64_t cout, int kh, int kw, int stride, int activation, hipStream_t st) {
  if (cout > 4 || cin % 64 || stride < 2) return -1;
  auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
  UpFusedGeom g{};
  g.N = n; g.H = static_cast<int>(h); g.W = static_cast<int>(wd); g.Cin = static_cast<int>(cin);
  g.Cout = static_cast<int>(cout); g.kh = kh; g.kw = kw; g.s = stride; g.activation = activation;
  g.dmax_y = fdiv(kh - 1 - kh / 2, stride);
  g.dmax_x = fdiv(kw - 1 - kw / 2, stride);
  const int dmin_y = -fdiv((stride - 1) + kh / 2, stride), dmin_x = -fdiv((stride - 1) + kw / 2, stride);
  g.PH = 8 + g.dmax_y - dmin_y; g.PW = 32 + g.dmax_x - dmin_x;
  g.BXn = (g.W + 31) / 32; g.BYn = (g.H + 7) / 8;
  g.NC = kh * kw * g.Cout;
  // all kh * kw * Cout tap products of a patch pixel in one LDS row; kernels with more of them (bls2017's 9 x 9 x 3)
  // keep the implicit GEMM over output pixels (a pass per kernel-row residue was built and measured level with it)
  if (g.NC > kUpZStride) return -1;
  const int tiles = (g.PH * g.PW + 31) / 32;
  const size_t lds = static_cast<size_t>(tiles) * 32 * kUpZStride * 4 + static_cast<size_t>(cin / 16) * kUpColTiles * 64 * 16;
  if (lds > 160 * 1024 || n * g.BXn * g.BYn >= (1ll << 31)) return -1;
  DevBuf wpk;
  const int frags = static_cast<int>(cin / 16) * kUpColTiles * 64;
  TFC_HIP(wpk.alloc(static_cast<size_t>(frags) * 16, st));
  hipLaunchKernelGGL(conv_up_fused_weights_kernel, dim3((frags + 255) / 256), dim3(256), 0, st, w, kh, kw,
                     static_cast<int>(cin), static_cast<int>(cout), wpk.as<bf16x8>());
  KernelTimer timer("conv2d", st);
#define TFC_UP_FUSED_LAUNCH(...)                                                                              \
  do {                                                                                                       \
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_up_fused_kernel<__VA_ARGS__>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));         \
    hipLaunchKernelGGL((conv_up_fused_kernel<__VA_ARGS__>), dim3(static_cast<unsigned>(n * g.BXn * g.BYn)),   \
                       dim3(kUpThreads), lds, st, static_cast<const __bf16*>(x), wpk.as<bf16x8>(), bias,     \
                       static_cast<__bf16*>(y), g);                                                          \
  } while (0)
  if (kh == 5 && kw == 5 && stride == 2 && cout == 3) TFC_UP_FUSED_LAUNCH(5, 5, 2, 3);          // bmshj2018
  else if (kh == 9 && kw == 9 && stride == 4 && cout == 3) TFC_UP_FUSED_LAUNCH(9, 9, 4, 3);     // bls2017
  else TFC_UP_FUSED_LAUNCH(0, 0, 0, 0);
#undef TFC_UP_FUSED_LAUNCH
  TFC_HIP(hipGetLastError());
  return 0;
}

int conv_up_small_cout(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t h,
                       int64_t wd, int64_t cin, int64_t cout, int kh, int kw, int stride, int activation,
                       hipStream_t st) {
  {
    const int rc = run_conv_up_fused(x, w, bias, y, n, h, wd, cin, cout, kh, kw, stride, activation, st);
    if (rc >= 0) return rc;
  }
  const int zc = kh * kw * 4;
  DevBuf w1, z;
  TFC_HIP(w1.alloc(sizeof(float) * cin * zc, st));
  TFC_HIP(z.alloc(sizeof(float) * static_cast<size_t>(n) * h * wd * zc, st));
  const long long wtotal = static_cast<long long>(cin) * zc;
  hipLaunchKernelGGL(conv_up_weights_kernel, dim3(static_cast<unsigned>(ceil_div(wtotal, 256))), dim3(256), 0, st, w,
                     kh, kw, static_cast<int>(cin), static_cast<int>(cout), w1.as<float>());
  // the 1x1 product: a "down" convolution with a 1x1 kernel, stride 1, no bias, no activation
  if (int rc = conv_entry(x, w1.p, nullptr, z.p, 1, n, h, wd, cin, zc, 1, 1, 1, 0, 0, st, true)) return rc;
  const long long outs = n * h * stride * wd * stride;
  if (ceil_div(outs, 256) >= (1ll << 31)) return fail("tfc_conv2d_up: problem too large for one launch");
  KernelTimer timer("conv2d", st);
  hipLaunchKernelGGL(conv_up_gather_kernel, dim3(static_cast<unsigned>(ceil_div(outs, 256))), dim3(256), 0, st,
                     z.as<float>(), bias, static_cast<__bf16*>(y), static_cast<long long>(n), static_cast<int>(h),
                     static_cast<int>(wd), kh, kw, stride, static_cast<int>(cout), activation);
  TFC_HIP(hipGetLastError());
  return 0;
}

int conv_entry(const void* x, const void* w, const float* bias, void* y, int dtype, int64_t n,
               int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh, int kw, int stride,
               int activation, int up, void* stream, bool out_f32, const tfc_gdn_params* gdn, int gdn_inverse,
               int* gdn_fused) {
  // the weights key named for this call (tfc_conv2d_weights_key), for the packing sites below
  struct KeyScope {
    KeyScope() { t_weights_key = t_next_weights_key; t_next_weights_key = 0; }
    ~KeyScope() { t_weights_key = 0; }
  } key_scope;

  if (gdn_fused) *gdn_fused = 0;
  if (dtype != 0 && dtype != 1) return fail("tfc_conv2d: dtype must be 0 (float32) or 1 (bfloat16)");
  if (kh < 1 || kw < 1 || stride < 1 || cin < 1 || cout < 1) return fail("tfc_conv2d: bad geometry");
  if (!(cin % 16 == 0 || cin <= 4))
    return fail("tfc_conv2d: input channels must be a multiple of 16 or <= 4 (got %lld)",
                static_cast<long long>(cin));
  if (activation != 0 && activation != 1) return fail("tfc_conv2d: activation must be 0 (none) or 1 (relu)");
  if (n == 0 || h == 0 || wd == 0) return 0;
  if (dtype == 0 && conv_f32_split_enabled() && cin % 16 == 0 && cout % 4 == 0 && !gdn &&
      6 * cin * static_cast<int64_t>(kh) * kw < (int64_t{1} << 24)) {
    // float32 on the bfloat16 matrix cores: three planes per operand, six products (see conv_split_x_kernel).  The planes
    // of the input are 3x its bytes: images go through in chunks of ~2 GB of planes (bmshj2018's first 192 -> 192 layer at
    // 128 x 768x512 would need 29 GB at once), each chunk split and convolved before the next — the kernel's six planes
    // are made and packed once, under the caller's weights key or a key of this call's own.
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long taps = static_cast<long long>(kh) * kw;
    const long long pix_image = static_cast<long long>(h) * wd;
    const size_t plane_bytes_image = static_cast<size_t>(pix_image) * 6 * cin * sizeof(__bf16);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, static_cast<int64_t>((size_t{2} << 30) / std::max<size_t>(plane_bytes_image, 1))));
    DevBuf xs, w6;
    TFC_HIP(xs.alloc(plane_bytes_image * static_cast<size_t>(chunk), st));
    TFC_HIP(w6.alloc(static_cast<size_t>(taps) * 6 * cin * cout * sizeof(float), st));
    const long long wthreads = taps * cin * cout;
    if (ceil_div(pix_image * chunk * (cin / 8), 256) >= (1ll << 31)) return fail("tfc_conv2d: problem too large for one launch");
    hipLaunchKernelGGL(conv_split_w_kernel, dim3(static_cast<unsigned>(ceil_div(wthreads, 256))), dim3(256), 0, st,
                       static_cast<const float*>(w), taps, static_cast<int>(cin), static_cast<int>(cout), w6.as<float>());
    static std::atomic<unsigned long long> own_keys{0};
    const unsigned long long caller_key = t_weights_key;
    const unsigned long long key = caller_key ? (caller_key ^ 0x5bf1600000000000ull)
                                              : (0xf32c000000000000ull | own_keys.fetch_add(1, std::memory_order_relaxed));
    const long long out_image = (up ? static_cast<long long>(h) * stride * wd * stride
                                    : static_cast<long long>((h + stride - 1) / stride) * ((wd + stride - 1) / stride)) * cout;
    int rc = 0;
    for (int64_t n0 = 0; n0 < n && rc == 0; n0 += chunk) {
      const int64_t nc = std::min<int64_t>(chunk, n - n0);
      const long long pixels = pix_image * nc;
      {
        KernelTimer timer("conv2d", st);
        hipLaunchKernelGGL(conv_split_x_kernel, dim3(static_cast<unsigned>(ceil_div(pixels * (cin / 8), 256))), dim3(256), 0, st,
                           static_cast<const float*>(x) + n0 * pix_image * cin, pixels, static_cast<int>(cin), xs.as<__bf16>());
      }
      t_next_weights_key = key;
      rc = conv_entry(xs.p, w6.p, bias, static_cast<float*>(y) + n0 * out_image, 1, nc, h, wd, 6 * cin, cout, kh, kw, stride,
                      activation, up, stream, true);
    }
    TFC_HIP(hipGetLastError());
    if (!caller_key) (void)tfc_conv2d_drop_weights(key);
    return rc;
  }
#ifndef TFC_CONV_NO_UP_GATHER
  // (up to 128 product columns, i.e. one column group: a 9x9 stride-4 kernel has 324 and measured the same
  // or slower this way — 0.19 against 0.16 ms at batch 64 — so it keeps the implicit GEMM over output pixels)
  if (up && dtype == 1 && cout == 3 && !out_f32) {
    // the synthesis transforms' last layer as a 3x3 convolution over the input grid: conv_up_phase_kernel
    const int rc = run_conv_up_phase(x, static_cast<const float*>(w), bias, y, n, h, wd, cin, cout, kh, kw, stride,
                                     activation, static_cast<hipStream_t>(stream));
    if (rc >= 0) return rc;
  }
  if (up && dtype == 1 && cout <= 4 && cin % 64 == 0 && stride >= 2 && !out_f32) {
    const int rc = run_conv_up_fused(x, static_cast<const float*>(w), bias, y, n, h, wd, cin, cout, kh, kw, stride,
                                     activation, static_cast<hipStream_t>(stream));
    if (rc >= 0) return rc;
  }
  if (up && dtype == 1 && cout <= 4 && cin % 16 == 0 && stride >= 2 && kh * kw * 4 <= 128 && !out_f32)
    return conv_up_small_cout(x, static_cast<const float*>(w), bias, y, n, h, wd, cin, cout, kh, kw, stride, activation,
                              static_cast<hipStream_t>(stream));
#endif
  if (!up && dtype == 1 && cin <= 4 && !out_f32) {
    // the image-side layer from the image's own rows (conv_image_direct_kernel), with its GDN where it is the 5x5 / 2 one
    const bool with_gdn = gdn && gdn_fused && !gdn_inverse && activation == 0 && kh == 5;
    const int rc = run_conv_image_direct(x, static_cast<const float*>(w), bias, y, n, h, wd, cin, cout, kh, kw, stride,
                                         activation, with_gdn ? gdn : nullptr, static_cast<hipStream_t>(stream));
    if (rc == 0 && with_gdn) {
      const_cast<tfc_gdn_params*>(gdn)->image.touch(static_cast<hipStream_t>(stream));
      *gdn_fused = 1;
    }
    if (rc >= 0) return rc;
  }
  if (!up && dtype == 1 && cin <= 4 && !out_f32) {
    const int rc = run_conv_image(x, static_cast<const float*>(w), bias, y, n, h, wd, cin, cout, kh, kw, stride, activation,
                                  static_cast<hipStream_t>(stream));
    if (rc >= 0) return rc;
  }
  ConvGeom c{};
  c.xcd = xcd_blocks();
  PackGeom g{};
  g.kh = kh; g.kw = kw; g.Cin_real = static_cast<int>(cin); g.Cout = static_cast<int>(cout);
  g.up = up; g.su = up ? stride : 1;
  c.N = n; c.H = static_cast<int>(h); c.W = static_cast<int>(wd);
  c.Cout = static_cast<int>(cout);
  c.activation = activation;
  c.out_f32 = out_f32 ? 1 : 0;        // only reached with the second-generation bf16 kernel (Cin % 16 == 0, Cout % 4 == 0)
  if (!up) {
    c.sd = stride; c.su = 1;
    c.Uy = kh; c.Ux = kw;
    c.py0 = kh / 2; c.px0 = kw / 2;
    c.OHq = static_cast<int>((h + stride - 1) / stride);
    c.OWq = static_cast<int>((wd + stride - 1) / stride);
  } else {
    // y[q*s + phi] = sum_d x[q - d] w[phi + d*s + k/2]; d in [dmin, dmax] over all phases
    auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    const int s = stride;
    g.dmax_y = fdiv(kh - 1 - kh / 2, s);
    g.dmax_x = fdiv(kw - 1 - kw / 2, s);
    const int dmin_y = -fdiv((s - 1) + kh / 2, s), dmin_x = -fdiv((s - 1) + kw / 2, s);
    c.sd = 1; c.su = s;
    c.Uy = g.dmax_y - dmin_y + 1; c.Ux = g.dmax_x - dmin_x + 1;
    c.py0 = g.dmax_y; c.px0 = g.dmax_x;
    c.OHq = c.H; c.OWq = c.W;
  }
  g.Uy = c.Uy; g.Ux = c.Ux;
  fast_div_setup(static_cast<unsigned int>(c.OWq), &c.owq_mul, &c.owq_sh);
  fast_div_setup(static_cast<unsigned int>(c.OHq), &c.ohq_mul, &c.ohq_sh);
  c.pix32 = static_cast<long long>(n) * c.OHq * c.OWq + 512 < (1ll << 31) ? 1 : 0;      // (+ the last workgroup's overhang)
  c.cols = c.su * c.su * c.Cout;
  c.OH = c.OHq * c.su; c.OW = c.OWq * c.su;
  if (cin <= 4) {
    c.small_cin = 1;
    c.Cin = 4;
    c.kw4 = (c.Ux * 4 + 15) / 16;
    c.ksteps = c.Uy * c.kw4;
    c.Hp = (c.OHq - 1) * c.sd + c.Uy;
    c.Wp = std::max((c.OWq - 1) * c.sd + c.Ux, c.px0 + c.W);
    c.Hp = std::max(c.Hp, c.py0 + c.H);
  } else {
    c.small_cin = 0;
    c.Cin = static_cast<int>(cin);
    c.Hp = c.H; c.Wp = c.W;
    c.ksteps = c.Uy * c.Ux * (c.Cin / 16);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* wf = static_cast<const float*>(w);
  if (gdn && gdn_fused && dtype == 1 && gdn->dtype == 1 && gdn->channels == cout && activation == 0 && !out_f32) {
    // GDN / IGDN as the activation: where the third-generation kernel takes the layer (it says so through *gdn_fused)
    c.gdn = gdn_inverse ? 2 : 1;
    c.gdn_image = gdn->image.p;
    const_cast<tfc_gdn_params*>(gdn)->image.touch(st);
  }
  return dtype == 1 ? run_conv<__bf16>(x, wf, bias, y, c, g, st, gdn_fused) : run_conv<float>(x, wf, bias, y, c, g, st, gdn_fused);
}

}  // namespace tfc

extern "C" void tfc_conv2d_weights_key(uint64_t key) { tfc::t_next_weights_key = key; }

extern "C" int tfc_conv2d_drop_weights(uint64_t key) {
  tfc::WeightsCache& c = tfc::WeightsCache::get();
  std::lock_guard<std::mutex> lock(c.mu);
  // A kernel on any stream that used an entry may still read its fragments.  The memory goes back in STREAM order: the
  // stream the entry was made on waits for an event on every other stream that read it, and the block returns to the
  // library's cache behind that (DevBuf::release: reusable on that stream at once, on another once its event has
  // completed).  Nothing here holds the host — model.eval() drops one key per layer.  Only when a reader's stream no
  // longer takes an event (destroyed by its owner) is that entry's device drained instead.
  int home = 0;
  (void)hipGetDevice(&home);
  for (auto it = c.entries.begin(); it != c.entries.end();) {
    // (and the six-plane kernel a float32 layer packed under the key derived from it: conv_entry's bf16 x 6 path)
    if (it->first.key != key && it->first.key != (key ^ 0x5bf1600000000000ull)) { ++it; continue; }
    tfc::WeightsCache::Entry& e = it->second;
    if (it->first.dev != home) (void)hipSetDevice(it->first.dev);
    bool ordered = true;
    for (hipStream_t s : e.users) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ordered = false; break; }
      if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(e.made_on, ev, 0) != hipSuccess) ordered = false;
      (void)hipEventDestroy(ev);
      if (!ordered) break;
    }
    if (!ordered) {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
    }
    e.buf.touch(e.made_on);
    (void)hipEventDestroy(e.ready);
    it = c.entries.erase(it);          // ~DevBuf: release() in the order of made_on
    if (it == c.entries.end() || it->first.dev != home) (void)hipSetDevice(home);
  }
  (void)hipSetDevice(home);
  return 0;
}

extern "C" int tfc_conv2d_down(const void* x, const void* w, const float* bias, void* y, int dtype,
                               int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh,
                               int kw, int stride, int activation, void* stream) {
  return tfc::conv_entry(x, w, bias, y, dtype, n, h, wd, cin, cout, kh, kw, stride, activation, 0, stream);
}

extern "C" int tfc_conv2d_gdn(const void* x, const void* w, const float* bias, void* y, int dtype,
                              int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh,
                              int kw, int stride, int up, const tfc_gdn_params* gdn, int inverse, int* fused,
                              void* stream) {
  if (!gdn || !fused) {
    tfc::t_next_weights_key = 0;       // the key named for THIS call must not reach another layer's
    return tfc::fail("tfc_conv2d_gdn: gdn and fused must not be null");
  }
  return tfc::conv_entry(x, w, bias, y, dtype, n, h, wd, cin, cout, kh, kw, stride, 0, up, stream, false, gdn, inverse,
                         fused);
}

extern "C" int tfc_conv2d_up(const void* x, const void* w, const float* bias, void* y, int dtype,
                             int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh,
                             int kw, int stride, int activation, void* stream) {
  return tfc::conv_entry(x, w, bias, y, dtype, n, h, wd, cin, cout, kh, kw, stride, activation, 1, stream);
}
