// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// Restatement of the reference's StochasticRound CPU kernel
// (tensorflow_compression/cc/kernels/quantization_kernels.cc:47-96):
//   * the generator state is 4 x u64, filled from the int32 seed tensor by the C++
//     standard's seed_seq::generate ([rand.util.seedseq], restated here by hand so that the
//     product's use of std::seed_seq is checked against something independent), 8 words of 32
//     bits stored little-endian (`:71-74`);
//   * one xoshiro256+ draw per element in flat order (`:35-45`), top 24 bits -> [0, 1) (`:88`);
//   * number = float(x) / step, out = floor(number), +1 when draw < number - floor (`:83-92`).
// Parity is pinned by tests/golden/stochastic_round.npz, generated with the reference file
// itself compiled behind oracle/shim (oracle/_ref/libtfc_ref.so).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace tfc_oracle {

inline void seed_words(const int32_t* seed, int64_t s, uint32_t* b, int64_t n) {
  auto mix = [](uint32_t x) { return x ^ (x >> 27); };
  for (int64_t i = 0; i < n; ++i) b[i] = 0x8b8b8b8bu;
  const int64_t t = n >= 623 ? 11 : n >= 68 ? 7 : n >= 39 ? 5 : n >= 7 ? 3 : (n - 1) / 2;
  const int64_t p = (n - t) / 2, q = p + t, m = std::max<int64_t>(s + 1, n);
  for (int64_t k = 0; k < m; ++k) {
    const uint32_t r1 = 1664525u * mix(b[k % n] ^ b[(k + p) % n] ^ b[(k + n - 1) % n]);
    uint32_t r2 = r1;
    if (k == 0) r2 += static_cast<uint32_t>(s);
    else if (k <= s) r2 += static_cast<uint32_t>(k % n) + static_cast<uint32_t>(seed[k - 1]);
    else r2 += static_cast<uint32_t>(k % n);
    b[(k + p) % n] += r1;
    b[(k + q) % n] += r2;
    b[k % n] = r2;
  }
  for (int64_t k = m; k < m + n; ++k) {
    const uint32_t r3 = 1566083941u * mix(b[k % n] + b[(k + p) % n] + b[(k + n - 1) % n]);
    const uint32_t r4 = r3 - static_cast<uint32_t>(k % n);
    b[(k + p) % n] ^= r3;
    b[(k + q) % n] ^= r4;
    b[k % n] = r4;
  }
}

struct Xoshiro256Plus {
  uint64_t s[4];
  uint64_t next() {
    const uint64_t out = s[0] + s[3];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = (s[3] << 45) | (s[3] >> 19);
    return out;
  }
};

// `x` already widened to float by the caller (bfloat16 / half -> float is exact).
inline void stochastic_round(const float* x, int64_t n, float step, const int32_t* seed, int64_t seed_len,
                             int32_t* out) {
  uint32_t w[8];
  seed_words(seed, seed_len, w, 8);
  Xoshiro256Plus g;
  for (int i = 0; i < 4; ++i) g.s[i] = static_cast<uint64_t>(w[2 * i]) | (static_cast<uint64_t>(w[2 * i + 1]) << 32);
  for (int64_t i = 0; i < n; ++i) {
    const float number = x[i] / step;
    const float integral = std::floor(number);
    int32_t v = static_cast<int32_t>(integral);
    const float frac = number - integral;
    const float draw = static_cast<float>(g.next() >> 40) * 0x1.0p-24f;
    if (draw < frac) ++v;
    out[i] = v;
  }
}

}  // namespace tfc_oracle
