// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// Restatement of PmfToQuantizedCdf's per-row normalisation
// (cc/kernels/pmf_to_cdf_kernels.cc:104-208).  The reference's result depends
// on the comparison order of libstdc++'s std::sort for tied penalties, so this
// restatement deliberately drives the same std::sort / std::find_if /
// std::rotate sequence with the same comparator outcomes; tie-exact tables are
// only claimed for builds against the same libstdc++ (the reference makes the
// same caveat, cc/ops/pmf_to_cdf_ops.cc:45-49).
//
// Parity status: PINNED — oracle/Makefile compiles the reference's own kernel file
// (cc/kernels/pmf_to_cdf_kernels.cc) verbatim behind a small OpKernel shim (oracle/shim/tensorflow)
// into oracle/_ref; tests/test_oracle.py checks restatement == that op on tie-heavy inputs and both ==
// tests/golden/pmf_to_cdf.npz (generated from it by oracle/make_golden.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <numeric>
#include <vector>

namespace tfc_oracle {

struct Shrink {           // PenaltyItem, :104-131
  int32_t* slot;
  double mass;
  double cost;
  Shrink(int32_t* s, double m) : slot(s), mass(m) { cost = next(); }
  double next() const {
    if (*slot <= 1) return std::numeric_limits<double>::infinity();
    return mass * (std::log2(*slot) - std::log2(*slot - 1));
  }
  void apply() { --*slot; cost = next(); }
  friend bool operator<(const Shrink& a, const Shrink& b) { return a.cost < b.cost; }
};

struct Grow {             // GainItem, :133-157
  int32_t* slot;
  double mass;
  double gain;
  Grow(int32_t* s, double m) : slot(s), mass(m) { gain = next(); }
  double next() const {
    if (*slot < 1) return -std::numeric_limits<double>::infinity();
    return mass * (std::log2(*slot + 1) - std::log2(*slot));
  }
  void apply() { ++*slot; gain = next(); }
  friend bool operator>(const Grow& a, const Grow& b) { return a.gain > b.gain; }
};

// pmf[n] -> cdf[n + 1] with cdf[0] = 0 and cdf[n] = 1 << precision.
inline void pmf_row_to_cdf(const float* pmf, int64_t n, int precision, int32_t* cdf) {
  const int32_t total = 1 << precision;
  int32_t* q = cdf + 1;
  cdf[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t v = static_cast<int32_t>(std::rint(pmf[i] * total));   // float * int -> float, :165
    q[i] = std::max(v, 1);
  }
  int32_t sum = std::accumulate(q, q + n, 0);
  if (sum > total) {
    std::vector<Shrink> heap;
    heap.reserve(n);
    for (int64_t i = 0; i < n; ++i) heap.emplace_back(&q[i], pmf[i]);
    std::sort(heap.begin(), heap.end());
    while (sum-- > total) {
      heap[0].apply();
      auto it = std::find_if(std::next(heap.begin()), heap.end(),
                             [&heap](const Shrink& r) { return heap[0] < r; });
      std::rotate(heap.begin(), std::next(heap.begin()), it);
    }
  } else if (sum < total) {
    std::vector<Grow> heap;
    heap.reserve(n);
    for (int64_t i = 0; i < n; ++i) heap.emplace_back(&q[i], pmf[i]);
    std::sort(heap.begin(), heap.end(), std::greater<Grow>());
    while (sum++ < total) {
      heap[0].apply();
      auto it = std::find_if(std::next(heap.begin()), heap.end(),
                             [&heap](const Grow& r) { return heap[0] > r; });
      std::rotate(heap.begin(), std::next(heap.begin()), it);
    }
  }
  std::partial_sum(q, q + n, q);
}

}  // namespace tfc_oracle
