// Test helper (host build of compression_amd/csrc/sort_order.h): sorts (key, tag) pairs with the
// restated algorithm and with std::sort itself and reports whether the permutations agree.
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

#include "../../compression_amd/csrc/sort_order.h"

namespace {
struct Pair {
  double key;
  unsigned int tag;
  friend bool operator<(const Pair& a, const Pair& b) { return a.key < b.key; }
  friend bool operator>(const Pair& a, const Pair& b) { return a.key > b.key; }
};
}  // namespace

// keys[n]; descending != 0 sorts with std::greater.  Writes both tag orders; returns 1 when equal.
extern "C" int sort_order_check(const double* keys, int n, int descending, uint32_t* mine, uint32_t* theirs) {
  std::vector<Pair> ref(n);
  std::vector<double> k(keys, keys + n);
  for (int i = 0; i < n; ++i) {
    ref[i] = {keys[i], static_cast<unsigned int>(i)};
    mine[i] = static_cast<uint32_t>(i);
  }
  if (descending) {
    std::sort(ref.begin(), ref.end(), std::greater<Pair>());
    tfc::SortOrder<tfc::KeyDescending>{k.data(), mine, {}}.sort(n);
  } else {
    std::sort(ref.begin(), ref.end());
    tfc::SortOrder<tfc::KeyAscending>{k.data(), mine, {}}.sort(n);
  }
  int same = 1;
  for (int i = 0; i < n; ++i) {
    theirs[i] = ref[i].tag;
    if (theirs[i] != mine[i] || k[i] != ref[i].key) same = 0;
  }
  return same;
}
