// The permutation libstdc++'s std::sort produces, restated so that it can run on the device.
//
// Why: PmfToQuantizedCdf (cc/kernels/pmf_to_cdf_kernels.cc:174-205) orders its candidates with
// std::sort and the FIRST of several equal penalties is the one adjusted.  Equal penalties are the
// normal case for symmetric tables (pmf[i] == pmf[n-1-i]), so which element of a tied pair moves
// depends on how an unstable sort happened to arrange them; the reference's Linux builds use
// libstdc++, and tables must be identical on both sides of a bit stream.  The algorithm restated here
// is the published one of libstdc++'s <bits/stl_algo.h> / <bits/stl_heap.h> (unchanged since GCC 4):
// introsort = median-of-three quicksort down to ranges of 16 with a depth limit of 2*floor(log2 n)
// (heap sort beyond it), then one insertion-sort pass.  Only the sequence of comparisons and moves
// matters; the recursion on the right-hand part is replaced by an explicit stack (the parts are
// disjoint, so the order in which they are finished does not change the result).
//
// tests/test_sort_order_cpu.py compiles this header for the host and checks it against std::sort
// itself on tie-heavy and adversarial inputs.
#pragma once

#ifdef __HIPCC__
#define TFC_HD __host__ __device__
#else
#define TFC_HD
#endif

namespace tfc {

// Items are (key, tag) pairs held in two parallel arrays; BEFORE(a, b) on keys is the strict weak order.
template <class Before>
struct SortOrder {
  double* key;
  unsigned int* tag;
  Before before;

  struct Item { double k; unsigned int t; };
  TFC_HD Item get(int i) const { return {key[i], tag[i]}; }
  TFC_HD void put(int i, const Item& v) const { key[i] = v.k; tag[i] = v.t; }
  TFC_HD void move(int to, int from) const { key[to] = key[from]; tag[to] = tag[from]; }
  TFC_HD void swap(int a, int b) const {
    const Item x = get(a);
    move(a, b);
    put(b, x);
  }
  TFC_HD bool lt(int a, int b) const { return before(key[a], key[b]); }

  // -- insertion sorts ---------------------------------------------------------------------------
  TFC_HD void linear_insert(int last) const {  // the element is known not to precede the first one
    const Item v = get(last);
    int next = last - 1;
    while (before(v.k, key[next])) {
      move(last, next);
      last = next;
      --next;
    }
    put(last, v);
  }
  TFC_HD void insertion_sort(int first, int last) const {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      if (lt(i, first)) {
        const Item v = get(i);
        for (int j = i; j > first; --j) move(j, j - 1);
        put(first, v);
      } else {
        linear_insert(i);
      }
    }
  }
  TFC_HD void final_insertion_sort(int first, int last) const {
    if (last - first > 16) {
      insertion_sort(first, first + 16);
      for (int i = first + 16; i != last; ++i) linear_insert(i);
    } else {
      insertion_sort(first, last);
    }
  }

  // -- heap sort (taken when the depth limit runs out) -------------------------------------------
  TFC_HD void push_heap(int first, int hole, int top, const Item& v) const {
    int parent = (hole - 1) / 2;
    while (hole > top && before(key[first + parent], v.k)) {
      move(first + hole, first + parent);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    put(first + hole, v);
  }
  TFC_HD void adjust_heap(int first, int hole, int len, const Item& v) const {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (lt(first + child, first + child - 1)) --child;
      move(first + hole, first + child);
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      move(first + hole, first + child - 1);
      hole = child - 1;
    }
    push_heap(first, hole, top, v);
  }
  TFC_HD void heap_sort(int first, int last) const {
    const int len = last - first;
    if (len >= 2) {
      for (int parent = (len - 2) / 2;; --parent) {
        const Item v = get(first + parent);
        adjust_heap(first, parent, len, v);
        if (parent == 0) break;
      }
    }
    while (last - first > 1) {
      --last;
      const Item v = get(last);
      move(last, first);
      adjust_heap(first, 0, last - first, v);
    }
  }

  // -- quicksort part ----------------------------------------------------------------------------
  TFC_HD void median_to_first(int result, int a, int b, int c) const {
    if (lt(a, b)) {
      if (lt(b, c)) swap(result, b);
      else if (lt(a, c)) swap(result, c);
      else swap(result, a);
    } else if (lt(a, c)) {
      swap(result, a);
    } else if (lt(b, c)) {
      swap(result, c);
    } else {
      swap(result, b);
    }
  }
  TFC_HD int partition(int first, int last, int pivot) const {
    while (true) {
      while (lt(first, pivot)) ++first;
      --last;
      while (lt(pivot, last)) --last;
      if (!(first < last)) return first;
      swap(first, last);
      ++first;
    }
  }
  TFC_HD int partition_pivot(int first, int last) const {
    const int mid = first + (last - first) / 2;
    median_to_first(first, first + 1, mid, last - 1);
    return partition(first + 1, last, first);
  }

  TFC_HD void sort(int n) const {
    if (n <= 0) return;
    int lg = 0;
    while ((n >> (lg + 1)) != 0) ++lg;
    struct Range { int first, last, depth; };
    Range stack[64];
    int top = 0;
    stack[top++] = {0, n, 2 * lg};
    while (top > 0) {
      Range r = stack[--top];
      while (r.last - r.first > 16) {
        if (r.depth == 0) {
          heap_sort(r.first, r.last);
          break;
        }
        --r.depth;
        const int cut = partition_pivot(r.first, r.last);
        stack[top++] = {cut, r.last, r.depth};
        r.last = cut;
      }
    }
    final_insertion_sort(0, n);
  }
};

struct KeyAscending { TFC_HD bool operator()(double a, double b) const { return a < b; } };
struct KeyDescending { TFC_HD bool operator()(double a, double b) const { return a > b; } };

}  // namespace tfc
