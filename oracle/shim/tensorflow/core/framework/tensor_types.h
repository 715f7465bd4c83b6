// Build shim (test infrastructure) for the Eigen views TTypes<T> names: plain pointer views.
// Not product code.
#pragma once
#include <cstdint>
#include <type_traits>
namespace tensorflow {
namespace shim {
template <class T>
struct FlatView {
  T* p = nullptr;
  int64_t n = 0;
  FlatView() = default;
  FlatView(T* ptr, int64_t size) : p(ptr), n(size) {}
  template <class U, class = std::enable_if_t<std::is_same<const U, T>::value>>
  FlatView(const FlatView<U>& o) : p(o.p), n(o.n) {}  // NOLINT
  T& operator()(int64_t i) const { return p[i]; }
  int64_t size() const { return n; }
  T* data() const { return p; }
  void setConstant(const std::remove_const_t<T>& v) const {
    for (int64_t i = 0; i < n; ++i) p[i] = v;
  }
};
template <class T>
struct ScalarView {
  T* p;
  T& operator()() const { return *p; }
};
template <class T>
struct MatrixView {
  T* p = nullptr;
  int64_t rows = 0, cols = 0;
  MatrixView() = default;
  MatrixView(T* ptr, int64_t r, int64_t c) : p(ptr), rows(r), cols(c) {}
  template <class U, class = std::enable_if_t<std::is_same<const U, T>::value>>
  MatrixView(const MatrixView<U>& o) : p(o.p), rows(o.rows), cols(o.cols) {}  // NOLINT
  int64_t dimension(int i) const { return i == 0 ? rows : cols; }
  int64_t size() const { return rows * cols; }
  T* data() const { return p; }
  T& operator()(int64_t i, int64_t j) const { return p[i * cols + j]; }
};
}  // namespace shim
template <class T, int NDIMS = 1>
struct TTypes {
  using Flat = shim::FlatView<T>;
  using ConstFlat = shim::FlatView<const T>;
  using Vec = shim::FlatView<T>;
  using ConstVec = shim::FlatView<const T>;
  using Matrix = shim::MatrixView<T>;
  using ConstMatrix = shim::MatrixView<const T>;
};
}  // namespace tensorflow
