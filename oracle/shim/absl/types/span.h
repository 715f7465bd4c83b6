// Build shim (test infrastructure) for absl::Span (read-mostly subset). Not product code.
#pragma once
#include <cstddef>
#include <type_traits>
#include <vector>
namespace absl {
template <typename T>
class Span {
 public:
  using size_type = std::size_t;
  using value_type = std::remove_cv_t<T>;
  Span() : p_(nullptr), n_(0) {}
  Span(T* p, size_type n) : p_(p), n_(n) {}
  template <size_type N>
  Span(T (&a)[N]) : p_(a), n_(N) {}  // NOLINT
  template <typename V, typename = std::enable_if_t<std::is_same<
                            std::remove_cv_t<T>, typename V::value_type>::value>>
  Span(const V& v) : p_(v.data()), n_(v.size()) {}  // NOLINT
  T* data() const { return p_; }
  size_type size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* begin() const { return p_; }
  T* end() const { return p_ + n_; }
  T& operator[](size_type i) const { return p_[i]; }
  T& back() const { return p_[n_ - 1]; }
  Span subspan(size_type pos) const { return Span(p_ + pos, n_ - pos); }
 private:
  T* p_;
  size_type n_;
};
}  // namespace absl
