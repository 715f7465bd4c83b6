// Build shim (test infrastructure) for tensorflow::TensorShape. Not product code.
#pragma once
#include <cstdint>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>
#include "absl/status/status.h"
namespace tensorflow {
class TensorShape {
 public:
  TensorShape() = default;
  explicit TensorShape(std::vector<int64_t> d) : d_(std::move(d)) {}
  int dims() const { return static_cast<int>(d_.size()); }
  int64_t dim_size(int i) const { return d_[i]; }
  void set_dim(int i, int64_t v) { d_[i] = v; }
  void AddDim(int64_t v) { d_.push_back(v); }
  void AppendShape(const TensorShape& o) { d_.insert(d_.end(), o.d_.begin(), o.d_.end()); }
  int64_t num_elements() const {
    int64_t n = 1;
    for (int64_t v : d_) n *= v;
    return n;
  }
  std::string DebugString() const {
    std::ostringstream os;
    os << *this;
    return os.str();
  }
  friend bool operator==(const TensorShape& a, const TensorShape& b) { return a.d_ == b.d_; }
  friend bool operator!=(const TensorShape& a, const TensorShape& b) { return !(a == b); }
  friend std::ostream& operator<<(std::ostream& os, const TensorShape& s) {
    os << "[";
    for (int i = 0; i < s.dims(); ++i) os << (i ? "," : "") << s.d_[i];
    return os << "]";
  }
 private:
  std::vector<int64_t> d_;
};
struct TensorShapeUtils {
  static bool IsVectorOrHigher(const TensorShape& s) { return s.dims() >= 1; }
  static bool IsScalar(const TensorShape& s) { return s.dims() == 0; }
  static bool IsVector(const TensorShape& s) { return s.dims() == 1; }
  static bool IsMatrix(const TensorShape& s) { return s.dims() == 2; }
  template <class View>
  static auto MakeShape(const View& dims, TensorShape* out) {
    *out = TensorShape();
    for (int64_t i = 0; i < dims.size(); ++i) out->AddDim(dims(i));
    return absl::OkStatus();
  }
  static bool StartsWith(const TensorShape& s, const TensorShape& prefix) {
    if (s.dims() < prefix.dims()) return false;
    for (int i = 0; i < prefix.dims(); ++i)
      if (s.dim_size(i) != prefix.dim_size(i)) return false;
    return true;
  }
};
}  // namespace tensorflow
