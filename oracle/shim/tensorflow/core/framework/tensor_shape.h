// Build shim (test infrastructure) for tensorflow::TensorShape. Not product code.
#pragma once
#include <cstdint>
#include <vector>
namespace tensorflow {
class TensorShape {
 public:
  TensorShape() = default;
  explicit TensorShape(std::vector<int64_t> d) : d_(std::move(d)) {}
  int dims() const { return static_cast<int>(d_.size()); }
  int64_t dim_size(int i) const { return d_[i]; }
  void set_dim(int i, int64_t v) { d_[i] = v; }
  int64_t num_elements() const {
    int64_t n = 1;
    for (int64_t v : d_) n *= v;
    return n;
  }
 private:
  std::vector<int64_t> d_;
};
struct TensorShapeUtils {
  static bool IsVectorOrHigher(const TensorShape& s) { return s.dims() >= 1; }
};
}  // namespace tensorflow
