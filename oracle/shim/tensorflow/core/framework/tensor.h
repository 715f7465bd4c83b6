// Build shim (test infrastructure) for tensorflow::Tensor: a typed view of caller memory.
// Not product code.
#pragma once
#include <cstdint>
#include "tensorflow/core/framework/tensor_shape.h"
namespace tensorflow {
class Tensor {
 public:
  template <class T>
  struct Flat {
    T* p;
    int64_t n;
    T& operator()(int64_t i) const { return p[i]; }
    int64_t size() const { return n; }
    T* data() const { return p; }
  };
  template <class T>
  struct Scalar {
    T* p;
    T& operator()() const { return *p; }
  };
  template <class T>
  struct Matrix {  // all leading dimensions merged
    T* p;
    int64_t rows, cols;
    int64_t dimension(int i) const { return i == 0 ? rows : cols; }
    T& operator()(int64_t i, int64_t j) const { return p[i * cols + j]; }
  };
  Tensor() = default;
  Tensor(void* data, TensorShape shape) : data_(data), shape_(std::move(shape)) {}
  template <class T> Flat<T> flat() const { return {static_cast<T*>(data_), shape_.num_elements()}; }
  template <class T> Scalar<T> scalar() const { return {static_cast<T*>(data_)}; }
  template <class T, int N> Matrix<T> flat_inner_dims() const {
    static_assert(N == 2, "shim: only the matrix view is provided");
    const int64_t cols = shape_.dim_size(shape_.dims() - 1);
    return {static_cast<T*>(data_), cols ? shape_.num_elements() / cols : 0, cols};
  }
  int dims() const { return shape_.dims(); }
  const TensorShape& shape() const { return shape_; }

 private:
  void* data_ = nullptr;
  TensorShape shape_;
};
}  // namespace tensorflow
