// Build shim (test infrastructure) for tensorflow::Tensor: a view of caller memory, or storage created
// on first typed access (allocate_output does not know the element type here).  Copies share storage,
// as in TensorFlow.  Not product code.
#pragma once
#include <cstdint>
#include <memory>
#include <typeindex>
#include <utility>
#include "tensorflow/core/framework/tensor_shape.h"
#include "tensorflow/core/framework/tensor_types.h"
namespace tensorflow {
class Tensor {
 public:
  Tensor() : buf_(std::make_shared<Buffer>()) {}
  explicit Tensor(TensorShape shape) : buf_(std::make_shared<Buffer>()), shape_(std::move(shape)) {}
  // view of caller memory
  template <class T>
  Tensor(T* data, TensorShape shape) : buf_(std::make_shared<Buffer>()), shape_(std::move(shape)) {
    buf_->data = const_cast<std::remove_const_t<T>*>(data);
    buf_->type = std::type_index(typeid(std::remove_const_t<T>));
  }
  Tensor(void* data, TensorShape shape) : buf_(std::make_shared<Buffer>()), shape_(std::move(shape)) {
    buf_->data = data;
  }

  template <class T> typename TTypes<T>::Flat flat() { return {typed<T>(), shape_.num_elements()}; }
  template <class T> typename TTypes<T>::ConstFlat flat() const { return {typed<T>(), shape_.num_elements()}; }
  template <class T> shim::ScalarView<const T> scalar() const { return {typed<T>()}; }
  template <class T> shim::ScalarView<T> scalar() { return {typed<T>()}; }
  template <class T> typename TTypes<T>::Vec vec() { return {typed<T>(), shape_.num_elements()}; }
  template <class T> typename TTypes<T>::ConstVec vec() const { return {typed<T>(), shape_.num_elements()}; }
  int64_t dim_size(int i) const { return shape_.dim_size(i); }
  template <class T> typename TTypes<T>::Matrix matrix() { return {typed<T>(), shape_.dim_size(0), shape_.dim_size(1)}; }
  template <class T> typename TTypes<T>::ConstMatrix matrix() const {
    return {typed<T>(), shape_.dim_size(0), shape_.dim_size(1)};
  }
  template <class T, int N = 2> typename TTypes<T>::Matrix flat_inner_dims() {
    static_assert(N == 2, "shim: matrix views only");
    const int64_t cols = shape_.dim_size(shape_.dims() - 1);
    return {typed<T>(), cols ? shape_.num_elements() / cols : 0, cols};
  }
  template <class T, int N = 2> typename TTypes<T>::ConstMatrix flat_inner_dims() const {
    static_assert(N == 2, "shim: matrix views only");
    const int64_t cols = shape_.dim_size(shape_.dims() - 1);
    return {typed<T>(), cols ? shape_.num_elements() / cols : 0, cols};
  }
  // Dimensions [0, begin] collapse into rows, the rest into columns (begin = -1: one row).
  template <class T, int N> typename TTypes<T>::Matrix flat_inner_outer_dims(int64_t begin) {
    static_assert(N == 2, "shim: matrix views only");
    const int64_t rows = outer(begin);
    return {typed<T>(), rows, rows ? shape_.num_elements() / rows : 0};
  }
  template <class T, int N> typename TTypes<T>::ConstMatrix flat_inner_outer_dims(int64_t begin) const {
    static_assert(N == 2, "shim: matrix views only");
    const int64_t rows = outer(begin);
    return {typed<T>(), rows, rows ? shape_.num_elements() / rows : 0};
  }
  int dims() const { return shape_.dims(); }
  const TensorShape& shape() const { return shape_; }
  std::type_index element_type() const { return buf_->type; }

 private:
  struct Buffer {
    void* data = nullptr;
    std::shared_ptr<void> owned;
    std::type_index type = std::type_index(typeid(void));
  };
  int64_t outer(int64_t begin) const {
    int64_t rows = 1;
    for (int64_t i = 0; i <= begin; ++i) rows *= shape_.dim_size(static_cast<int>(i));
    return rows;
  }
  template <class T> T* typed() const {
    if (buf_->data == nullptr) {
      const int64_t n = shape_.num_elements();
      T* p = new T[n > 0 ? n : 1]();
      buf_->owned = std::shared_ptr<void>(p, [](void* q) { delete[] static_cast<T*>(q); });
      buf_->data = p;
      buf_->type = std::type_index(typeid(T));
    }
    return static_cast<T*>(buf_->data);
  }
  std::shared_ptr<Buffer> buf_;
  TensorShape shape_;
};
}  // namespace tensorflow
