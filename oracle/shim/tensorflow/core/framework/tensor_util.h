// Build shim (test infrastructure) for tensor::MakeShape. Not product code.
#pragma once
#include <cstdint>
#include "tensorflow/core/framework/tensor.h"
#include "tensorflow/core/platform/status.h"
namespace tensorflow {
namespace tensor {
inline Status MakeShape(const Tensor& shape, TensorShape* out) {
  if (shape.dims() != 1) return errors::InvalidArgument("shape must be a vector");
  *out = TensorShape();
  if (shape.element_type() == std::type_index(typeid(int64_t))) {
    auto v = shape.flat<int64_t>();
    for (int64_t i = 0; i < v.size(); ++i) out->AddDim(v(i));
  } else {
    auto v = shape.flat<int32_t>();
    for (int64_t i = 0; i < v.size(); ++i) out->AddDim(v(i));
  }
  return Status();
}
}  // namespace tensor
}  // namespace tensorflow
