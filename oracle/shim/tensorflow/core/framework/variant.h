// Build shim (test infrastructure) for tensorflow::Variant: a type-erased holder. Not product code.
#pragma once
#include <memory>
#include <typeindex>
#include <utility>
namespace tensorflow {
class VariantTensorData {};
class Variant {
 public:
  Variant() = default;
  template <class T, class = std::enable_if_t<!std::is_same<std::decay_t<T>, Variant>::value>>
  Variant& operator=(T&& v) {
    using V = std::decay_t<T>;
    held_ = std::make_shared<V>(std::forward<T>(v));
    type_ = std::type_index(typeid(V));
    return *this;
  }
  template <class T> T* get() {
    return held_ && type_ == std::type_index(typeid(T)) ? static_cast<T*>(held_.get()) : nullptr;
  }
 private:
  std::shared_ptr<void> held_;
  std::type_index type_ = std::type_index(typeid(void));
};
}  // namespace tensorflow
