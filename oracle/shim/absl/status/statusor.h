// Build shim (test infrastructure) for absl::StatusOr. Not product code.
#pragma once
#include <utility>
#include "absl/status/status.h"
namespace absl {
template <class T>
class StatusOr {
 public:
  StatusOr(const Status& s) : status_(s) {}          // NOLINT
  StatusOr(T v) : value_(std::move(v)) {}            // NOLINT
  template <class U, class = std::enable_if_t<std::is_convertible<U, T>::value>>
  StatusOr(U&& v) : value_(std::forward<U>(v)) {}    // NOLINT
  bool ok() const { return status_.ok(); }
  const Status& status() const { return status_; }
  T& value() { return value_; }
  T& operator*() { return value_; }
 private:
  Status status_;
  T value_{};
};
}  // namespace absl
