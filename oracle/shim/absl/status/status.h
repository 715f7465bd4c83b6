// Build shim (test infrastructure) for absl::Status. Not product code.
#pragma once
#include <string>
#include <utility>
namespace absl {
class Status {
 public:
  Status() = default;
  Status(int code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  bool ok() const { return code_ == 0; }
  const std::string& message() const { return msg_; }
 private:
  int code_ = 0;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(std::string msg) { return Status(3, std::move(msg)); }
}  // namespace absl
