// Build shim (test infrastructure) for tensorflow::Status / errors. Not product code.
#pragma once
#include <sstream>
#include <string>
#include "absl/status/status.h"
#include "absl/status/statusor.h"
namespace tensorflow {
using Status = absl::Status;
template <class T> using StatusOr = absl::StatusOr<T>;
namespace errors {
template <class... A>
Status InvalidArgument(const A&... parts) {
  std::ostringstream os;
  (os << ... << parts);
  return absl::InvalidArgumentError(os.str());
}
}  // namespace errors
}  // namespace tensorflow
#define TF_RETURN_IF_ERROR(expr)              \
  do {                                        \
    ::tensorflow::Status s__ = (expr);        \
    if (!s__.ok()) return s__;                \
  } while (0)
