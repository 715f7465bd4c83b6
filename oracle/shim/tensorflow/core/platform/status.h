// Build shim (test infrastructure) for tensorflow::Status / errors. Not product code.
#pragma once
#include <sstream>
#include <string>
#include "absl/status/status.h"
namespace tensorflow {
using Status = absl::Status;
namespace errors {
template <class... A>
Status InvalidArgument(const A&... parts) {
  std::ostringstream os;
  (os << ... << parts);
  return absl::InvalidArgumentError(os.str());
}
}  // namespace errors
}  // namespace tensorflow
