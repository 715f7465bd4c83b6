// Build shim (test infrastructure) for absl::StrCat. Not product code.
#pragma once
#include <sstream>
#include <string>
namespace absl {
template <typename... Args>
std::string StrCat(const Args&... args) {
  std::ostringstream os;
  (void)std::initializer_list<int>{((os << args), 0)...};
  return os.str();
}
}  // namespace absl
