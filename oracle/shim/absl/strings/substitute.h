// Build shim (test infrastructure) for absl::Substitute ($0..$9). Not product code.
#pragma once
#include <sstream>
#include <string>
#include <string_view>
#include <vector>
namespace absl {
template <class... A>
std::string Substitute(std::string_view format, const A&... args) {
  std::vector<std::string> parts;
  auto one = [&parts](const auto& a) {
    std::ostringstream os;
    os << a;
    parts.push_back(os.str());
  };
  (one(args), ...);
  std::string out;
  for (size_t i = 0; i < format.size(); ++i) {
    if (format[i] == '$' && i + 1 < format.size() && format[i + 1] >= '0' && format[i + 1] <= '9') {
      out += parts.at(static_cast<size_t>(format[i + 1] - '0'));
      ++i;
    } else {
      out += format[i];
    }
  }
  return out;
}
}  // namespace absl
