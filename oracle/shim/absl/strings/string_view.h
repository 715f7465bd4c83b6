// Build shim (test infrastructure): absl::string_view == std::string_view. Not product code.
#pragma once
#include <algorithm>
#include <string_view>
namespace absl {
using string_view = std::string_view;
}
