// Build shim (test infrastructure). Not product code.
#pragma once
#include <cstdint>
#include <string>
namespace tensorflow {
using string = std::string;
}  // namespace tensorflow
