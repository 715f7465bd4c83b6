// Build shim (test infrastructure) for the element types TF names. Not product code.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
namespace tensorflow {
using tstring = std::string;
struct bfloat16 {
  uint16_t bits;
  explicit operator float() const {
    const uint32_t u = static_cast<uint32_t>(bits) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  }
};
}  // namespace tensorflow
namespace Eigen {
struct half {
  uint16_t bits;
  explicit operator float() const {
    const uint32_t sign = static_cast<uint32_t>(bits & 0x8000u) << 16;
    const uint32_t exp = (bits >> 10) & 0x1Fu;
    uint32_t man = bits & 0x3FFu;
    uint32_t u;
    if (exp == 0x1F) {
      u = sign | 0x7F800000u | (man << 13);
    } else if (exp != 0) {
      u = sign | ((exp + 112u) << 23) | (man << 13);
    } else if (man == 0) {
      u = sign;
    } else {  // subnormal half: normalise
      int e = -1;
      do { ++e; man <<= 1; } while ((man & 0x400u) == 0);
      u = sign | ((112u - e) << 23) | ((man & 0x3FFu) << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  }
};
}  // namespace Eigen
