// Build shim (test infrastructure) for tensorflow logging macros. Not product code.
#pragma once
#include <iostream>
#include "absl/log/check.h"
namespace tfc_shim {
struct NullStream {
  template <class T> NullStream& operator<<(const T&) { return *this; }
};
}  // namespace tfc_shim
#ifndef LOG
#define LOG(severity) ::tfc_shim::NullStream()
#define VLOG(level) ::tfc_shim::NullStream()
#endif
