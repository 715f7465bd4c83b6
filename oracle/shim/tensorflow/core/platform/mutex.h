// Build shim (test infrastructure) for tensorflow::mutex. Not product code.
#pragma once
#include <mutex>
#include "absl/base/optimization.h"
#define ABSL_EXCLUSIVE_LOCKS_REQUIRED(...)
#define ABSL_GUARDED_BY(...)
namespace tensorflow {
using mutex = std::mutex;
using mutex_lock = std::lock_guard<std::mutex>;
}  // namespace tensorflow
