// Build shim (test infrastructure) for tensorflow logging macros. Not product code.
#pragma once
#include "absl/log/check.h"
