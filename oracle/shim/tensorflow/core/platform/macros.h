// Build shim (test infrastructure). Not product code.
#pragma once
#include "absl/log/check.h"
#define TF_PREDICT_FALSE(x) (__builtin_expect(false || (x), false))
#define TF_PREDICT_TRUE(x) (__builtin_expect(false || (x), true))
