// Build shim (test infrastructure): stands in for abseil so that the reference's
// cc/lib/range_coder.{h,cc} compile unmodified from /root/reference. Not product code.
#pragma once
#define ABSL_PREDICT_FALSE(x) (__builtin_expect(false || (x), false))
#define ABSL_PREDICT_TRUE(x) (__builtin_expect(false || (x), true))
