// Device-side state of one range-coder stream (wave-uniform values).
#pragma once

// clang in ROCm 7.2 has no __builtin_amdgcn_writelane; bind the LLVM intrinsic
// (v_writelane_b32: write a wave-uniform value into one lane of a VGPR).
extern "C" __device__ int tfc_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

namespace tfc {

// Interval [base, base + span_m1] in a 32-bit window plus the unresolved carry:
// pend_digit = (undecided 16-bit digit + 1) or 0; pend_bytes = further
// undecided bytes behind it (always even).  Same information as
// RangeEncoder::{base_, size_minus1_, delay_} (cc/lib/range_coder.h:67-69).
struct EncoderState {
  unsigned int base;
  unsigned int span_m1;
  unsigned int pend_digit;
  unsigned int pend_bytes;
};

// Directory entry of the lane-per-stream kernels' LDS image (range_lanes.h): byte offsets inside the
// image of the row's cdf entries (minus 2: the lower bound of symbol s is at cdf + 2 s + 2), boundary
// bitmap and running counts, and
//   info = limit | has_escape << 31,  limit = number of plain symbols (= index of the escape symbol).
struct LaneRow { unsigned int cdf, info, bits, cum; };

}  // namespace tfc
