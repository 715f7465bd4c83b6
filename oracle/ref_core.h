// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// Adapter that plugs the reference's OWN RangeEncoder / RangeDecoder classes
// (compiled verbatim from /root/reference/tensorflow_compression/cc/lib/
// range_coder.{h,cc} with the shim headers in oracle/shim) into drivers.h.
// No reference source is copied into this repository; the build reads it in
// place and writes only to oracle/_ref/.
#pragma once
#include <cstdint>
#include <string>

#include "absl/strings/string_view.h"
#include "absl/types/span.h"
#include "tensorflow_compression/cc/lib/range_coder.h"

namespace tfc_oracle {

struct RefDec {
  // RangeDecoder has no default constructor and holds a view of the bytes.
  alignas(tensorflow_compression::RangeDecoder) unsigned char buf[sizeof(tensorflow_compression::RangeDecoder)];
  bool live = false;
  tensorflow_compression::RangeDecoder* get() { return reinterpret_cast<tensorflow_compression::RangeDecoder*>(buf); }
  const tensorflow_compression::RangeDecoder* get() const { return reinterpret_cast<const tensorflow_compression::RangeDecoder*>(buf); }
};

struct RefCore {
  using Enc = tensorflow_compression::RangeEncoder;
  using Dec = RefDec;
  static void encode(Enc& e, int32_t lo, int32_t hi, int prec, std::string* out) { e.Encode(lo, hi, prec, out); }
  static void flush(const Enc& e, std::string* out) { e.Finalize(out); }
  static void open(Dec& d, const uint8_t* p, size_t n) {
    new (d.buf) tensorflow_compression::RangeDecoder(absl::string_view(reinterpret_cast<const char*>(p), n));
    d.live = true;
  }
  static int decode(Dec& d, const int32_t* cdf, int64_t n, int prec) {
    return d.get()->Decode(absl::Span<const int32_t>(cdf, static_cast<size_t>(n)), prec);
  }
  static int decode_linear(Dec& d, const int32_t* cdf, int64_t n, int prec) {
    return d.get()->DecodeLinearly(absl::Span<const int32_t>(cdf, static_cast<size_t>(n)), prec);
  }
  static bool close(const Dec& d) { return d.get()->Finalize(); }
};

}  // namespace tfc_oracle
