// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// Plain restatement of the reference's 32-bit range coder arithmetic
// (tensorflow_compression/cc/lib/range_coder.{h,cc}).  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may use anything under oracle/.
//
// Parity is PINNED: tests/test_oracle_vs_ref.py runs this restatement against
// the reference's own range_coder.cc compiled verbatim into oracle/_ref/ and
// against the committed golden vectors in tests/golden/ (generated from that
// compiled reference by oracle/make_golden.py).
#pragma once
#include <cstdint>
#include <string>

namespace tfc_oracle {

// Encoder state.  Mirrors RangeEncoder's three members (range_coder.h:67-69):
// the interval is [base, base + span_m1] in a 32-bit window, and `pending`
// holds an unresolved carry digit (low 16 bits = digit + 1) together with the
// number of further pending bytes (bits 16+).
struct EncState {
  uint32_t base = 0;
  uint32_t span_m1 = 0xFFFFFFFFu;
  uint64_t pending = 0;
};

inline void put16(std::string* out, uint32_t v) {
  out->push_back(static_cast<char>((v >> 8) & 0xFF));
  out->push_back(static_cast<char>(v & 0xFF));
}

// One coding step for the sub-interval [lo, hi) / 2^prec.
// Follows RangeEncoder::Encode, range_coder.cc:37-264.
inline void enc_step(EncState& st, int32_t lo, int32_t hi, int prec,
                     std::string* out) {
  const uint64_t span = static_cast<uint64_t>(st.span_m1) + 1;      // :52
  const uint32_t off_lo = static_cast<uint32_t>((span * static_cast<uint64_t>(lo)) >> prec);       // :69
  const uint32_t off_hi = static_cast<uint32_t>(((span * static_cast<uint64_t>(hi)) >> prec) - 1); // :70
  st.base += off_lo;                                                // :82 (wraps)
  st.span_m1 = off_hi - off_lo;                                     // :83
  const bool wrapped = st.base < off_lo;                            // :84

  const bool straddles = static_cast<uint32_t>(st.base + st.span_m1) < st.base;  // :167
  if (straddles) {
    // Carry still undecided.  If the span got short, widen it and remember two
    // more undecided bytes (:194-207).
    if ((st.span_m1 >> 16) == 0) {
      st.base <<= 16;
      st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
      st.pending += 0x20000;
    }
    return;
  }

  if (st.pending != 0) {
    // Carry is now decided (:213-245): either it happened (digit as stored,
    // then 0x00 fill) or it did not (digit - 1, then 0xFF fill).
    uint64_t d = st.pending;
    char fill = 0;
    if (!wrapped) {
      d -= 1;
      fill = static_cast<char>(0xFF);
    }
    put16(out, static_cast<uint32_t>(d & 0xFFFF));
    out->append(static_cast<size_t>(d >> 16), fill);
    st.pending = 0;
  }

  if ((st.span_m1 >> 16) == 0) {                                    // :247-263
    const uint32_t top = st.base >> 16;
    st.base <<= 16;
    st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
    if (st.base <= static_cast<uint32_t>(st.base + st.span_m1)) {
      put16(out, top);
    } else {
      st.pending = static_cast<uint64_t>(top) + 1;
    }
  }
}

// Flush.  Follows RangeEncoder::Finalize, range_coder.cc:266-307.
inline void enc_flush(const EncState& st, std::string* out) {
  if (st.pending != 0) {
    out->push_back(static_cast<char>((st.pending >> 8) & 0xFF));
    if ((st.pending & 0xFF) != 0) out->push_back(static_cast<char>(st.pending & 0xFF));
    return;
  }
  if (st.base == 0) return;
  const uint32_t top = st.base + st.span_m1;
  const uint32_t r24 = ((st.base - 1) >> 24) + 1;
  if (r24 <= (top >> 24)) {
    out->push_back(static_cast<char>(r24 & 0xFF));
    return;
  }
  const uint32_t r16 = ((st.base - 1) >> 16) + 1;
  out->push_back(static_cast<char>((r16 >> 8) & 0xFF));
  if ((r16 & 0xFF) != 0) out->push_back(static_cast<char>(r16 & 0xFF));
}

// Decoder state.  Mirrors RangeDecoder (range_coder.h:287-293).
struct DecState {
  uint32_t base = 0;
  uint32_t span_m1 = 0xFFFFFFFFu;
  uint32_t window = 0;
  const uint8_t* cur = nullptr;
  const uint8_t* end = nullptr;
};

inline void dec_pull16(DecState& st) {                               // range_coder.h:273-282
  for (int i = 0; i < 2; ++i) {
    st.window <<= 8;
    if (st.cur != st.end) st.window |= *st.cur++;
  }
}

inline void dec_open(DecState& st, const uint8_t* p, size_t n) {     // range_coder.h:79-83
  st = DecState();
  st.cur = p;
  st.end = p + n;
  dec_pull16(st);
  dec_pull16(st);
}

// Decode one symbol against cdf[0..n) (cdf[0] is assumed 0 and not looked at).
// Follows RangeDecoder::DecodeInternal, range_coder.h:224-271.  `linear` picks
// LinearSearch (:193-202) instead of BinarySearch (:204-222); both return the
// first position k >= 1 with target <= span * cdf[k].
template <typename T>
inline int dec_step(DecState& st, const T* cdf, int64_t n, int prec, bool linear) {
  const uint64_t span = static_cast<uint64_t>(st.span_m1) + 1;
  const uint64_t target =
      (static_cast<uint64_t>(static_cast<uint32_t>(st.window - st.base)) + 1) << prec;
  const T* first = cdf + 1;
  int64_t len = n - 1;
  const T* hit;
  if (linear) {
    hit = first;
    while (hit != first + len && !(target <= span * static_cast<uint64_t>(*hit))) ++hit;
  } else {
    const T* pv = first;
    do {
      const int64_t half = len / 2;
      const T* mid = pv + half;
      if (target <= span * static_cast<uint64_t>(*mid)) {
        len = half;
      } else {
        pv = mid + 1;
        len -= half + 1;
      }
    } while (len > 0);
    hit = pv;
  }
  const uint32_t off_lo = static_cast<uint32_t>((span * static_cast<uint64_t>(*(hit - 1))) >> prec);
  const uint32_t off_hi = static_cast<uint32_t>(((span * static_cast<uint64_t>(*hit)) >> prec) - 1);
  st.base += off_lo;
  st.span_m1 = off_hi - off_lo;
  if ((st.span_m1 >> 16) == 0) {
    st.base <<= 16;
    st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
    dec_pull16(st);
  }
  return static_cast<int>(hit - cdf - 1);
}

// Weak end-of-stream check.  Follows RangeDecoder::Finalize, range_coder.h:144-169.
inline bool dec_close(const DecState& st) {
  if (st.cur != st.end) return false;
  const uint32_t top = st.base + st.span_m1;
  if (st.base == 0 || top < st.base) return st.window == 0;
  const int sh = (((st.base - 1) >> 24) < (top >> 24)) ? 24 : 16;
  const uint32_t r = ((st.base - 1) >> sh) + 1;
  return (r << sh) == st.window;
}

// Adapter exposing the restated core under the interface drivers.h expects.
struct OracleCore {
  using Enc = EncState;
  using Dec = DecState;
  static void encode(Enc& e, int32_t lo, int32_t hi, int prec, std::string* out) { enc_step(e, lo, hi, prec, out); }
  static void flush(const Enc& e, std::string* out) { enc_flush(e, out); }
  static void open(Dec& d, const uint8_t* p, size_t n) { dec_open(d, p, n); }
  static int decode(Dec& d, const int32_t* cdf, int64_t n, int prec) { return dec_step(d, cdf, n, prec, false); }
  static int decode_linear(Dec& d, const int32_t* cdf, int64_t n, int prec) { return dec_step(d, cdf, n, prec, true); }
  static bool close(const Dec& d) { return dec_close(d); }
};

}  // namespace tfc_oracle
