// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// Restatement of the reference's op-level drivers around the range coder core,
// templated on the core so the same drivers run (a) on the restated core in
// coder_core.h and (b) on the reference's own RangeEncoder/RangeDecoder compiled
// verbatim from /root/reference (oracle/_ref build, see ref_core.h).
//
// Follows:
//   table scan           cc/kernels/range_coder_kernels.cc:101-164
//   channel/index encode cc/kernels/range_coder_kernels.cc:191-272, 290-322
//   channel/index decode cc/kernels/range_coder_kernels.cc:360-429, 449-471
//   finalize             cc/kernels/range_coder_kernels.cc:274-287, 431-446
//   legacy RangeEncode/RangeDecode + broadcast
//                        cc/kernels/range_coding_kernels.cc:60-379,
//                        cc/kernels/range_coding_kernels_util.cc:34-91
//   ParallelFor over streams -> std::thread shards of contiguous streams.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace tfc_oracle {

struct TableRef {
  const int32_t* p;   // p[0] = +-precision, p[1..] = cdf values
  int64_t n;          // number of int32 in the row including p[0]
};

inline std::string range_msg(const char* name, int64_t v, int64_t lo, int64_t hi) {
  std::ostringstream os;
  os << name << "=" << v << " not in range [" << lo << ", " << hi << ")";
  return os.str();
}

// One table starting at *cur.  range_coder_kernels.cc:110-137.
inline bool scan_one(const int32_t* end, const int32_t** cur,
                     std::vector<TableRef>* out, std::string* err) {
  const int32_t* p = *cur;
  if (end < p + 3) { *err = "CDF ended prematurely."; return false; }
  const int32_t* head = p;
  const int64_t aprec = std::abs(static_cast<int64_t>(*head));
  if (aprec < 1 || aprec >= 17) { *err = range_msg("precision", aprec, 1, 17); return false; }
  const int32_t last = 1 << aprec;
  ++p;
  if (*p != 0) { *err = "CDF must start with 0."; return false; }
  do {
    ++p;
    if (p == end) { *err = "CDF must end with 1 << precision."; return false; }
    if (p[0] < p[-1]) { *err = "CDF must be monotonically increasing."; return false; }
  } while (*p != last);
  ++p;
  out->push_back(TableRef{head, static_cast<int64_t>(p - head)});
  while (p != end && *p == last) ++p;
  *cur = p;
  return true;
}

// rank 1: ragged concatenation; rank 2: one table per row, padded with 1<<prec.
// range_coder_kernels.cc:139-164.
inline bool scan_tables(const int32_t* lookup, int rank, int64_t rows, int64_t cols,
                        std::vector<TableRef>* out, std::string* err) {
  out->clear();
  if (rank == 1) {
    const int32_t* end = lookup + cols;
    for (const int32_t* cur = lookup; cur != end;) {
      if (!scan_one(end, &cur, out, err)) return false;
    }
    return true;
  }
  if (rank == 2) {
    const int32_t* end = lookup + rows * cols;
    for (const int32_t* cur = lookup; cur != end;) {
      const int32_t* row_end = cur + cols;
      if (!scan_one(row_end, &cur, out, err)) return false;
      if (cur != row_end) { *err = "CDF must end with 1 << precision."; return false; }
    }
    return true;
  }
  *err = "`lookup` must be rank 1 or 2";
  return false;
}

template <typename F>
inline void shard_streams(int64_t streams, int threads, F&& body) {
  if (threads <= 1 || streams <= 1) { body(0, streams); return; }
  const int64_t t = std::min<int64_t>(threads, streams);
  std::vector<std::thread> pool;
  for (int64_t k = 0; k < t; ++k) {
    const int64_t lo = streams * k / t, hi = streams * (k + 1) / t;
    pool.emplace_back([&body, lo, hi] { body(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

template <typename Core>
struct StreamEncoder {
  std::vector<int32_t> lookup;        // owned copy (the reference ref-holds the tensor)
  std::vector<TableRef> tables;
  std::vector<typename Core::Enc> enc;
  std::vector<std::string> sink;
  std::string err;

  bool init(const int32_t* lk, int rank, int64_t rows, int64_t cols, int64_t streams) {
    lookup.assign(lk, lk + (rank == 2 ? rows * cols : cols));
    enc.assign(streams, typename Core::Enc());
    sink.assign(streams, std::string());
    return scan_tables(lookup.data(), rank, rows, cols, &tables, &err);
  }

  // Escape coding of out-of-range values.  range_coder_kernels.cc:290-322.
  static void escape_encode(typename Core::Enc& e, std::string* out, const TableRef& row, int32_t v) {
    const int32_t vmax = static_cast<int32_t>(row.n) - 3;
    const int32_t neg = v < 0;
    int32_t g = 0;
    if (neg) { g = -v; v = vmax; }
    else if (v >= vmax) { g = v - vmax + 1; v = vmax; }
    Core::encode(e, row.p[v + 1], row.p[v + 2], -row.p[0], out);
    if (v != vmax) return;
    int32_t nb = 1;
    while (g >= (1 << nb)) { Core::encode(e, 0, 1, 1, out); ++nb; }
    while (--nb >= 0) { const int32_t b = (g >> nb) & 1; Core::encode(e, b, b + 1, 1, out); }
    Core::encode(e, neg, neg + 1, 1, out);
  }

  // value/index: [streams, elems] row-major; index may be null (channel mode).
  bool encode(const int32_t* value, const int32_t* index, int64_t elems, int threads) {
    const int64_t ntab = static_cast<int64_t>(tables.size());
    std::mutex mu;
    std::string first_err;
    shard_streams(static_cast<int64_t>(enc.size()), threads, [&](int64_t lo, int64_t hi) {
      for (int64_t s = lo; s < hi; ++s) {
        typename Core::Enc& e = enc[s];
        std::string* out = &sink[s];
        const int32_t* pv = value + s * elems;
        const int32_t* pi = index ? index + s * elems : nullptr;
        int64_t ch = 0;
        for (int64_t j = 0; j < elems; ++j, ++ch) {
          int64_t t;
          if (pi) {
            t = pi[j];
            if (t < 0 || t >= ntab) {
              std::lock_guard<std::mutex> g(mu);
              if (first_err.empty()) first_err = range_msg("index", t, 0, ntab);
              return;
            }
          } else {
            if (ch >= ntab) ch = 0;                      // :257
            t = ch;
          }
          const TableRef& row = tables[t];
          const int32_t v = pv[j];
          if (row.p[0] > 0) {
            if (v < 0 || v >= row.n - 2) {
              std::lock_guard<std::mutex> g(mu);
              if (first_err.empty()) first_err = range_msg("value", v, 0, row.n - 2);
              return;
            }
            Core::encode(e, row.p[v + 1], row.p[v + 2], row.p[0], out);
          } else {
            escape_encode(e, out, row, v);
          }
        }
      }
    });
    if (!first_err.empty()) { err = first_err; return false; }
    return true;
  }

  void finalize() {
    for (size_t s = 0; s < enc.size(); ++s) Core::flush(enc[s], &sink[s]);
  }
};

template <typename Core>
struct StreamDecoder {
  std::vector<int32_t> lookup;
  std::vector<TableRef> tables;
  std::vector<uint8_t> blob;
  std::vector<int64_t> offs;
  std::vector<typename Core::Dec> dec;
  std::string err;

  bool init(const uint8_t* bytes, const int64_t* offsets, int64_t streams,
            const int32_t* lk, int rank, int64_t rows, int64_t cols) {
    lookup.assign(lk, lk + (rank == 2 ? rows * cols : cols));
    offs.assign(offsets, offsets + streams + 1);
    blob.assign(bytes, bytes + offs[streams]);
    blob.push_back(0);  // keep data() non-null for empty input
    dec.resize(streams);
    for (int64_t s = 0; s < streams; ++s)
      Core::open(dec[s], blob.data() + offs[s], static_cast<size_t>(offs[s + 1] - offs[s]));
    return scan_tables(lookup.data(), rank, rows, cols, &tables, &err);
  }

  // range_coder_kernels.cc:449-471.
  static int32_t escape_decode(typename Core::Dec& d, const TableRef& row) {
    static const int32_t kBit[3] = {0, 1, 2};
    const int32_t vmax = static_cast<int32_t>(row.n) - 3;
    int32_t v = Core::decode(d, row.p + 1, row.n - 1, -row.p[0]);
    if (v != vmax) return v;
    int32_t nb = 0;
    while (Core::decode_linear(d, kBit, 3, 1) == 0) ++nb;
    v = 1 << nb;
    while (--nb >= 0) v |= Core::decode_linear(d, kBit, 3, 1) << nb;
    const int32_t neg = Core::decode_linear(d, kBit, 3, 1);
    return neg ? -v : v + vmax - 1;
  }

  bool decode(const int32_t* index, int32_t* output, int64_t elems, int threads) {
    const int64_t ntab = static_cast<int64_t>(tables.size());
    std::mutex mu;
    std::string first_err;
    shard_streams(static_cast<int64_t>(dec.size()), threads, [&](int64_t lo, int64_t hi) {
      for (int64_t s = lo; s < hi; ++s) {
        typename Core::Dec& d = dec[s];
        int32_t* po = output + s * elems;
        const int32_t* pi = index ? index + s * elems : nullptr;
        int64_t ch = 0;
        for (int64_t j = 0; j < elems; ++j, ++ch) {
          int64_t t;
          if (pi) {
            t = pi[j];
            if (t < 0 || t >= ntab) {
              std::lock_guard<std::mutex> g(mu);
              if (first_err.empty()) first_err = range_msg("index", t, 0, ntab);
              return;
            }
          } else {
            if (ch >= ntab) ch = 0;
            t = ch;
          }
          const TableRef& row = tables[t];
          po[j] = (row.p[0] > 0) ? Core::decode(d, row.p + 1, row.n - 1, row.p[0])
                                 : escape_decode(d, row);
        }
      }
    });
    if (!first_err.empty()) { err = first_err; return false; }
    return true;
  }
};

// ---------------------------------------------------------------------------
// Legacy single-stream ops.
// ---------------------------------------------------------------------------

// range_coding_kernels_util.cc:34-91.  data_shape has nd dims, cdf_shape nd+1.
inline bool merge_axes(const int64_t* data_shape, const int64_t* cdf_shape, int nd,
                       std::vector<int64_t>* md, std::vector<int64_t>* mc, std::string* err) {
  md->assign(1, 1);
  mc->assign(1, 1);
  size_t i = 0;
  for (int j = 0; j < nd; ++j) {
    if (data_shape[j] != cdf_shape[j] && cdf_shape[j] != 1) {
      std::ostringstream os;
      os << "Cannot broadcast shape [";
      for (int k = 0; k <= nd; ++k) os << (k ? "," : "") << cdf_shape[k];
      os << "] to [";
      for (int k = 0; k < nd; ++k) os << (k ? "," : "") << data_shape[k];
      os << "]";
      *err = os.str();
      return false;
    }
    const bool was_b = ((*mc)[i] == 1);
    const bool is_b = (cdf_shape[j] == 1);
    const bool merge = (was_b == is_b) || (data_shape[j] <= 1) || ((*md)[i] <= 1);
    if (merge) {
      (*md)[i] *= data_shape[j];
      (*mc)[i] *= cdf_shape[j];
    } else {
      md->push_back(data_shape[j]);
      mc->push_back(cdf_shape[j]);
      ++i;
    }
  }
  mc->push_back(cdf_shape[nd]);
  return true;
}

// Walks data linearly while tracking the broadcast cdf row.
// Same traversal as BroadcastRange (range_coding_kernels.cc:60-132).
struct BroadcastWalk {
  std::vector<int64_t> shape, step, idx;
  int64_t cdf_pos = 0;
  BroadcastWalk(const std::vector<int64_t>& md, const std::vector<int64_t>& mc) {
    const int n = static_cast<int>(md.size());
    shape = md;
    idx.assign(n, 0);
    const int64_t inner = mc[n];
    step.assign(n, inner);
    int64_t stride = inner;
    for (int i = n - 1; i >= 0; --i) {
      if (mc[i] <= 1) step[i] -= stride;
      stride *= mc[i];
    }
  }
  int64_t next() {
    const int64_t here = cdf_pos;
    int i = static_cast<int>(shape.size()) - 1;
    for (; i > 0; --i) {
      if (++idx[i] < shape[i]) break;
      idx[i] = 0;
    }
    cdf_pos += step[i];
    return here;
  }
};

inline bool check_cdf_shape(const int64_t* data_shape, int nd, const int64_t* cdf_shape, int nc,
                            std::string* err) {
  (void)data_shape;
  if (nc != nd + 1) { *err = "`cdf` should have one more axis than `data`"; return false; }
  if (cdf_shape[nc - 1] <= 1) { *err = "The last dimension of `cdf` should be > 1"; return false; }
  return true;
}

inline bool check_cdf_values(int prec, const int32_t* cdf, int64_t rows, int64_t width, std::string* err) {
  if (width <= 2) { std::ostringstream os; os << "CDF size should be > 2: " << width; *err = os.str(); return false; }
  const int32_t ub = 1 << prec;
  for (int64_t r = 0; r < rows; ++r) {
    const int32_t* s = cdf + r * width;
    if (s[0] != 0 || s[width - 1] != ub) {
      std::ostringstream os;
      os << "CDF should start from 0 and end at " << ub << ": cdf[0]=" << s[0] << ", cdf[^1]=" << s[width - 1];
      *err = os.str();
      return false;
    }
    for (int64_t j = 0; j + 1 < width; ++j)
      if (s[j + 1] <= s[j]) { *err = "CDF is not monotonic"; return false; }
  }
  return true;
}

template <typename Core>
inline bool legacy_encode(const int16_t* data, const int64_t* data_shape, int nd,
                          const int32_t* cdf, const int64_t* cdf_shape, int nc,
                          int prec, int debug_level, std::string* out, std::string* err) {
  if (!check_cdf_shape(data_shape, nd, cdf_shape, nc, err)) return false;
  int64_t cdf_total = 1, data_total = 1;
  for (int i = 0; i < nc; ++i) cdf_total *= cdf_shape[i];
  for (int i = 0; i < nd; ++i) data_total *= data_shape[i];
  const int64_t width = cdf_shape[nc - 1];
  if (debug_level > 0 && !check_cdf_values(prec, cdf, cdf_total / width, width, err)) return false;
  std::vector<int64_t> md, mc;
  if (!merge_axes(data_shape, cdf_shape, nd, &md, &mc, err)) return false;
  if (md.size() > 6) { *err = "Irregular broadcast pattern"; return false; }
  BroadcastWalk walk(md, mc);
  typename Core::Enc e;
  for (int64_t k = 0; k < data_total; ++k) {
    const int64_t row = walk.next();
    const int64_t v = data[k];
    if (debug_level > 0 && (v < 0 || width <= v + 1)) {
      std::ostringstream os;
      os << "'data' value not in [0, " << width - 1 << "): value=" << v;
      *err = os.str();
      return false;
    }
    Core::encode(e, cdf[row + v], cdf[row + v + 1], prec, out);
  }
  Core::flush(e, out);
  return true;
}

template <typename Core>
inline bool legacy_decode(const uint8_t* bytes, int64_t nbytes, const int64_t* out_shape, int nd,
                          const int32_t* cdf, const int64_t* cdf_shape, int nc,
                          int prec, int debug_level, int16_t* out, std::string* err) {
  if (!check_cdf_shape(out_shape, nd, cdf_shape, nc, err)) return false;
  int64_t cdf_total = 1, out_total = 1;
  for (int i = 0; i < nc; ++i) cdf_total *= cdf_shape[i];
  for (int i = 0; i < nd; ++i) out_total *= out_shape[i];
  const int64_t width = cdf_shape[nc - 1];
  if (debug_level > 0 && !check_cdf_values(prec, cdf, cdf_total / width, width, err)) return false;
  std::vector<int64_t> md, mc;
  if (!merge_axes(out_shape, cdf_shape, nd, &md, &mc, err)) return false;
  if (md.size() > 6) { *err = "Irregular broadcast pattern"; return false; }
  BroadcastWalk walk(md, mc);
  std::vector<uint8_t> copy(bytes, bytes + nbytes);
  copy.push_back(0);
  typename Core::Dec d;
  Core::open(d, copy.data(), static_cast<size_t>(nbytes));
  for (int64_t k = 0; k < out_total; ++k) {
    const int64_t row = walk.next();
    out[k] = static_cast<int16_t>(Core::decode(d, cdf + row, width, prec));
  }
  return true;
}


// ---- deprecated UnboundedIndexRangeEncode / Decode ------------------------------------
// cc/kernels/unbounded_index_range_coding_kernels.cc:54-143 (argument checks), :185-249 (encode),
// :307-367 (decode).  One stream for the whole tensor; out-of-range values are clamped to the
// row's last symbol and followed by a variable-length code in `overflow_width`-bit digits:
// the digit count in unary-of-max-digits form, then the digits, least significant first; the
// overflow value is 2(v - max) for v >= max and -2v - 1 for v < 0.
inline bool unbounded_check(const int32_t* index, int64_t total, const int32_t* cdf, int64_t rows,
                            int64_t width, const int32_t* cdf_size, int prec, int debug_level,
                            std::string* err) {
  if (width < 3) { *err = "'cdf' should be 2-D and cdf.dim_size(1) >= 3"; return false; }
  if (debug_level <= 0) return true;
  for (int64_t i = 0; i < total; ++i)                               // CheckIndex :54-63
    if (index[i] < 0 || rows <= index[i]) {
      std::ostringstream os;
      os << "'index' has a value not in [0, " << rows << "): value=" << index[i];
      *err = os.str();
      return false;
    }
  for (int64_t i = 0; i < rows; ++i)                                // CheckCdfSize :65-74
    if (cdf_size[i] < 3 || width < cdf_size[i]) {
      std::ostringstream os;
      os << "'cdf_size' has a value not in [3, " << width << "]: value=" << cdf_size[i];
      *err = os.str();
      return false;
    }
  const int32_t upper = 1 << prec;                                  // CheckCdf :76-101
  for (int64_t i = 0; i < rows; ++i) {
    const int32_t* s = cdf + i * width;
    const int32_t n = cdf_size[i];
    if (s[0] != 0 || s[n - 1] != upper) {
      std::ostringstream os;
      os << "Each cdf should start from 0 and end at " << upper << ": cdf[0]=" << s[0]
         << ", cdf[^1]=" << s[n - 1];
      *err = os.str();
      return false;
    }
    for (int32_t j = 0; j + 1 < n; ++j)
      if (s[j + 1] <= s[j]) { *err = "CDF is not monotonic"; return false; }
  }
  return true;
}

template <typename Core>
inline bool unbounded_encode(const int32_t* data, const int32_t* index, int64_t total, const int32_t* cdf,
                             int64_t rows, int64_t width, const int32_t* cdf_size, const int32_t* offset,
                             int prec, int overflow_width, int debug_level, std::string* out,
                             std::string* err) {
  if (!unbounded_check(index, total, cdf, rows, width, cdf_size, prec, debug_level, err)) return false;
  typename Core::Enc e;
  const uint32_t max_overflow = (1u << overflow_width) - 1;
  for (int64_t i = 0; i < total; ++i) {
    const int32_t row = index[i];
    const int32_t max_value = cdf_size[row] - 2;
    int32_t value = data[i] - offset[row];
    uint32_t overflow = 0;
    if (value < 0) {
      overflow = static_cast<uint32_t>(-2 * value - 1);
      value = max_value;
    } else if (value >= max_value) {
      overflow = static_cast<uint32_t>(2 * (value - max_value));
      value = max_value;
    }
    const int32_t* s = cdf + static_cast<int64_t>(row) * width;
    Core::encode(e, s[value], s[value + 1], prec, out);
    if (value == max_value) {
      int32_t widths = 0;
      while (widths * overflow_width < 32 && (overflow >> (widths * overflow_width)) != 0) ++widths;
      uint32_t val = static_cast<uint32_t>(widths);
      while (val >= max_overflow) {
        Core::encode(e, max_overflow, max_overflow + 1, overflow_width, out);
        val -= max_overflow;
      }
      Core::encode(e, val, val + 1, overflow_width, out);
      for (int32_t j = 0; j < widths; ++j) {
        const uint32_t d = (overflow >> (j * overflow_width)) & max_overflow;
        Core::encode(e, d, d + 1, overflow_width, out);
      }
    }
  }
  Core::flush(e, out);
  return true;
}

template <typename Core>
inline bool unbounded_decode(const uint8_t* bytes, int64_t nbytes, const int32_t* index, int64_t total,
                             const int32_t* cdf, int64_t rows, int64_t width, const int32_t* cdf_size,
                             const int32_t* offset, int prec, int overflow_width, int debug_level,
                             int32_t* out, std::string* err) {
  if (!unbounded_check(index, total, cdf, rows, width, cdf_size, prec, debug_level, err)) return false;
  std::vector<uint8_t> copy(bytes, bytes + nbytes);
  copy.push_back(0);
  typename Core::Dec d;
  Core::open(d, copy.data(), static_cast<size_t>(nbytes));
  const uint32_t max_overflow = (1u << overflow_width) - 1;
  std::vector<int32_t> digits((1 << overflow_width) + 1);
  for (size_t i = 0; i < digits.size(); ++i) digits[i] = static_cast<int32_t>(i);
  for (int64_t i = 0; i < total; ++i) {
    const int32_t row = index[i];
    const int32_t max_value = cdf_size[row] - 2;
    const int32_t* s = cdf + static_cast<int64_t>(row) * width;
    int32_t value = Core::decode(d, s, max_value + 2, prec);
    if (value == max_value) {
      int32_t widths = 0;
      uint32_t val;
      do {
        val = static_cast<uint32_t>(Core::decode(d, digits.data(), static_cast<int64_t>(digits.size()), overflow_width));
        widths += static_cast<int32_t>(val);
      } while (val == max_overflow && widths < 64);      // bounded: damaged input cannot spin
      uint32_t overflow = 0;
      for (int32_t j = 0; j < widths; ++j) {
        const uint32_t v = static_cast<uint32_t>(Core::decode(d, digits.data(), static_cast<int64_t>(digits.size()), overflow_width));
        if (j * overflow_width < 32) overflow |= v << (j * overflow_width);
      }
      value = static_cast<int32_t>(overflow >> 1);
      if (overflow & 1) value = -value - 1; else value += max_value;
    }
    out[i] = value + offset[row];
  }
  return true;
}

}  // namespace tfc_oracle
