// TEST INFRASTRUCTURE — CPU oracle, not product code.
//
// extern "C" surface over drivers.h.  Built twice by oracle/Makefile:
//   libtfc_oracle.so        restated core (coder_core.h), symbols tfco_*
//   _ref/libtfc_ref.so      reference core compiled verbatim from
//                           /root/reference (ref_core.h), symbols tfcr_*
// Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() use it.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#ifdef TFC_USE_REF
#include "ref_core.h"
#define SYM(name) tfcr_##name
using Core = tfc_oracle::RefCore;
#else
#include "coder_core.h"
#define SYM(name) tfco_##name
using Core = tfc_oracle::OracleCore;
#endif
#include "drivers.h"
#include "pmf_to_cdf.h"
#ifdef TFC_USE_REF
#include <map>
#include <memory>
#include "tensorflow/core/framework/op_kernel.h"  // oracle/shim: registry of the reference's kernels
#else
#include "quantization.h"
#endif

using tfc_oracle::StreamDecoder;
using tfc_oracle::StreamEncoder;

static thread_local std::string g_err;

struct EncHandle {
  StreamEncoder<Core> impl;
  std::vector<int64_t> offs;
};
struct DecHandle {
  StreamDecoder<Core> impl;
};

extern "C" {

const char* SYM(last_error)() { return g_err.c_str(); }

// ---- raw coder calls (known-answer vectors) --------------------------------
// Encodes n intervals [lower[i], upper[i]) / 2^precision[i]; returns the byte
// count (bytes copied into out if it fits in cap).
int64_t SYM(raw_encode)(int64_t n, const int32_t* lower, const int32_t* upper,
                        const int32_t* precision, uint8_t* out, int64_t cap) {
  typename Core::Enc e{};
  std::string s;
  for (int64_t i = 0; i < n; ++i) Core::encode(e, lower[i], upper[i], precision[i], &s);
  Core::flush(e, &s);
  if (static_cast<int64_t>(s.size()) <= cap && !s.empty()) std::memcpy(out, s.data(), s.size());
  return static_cast<int64_t>(s.size());
}

// Decodes n symbols against one cdf (binary search); returns Finalize().
int SYM(raw_decode)(const uint8_t* bytes, int64_t nbytes, const int32_t* cdf, int64_t cdf_n,
                    int precision, int64_t n, int32_t* out) {
  std::vector<uint8_t> copy(bytes, bytes + nbytes);
  copy.push_back(0);
  typename Core::Dec d;
  Core::open(d, copy.data(), static_cast<size_t>(nbytes));
  for (int64_t i = 0; i < n; ++i) out[i] = Core::decode(d, cdf, cdf_n, precision);
  return Core::close(d) ? 1 : 0;
}

// ---- multi-stream encoder handle -------------------------------------------
void* SYM(encoder_create)(const int32_t* lookup, int rank, int64_t rows, int64_t cols,
                          int64_t streams) {
  auto* h = new EncHandle;
  if (!h->impl.init(lookup, rank, rows, cols, streams)) {
    g_err = h->impl.err;
    delete h;
    return nullptr;
  }
  return h;
}

int SYM(encoder_encode)(void* handle, const int32_t* value, const int32_t* index,
                        int64_t elems, int threads) {
  auto* h = static_cast<EncHandle*>(handle);
  if (!h->impl.encode(value, index, elems, threads)) { g_err = h->impl.err; return 1; }
  return 0;
}

// Flushes every stream; returns total byte count.
int64_t SYM(encoder_finalize)(void* handle) {
  auto* h = static_cast<EncHandle*>(handle);
  h->impl.finalize();
  h->offs.assign(1, 0);
  for (auto& s : h->impl.sink) h->offs.push_back(h->offs.back() + static_cast<int64_t>(s.size()));
  return h->offs.back();
}

void SYM(encoder_output)(void* handle, uint8_t* blob, int64_t* offsets) {
  auto* h = static_cast<EncHandle*>(handle);
  std::memcpy(offsets, h->offs.data(), h->offs.size() * sizeof(int64_t));
  for (size_t s = 0; s < h->impl.sink.size(); ++s)
    if (!h->impl.sink[s].empty())
      std::memcpy(blob + h->offs[s], h->impl.sink[s].data(), h->impl.sink[s].size());
}

void SYM(encoder_free)(void* handle) { delete static_cast<EncHandle*>(handle); }

// ---- multi-stream decoder handle -------------------------------------------
void* SYM(decoder_create)(const uint8_t* blob, const int64_t* offsets, int64_t streams,
                          const int32_t* lookup, int rank, int64_t rows, int64_t cols) {
  auto* h = new DecHandle;
  if (!h->impl.init(blob, offsets, streams, lookup, rank, rows, cols)) {
    g_err = h->impl.err;
    delete h;
    return nullptr;
  }
  return h;
}

int SYM(decoder_decode)(void* handle, const int32_t* index, int32_t* output, int64_t elems,
                        int threads) {
  auto* h = static_cast<DecHandle*>(handle);
  if (!h->impl.decode(index, output, elems, threads)) { g_err = h->impl.err; return 1; }
  return 0;
}

void SYM(decoder_finalize)(void* handle, uint8_t* ok) {
  auto* h = static_cast<DecHandle*>(handle);
  for (size_t s = 0; s < h->impl.dec.size(); ++s) ok[s] = Core::close(h->impl.dec[s]) ? 1 : 0;
}

void SYM(decoder_free)(void* handle) { delete static_cast<DecHandle*>(handle); }

// ---- legacy single-stream ops ----------------------------------------------
// Returns byte count (>= 0) or -1 on error.
int64_t SYM(range_encode)(const int16_t* data, const int64_t* data_shape, int nd,
                          const int32_t* cdf, const int64_t* cdf_shape, int nc, int precision,
                          int debug_level, uint8_t* out, int64_t cap) {
  std::string s, err;
  if (!tfc_oracle::legacy_encode<Core>(data, data_shape, nd, cdf, cdf_shape, nc, precision,
                                       debug_level, &s, &err)) {
    g_err = err;
    return -1;
  }
  if (static_cast<int64_t>(s.size()) <= cap && !s.empty()) std::memcpy(out, s.data(), s.size());
  return static_cast<int64_t>(s.size());
}

int SYM(range_decode)(const uint8_t* bytes, int64_t nbytes, const int64_t* out_shape, int nd,
                      const int32_t* cdf, const int64_t* cdf_shape, int nc, int precision,
                      int debug_level, int16_t* out) {
  std::string err;
  if (!tfc_oracle::legacy_decode<Core>(bytes, nbytes, out_shape, nd, cdf, cdf_shape, nc,
                                       precision, debug_level, out, &err)) {
    g_err = err;
    return 1;
  }
  return 0;
}

// ---- deprecated UnboundedIndexRangeEncode / Decode -------------------------------------
int64_t SYM(unbounded_index_range_encode)(const int32_t* data, const int32_t* index, int64_t total,
                                          const int32_t* cdf, int64_t rows, int64_t width,
                                          const int32_t* cdf_size, const int32_t* offset, int precision,
                                          int overflow_width, int debug_level, uint8_t* out, int64_t cap) {
  std::string s, err;
  if (!tfc_oracle::unbounded_encode<Core>(data, index, total, cdf, rows, width, cdf_size, offset, precision,
                                          overflow_width, debug_level, &s, &err)) {
    g_err = err;
    return -1;
  }
  if (static_cast<int64_t>(s.size()) <= cap && !s.empty()) std::memcpy(out, s.data(), s.size());
  return static_cast<int64_t>(s.size());
}

int SYM(unbounded_index_range_decode)(const uint8_t* bytes, int64_t nbytes, const int32_t* index,
                                      int64_t total, const int32_t* cdf, int64_t rows, int64_t width,
                                      const int32_t* cdf_size, const int32_t* offset, int precision,
                                      int overflow_width, int debug_level, int32_t* out) {
  std::string err;
  if (!tfc_oracle::unbounded_decode<Core>(bytes, nbytes, index, total, cdf, rows, width, cdf_size, offset,
                                          precision, overflow_width, debug_level, out, &err)) {
    g_err = err;
    return 1;
  }
  return 0;
}

// ---- PmfToQuantizedCdf -----------------------------------------------------
// pmf [rows, n] float32 -> cdf [rows, n + 1] int32.  Validation as in
// pmf_to_cdf_kernels.cc:58-86.
int SYM(pmf_to_quantized_cdf)(const float* pmf, int64_t rows, int64_t n, int precision,
                              int32_t* cdf) {
#ifdef TFC_USE_REF
  // the reference's own PmfToCdfOp (pmf_to_cdf_kernels.cc, compiled verbatim behind oracle/shim)
  auto it = tfc_shim::registry().find({"PmfToQuantizedCdf", std::type_index(typeid(void))});
  if (it == tfc_shim::registry().end()) { g_err = "PmfToQuantizedCdf: reference kernel not registered"; return 1; }
  tensorflow::OpKernelConstruction construction;
  construction.int_attrs["precision"] = precision;
  std::unique_ptr<tensorflow::OpKernel> op(it->second(&construction));
  if (!construction.status.ok()) { g_err = construction.status.message(); return 1; }
  tensorflow::OpKernelContext ctx;
  ctx.inputs.emplace_back(pmf, tensorflow::TensorShape({rows, n}));
  ctx.output = tensorflow::Tensor(cdf, tensorflow::TensorShape({rows, n + 1}));
  ctx.caller_output = true;
  op->Compute(&ctx);
  if (!ctx.status.ok()) { g_err = ctx.status.message(); return 1; }
  return 0;
#endif
  if (!(0 < precision && precision <= 16)) { g_err = "`precision` must be in [1, 16]"; return 1; }
  if (n <= 1) { g_err = "`pmf` size should be at least 2 in the last axis."; return 1; }
  for (int64_t i = 0; i < rows * n; ++i)
    if (!(std::isfinite(pmf[i]) && pmf[i] >= 0)) {
      g_err = "`pmf` has non-finite or negative element";
      return 1;
    }
  for (int64_t r = 0; r < rows; ++r)
    tfc_oracle::pmf_row_to_cdf(pmf + r * n, n, precision, cdf + r * (n + 1));
  return 0;
}

// StochasticRound; dtype 0 float32, 1 bfloat16, 2 float16 (bit patterns).  Empty seed (clock) is not
// offered: there is nothing to compare.
int SYM(stochastic_round)(const void* x, int dtype, int64_t n, float step, const int32_t* seed,
                          int64_t seed_len, int32_t* out) {
  if (dtype < 0 || dtype > 2 || seed_len <= 0) {
    g_err = "stochastic_round: dtype must be 0..2 and the seed non-empty";
    return 1;
  }
#ifdef TFC_USE_REF
  const std::type_index ti = dtype == 0   ? std::type_index(typeid(float))
                             : dtype == 1 ? std::type_index(typeid(tensorflow::bfloat16))
                                          : std::type_index(typeid(Eigen::half));
  auto it = tfc_shim::registry().find({"StochasticRound", ti});
  if (it == tfc_shim::registry().end()) {
    g_err = "stochastic_round: reference kernel not registered";
    return 1;
  }
  tensorflow::OpKernelConstruction construction;
  std::unique_ptr<tensorflow::OpKernel> op(it->second(&construction));
  tensorflow::OpKernelContext ctx;
  ctx.inputs.emplace_back(const_cast<void*>(x), tensorflow::TensorShape({n}));
  ctx.inputs.emplace_back(&step, tensorflow::TensorShape());
  ctx.inputs.emplace_back(const_cast<int32_t*>(seed), tensorflow::TensorShape({seed_len}));
  ctx.output = tensorflow::Tensor(out, tensorflow::TensorShape({n}));
  ctx.caller_output = true;
  op->Compute(&ctx);
  if (!ctx.status.ok()) {
    g_err = ctx.status.message();
    return 1;
  }
  return 0;
#else
  std::vector<float> wide(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) {
    if (dtype == 0) {
      wide[i] = static_cast<const float*>(x)[i];
    } else {
      const uint16_t h = static_cast<const uint16_t*>(x)[i];
      uint32_t u;
      if (dtype == 1) {
        u = static_cast<uint32_t>(h) << 16;
      } else {  // IEEE half -> float through the exponent re-bias; subnormals via scaling
        const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
        const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
        if (e == 0) {
          float f = static_cast<float>(m) * 0x1.0p-24f;
          std::memcpy(&u, &f, 4);
          u |= sign;
        } else if (e == 31) {
          u = sign | 0x7F800000u | (m << 13);
        } else {
          u = sign | ((e + 112u) << 23) | (m << 13);
        }
      }
      std::memcpy(&wide[i], &u, 4);
    }
  }
  tfc_oracle::stochastic_round(wide.data(), n, step, seed, seed_len, out);
  return 0;
#endif
}

#ifdef TFC_USE_REF
// ---- the reference's OWN op kernels (cc/kernels/range_coder_kernels.cc compiled verbatim) -------
// CreateRangeEncoder -> EntropyEncodeChannel / EntropyEncodeIndex -> EntropyEncodeFinalize and the
// decoder counterparts, run through the OpKernel shim on caller memory.
}  // extern "C" (helpers below are C++)
namespace {
namespace tf = tensorflow;
struct OpsHandle {
  std::vector<int32_t> lookup;      // the kernels keep spans into the lookup tensor
  std::vector<std::string> strings;  // decoder input
  tf::Tensor lookup_tensor, handle, encoded;
};
bool run_op(const char* name, tf::OpKernelContext* ctx, const std::map<std::string, int>& attrs = {}) {
  auto it = tfc_shim::registry().find({name, std::type_index(typeid(void))});
  if (it == tfc_shim::registry().end()) { g_err = std::string(name) + ": reference kernel not registered"; return false; }
  tf::OpKernelConstruction construction;
  construction.int_attrs = attrs;
  std::unique_ptr<tf::OpKernel> op(it->second(&construction));
  if (!construction.status.ok()) { g_err = construction.status.message(); return false; }
  op->Compute(ctx);
  if (!ctx->status.ok()) { g_err = ctx->status.message(); return false; }
  return true;
}
tf::TensorShape shape_of(const int64_t* d, int n) { return tf::TensorShape(std::vector<int64_t>(d, d + n)); }
void set_lookup(OpsHandle* h, const int32_t* lookup, int rank, int64_t rows, int64_t cols) {
  const int64_t n = rank == 1 ? cols : rows * cols;
  h->lookup.assign(lookup, lookup + n);
  h->lookup_tensor = tf::Tensor(h->lookup.data(), rank == 1 ? tf::TensorShape({cols}) : tf::TensorShape({rows, cols}));
}
}  // namespace
extern "C" {

void* SYM(ops_encoder_create)(const int64_t* handle_shape, int nh, const int32_t* lookup, int rank,
                              int64_t rows, int64_t cols) {
  auto h = std::make_unique<OpsHandle>();
  set_lookup(h.get(), lookup, rank, rows, cols);
  std::vector<int32_t> dims(handle_shape, handle_shape + nh);
  tf::OpKernelContext ctx;
  ctx.inputs.emplace_back(dims.data(), tf::TensorShape({nh}));
  ctx.inputs.push_back(h->lookup_tensor);
  if (!run_op("CreateRangeEncoder", &ctx)) return nullptr;
  h->handle = ctx.outputs.at(0);
  return h.release();
}

// value / index: int32 with shape[nd] = handle shape + suffix; index may be null (channel mode).
int SYM(ops_encoder_encode)(void* handle, const int32_t* value, const int32_t* index, const int64_t* shape, int nd) {
  auto* h = static_cast<OpsHandle*>(handle);
  tf::OpKernelContext ctx;
  ctx.inputs.push_back(h->handle);
  ctx.input_names = {"handle"};
  if (index != nullptr) {
    ctx.inputs.emplace_back(index, shape_of(shape, nd));
    ctx.input_names.push_back("index");
  }
  ctx.inputs.emplace_back(value, shape_of(shape, nd));
  ctx.input_names.push_back("value");
  if (!run_op(index ? "EntropyEncodeIndex" : "EntropyEncodeChannel", &ctx)) return 1;
  h->handle = ctx.outputs.at(0);
  return 0;
}

// Writes the strings back to back into blob (capacity cap) and their offsets[count + 1]; returns the
// total size (call again with a larger blob if it exceeds cap), -1 on error.
int64_t SYM(ops_encoder_finalize)(void* handle, uint8_t* blob, int64_t cap, int64_t* offsets) {
  auto* h = static_cast<OpsHandle*>(handle);
  if (h->strings.empty()) {
    tf::OpKernelContext ctx;
    ctx.inputs.push_back(h->handle);
    if (!run_op("EntropyEncodeFinalize", &ctx)) return -1;
    auto out = ctx.outputs.at(0).flat<tf::tstring>();
    for (int64_t i = 0; i < out.size(); ++i) h->strings.push_back(out(i));
  }
  int64_t total = 0;
  for (size_t i = 0; i < h->strings.size(); ++i) {
    offsets[i] = total;
    if (total + static_cast<int64_t>(h->strings[i].size()) <= cap)
      std::memcpy(blob + total, h->strings[i].data(), h->strings[i].size());
    total += static_cast<int64_t>(h->strings[i].size());
  }
  offsets[h->strings.size()] = total;
  return total;
}

void* SYM(ops_decoder_create)(const uint8_t* blob, const int64_t* offsets, const int64_t* handle_shape, int nh,
                              const int32_t* lookup, int rank, int64_t rows, int64_t cols) {
  auto h = std::make_unique<OpsHandle>();
  set_lookup(h.get(), lookup, rank, rows, cols);
  const tf::TensorShape shape = shape_of(handle_shape, nh);
  for (int64_t i = 0; i < shape.num_elements(); ++i)
    h->strings.emplace_back(reinterpret_cast<const char*>(blob) + offsets[i], static_cast<size_t>(offsets[i + 1] - offsets[i]));
  h->encoded = tf::Tensor(h->strings.data(), shape);
  tf::OpKernelContext ctx;
  ctx.inputs.push_back(h->encoded);
  ctx.inputs.push_back(h->lookup_tensor);
  if (!run_op("CreateRangeDecoder", &ctx)) return nullptr;
  h->handle = ctx.outputs.at(0);
  return h.release();
}

// out: int32 [handle shape + suffix]; index null (channel mode) or the same shape.
int SYM(ops_decoder_decode)(void* handle, const int64_t* suffix, int ns, const int32_t* index, int32_t* out) {
  auto* h = static_cast<OpsHandle*>(handle);
  std::vector<int32_t> dims(suffix, suffix + ns);
  tf::TensorShape full = h->handle.shape();
  full.AppendShape(shape_of(suffix, ns));
  tf::OpKernelContext ctx;
  ctx.inputs.push_back(h->handle);
  ctx.input_names = {"handle"};
  if (index != nullptr) {
    ctx.inputs.emplace_back(index, full);
    ctx.input_names.push_back("index");
  }
  ctx.inputs.emplace_back(dims.data(), tf::TensorShape({ns}));
  ctx.input_names.push_back("shape");
  if (!run_op(index ? "EntropyDecodeIndex" : "EntropyDecodeChannel", &ctx)) return 1;
  h->handle = ctx.outputs.at(0);
  auto got = ctx.outputs.at(1).flat<int32_t>();
  std::memcpy(out, got.data(), static_cast<size_t>(got.size()) * sizeof(int32_t));
  return 0;
}

int SYM(ops_decoder_finalize)(void* handle, uint8_t* ok) {
  auto* h = static_cast<OpsHandle*>(handle);
  tf::OpKernelContext ctx;
  ctx.inputs.push_back(h->handle);
  if (!run_op("EntropyDecodeFinalize", &ctx)) return 1;
  auto got = ctx.outputs.at(0).flat<bool>();
  for (int64_t i = 0; i < got.size(); ++i) ok[i] = got(i) ? 1 : 0;
  return 0;
}

void SYM(ops_free)(void* handle) { delete static_cast<OpsHandle*>(handle); }

// Legacy RangeEncode / RangeDecode ops (range_coding_kernels.cc + range_coding_kernels_util.cc).
int64_t SYM(op_range_encode)(const int16_t* data, const int64_t* data_shape, int nd, const int32_t* cdf,
                             const int64_t* cdf_shape, int nc, int precision, int debug_level, uint8_t* out,
                             int64_t cap) {
  tf::OpKernelContext ctx;
  ctx.inputs.emplace_back(data, shape_of(data_shape, nd));
  ctx.inputs.emplace_back(cdf, shape_of(cdf_shape, nc));
  if (!run_op("RangeEncode", &ctx, {{"precision", precision}, {"debug_level", debug_level}})) return -1;
  const std::string& s = ctx.outputs.at(0).scalar<tf::tstring>()();
  if (static_cast<int64_t>(s.size()) <= cap && !s.empty()) std::memcpy(out, s.data(), s.size());
  return static_cast<int64_t>(s.size());
}

int SYM(op_range_decode)(const uint8_t* bytes, int64_t nbytes, const int64_t* out_shape, int nd,
                         const int32_t* cdf, const int64_t* cdf_shape, int nc, int precision, int debug_level,
                         int16_t* out) {
  std::string encoded(reinterpret_cast<const char*>(bytes), static_cast<size_t>(nbytes));
  std::vector<int32_t> dims(out_shape, out_shape + nd);
  tf::OpKernelContext ctx;
  ctx.inputs.emplace_back(&encoded, tf::TensorShape());
  ctx.inputs.emplace_back(dims.data(), tf::TensorShape({nd}));
  ctx.inputs.emplace_back(cdf, shape_of(cdf_shape, nc));
  if (!run_op("RangeDecode", &ctx, {{"precision", precision}, {"debug_level", debug_level}})) return 1;
  auto got = ctx.outputs.at(0).flat<int16_t>();
  std::memcpy(out, got.data(), static_cast<size_t>(got.size()) * sizeof(int16_t));
  return 0;
}

// Deprecated UnboundedIndexRangeEncode / Decode ops (unbounded_index_range_coding_kernels.cc).
int64_t SYM(op_unbounded_index_range_encode)(const int32_t* data, const int32_t* index, int64_t total,
                                             const int32_t* cdf, int64_t rows, int64_t width,
                                             const int32_t* cdf_size, const int32_t* offset, int precision,
                                             int overflow_width, int debug_level, uint8_t* out, int64_t cap) {
  tf::OpKernelContext ctx;
  ctx.inputs.emplace_back(data, tf::TensorShape({total}));
  ctx.inputs.emplace_back(index, tf::TensorShape({total}));
  ctx.inputs.emplace_back(cdf, tf::TensorShape({rows, width}));
  ctx.inputs.emplace_back(cdf_size, tf::TensorShape({rows}));
  ctx.inputs.emplace_back(offset, tf::TensorShape({rows}));
  if (!run_op("UnboundedIndexRangeEncode", &ctx,
              {{"precision", precision}, {"overflow_width", overflow_width}, {"debug_level", debug_level}}))
    return -1;
  const std::string& s = ctx.outputs.at(0).flat<tf::tstring>()(0);
  if (static_cast<int64_t>(s.size()) <= cap && !s.empty()) std::memcpy(out, s.data(), s.size());
  return static_cast<int64_t>(s.size());
}

int SYM(op_unbounded_index_range_decode)(const uint8_t* bytes, int64_t nbytes, const int32_t* index,
                                         int64_t total, const int32_t* cdf, int64_t rows, int64_t width,
                                         const int32_t* cdf_size, const int32_t* offset, int precision,
                                         int overflow_width, int debug_level, int32_t* out) {
  std::string encoded(reinterpret_cast<const char*>(bytes), static_cast<size_t>(nbytes));
  tf::OpKernelContext ctx;
  ctx.inputs.emplace_back(&encoded, tf::TensorShape());
  ctx.inputs.emplace_back(index, tf::TensorShape({total}));
  ctx.inputs.emplace_back(cdf, tf::TensorShape({rows, width}));
  ctx.inputs.emplace_back(cdf_size, tf::TensorShape({rows}));
  ctx.inputs.emplace_back(offset, tf::TensorShape({rows}));
  if (!run_op("UnboundedIndexRangeDecode", &ctx,
              {{"precision", precision}, {"overflow_width", overflow_width}, {"debug_level", debug_level}}))
    return 1;
  auto got = ctx.outputs.at(0).flat<int32_t>();
  std::memcpy(out, got.data(), static_cast<size_t>(got.size()) * sizeof(int32_t));
  return 0;
}
#endif  // TFC_USE_REF

}  // extern "C"

// ---- timed round trip for bench.py's cpu_baseline ---------------------------
// Mirrors how the reference ops run: streams sharded over a persistent pool of
// `threads` workers (ThreadPool::ParallelFor over streams,
// range_coder_kernels.cc:212-267,377-424), output buffers reused between
// repetitions.  Times only the coding loops (encode: Encode+Finalize per
// stream; decode: ctor+Decode+Finalize per stream), per repetition.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>

namespace {
struct Gate {
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0, generation = 0, parties;
  explicit Gate(int n) : parties(n) {}
  void arrive() {
    std::unique_lock<std::mutex> lk(m);
    const int gen = generation;
    if (++waiting == parties) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != generation; });
    }
  }
};
}  // namespace

extern "C" int SYM(bench_roundtrip)(const int32_t* lookup, int rank, int64_t rows, int64_t cols,
                                    const int32_t* value, int64_t streams, int64_t elems,
                                    int threads, int reps, double* enc_seconds,
                                    double* dec_seconds, int64_t* total_bytes, int* all_ok) {
  std::vector<tfc_oracle::TableRef> tables;
  std::string err;
  if (!tfc_oracle::scan_tables(lookup, rank, rows, cols, &tables, &err)) { g_err = err; return 1; }
  threads = std::max(1, std::min<int>(threads, static_cast<int>(streams)));
  std::vector<std::string> sink(streams);
  std::vector<int32_t> decoded(static_cast<size_t>(streams) * elems);
  std::vector<uint8_t> okv(streams, 0);
  for (auto& s : sink) s.reserve(static_cast<size_t>(elems));
  Gate gate(threads + 1);
  const int64_t ntab = static_cast<int64_t>(tables.size());
  auto worker = [&](int64_t lo, int64_t hi) {
    for (int r = 0; r < reps; ++r) {
      gate.arrive();  // start encode
      for (int64_t s = lo; s < hi; ++s) {
        typename Core::Enc e{};
        std::string* out = &sink[s];
        out->clear();
        const int32_t* pv = value + s * elems;
        int64_t ch = 0;
        for (int64_t j = 0; j < elems; ++j, ++ch) {
          if (ch >= ntab) ch = 0;
          const tfc_oracle::TableRef& row = tables[ch];
          if (row.p[0] > 0) Core::encode(e, row.p[pv[j] + 1], row.p[pv[j] + 2], row.p[0], out);
          else StreamEncoder<Core>::escape_encode(e, out, row, pv[j]);
        }
        Core::flush(e, out);
      }
      gate.arrive();  // end encode
      gate.arrive();  // start decode
      for (int64_t s = lo; s < hi; ++s) {
        typename Core::Dec d;
        Core::open(d, reinterpret_cast<const uint8_t*>(sink[s].data()), sink[s].size());
        int32_t* po = decoded.data() + s * elems;
        int64_t ch = 0;
        for (int64_t j = 0; j < elems; ++j, ++ch) {
          if (ch >= ntab) ch = 0;
          const tfc_oracle::TableRef& row = tables[ch];
          po[j] = (row.p[0] > 0) ? Core::decode(d, row.p + 1, row.n - 1, row.p[0])
                                 : StreamDecoder<Core>::escape_decode(d, row);
        }
        okv[s] = Core::close(d) ? 1 : 0;
      }
      gate.arrive();  // end decode
    }
  };
  std::vector<std::thread> pool;
  for (int k = 0; k < threads; ++k)
    pool.emplace_back(worker, streams * k / threads, streams * (k + 1) / threads);
  using clk = std::chrono::steady_clock;
  for (int r = 0; r < reps; ++r) {
    auto t0 = clk::now();
    gate.arrive();
    gate.arrive();
    auto t1 = clk::now();
    gate.arrive();
    gate.arrive();
    auto t2 = clk::now();
    (void)t0;
    enc_seconds[r] = std::chrono::duration<double>(t1 - t0).count();
    dec_seconds[r] = std::chrono::duration<double>(t2 - t1).count();
  }
  for (auto& th : pool) th.join();
  int64_t total = 0;
  for (auto& s : sink) total += static_cast<int64_t>(s.size());
  *total_bytes = total;
  int ok = 1;
  for (int64_t s = 0; s < streams; ++s) ok &= okv[s];
  ok &= std::memcmp(decoded.data(), value, sizeof(int32_t) * static_cast<size_t>(streams) * elems) == 0;
  *all_ok = ok;
  return 0;
}
