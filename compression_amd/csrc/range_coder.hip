// Multi-stream range coder for gfx950 (MI355X): one 64-lane wavefront per code
// stream.
//
// What runs where
//   * The arithmetic of one stream is a strict chain (every interval update
//     needs the previous span), so parallelism = number of streams.  Each
//     stream gets a whole wave; the wave's 64 lanes do everything that is NOT
//     on the chain in parallel — coalesced symbol loads, CDF gathers from the
//     LDS-resident table, escape detection, the decoder's CDF search
//     (64 candidate symbols per compare + ballot), byte-window refills and
//     coalesced output stores — while the chain itself runs on wave-uniform
//     values that hipcc keeps in scalar registers.
//   * Output is append-only 16-bit digits (the delayed-carry scheme never
//     rewrites emitted bytes), collected 64 at a time in one VGPR via
//     v_writelane and flushed as one coalesced 128-byte store.
//
// Behaviour follows (bit-exact, checked by tests/ against oracle/):
//   cc/lib/range_coder.cc:37-307, cc/lib/range_coder.h:79-282,
//   cc/kernels/range_coder_kernels.cc:101-471.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <type_traits>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tfc_hip.h"
#include "common.h"
#include "range_coder_device.h"

namespace tfc {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}

double slow_call_threshold_ms() {
  static const double ms = [] {
    const char* e = std::getenv("TFC_SLOW_CALL_MS");
    return e ? std::atof(e) : 0.0;
  }();
  return ms;
}

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

SlowCall::SlowCall(const char* w, const char* f, int l)
    : what(w), file(f), line(l), t0(slow_call_threshold_ms() > 0.0 ? now_ms() : 0.0) {}

SlowCall::~SlowCall() {
  if (t0 == 0.0) return;
  const double dt = now_ms() - t0;
  if (dt >= slow_call_threshold_ms())
    std::fprintf(stderr, "[tfc slow call] %8.2f ms  %s:%d  %s  (%llu)\n", dt, file, line, what, detail);
}

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return 1;
}

// ---- optional kernel timing --------------------------------------------------
namespace {
struct ProfileEntry {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0;
  int64_t launches = 0;
};
bool g_profile_on = false;
std::mutex g_profile_mutex;      // encoders / decoders may run on several host threads
std::vector<ProfileEntry>& profile_table() {
  static std::vector<ProfileEntry> t;
  return t;
}
ProfileEntry& profile_entry(const char* name) {
  for (auto& e : profile_table())
    if (e.name == name) return e;
  profile_table().push_back(ProfileEntry{name, {}, 0, 0});
  return profile_table().back();
}
}  // namespace

bool profiling_enabled() { return g_profile_on; }

KernelTimer::KernelTimer(const char* n, hipStream_t s) : name(n), st(s), on(g_profile_on), slow(n, "launch scope", 0) {
  if (!on) return;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, st);
}

KernelTimer::~KernelTimer() {
  if (!on) return;
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lock(g_profile_mutex);
  profile_entry(name).pending.emplace_back(a, b);
}

}  // namespace tfc

using namespace tfc;

namespace {
std::atomic<int> g_chip_shared{0};      // tfc_set_chip_shared
}

// -> the previous value, so that nested users can restore it
extern "C" int tfc_set_chip_shared(int shared) {
  return g_chip_shared.exchange(shared ? 1 : 0, std::memory_order_relaxed);
}

namespace {
std::atomic<int>& default_mode() {
  static std::atomic<int> mode{[] {
    const char* e = std::getenv("TFC_DEFAULT_MODE");
    if (e && !std::strcmp(e, "latency")) return TFC_MODE_LATENCY;
    if (e && !std::strcmp(e, "throughput")) return TFC_MODE_THROUGHPUT;
    return TFC_MODE_AUTO;
  }()};
  return mode;
}
}  // namespace
extern "C" int tfc_set_default_mode(int mode) {
  if (mode != TFC_MODE_AUTO && mode != TFC_MODE_LATENCY && mode != TFC_MODE_THROUGHPUT)
    return fail("unknown mode %d", mode);
  default_mode().store(mode);
  return 0;
}
extern "C" int tfc_get_default_mode(void) { return default_mode().load(); }

extern "C" int tfc_pipe_counters(int64_t* launches, int64_t* fallback_blocks);

extern "C" void tfc_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_profile_mutex);
  g_profile_on = on != 0;
  if (on) {
    for (auto& e : profile_table()) {
      for (auto& p : e.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
      e.pending.clear();
      e.total_ms = 0;
      e.launches = 0;
    }
  }
}

extern "C" int tfc_profile_query(const char* kernel, double* total_ms, int64_t* launches) {
  std::lock_guard<std::mutex> lock(g_profile_mutex);
  ProfileEntry& e = profile_entry(kernel);
  for (auto& p : e.pending) {
    float ms = 0;
    if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      e.total_ms += ms;
      e.launches += 1;
    }
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  e.pending.clear();
  *total_ms = e.total_ms;
  *launches = e.launches;
  return 0;
}

// ===========================================================================
// Tables
// ===========================================================================

struct tfc_tables {
  std::vector<int32_t> host;       // raw lookup
  std::vector<int2> rows;          // (start of header, ints incl. header)
  DevBuf d_data, d_rows;
  DevBuf d_fast, d_rows_fast;      // encoder LDS image (uint16, entries scaled to 16-bit precision) + its rows
  DevBuf d_dec_image, d_dec_dir;   // decoder LDS image: d_fast + pad + pivot arrays; row directory
  int dec_words = 0;
  bool dec_fast_ok = false;
  // lane-per-stream kernels (range_lanes.h): one LDS image; the encoder uses its first lane_enc_bytes
  DevBuf d_lane_image;
  int lane_enc_bytes = 0, lane_dec_bytes = 0, lane_precision = 0;
  bool lanes_ok = false;
  // the pipelined decoder's COMPACT image (range_pipe.h, dec_chain_kernel<..., true>): bitmaps of every second bound at
  // PAIR resolution (one bit per two quotient values: half the bitmaps' bytes), and per row what dec_parse_kernel adds
  // to a raw entry to have the symbol
  DevBuf d_pair_image, d_pair_adjust;
  int pair_dec_bytes = 0;
  bool pairs_ok = false;
  int max_abs_prec = 0;
  bool any_escape = false;
  int64_t max_row = 0;
};

namespace {

int scan_row(const std::vector<int32_t>& v, int64_t end, int64_t* cur, std::vector<int2>* rows) {
  int64_t p = *cur;
  if (end < p + 3) return fail("CDF ended prematurely.");
  const int64_t head = p;
  const int64_t ap = std::llabs(static_cast<long long>(v[head]));
  if (ap < 1 || ap >= 17)
    return fail("precision=%lld not in range [1, 17)", static_cast<long long>(ap));
  const int32_t last = 1 << ap;
  ++p;
  if (v[p] != 0) return fail("CDF must start with 0.");
  do {
    ++p;
    if (p == end) return fail("CDF must end with 1 << precision.");
    if (v[p] < v[p - 1]) return fail("CDF must be monotonically increasing.");
  } while (v[p] != last);
  ++p;
  rows->push_back(make_int2(static_cast<int>(head), static_cast<int>(p - head)));
  while (p != end && v[p] == last) ++p;
  *cur = p;
  return 0;
}

}  // namespace

extern "C" int tfc_abi_version(void) { return TFC_ABI_VERSION; }
extern "C" const char* tfc_last_error(void) { return last_error().c_str(); }
extern "C" void tfc_free(void* p) { std::free(p); }

extern "C" int tfc_tables_create(const int32_t* lookup, int rank, int64_t rows, int64_t cols,
                                 void* stream, tfc_tables** out) {
  *out = nullptr;
  if (rank != 1 && rank != 2) return fail("`lookup` must be rank 1 or 2: rank=%d", rank);
  const int64_t total = rank == 1 ? cols : rows * cols;
  if (total >= (int64_t{1} << 31)) return fail("`lookup` too large");
  std::unique_ptr<tfc_tables> t(new tfc_tables);
  t->host.assign(lookup, lookup + total);
  if (rank == 1) {
    for (int64_t cur = 0; cur != total;)
      if (scan_row(t->host, total, &cur, &t->rows)) return 1;
  } else {
    for (int64_t cur = 0; cur != total;) {
      const int64_t row_end = cur + cols;
      if (scan_row(t->host, row_end, &cur, &t->rows)) return 1;
      if (cur != row_end) return fail("CDF must end with 1 << precision.");
    }
  }
  for (const int2& r : t->rows) {
    const int32_t sp = t->host[r.x];
    t->max_abs_prec = std::max(t->max_abs_prec, std::abs(sp));
    t->any_escape |= sp < 0;
    t->max_row = std::max<int64_t>(t->max_row, r.y);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  TFC_HIP(t->d_data.alloc(sizeof(int32_t) * std::max<int64_t>(total, 1), st));
  TFC_HIP(t->d_rows.alloc(sizeof(int2) * std::max<size_t>(t->rows.size(), 1), st));
  if (total)
    TFC_HIP(hipMemcpyAsync(t->d_data.p, t->host.data(), sizeof(int32_t) * total,
                           hipMemcpyHostToDevice, st));
  if (!t->rows.empty())
    TFC_HIP(hipMemcpyAsync(t->d_rows.p, t->rows.data(), sizeof(int2) * t->rows.size(),
                           hipMemcpyHostToDevice, st));
  {
    std::vector<int32_t> fast(t->host);
    for (const int2& r : t->rows) {
      const int sh = 16 - std::abs(t->host[r.x]);
      for (int i = 1; i < r.y; ++i) fast[r.x + i] = t->host[r.x + i] << sh;
    }
    // Encoder image: 16 bits per entry.  A scaled entry is at most 65536 and only ever used as
    // "lower" (< 65536) or as "upper - 1" (the coder call word holds upper - 1), so it is stored modulo
    // 2^16; whether a row has the escape symbol moves from the header's sign into the row directory.
    // Half the LDS of the int32 image = twice the encoder workgroups per CU.
    std::vector<uint16_t> fast16(std::max<int64_t>(total, 1));
    for (int64_t i = 0; i < total; ++i) fast16[i] = static_cast<uint16_t>(fast[i]);
    std::vector<int2> rows_fast(t->rows);
    for (int2& r : rows_fast)
      if (t->host[r.x] < 0) r.y |= static_cast<int>(0x80000000u);
    TFC_HIP(t->d_fast.alloc(sizeof(uint16_t) * fast16.size(), st));
    TFC_HIP(hipMemcpyAsync(t->d_fast.p, fast16.data(), sizeof(uint16_t) * fast16.size(), hipMemcpyHostToDevice, st));
    TFC_HIP(t->d_rows_fast.alloc(sizeof(int2) * std::max<size_t>(rows_fast.size(), 1), st));
    if (!rows_fast.empty())
      TFC_HIP(hipMemcpyAsync(t->d_rows_fast.p, rows_fast.data(), sizeof(int2) * rows_fast.size(),
                             hipMemcpyHostToDevice, st));
    // Decoder image: the scaled table, 64 words of padding (lanes past a row's end
    // read harmless data), then 64 pivots per wide row (> 64 symbols).
    std::vector<int32_t> image(fast);
    image.resize(image.size() + 64, 65536);
    std::vector<int4> dir;
    bool ok = true;
    for (const int2& r : t->rows) {
      const int nsym = r.y - 2;
      const int cdf0 = r.x + 1;
      const int chunk = (nsym + 63) / 64;
      if (chunk > 64) ok = false;
      // a zero-width FIRST symbol has the upper bound 0, whose "bound - 1" form wraps and matches every
      // offset: such tables keep the generic decoder (64-bit comparison, range_coder.h:204-222)
      if (nsym > 1 && fast[cdf0 + 1] == 0) ok = false;
      int4 d;
      d.y = cdf0;
      d.z = nsym | (std::max(chunk, 1) << 16);
      d.w = t->host[r.x] < 0 ? nsym - 1 : -1;
      if (chunk <= 1) {
        d.x = cdf0 + 1;
      } else {
        d.x = static_cast<int>(image.size());
        for (int j = 0; j < 64; ++j)
          image.push_back(fast[cdf0 + std::min((j + 1) * chunk, nsym)]);
      }
      dir.push_back(d);
    }
    image.resize(image.size() + 64, 65536);
    t->dec_words = static_cast<int>(image.size());
    t->dec_fast_ok = ok && !t->rows.empty();
    TFC_HIP(t->d_dec_image.alloc(sizeof(int32_t) * image.size(), st));
    TFC_HIP(t->d_dec_dir.alloc(sizeof(int4) * std::max<size_t>(dir.size(), 1), st));
    TFC_HIP(hipMemcpyAsync(t->d_dec_image.p, image.data(), sizeof(int32_t) * image.size(),
                           hipMemcpyHostToDevice, st));
    if (!dir.empty())
      TFC_HIP(hipMemcpyAsync(t->d_dec_dir.p, dir.data(), sizeof(int4) * dir.size(),
                             hipMemcpyHostToDevice, st));
    TFC_HIP(hipStreamSynchronize(st));
  }
  {
    // Image of the lane-per-stream kernels: directory (one entry per table), 16-bit scaled cdf entries
    // (modulo 2^16: only a row's last entry is 2^16), then per row the bitmap of its boundaries over
    // [0, 2^precision) and the running count of boundaries before each 64-bit word.  The counts of a row have one
    // more entry, for the word BEHIND the row (the next row's first, or one empty word behind the last row): the
    // quotient estimate of an offset at the very top of the span is 2^precision, and the pipelined decoder
    // (range_pipe.h) reads that word — of which its shift keeps bit 0 only, which it has cleared in its own copy —
    // instead of clamping.  Rows must be strictly increasing (rank = popcount) and share one precision (the
    // quotient scale is a kernel constant) — other tables keep the wave-per-stream kernels.
    const size_t ntab = t->rows.size();
    bool ok = ntab > 0;
    const int prec = ok ? std::abs(t->host[t->rows[0].x]) : 0;
    const size_t nw = std::max<size_t>(1, (size_t{1} << prec) / 64);
    size_t cdf_entries = 0;
    for (const int2& r : t->rows) {
      const int nsym = r.y - 2;
      if (std::abs(t->host[r.x]) != prec) ok = false;
      if (nsym > 32767) ok = false;
      for (int k = 1; ok && k <= nsym; ++k)
        if (t->host[r.x + 1 + k] <= t->host[r.x + k]) ok = false;
      cdf_entries += static_cast<size_t>(nsym + 1);
    }
    // behind the tables' rows: the uniform binary row {0, 1/2, 1} the pipelined decoder (range_pipe.h) decodes the
    // bits of an escape code from (range_coder_kernels.cc:449-471: DecodeLinearly on {0, 1, 2} at precision 1)
    cdf_entries += 3;
    const size_t words = (ntab + 1) * nw + 1, counts = (ntab + 1) * (nw + 1);
    // the directory repeats its first entries behind its end: a block of kEncCadence / kDecCadence steps
    // reads that many consecutive entries without a wrap test per step
    constexpr size_t kDirRepeat = 16;
    // (>= kEncCadence and kDecCadence of range_lanes.h, which asserts it)
    const size_t dir_bytes = sizeof(tfc::LaneRow) * (ntab + kDirRepeat + 1);
    const size_t cdf_bytes = (2 * cdf_entries + 15) & ~size_t{15};
    const size_t enc_bytes = dir_bytes + cdf_bytes;
    const size_t dec_bytes = enc_bytes + 8 * words + ((2 * counts + 15) & ~size_t{15});
    // (the encoder needs directory + cdf entries only; a decoder image over the CU's LDS keeps the decoder on the
    // wave-per-stream kernels — decoder_family checks — and the encoder may still run lane-per-stream)
    if (enc_bytes > 128 * 1024) ok = false;
    if (ok) {
      std::vector<uint8_t> image(dec_bytes, 0);
      tfc::LaneRow* dir = reinterpret_cast<tfc::LaneRow*>(image.data());
      uint16_t* cdf16 = reinterpret_cast<uint16_t*>(image.data() + dir_bytes);
      uint64_t* bits = reinterpret_cast<uint64_t*>(image.data() + enc_bytes);
      uint16_t* cum = reinterpret_cast<uint16_t*>(image.data() + enc_bytes + 8 * words);
      size_t ce = 0;
      // per word: the boundaries before it MINUS ONE, as int16 (rank - 1 = symbol: the decoder adds the
      // popcount inside the word and has the symbol; -1 for the first word), and once more behind the row
      auto count_row = [&](size_t i) {
        unsigned int run = 0;
        for (size_t w = 0; w <= nw; ++w) {
          cum[i * (nw + 1) + w] = static_cast<uint16_t>(static_cast<int16_t>(static_cast<int>(run) - 1));
          if (w < nw) run += static_cast<unsigned int>(__builtin_popcountll(bits[i * nw + w]));
        }
      };
      auto place_row = [&](tfc::LaneRow& d, size_t i) {
        d.cdf = static_cast<unsigned int>(dir_bytes + 2 * ce) - 2u;     // of cdf[0], minus 2: lo / hi of symbol s at + 2 s + 2 / + 4
        d.bits = static_cast<unsigned int>(enc_bytes + 8 * i * nw);
        d.cum = static_cast<unsigned int>(enc_bytes + 8 * words + 2 * i * (nw + 1));
      };
      for (size_t i = 0; i < ntab; ++i) {
        const int2 r = t->rows[i];
        const int32_t* cdf = &t->host[r.x + 1];
        const int nsym = r.y - 2;
        const bool esc = t->host[r.x] < 0;
        tfc::LaneRow& d = dir[i];
        place_row(d, i);
        d.info = static_cast<unsigned int>(esc ? nsym - 1 : nsym) | (esc ? 0x80000000u : 0u);
        for (int k = 0; k <= nsym; ++k) cdf16[ce + k] = static_cast<uint16_t>(cdf[k] << (16 - prec));
        for (int k = 0; k < nsym; ++k) bits[i * nw + (cdf[k] >> 6)] |= uint64_t{1} << (cdf[k] & 63);
        count_row(i);
        ce += static_cast<size_t>(nsym + 1);
      }
      for (size_t i = 0; i < kDirRepeat; ++i) dir[ntab + i] = dir[i % ntab];
      {
        // the binary row, at the table set's precision (the quotient scale is a kernel constant); at precision 0
        // (no such tables: precision >= 1) its two boundaries would coincide
        tfc::LaneRow& d = dir[ntab + kDirRepeat];
        place_row(d, ntab);
        d.info = 2u;
        cdf16[ce] = 0; cdf16[ce + 1] = 0x8000; cdf16[ce + 2] = 0;
        const unsigned int mid = 1u << (prec - 1);
        bits[ntab * nw] |= 1ull;
        bits[ntab * nw + (mid >> 6)] |= uint64_t{1} << (mid & 63);
        count_row(ntab);
      }
      TFC_HIP(t->d_lane_image.alloc(image.size(), st));
      TFC_HIP(hipMemcpyAsync(t->d_lane_image.p, image.data(), image.size(), hipMemcpyHostToDevice, st));
      TFC_HIP(hipStreamSynchronize(st));
      t->lane_enc_bytes = static_cast<int>(enc_bytes);
      t->lane_dec_bytes = static_cast<int>(dec_bytes);
      t->lane_precision = prec;
      t->lanes_ok = true;
    }
  }
  if (t->lanes_ok && t->lane_precision <= 15) {
    // Compact image of the pipelined decoder (round 6).  The boundary bitmaps are 2/3 of the lane image (98 of 154 KB for
    // BASELINE config 2's tables; bls2017's 192 x 128-symbol tables need 176 KB and do not fit a CU at all): one bit per
    // quotient value, because two bounds may be neighbours.  EVERY SECOND bound of a strictly increasing row is at least
    // two apart from the next one marked, so a bitmap of those needs one bit per PAIR of quotient values {2 j, 2 j + 1}
    // only — half the bytes — and its rank i says: bound k = 2 i + o is the last marked one whose pair is not behind
    // q's, hence  cdf[k] - 1 <= q < cdf[k + 2]  and the symbol is k - 1, k or k + 1.  The step (TFC_PDEC_STEP_H) reads
    // the four entries cdf[k - 1 .. k + 2] with one ds_read2_b32 and settles it with two comparisons of the quotient
    // (t0 = [q >= cdf[k]], t1 = [q >= cdf[k + 1]]: lower / upper bound by four selects, symbol = k - 1 + t0 + t1) —
    // verified by the exact interval test like every estimate.
    //   * which bounds are marked: those with k = o (mod 2), o = symbols of the row (mod 2) — then the last marked one is
    //     k = n - 2 and the row's END (2^16, stored as 0: the only entry a comparison must not meet) is only ever an
    //     upper bound;
    //   * in front of cdf[0] a row has two zero entries (an even row's first window starts at cdf[-1]; a row of one symbol
    //     takes k = -1: both comparisons true, the window slides to (cdf[0], cdf[1])), and rows are placed so that the
    //     window of rank i starts at a multiple of four bytes: the directory's cdf pointer is that address for i = 0;
    //   * the step's raw entry is 2 i + t0 + t1 = symbol - (o - 1): dec_parse_kernel adds the row's o - 1 (pair_adjust).
    // Layout: directory (final form: what dec_chain_kernel<..., false> makes of its copy of the lane image), entries,
    // bitmaps (row i at word i * nw2, one spare word behind the last row), counts (nw2 + 1 per row).
    const size_t ntab = t->rows.size();
    const int prec = t->lane_precision, sh = 16 - prec;
    const size_t npairs = size_t{1} << (prec - 1);
    const size_t nw2 = std::max<size_t>(1, npairs / 64);
    constexpr size_t kDirRepeat = 16;
    const size_t dir_bytes = sizeof(tfc::LaneRow) * (ntab + kDirRepeat + 1);
    std::vector<uint16_t> entries;            // all rows: [pad ... 0, 0, cdf[0] ... cdf[n]]
    std::vector<int> adjust(ntab);
    std::vector<uint64_t> bits((ntab + 1) * nw2 + 1, 0);
    std::vector<uint16_t> cum((ntab + 1) * (nw2 + 1), 0);
    std::vector<size_t> window0(ntab + 1);    // entry index of the window of rank 0: cdf[o - 1]
    auto add_row = [&](size_t i, const int32_t* cdf, int nsym, unsigned int carry) {
      // (a row of ONE symbol has no bound a comparison may meet — cdf[1] is its end: its window starts two entries in
      // front of cdf[0], in the pad: o = -1, both comparisons true)
      const int o = nsym == 1 ? -1 : (nsym & 1);
      // entry index of cdf[0] such that the byte address of cdf[o - 1] (dir_bytes is a multiple of 16) is a multiple of 4
      size_t at0 = entries.size() + 2;
      if (((at0 + o - 1) & 1) != 0) ++at0;
      entries.resize(at0, 0);
      for (int k = 0; k <= nsym; ++k) entries.push_back(static_cast<uint16_t>(static_cast<unsigned int>(cdf[k]) << sh));
      window0[i] = at0 + o - 1;
      // (the first marked bound, k = o, is left out: the rank is then the index i of the last marked bound, and the
      // quotients below it share its window)
      for (int k = o + 2; k < nsym; k += 2) {
        const unsigned int pair = static_cast<unsigned int>(cdf[k]) >> 1;
        bits[i * nw2 + (pair >> 6)] |= uint64_t{1} << (pair & 63);
      }
      unsigned int run = carry;
      for (size_t w = 0; w <= nw2; ++w) {
        cum[i * (nw2 + 1) + w] = static_cast<uint16_t>(run);
        if (w < nw2) run += static_cast<unsigned int>(__builtin_popcountll(bits[i * nw2 + w]));
      }
    };
    for (size_t i = 0; i < ntab; ++i) {
      const int2 r = t->rows[i];
      const int nsym = r.y - 2;
      adjust[i] = (nsym == 1 ? -1 : (nsym & 1)) - 1;
      add_row(i, &t->host[r.x + 1], nsym, 0u);
    }
    {
      // the binary row of an escape code's bits {0, 1/2, 1}: two symbols, bound 0 marked, window (pad, 0, 1/2, end):
      // t0 = 1, t1 = the bit; its ranks carry 0x4000, so that 2 i + t0 + t1 = 0x8001 + bit — the raw entry of a bit row
      // as it is stored (the bit is entry >> 1 & 1; 0xFFFF stays the mark of a row a lane sat out)
      const int32_t bin[3] = {0, 1 << (prec - 1), 1 << prec};
      add_row(ntab, bin, 2, 0x4000u);
    }
    entries.resize(entries.size() + 2, 0);    // (the last window's fourth entry)
    const size_t cdf_bytes = (2 * entries.size() + 15) & ~size_t{15};
    const size_t bits_off = dir_bytes + cdf_bytes;
    const size_t cum_off = bits_off + 8 * bits.size();
    const size_t total = cum_off + ((2 * cum.size() + 15) & ~size_t{15});
    if (total <= 160 * 1024) {
      std::vector<uint8_t> image(total, 0);
      tfc::LaneRow* dir = reinterpret_cast<tfc::LaneRow*>(image.data());
      auto entry_of = [&](size_t i, unsigned int limit, bool esc, unsigned int esclo, bool binary) {
        tfc::LaneRow d;
        // (the step addresses the window as cdf + 4 i; the binary row's ranks carry 0x4000)
        d.cdf = static_cast<unsigned int>(dir_bytes + 2 * window0[i]) - (binary ? 0x10000u : 0u);
        d.info = (limit & 0x7FFFu) | (esc ? 0x8000u : 0u) | ((0xFFFFu - esclo) << 16);
        d.bits = static_cast<unsigned int>(bits_off + 8 * i * nw2) - 8u;
        d.cum = static_cast<unsigned int>(cum_off + 2 * i * (nw2 + 1)) - 2u;
        return d;
      };
      for (size_t i = 0; i < ntab; ++i) {
        const int2 r = t->rows[i];
        const bool esc = t->host[r.x] < 0;
        const int nsym = r.y - 2;
        // limit: plain symbols (= the escape symbol's index), as in the lane image; ESCLO: the escape symbol's lower bound
        // on the tables' own scale (0xFFFF, which no quotient reaches at precision <= 15, for a row without one)
        dir[i] = entry_of(i, static_cast<unsigned int>(esc ? nsym - 1 : nsym), esc,
                          esc ? static_cast<unsigned int>(t->host[r.x + 1 + nsym - 1]) : 0xFFFFu, false);
      }
      for (size_t i = 0; i < kDirRepeat; ++i) dir[ntab + i] = dir[i % ntab];
      dir[ntab + kDirRepeat] = entry_of(ntab, 2u, false, 1u << (prec - 1), true);
      std::memcpy(image.data() + dir_bytes, entries.data(), 2 * entries.size());
      std::memcpy(image.data() + bits_off, bits.data(), 8 * bits.size());
      std::memcpy(image.data() + cum_off, cum.data(), 2 * cum.size());
      TFC_HIP(t->d_pair_image.alloc(image.size(), st));
      TFC_HIP(hipMemcpyAsync(t->d_pair_image.p, image.data(), image.size(), hipMemcpyHostToDevice, st));
      TFC_HIP(t->d_pair_adjust.alloc(sizeof(int) * ntab, st));
      TFC_HIP(hipMemcpyAsync(t->d_pair_adjust.p, adjust.data(), sizeof(int) * ntab, hipMemcpyHostToDevice, st));
      TFC_HIP(hipStreamSynchronize(st));
      t->pair_dec_bytes = static_cast<int>(total);
      t->pairs_ok = true;
    }
  }
  TFC_HIP(hipStreamSynchronize(st));
  *out = t.release();
  return 0;
}

extern "C" int64_t tfc_tables_count(const tfc_tables* t) { return static_cast<int64_t>(t->rows.size()); }
extern "C" void tfc_tables_destroy(tfc_tables* t) { delete t; }

// ===========================================================================
// Kernels
// ===========================================================================

namespace tfc {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * 64;
// Tables up to this many bytes are staged in LDS (160 KiB per CU on gfx950).
constexpr size_t kLdsTableBytes = 144 * 1024;

struct DecRow;
struct TableView {
  const int32_t* data;
  const uint16_t* fast16;     // encoder LDS image: entries scaled to 16-bit precision, modulo 2^16
  const int2* rows_fast;      // (offset, length | escape row << 31) per table, for that image
  const int32_t* dec_image;   // decoder LDS image (see tfc_tables_create)
  const struct DecRow* dec_dir;
  int dec_words;
  const int2* rows;
  int ntab;
  int total;
};

// Where symbols come from (encode) / go to (decode).
// Loads and stores of the kernels' tensor arguments go through global-address-space pointers: the functors below travel
// inside job arrays indexed at run time, where hipcc cannot tell that their pointers are global and emits FLAT
// instructions — which count on lgkmcnt as well as vmcnt, so that every wait for an LDS read in the same loop becomes a
// wait for the loop's stores too (seen in dec_parse_kernel and enc_expand_kernel, round 6).
#ifndef TFC_AS1
#define TFC_AS1 __attribute__((address_space(1)))
#endif
template <typename T>
__device__ inline T tfc_gload(const T* p) {
  static_assert(sizeof(T) == 2 || sizeof(T) == 4, "16- and 32-bit elements");
  if constexpr (sizeof(T) == 2) {
    return __builtin_bit_cast(T, *reinterpret_cast<const TFC_AS1 unsigned short*>((const TFC_AS1 void*)p));
  } else {
    return __builtin_bit_cast(T, *reinterpret_cast<const TFC_AS1 unsigned int*>((const TFC_AS1 void*)p));
  }
}
template <typename T>
__device__ inline void tfc_gstore(T* p, T v) {
  static_assert(sizeof(T) == 2 || sizeof(T) == 4, "16- and 32-bit elements");
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<TFC_AS1 unsigned short*>((TFC_AS1 void*)p) = __builtin_bit_cast(unsigned short, v);
  } else {
    *reinterpret_cast<TFC_AS1 unsigned int*>((TFC_AS1 void*)p) = __builtin_bit_cast(unsigned int, v);
  }
}

struct SymInt32 {          // plain int32 symbols
  const int32_t* value;
  __device__ int32_t load(int64_t pos, int /*table*/) const { return tfc_gload(value + pos); }
  // split form for kernels that request an element before they know its table
  __device__ int32_t raw(int64_t pos) const { return tfc_gload(value + pos); }
  __device__ int32_t quant(int32_t r, int /*table*/) const { return r; }
  __device__ const int32_t* base() const { return value; }
  using raw_type = int32_t;
};


template <typename T>
__device__ inline float to_float(T v);
template <> __device__ inline float to_float<float>(float v) { return v; }
template <> __device__ inline float to_float<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ inline float to_float<__half>(__half v) { return __half2float(v); }
template <typename T>
__device__ inline T from_float(float v);
template <> __device__ inline float from_float<float>(float v) { return v; }
template <> __device__ inline __hip_bfloat16 from_float<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ inline __half from_float<__half>(float v) { return __float2half(v); }

// Fused quantisation: sym = int32(rint(y - qoff[t])) - cdf_offset[t].  The
// subtraction happens in the bottleneck dtype like the reference's
// `bottleneck -= offset` (continuous_batched.py:375-378); rintf is
// round-half-to-even like tf.round.
template <typename T>
struct SymQuant {
  const T* y;
  const float* qoffset;        // may be null
  const int32_t* cdf_offset;
  __device__ int32_t load(int64_t pos, int table) const { return quant(tfc_gload(y + pos), table); }
  __device__ T raw(int64_t pos) const { return tfc_gload(y + pos); }
  __device__ const T* base() const { return y; }
  using raw_type = T;
  __device__ int32_t quant(T r, int table) const {
    float f = to_float<T>(r);
    if (qoffset) f = to_float<T>(from_float<T>(f - to_float<T>(from_float<T>(tfc_gload(qoffset + table)))));
    return static_cast<int32_t>(rintf(f)) - tfc_gload(cdf_offset + table);
  }
};

struct EncParams {
  TableView tab;
  const int32_t* index;     // null => channel mode
  int64_t streams;
  int64_t elems;
  // outputs of the counting pass / inputs of the coding pass
  unsigned long long* calls;        // [streams]
  unsigned long long* first_error;  // [1], linear position of the first range error
  // coding pass
  uint4* state;                     // [streams] base, span_m1, pend_digit, pend_bytes
  uint8_t* chunk;
  const long long* chunk_off;       // [streams + 1]
  unsigned int* chunk_len;          // [streams]
  unsigned int* overflow_flag;      // [1]
  // deferred-error handles (no read-back between the counting and the coding pass): the coding pass
  // itself looks at the counting pass's verdict — guard[0] = first range error (~0 = none), guard[1] =
  // bytes the per-stream slabs need — and appends nothing if there was an error or the slab the host
  // sized without knowing the data is too small (then it raises the overflow flag).  null: checked by the host.
  const unsigned long long* guard;
  unsigned long long cap_total;
};

// true: this call appends nothing (see EncParams::guard); the stream's piece gets length 0
__device__ inline bool enc_guard_skips(const EncParams& p, int64_t s, int lane) {
  if (!p.guard) return false;
  const unsigned long long err = p.guard[0], need = p.guard[1];
  if (err == ~0ull && need <= p.cap_total) return false;
  if (lane == 0) {
    p.chunk_len[s] = 0u;
    if (need > p.cap_total) atomicOr(p.overflow_flag, 1u);
  }
  return true;
}

// Per-element classification shared by the counting and the coding pass.
struct Call {
  int32_t lo16, hi16;   // interval scaled to 16-bit precision
  int32_t gamma;        // > 0 => escape follows
  int32_t neg;
  int32_t bad;          // 1: index out of range, 2: value out of range
};

template <bool NORMALISED, typename TabFn>
__device__ inline Call classify_impl(const TabFn& T, const int2 row, int32_t v) {
  Call c;
  c.gamma = 0;
  c.neg = 0;
  c.bad = 0;
  const int32_t sp = T(row.x);
  const int32_t prec = sp < 0 ? -sp : sp;
  int32_t sym = v;
  if (sp > 0) {
    if (v < 0 || v >= row.y - 2) {
      c.bad = 2;
      sym = 0;
    }
  } else {
    const int32_t vmax = row.y - 3;
    if (v < 0) {
      c.neg = 1;
      c.gamma = -v;
      sym = vmax;
    } else if (v >= vmax) {
      c.gamma = v - vmax + 1;
      sym = vmax;
    }
  }
  const int sh = NORMALISED ? 0 : 16 - prec;
  c.lo16 = T(row.x + 1 + sym) << sh;
  c.hi16 = T(row.x + 2 + sym) << sh;
  return c;
}

template <typename TabFn>
__device__ inline Call classify(const TabFn& T, const int2 row, int32_t v) {
  return classify_impl<false>(T, row, v);
}
template <typename TabFn>
__device__ inline Call classify_normalised(const TabFn& T, const int2 row, int32_t v) {
  return classify_impl<true>(T, row, v);
}

// The same classification on the encoder's 16-bit LDS image (tfc_tables_create): hi16 is the upper
// bound modulo 2^16 — its only use is "upper - 1" in the call word, which is right either way.
__device__ inline Call classify_fast(const uint16_t* tab, const int2 row, int32_t v) {
  Call c;
  c.gamma = 0;
  c.neg = 0;
  c.bad = 0;
  const int len = row.y & 0x7FFFFFFF;
  int32_t sym = v;
  if (row.y >= 0) {
    if (v < 0 || v >= len - 2) {
      c.bad = 2;
      sym = 0;
    }
  } else {
    const int32_t vmax = len - 3;
    if (v < 0) {
      c.neg = 1;
      c.gamma = -v;
      sym = vmax;
    } else if (v >= vmax) {
      c.gamma = v - vmax + 1;
      sym = vmax;
    }
  }
  c.lo16 = tab[row.x + 1 + sym];
  c.hi16 = tab[row.x + 2 + sym];
  return c;
}

__device__ inline int escape_calls(int32_t gamma) {
  // 1 + 2*floor(log2 gamma) bits for the Elias-gamma code, plus one sign bit
  // (range_coder_kernels.cc:304-321).
  const int nb = 31 - __clz(gamma);
  return 2 * nb + 2;
}

// Counting + validation pass: fully parallel and HBM-bound (one coalesced read of the
// symbols).  Only the row's symbol count and escape flag are needed, so they are staged in
// LDS ((nsym << 1) | escape per table) instead of gathering table entries.
constexpr int kCountTile = 4096;      // elements of one stream per workgroup
constexpr int kCountLdsRows = 8192;

template <typename Src>
__global__ void __launch_bounds__(256) enc_count_kernel(EncParams p, Src src) {
  // rowinfo[min(ntab, kCountLdsRows)], sized at launch: this kernel runs beside the coding kernels of
  // other steps, and LDS it does not need is LDS their workgroups cannot get
  extern __shared__ int rowinfo[];
  __shared__ unsigned int part[4];
  const bool in_lds = p.tab.ntab <= kCountLdsRows;
  if (in_lds) {
    for (int i = threadIdx.x; i < p.tab.ntab; i += 256) {
      const int2 r = p.tab.rows[i];
      rowinfo[i] = ((r.y - 2) << 1) | (p.tab.data[r.x] < 0 ? 1 : 0);
    }
    __syncthreads();
  }
  const int64_t tiles = (p.elems + kCountTile - 1) / kCountTile;
  const int64_t s = blockIdx.x / tiles;
  const int64_t j0 = (blockIdx.x % tiles) * kCountTile;
  const unsigned int ntab = static_cast<unsigned int>(p.tab.ntab);
  unsigned int ch = static_cast<unsigned int>((j0 + threadIdx.x) % ntab);
  const unsigned int step = 256u % ntab;
  unsigned int calls = 0;
  for (int k = 0; k < kCountTile / 256; ++k) {
    const int64_t j = j0 + k * 256 + threadIdx.x;
    if (j < p.elems) {
      const int64_t pos = s * p.elems + j;
      int t = static_cast<int>(ch);
      bool bad = false;
      if (p.index) {
        t = p.index[pos];
        if (t < 0 || t >= p.tab.ntab) { bad = true; t = 0; }
      }
      int info;
      if (in_lds) {
        info = rowinfo[t];
      } else {
        const int2 r = p.tab.rows[t];
        info = ((r.y - 2) << 1) | (p.tab.data[r.x] < 0 ? 1 : 0);
      }
      const int nsym = info >> 1;
      const int32_t v = src.load(pos, t);
      unsigned int c = 1;
      if (info & 1) {
        const int vmax = nsym - 1;                  // last interval is the escape symbol
        const int gamma = v < 0 ? -v : (v >= vmax ? v - vmax + 1 : 0);
        if (gamma > 0) c += escape_calls(gamma);
      } else if (v < 0 || v >= nsym) {
        bad = true;
      }
      if (bad) atomicMin(p.first_error, static_cast<unsigned long long>(pos));
      calls += c;
    }
    ch += step;
    if (ch >= ntab) ch -= ntab;
  }
  for (int off = 32; off > 0; off >>= 1) calls += __shfl_down(calls, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = calls;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int tot = part[0] + part[1] + part[2] + part[3];
    atomicAdd(&p.calls[s], static_cast<unsigned long long>(tot));
    atomicAdd(p.first_error + 2, static_cast<unsigned long long>(tot));   // status[2]: calls of all streams
  }
}

// calls[s] -> byte capacity 2*calls + 4 rounded to 16, exclusive scan -> off.
__global__ void enc_offsets_kernel(const unsigned long long* calls, const uint4* state,
                                   int held_digits, int64_t streams,
                                   long long* off, unsigned long long* total) {
  // single block; streams is usually 1e2..1e5
  __shared__ long long carry;
  __shared__ long long tmp[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < streams; base += 1024) {
    const int64_t i = base + threadIdx.x;
    long long cap = 0;
    if (i < streams) {
      // 2 bytes per coder call, plus (fast kernels) the digits the previous calls held back
      long long digits = static_cast<long long>(calls[i]);
      if (held_digits) digits += 1 + static_cast<long long>(state[i].w);
      cap = ((2 * digits + 4 + 15) / 16) * 16;
    }
    tmp[threadIdx.x] = cap;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      long long v = threadIdx.x >= d ? tmp[threadIdx.x - d] : 0;
      __syncthreads();
      tmp[threadIdx.x] += v;
      __syncthreads();
    }
    if (i < streams) off[i] = carry + tmp[threadIdx.x] - cap;
    __syncthreads();
    if (threadIdx.x == 1023) carry += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    off[streams] = carry;
    *total = static_cast<unsigned long long>(carry);
  }
}

// 16-bit digit collector: 64 digits per VGPR, one coalesced store per flush.
struct DigitSink {
  uint8_t* dst;        // 2-byte aligned
  unsigned int nbytes; // bytes already stored
  unsigned int cap;
  int n;               // digits waiting in reg
  int reg;
  unsigned int overflow;
};

__device__ inline void sink_flush(DigitSink& o, int lane) {
  if (o.nbytes + 2u * o.n > o.cap) {
    o.overflow = 1;
  } else if (lane < o.n) {
    const unsigned int d = static_cast<unsigned int>(o.reg);
    const unsigned short be = static_cast<unsigned short>(((d & 0xFF) << 8) | ((d >> 8) & 0xFF));
    reinterpret_cast<unsigned short*>(o.dst + o.nbytes)[lane] = be;
  }
  o.nbytes += 2u * o.n;
  o.n = 0;
}

__device__ inline void sink_put(DigitSink& o, unsigned int digit, int lane) {
  o.reg = tfc_writelane(static_cast<int>(digit), o.n, o.reg);
  ++o.n;
  if (o.n == 64) sink_flush(o, lane);
}

// One interval update on wave-uniform state; [lo16, hi16) / 2^16.
__device__ inline void enc_update(EncoderState& st, unsigned int lo16, unsigned int hi16,
                                  DigitSink& o, int lane) {
  const unsigned long long span = static_cast<unsigned long long>(st.span_m1) + 1;
  const unsigned int a = static_cast<unsigned int>((span * lo16) >> 16);
  const unsigned int b = static_cast<unsigned int>(((span * hi16) >> 16) - 1);
  st.base += a;
  st.span_m1 = b - a;
  const bool wrapped = st.base < a;
  if (static_cast<unsigned int>(st.base + st.span_m1) < st.base) {
    if ((st.span_m1 >> 16) == 0) {
      st.base <<= 16;
      st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
      st.pend_bytes += 2;
    }
    return;
  }
  if (st.pend_digit != 0) {
    unsigned int d = st.pend_digit;
    unsigned int fill = 0;
    if (!wrapped) {
      d -= 1;
      fill = 0xFFFFu;
    }
    sink_put(o, d, lane);
    for (unsigned int k = 0; k < st.pend_bytes; k += 2) sink_put(o, fill, lane);
    st.pend_digit = 0;
    st.pend_bytes = 0;
  }
  if ((st.span_m1 >> 16) == 0) {
    const unsigned int top = st.base >> 16;
    st.base <<= 16;
    st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
    if (st.base <= static_cast<unsigned int>(st.base + st.span_m1)) {
      sink_put(o, top, lane);
    } else {
      st.pend_digit = top + 1;
    }
  }
}

template <bool LDS_TAB, typename Src>
__global__ void __launch_bounds__(kBlock) enc_kernel(EncParams p, Src src) {
  extern __shared__ int32_t lds_tab[];
  if (LDS_TAB) {
    for (int i = threadIdx.x; i < p.tab.total; i += kBlock) lds_tab[i] = p.tab.data[i];
    __syncthreads();
  }
  auto T = [&](int i) -> int32_t { return LDS_TAB ? lds_tab[i] : p.tab.data[i]; };

  const int lane = threadIdx.x & 63;
  const int64_t s = __builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
  if (s >= p.streams) return;
  if (enc_guard_skips(p, s, lane)) return;

  const uint4 st0 = p.state[s];
  EncoderState st;
  st.base = __builtin_amdgcn_readfirstlane(st0.x);
  st.span_m1 = __builtin_amdgcn_readfirstlane(st0.y);
  st.pend_digit = __builtin_amdgcn_readfirstlane(st0.z);
  st.pend_bytes = __builtin_amdgcn_readfirstlane(st0.w);

  DigitSink o;
  const long long off0 = p.chunk_off[s];
  o.dst = p.chunk + off0;
  o.cap = static_cast<unsigned int>(p.chunk_off[s + 1] - off0);
  o.nbytes = 0;
  o.n = 0;
  o.reg = 0;
  o.overflow = 0;

  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
    // ---- vector phase: 64 symbols at once --------------------------------
    const int64_t j = j0 + lane;
    Call c;
    c.lo16 = 0; c.hi16 = 0; c.gamma = 0; c.neg = 0; c.bad = 0;
    if (j < p.elems) {
      const int64_t pos = s * p.elems + j;
      int t = p.index ? p.index[pos] : static_cast<int>(j % p.tab.ntab);
      t = min(max(t, 0), p.tab.ntab - 1);  // range errors were reported by the counting pass
      const int2 row = p.tab.rows[t];
      c = classify(T, row, src.load(pos, t));
    }
    const unsigned long long esc_mask = __ballot(c.gamma > 0);
    const int cnt = static_cast<int>(min<int64_t>(64, p.elems - j0));
    // ---- serial phase: the chain -----------------------------------------
    for (int n = 0; n < cnt; ++n) {
      const unsigned int lo = __builtin_amdgcn_readlane(c.lo16, n);
      const unsigned int hi = __builtin_amdgcn_readlane(c.hi16, n);
      enc_update(st, lo, hi, o, lane);
      if ((esc_mask >> n) & 1) {
        const int g = __builtin_amdgcn_readlane(c.gamma, n);
        const int neg = __builtin_amdgcn_readlane(c.neg, n);
        int nb = 31 - __clz(g);             // floor(log2 g)
        for (int k = 0; k < nb; ++k) enc_update(st, 0, 0x8000u, o, lane);   // zeros
        for (int k = nb; k >= 0; --k) {
          const unsigned int bit = (g >> k) & 1;
          enc_update(st, bit << 15, (bit + 1) << 15, o, lane);
        }
        enc_update(st, static_cast<unsigned int>(neg) << 15,
                   static_cast<unsigned int>(neg + 1) << 15, o, lane);
      }
    }
  }
  sink_flush(o, lane);
  if (lane == 0) {
    p.state[s] = make_uint4(st.base, st.span_m1, st.pend_digit, st.pend_bytes);
    p.chunk_len[s] = o.nbytes;
    if (o.overflow) atomicOr(p.overflow_flag, 1u);
  }
}

}  // namespace tfc
#include "range_encoder_fast.h"
namespace tfc {

// ---- finalize -------------------------------------------------------------

struct ChunkRef {
  const uint8_t* data;
  const long long* off;      // start of every stream's piece, or null: stream s starts at s * stride
  const unsigned int* len;
  long long stride;
};
__device__ inline const uint8_t* chunk_piece(const ChunkRef& c, int64_t s) {
  return c.data + (c.off ? c.off[s] : s * c.stride);
}
// The pieces of a handle travel to the finalize kernels BY VALUE (kernel arguments are captured at
// launch): an asynchronous copy from a host vector would still be reading it after a stream-ordered
// finalize has returned.  Handles with more encode calls than this take a synchronising copy.
constexpr int kInlineChunks = 8;
struct ChunkList {
  ChunkRef inline_refs[kInlineChunks];
  const ChunkRef* more;      // all of them, when there are more than kInlineChunks
  int n;
  __device__ const ChunkRef& operator[](int i) const { return more ? more[i] : inline_refs[i]; }
};

// Tail of every stream per RangeEncoder::Finalize (range_coder.cc:266-307); one
// thread per stream.  tail[s] = {head bytes (<= 2), 0xFFFF digits to insert,
// end bytes (<= 2)}; also sums the stream's total length.
//   generic kernels:        state = (base, span-1, delay digit + 1, delayed bytes)
//   fast and lane kernels:  state = (base, span-1, valid<<31 | held digit, held 0xFFFF run)
struct Tail {
  unsigned char head[2];
  unsigned char end[2];
  unsigned int nhead, nend, run;
};

__device__ inline void enc_tail_one(const uint4* state, int64_t streams, const ChunkList& chunks,
                                    int nchunks, int fast_state, Tail* tail, long long* length) {
  const int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (s >= streams) return;
  const uint4 st = state[s];
  Tail t;
  t.nhead = t.nend = t.run = 0;
  t.head[0] = t.head[1] = t.end[0] = t.end[1] = 0;
  const bool state1 = static_cast<unsigned int>(st.x + st.y) < st.x;   // carry still undecided
  unsigned int delay = 0;                                              // reference's delay_ & 0xFFFF
  if (fast_state) {
    if (state1) {
      delay = (st.z & 0xFFFFu) + 1u;
    } else if (st.z >> 31) {
      t.head[0] = (st.z >> 8) & 0xFF;
      t.head[1] = st.z & 0xFF;
      t.nhead = 2;
      t.run = st.w;
    }
  } else {
    delay = st.z;
  }
  if (delay != 0) {
    t.end[0] = (delay >> 8) & 0xFF;
    t.nend = 1;
    if ((delay & 0xFF) != 0) { t.end[1] = delay & 0xFF; t.nend = 2; }
  } else if (st.x != 0) {
    const unsigned int top = st.x + st.y;
    const unsigned int r24 = ((st.x - 1) >> 24) + 1;
    if (r24 <= (top >> 24)) {
      t.end[0] = r24 & 0xFF;
      t.nend = 1;
    } else {
      const unsigned int r16 = ((st.x - 1) >> 16) + 1;
      t.end[0] = (r16 >> 8) & 0xFF;
      t.nend = 1;
      if ((r16 & 0xFF) != 0) { t.end[1] = r16 & 0xFF; t.nend = 2; }
    }
  }
  tail[s] = t;
  long long len = static_cast<long long>(t.nhead) + 2ll * t.run + t.nend;
  for (int c = 0; c < nchunks; ++c) len += chunks[c].len[s];
  length[s] = len;
}
__global__ void enc_tail_kernel(const uint4* state, int64_t streams, const ChunkList chunks,
                                int nchunks, int fast_state, Tail* tail, long long* length) {
  enc_tail_one(state, streams, chunks, nchunks, fast_state, tail, length);
}

// Finalize of n single-call handles of the lane family in three launches (blockIdx.y = handle): their
// temporaries, offsets and packed blobs live at a fixed stride in ONE allocation.
constexpr int kMaxFinalizeJobs = 64;
struct FinalizeJobs {
  int64_t streams;
  int n;
  uint8_t* base;             // job k: base + k * job_bytes = [Tail x streams][length x streams][offsets x (streams + 1)][blob]
  long long job_bytes, length_off, offsets_off, blob_off;
  struct { const uint4* state; ChunkRef chunk; } job[kMaxFinalizeJobs];
};
__device__ inline Tail* fin_tail(const FinalizeJobs& f, int k) { return reinterpret_cast<Tail*>(f.base + k * f.job_bytes); }
__device__ inline long long* fin_length(const FinalizeJobs& f, int k) { return reinterpret_cast<long long*>(f.base + k * f.job_bytes + f.length_off); }
__device__ inline long long* fin_offsets(const FinalizeJobs& f, int k) { return reinterpret_cast<long long*>(f.base + k * f.job_bytes + f.offsets_off); }
__global__ void enc_tail_many_kernel(const FinalizeJobs f) {
  const int k = blockIdx.y;
  ChunkList one;
  one.inline_refs[0] = f.job[k].chunk;
  one.more = nullptr;
  one.n = 1;
  enc_tail_one(f.job[k].state, f.streams, one, 1, 1, fin_tail(f, k), fin_length(f, k));     // lane family: held-digit state
}

__device__ inline void scan_lengths_one(const long long* length, int64_t streams, long long* off) {
  __shared__ long long carry;
  __shared__ long long tmp[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < streams; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const long long v0 = i < streams ? length[i] : 0;
    tmp[threadIdx.x] = v0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      long long v = threadIdx.x >= d ? tmp[threadIdx.x - d] : 0;
      __syncthreads();
      tmp[threadIdx.x] += v;
      __syncthreads();
    }
    if (i < streams) off[i] = carry + tmp[threadIdx.x] - v0;
    __syncthreads();
    if (threadIdx.x == 1023) carry += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[streams] = carry;
}
__global__ void scan_lengths_kernel(const long long* length, int64_t streams, long long* off) {
  scan_lengths_one(length, streams, off);
}
__global__ void scan_lengths_many_kernel(const FinalizeJobs f) {
  scan_lengths_one(fin_length(f, blockIdx.x), f.streams, fin_offsets(f, blockIdx.x));
}

// Wave-cooperative byte copy with 4-byte stores: dst is brought to dword alignment, the
// (generally misaligned) source is read as aligned dwords and funnel-shifted.  The source
// slabs carry >= 4 bytes of slack behind every stream, so reading one dword past n is safe.
__device__ inline void wave_copy(uint8_t* dst, const uint8_t* src, unsigned int n, int lane) {
  const unsigned int head = min(n, static_cast<unsigned int>((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3));
  if (lane < static_cast<int>(head)) dst[lane] = src[lane];
  dst += head; src += head; n -= head;
  const unsigned int nd = n >> 2;
  const unsigned int sh = static_cast<unsigned int>(reinterpret_cast<uintptr_t>(src) & 3) * 8;
  const unsigned int* s32 = reinterpret_cast<const unsigned int*>(reinterpret_cast<uintptr_t>(src) & ~uintptr_t{3});
  unsigned int* d32 = reinterpret_cast<unsigned int*>(dst);
  if (sh == 0) {
    for (unsigned int i = lane; i < nd; i += 64) d32[i] = s32[i];
  } else {
    for (unsigned int i = lane; i < nd; i += 64) d32[i] = (s32[i] >> sh) | (s32[i + 1] << (32 - sh));
  }
  const unsigned int rem = n & 3;
  if (lane < static_cast<int>(rem)) dst[4 * nd + lane] = src[4 * nd + lane];
}

// One wave per stream: copy the stream's chunk pieces then its tail.
__device__ inline void enc_pack_one(int64_t streams, const ChunkList& chunks, int nchunks, const Tail* tail,
                                    const long long* off, uint8_t* blob) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (s >= streams) return;
  uint8_t* dst = blob + off[s];
  long long done = 0;
  for (int c = 0; c < nchunks; ++c) {
    const unsigned int n = chunks[c].len[s];
    wave_copy(dst + done, chunk_piece(chunks[c], s), n, lane);
    done += n;
  }
  const Tail t = tail[s];
  if (lane < static_cast<int>(t.nhead)) dst[done + lane] = t.head[lane];
  done += t.nhead;
  for (unsigned long long i = lane; i < 2ull * t.run; i += 64) dst[done + i] = 0xFF;
  done += 2ll * t.run;
  if (lane < static_cast<int>(t.nend)) dst[done + lane] = t.end[lane];
}
__global__ void __launch_bounds__(kBlock) enc_pack_kernel(int64_t streams, const ChunkList chunks,
                                                         int nchunks, const Tail* tail,
                                                         const long long* off, uint8_t* blob) {
  enc_pack_one(streams, chunks, nchunks, tail, off, blob);
}
__global__ void __launch_bounds__(kBlock) enc_pack_many_kernel(const FinalizeJobs f) {
  const int k = blockIdx.y;
  ChunkList one;
  one.inline_refs[0] = f.job[k].chunk;
  one.more = nullptr;
  one.n = 1;
  enc_pack_one(f.streams, one, 1, fin_tail(f, k), fin_offsets(f, k), f.base + k * f.job_bytes + f.blob_off);
}

// ---- decoder --------------------------------------------------------------

struct DecParams {
  TableView tab;
  const int32_t* index;
  int64_t streams;
  int64_t elems;
  const uint8_t* blob;
  const long long* off;            // [streams + 1]
  uint4* state;                    // base, span_m1, window, pulls
  unsigned long long* first_error; // index range error
  const unsigned int* only_flagged; // dec_fast_kernel: if set, decode only streams with a nonzero flag
  const unsigned int* job_guard;    // null, or a flag: the whole launch decodes only if it is set (fallback of range_pipe.h
                                    // for tables whose lane-per-stream image does not fit a CU)
  int blocks_after_escape;         // dec_fast_kernel: batches decoded as checked 8-symbol blocks after an escape
};

// 64 upcoming big-endian digits of the stream, one per lane.
struct DigitWindow {
  const uint8_t* src;
  long long len;        // stream length in bytes
  unsigned int pulls;   // digits consumed so far (including the two of the ctor)
  unsigned int base;    // digit index held by lane 0
  int reg;
};

__device__ inline void window_load(DigitWindow& w, int lane) {
  const long long b = 2ll * (static_cast<long long>(w.base) + lane);
  unsigned int hi = b < w.len ? w.src[b] : 0u;
  unsigned int lo = b + 1 < w.len ? w.src[b + 1] : 0u;
  w.reg = static_cast<int>((hi << 8) | lo);
}

__device__ inline unsigned int window_pull(DigitWindow& w, int lane) {
  if (w.pulls - w.base >= 64u) {
    w.base = w.pulls;
    window_load(w, lane);
  }
  const unsigned int d = __builtin_amdgcn_readlane(w.reg, static_cast<int>(w.pulls - w.base));
  ++w.pulls;
  return d;
}

struct DecoderState {
  unsigned int base, span_m1, window;
};

__device__ inline void dec_narrow(DecoderState& st, unsigned int lo, unsigned int hi, int prec,
                                  DigitWindow& w, int lane) {
  const unsigned long long span = static_cast<unsigned long long>(st.span_m1) + 1;
  const unsigned int a = static_cast<unsigned int>((span * lo) >> prec);
  const unsigned int b = static_cast<unsigned int>(((span * hi) >> prec) - 1);
  st.base += a;
  st.span_m1 = b - a;
  if ((st.span_m1 >> 16) == 0) {
    st.base <<= 16;
    st.span_m1 = (st.span_m1 << 16) | 0xFFFFu;
    st.window = (st.window << 16) | window_pull(w, lane);
  }
}

// Decode one binary digit with the uniform cdf {0,1,2}, precision 1
// (DecodeLinearly, range_coder.h:193-202 with the call at
// range_coder_kernels.cc:449-471).
__device__ inline int dec_bit(DecoderState& st, DigitWindow& w, int lane) {
  const unsigned long long span = static_cast<unsigned long long>(st.span_m1) + 1;
  const unsigned long long target =
      (static_cast<unsigned long long>(static_cast<unsigned int>(st.window - st.base)) + 1) << 1;
  const int bit = (target <= span) ? 0 : 1;
  dec_narrow(st, bit, bit + 1, 1, w, lane);
  return bit;
}

// Finds the first symbol k with target <= span * cdf[k + 1]; all 64 lanes test
// one candidate each.  `cdf0` = position of cdf[0], `ncdf` = number of cdf
// entries.  On damaged input (no candidate matches) the last symbol is taken.
template <typename TabFn>
__device__ inline int dec_symbol(const TabFn& T, DecoderState& st, int cdf0, int ncdf, int prec,
                                 DigitWindow& w, int lane) {
  const unsigned long long span = static_cast<unsigned long long>(st.span_m1) + 1;
  const unsigned long long target =
      (static_cast<unsigned long long>(static_cast<unsigned int>(st.window - st.base)) + 1) << prec;
  const int nsym = ncdf - 1;
  int sym = nsym - 1;
  unsigned int lo = 0, hi = 0;
  bool found = false;
  for (int c0 = 0; c0 < nsym; c0 += 64) {
    const int k = c0 + lane;
    unsigned int lo_k = 0, hi_k = 0;
    if (k < nsym) {
      lo_k = static_cast<unsigned int>(T(cdf0 + k));
      hi_k = static_cast<unsigned int>(T(cdf0 + k + 1));
    }
    const bool pred = (k < nsym) && (target <= span * hi_k);
    const unsigned long long m = __ballot(pred);
    if (m != 0) {
      const int kk = __builtin_ctzll(m);
      lo = __builtin_amdgcn_readlane(static_cast<int>(lo_k), kk);
      hi = __builtin_amdgcn_readlane(static_cast<int>(hi_k), kk);
      sym = c0 + kk;
      found = true;
      break;
    }
  }
  if (!found) {
    lo = static_cast<unsigned int>(T(cdf0 + nsym - 1));
    hi = static_cast<unsigned int>(T(cdf0 + nsym));
  }
  dec_narrow(st, lo, hi, prec, w, lane);
  return sym;
}

struct OutInt32 {
  int32_t* out;
  __device__ void store(int64_t pos, int /*table*/, int32_t sym) const { tfc_gstore(out + pos, sym); }
  // split form for kernels that collect several elements per store
  using elem = int32_t;
  __device__ int32_t make(int /*table*/, int32_t sym) const { return sym; }
  __device__ int32_t* ptr() const { return out; }
};

template <typename T>
struct OutDequant {
  T* y;
  const float* qoffset;
  const int32_t* cdf_offset;
  __device__ void store(int64_t pos, int table, int32_t sym) const { tfc_gstore(y + pos, make(table, sym)); }
  using elem = T;
  __device__ T make(int table, int32_t sym) const {
    // outputs = cast(symbols + cdf_offset, dtype) (+ quantization_offset)
    T v = from_float<T>(static_cast<float>(sym + tfc_gload(cdf_offset + table)));
    if (qoffset) v = from_float<T>(to_float<T>(v) + to_float<T>(from_float<T>(tfc_gload(qoffset + table))));
    return v;
  }
  __device__ T* ptr() const { return y; }
};

template <bool LDS_TAB, typename Dst>
__global__ void __launch_bounds__(kBlock) dec_kernel(DecParams p, Dst dst) {
  extern __shared__ int32_t lds_tab[];
  if (p.job_guard != nullptr && *p.job_guard == 0u) return;
  if (LDS_TAB) {
    for (int i = threadIdx.x; i < p.tab.total; i += kBlock) lds_tab[i] = p.tab.data[i];
    __syncthreads();
  }
  auto T = [&](int i) -> int32_t { return LDS_TAB ? lds_tab[i] : p.tab.data[i]; };

  const int lane = threadIdx.x & 63;
  const int64_t s = __builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
  if (s >= p.streams) return;

  const uint4 st0 = p.state[s];
  DecoderState st;
  st.base = __builtin_amdgcn_readfirstlane(st0.x);
  st.span_m1 = __builtin_amdgcn_readfirstlane(st0.y);
  st.window = __builtin_amdgcn_readfirstlane(st0.z);
  DigitWindow w;
  const long long o0 = p.off[s];
  w.src = p.blob + o0;
  w.len = p.off[s + 1] - o0;
  w.pulls = __builtin_amdgcn_readfirstlane(st0.w);
  w.base = w.pulls;
  window_load(w, lane);

  for (int64_t j0 = 0; j0 < p.elems; j0 += 64) {
    const int64_t j = j0 + lane;
    int t = 0;
    if (j < p.elems) {
      const int64_t pos = s * p.elems + j;
      if (p.index) {
        t = p.index[pos];
        if (t < 0 || t >= p.tab.ntab) {
          atomicMin(p.first_error, static_cast<unsigned long long>(pos));
          t = 0;
        }
      } else {
        t = static_cast<int>(j % p.tab.ntab);
      }
    }
    const int cnt = static_cast<int>(min<int64_t>(64, p.elems - j0));
    int outv = 0;
    for (int n = 0; n < cnt; ++n) {
      const int tn = __builtin_amdgcn_readlane(t, n);
      const int2 row = p.tab.rows[tn];
      const int start = __builtin_amdgcn_readfirstlane(row.x);
      const int nints = __builtin_amdgcn_readfirstlane(row.y);
      const int sp = __builtin_amdgcn_readfirstlane(T(start));
      const int prec = sp < 0 ? -sp : sp;
      int sym = dec_symbol(T, st, start + 1, nints - 1, prec, w, lane);
      if (sp < 0 && sym == nints - 3) {
        int nb = 0;
        // bound the unary prefix so damaged input cannot spin forever
        while (nb < 31 && dec_bit(st, w, lane) == 0) ++nb;
        int v = 1 << nb;
        while (--nb >= 0) v |= dec_bit(st, w, lane) << nb;
        const int neg = dec_bit(st, w, lane);
        sym = neg ? -v : v + (nints - 3) - 1;
      }
      outv = tfc_writelane(sym, n, outv);
    }
    if (j < p.elems) dst.store(s * p.elems + j, t, outv);
  }
  if (lane == 0) p.state[s] = make_uint4(st.base, st.span_m1, st.window, w.pulls);
}

}  // namespace tfc
#include "range_decoder_fast.h"
#include "range_lanes.h"
#include "range_pipe.h"
namespace tfc {

// Reads the first four bytes of every stream (RangeDecoder ctor,
// range_coder.h:79-83).
__device__ inline void dec_open_one(const uint8_t* blob, const long long* off, int64_t streams,
                                    uint4* state, unsigned long long* status) {
  const int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (s == 0) *status = ~0ull;       // no index error met yet
  if (s >= streams) return;
  const uint8_t* src = blob + off[s];
  const long long len = off[s + 1] - off[s];
  unsigned int w = 0;
  for (int i = 0; i < 4; ++i) w = (w << 8) | (i < len ? src[i] : 0u);
  state[s] = make_uint4(0u, 0xFFFFFFFFu, w, 2u);
}
__global__ void dec_open_kernel(const uint8_t* blob, const long long* off, int64_t streams,
                                uint4* state, unsigned long long* status) {
  dec_open_one(blob, off, streams, state, status);
}
// n decoders at once (blockIdx.y = handle): their control blocks ([status 16 B][state]) sit at a fixed
// stride in one allocation
constexpr int kMaxDecoderJobs = 64;
struct DecoderJobs {
  int64_t streams;
  int n;
  uint8_t* ctl;
  long long ctl_bytes;
  struct { const uint8_t* blob; const long long* off; uint8_t* ok; } job[kMaxDecoderJobs];
};
__global__ void dec_open_many_kernel(const DecoderJobs f) {
  const int k = blockIdx.y;
  uint8_t* ctl = f.ctl + k * f.ctl_bytes;
  dec_open_one(f.job[k].blob, f.job[k].off, f.streams, reinterpret_cast<uint4*>(ctl + 16),
               reinterpret_cast<unsigned long long*>(ctl));
}

// RangeDecoder::Finalize (range_coder.h:144-169).
__device__ inline void dec_close_one(const uint4* state, const long long* off, int64_t streams,
                                     uint8_t* ok) {
  const int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (s >= streams) return;
  const uint4 st = state[s];
  const long long len = off[s + 1] - off[s];
  bool good;
  if (2ll * st.w < len) {
    good = false;
  } else {
    const unsigned int top = st.x + st.y;
    if (st.x == 0 || top < st.x) {
      good = st.z == 0;
    } else {
      const int sh = (((st.x - 1) >> 24) < (top >> 24)) ? 24 : 16;
      const unsigned int r = ((st.x - 1) >> sh) + 1;
      good = (r << sh) == st.z;
    }
  }
  ok[s] = good ? 1 : 0;
}
__global__ void dec_close_kernel(const uint4* state, const long long* off, int64_t streams,
                                 uint8_t* ok) {
  dec_close_one(state, off, streams, ok);
}
__global__ void dec_close_many_kernel(const DecoderJobs f) {
  const int k = blockIdx.y;
  dec_close_one(reinterpret_cast<const uint4*>(f.ctl + k * f.ctl_bytes + 16), f.job[k].off, f.streams, f.job[k].ok);
}

// Initial coder state of every stream, plus the handle's status words (first error position = none,
// value, index, filled flag) and overflow flag.
__device__ inline void fill_state_one(uint4* state, int64_t n, uint4 v, unsigned long long* status,
                                      unsigned int* oflag) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) state[i] = v;
  if (i == 0) {
    status[0] = ~0ull;
    status[1] = status[2] = status[3] = 0ull;
    *oflag = 0u;
  }
}
__global__ void fill_state_kernel(uint4* state, int64_t n, uint4 v, unsigned long long* status,
                                  unsigned int* oflag) {
  fill_state_one(state, n, v, status, oflag);
}
// n encoder control blocks ([status 32 B][overflow flag .. 64 B][state]) at a fixed stride (blockIdx.y)
__global__ void fill_state_many_kernel(uint8_t* ctl, long long ctl_bytes, int64_t n, uint4 v) {
  uint8_t* c = ctl + blockIdx.y * ctl_bytes;
  fill_state_one(reinterpret_cast<uint4*>(c + 64), n, v, reinterpret_cast<unsigned long long*>(c),
                 reinterpret_cast<unsigned int*>(c + 32));
}

}  // namespace tfc

// ===========================================================================
// Host side: encoder
// ===========================================================================

struct EncChunk {
  DevBuf data, off, len;
  long long stride = 0;         // lane kernels: stream s starts at s * stride (off unused)
  unsigned int* len_p = nullptr;    // the lengths: `len`, or the tail of `data` (lane kernels: one allocation)
  size_t data_bytes = 0;
};

// Kernel family of a handle (fixed at its first coding call; the families keep different state
// encodings between calls).
enum Family { kGeneric = 0, kFast = 1, kLanes = 2 };

struct tfc_encoder {
  const tfc_tables* tables = nullptr;
  int64_t streams = 0;
  int mode = TFC_MODE_AUTO;
  int family = -1;              // Family, chosen by the first encode call
  bool fast = false;            // wave-per-stream fast kernels are possible for these tables
  int fast_waves = 0;           // waves per workgroup for the fast kernels
  size_t fast_lds = 0;
  bool deferred = false;        // range errors are reported by finalize / status instead of encode
  bool poisoned = false;        // a range error was reported: the streams are no longer meaningful
  int64_t elems_last = 0;       // geometry of the call the recorded error belongs to
  bool indexed_last = false;
  DevBuf ctl;                   // one allocation behind the three views below ...
  std::shared_ptr<DevBuf> ctl_group;      // ... or a slice of the allocation shared by a create_many group
  DevView state;                // uint4 [streams]
  DevView oflag;                // unsigned int: a stream outgrew its output slab
  DevView status;               // u64[4]: first error position, its value, its index, filled flag
  std::vector<EncChunk> chunks;
  // results
  bool finalized = false;
  bool total_known = false;
  DevBuf blob_own, offsets_own;
  std::shared_ptr<DevBuf> result_group;   // finalize_device_many: blob / offsets are slices of one allocation
  DevView blob, offsets;
  int64_t total = 0;
  int64_t blob_capacity = 0;    // bytes behind `blob`
  // Every entry point that takes a stream retargets the handle's buffers to it: they are released in the
  // order of the stream that used them LAST, not of the one they were allocated under.
  void touch(hipStream_t s) {
    ctl.touch(s);
    if (ctl_group) ctl_group->touch(s);
    for (auto& c : chunks) { c.data.touch(s); c.off.touch(s); c.len.touch(s); }
    blob_own.touch(s);
    offsets_own.touch(s);
    if (result_group) result_group->touch(s);
  }
};

namespace {

size_t table_lds_bytes(const tfc_tables* t) {
  const size_t b = t->host.size() * sizeof(int32_t);
  return b <= kLdsTableBytes ? b : 0;
}

TableView view_of(const tfc_tables* t) {
  TableView v;
  v.data = t->d_data.as<int32_t>();
  v.fast16 = t->d_fast.as<uint16_t>();
  v.rows_fast = t->d_rows_fast.as<int2>();
  v.dec_image = t->d_dec_image.as<int32_t>();
  v.dec_dir = t->d_dec_dir.as<DecRow>();
  v.dec_words = t->dec_words;
  v.rows = t->d_rows.as<int2>();
  v.ntab = static_cast<int>(t->rows.size());
  v.total = static_cast<int>(t->host.size());
  return v;
}

inline size_t lds_request(size_t need) { return need; }

// Waves (= code streams) of a workgroup of the wave-per-stream kernels, sharing one LDS copy of the tables: one wave
// per SIMD of a CU whenever there are that many streams.  Fewer, fuller workgroups: a CU that hosts even one coder wave
// is lost to a convolution workgroup — which wants all four SIMDs' registers — for as long as that wave runs, so with
// model steps in flight 128 streams as 64 workgroups of 2 waves cost the transforms twice the CUs of 32 workgroups of
// 4 (C4: 47.2 -> 45.2 ms per step; 8 or 16 waves per workgroup, two or four per SIMD: 51.2 / 63.5; measured in round 3).
inline int64_t waves_wanted(int64_t streams) {
  // tfc_set_chip_shared(1) — the caller keeps other kernels in flight beside the coder's: 512 streams and more get two
  // waves per SIMD, i.e. 64 instead of 128 CUs under a bls2017 batch (C1: 12.0 -> 11.1 ms per step with 8 steps in
  // flight; a lone 512-stream call is 1.4x slower that way, 6.5 -> 8.8 ms, hence not the default); for 128 streams the
  // same packing costs more than it frees (profiles/r03_notes.md)
  const bool shared = g_chip_shared.load(std::memory_order_relaxed) != 0;
  const int64_t limit = shared && streams >= 512 ? 8 : 4;
  return std::min<int64_t>(limit, std::max<int64_t>(1, streams));
}

// Which kernels a handle uses.  The lane-per-stream kernels issue ~1 vector instruction per symbol
// and 64 streams against 15-28 and 1 for the wave-per-stream kernels, but a lone call takes
// elems x ~400 cycles against elems x ~90-160: they win when the streams of all calls in flight
// outnumber the chip's SIMDs several times.  AUTO decides on the stream count of the handle alone;
// a caller that keeps many calls in flight asks for TFC_MODE_THROUGHPUT.
int select_family(const tfc_tables* t, int mode, int64_t streams, int64_t elems, bool fast_ok) {
  // the lane kernels address a stream's bytes with 32 bits, and their LDS plan (image + one wave's
  // staging planes, any variant of the encoder) has to fit the CU
  const bool lanes_ok = t->lanes_ok && elems < (int64_t{1} << 29) &&
                        ((t->lane_enc_bytes + 1023) & ~1023) + EncWaveLds<int32_t>::kBytes <= 160 * 1024;
  if (mode == TFC_MODE_AUTO) mode = tfc_get_default_mode();
  if (mode == TFC_MODE_THROUGHPUT && lanes_ok) return kLanes;
  if (mode == TFC_MODE_AUTO && lanes_ok && streams >= 4096) return kLanes;
  // tfc_set_chip_shared(1): a batch of 256 streams and more goes to the lane kernels where they fit — a wave per 64
  // streams on a handful of CUs instead of a wave per stream on 64-128 CUs that the convolutions then cannot use
  if (mode == TFC_MODE_AUTO && lanes_ok && streams >= 256 && g_chip_shared.load(std::memory_order_relaxed) != 0)
    return kLanes;
  return fast_ok ? kFast : kGeneric;
}

// Blocks a wave of the lane kernels lets lanes wait in front of an escape code before it takes the codes of all
// waiting lanes together (LaneArgs::defer; measured in round 3: 3).
inline int lanes_escape_defer() { return 3; }

// Streams per workgroup of the lane kernels: one wave (64 streams) until the launch fills the chip,
// so that a 512-stream call spreads over eight CUs (and XCDs), then more waves behind each LDS image.
inline int lanes_block(int64_t streams) {
  const int64_t waves = ceil_div(streams, 64);
  const int64_t per = std::min<int64_t>(8, std::max<int64_t>(1, waves / 256));
  return static_cast<int>(64 * per);
}

// Slab bytes per stream that a lane-encoder call can never exceed, from the table headers alone (no
// counting pass, no read-back).  A call emits at most one 16-bit digit; and a call of probability P
// shrinks the span by at most 1 + log2(1/P) bits, a digit leaves every 16 bits: a symbol of a
// precision-p row costs <= p + 1 bits, each of the <= 64 binary calls of an escape code
// 1 + 2^-15 bits.  `carry` covers digits an earlier call left delayed.
// With escape rows that worst case is ~10 bytes per symbol against ~0.5 produced, so a call first
// runs with the no-escape bound (16 bits per symbol on average: exceeded only by streams that are
// mostly long escape codes); a stream that outgrows it raises the handle's overflow flag, and the call
// is repeated with the worst-case slab (handles that synchronise) or reported (deferred handles).
unsigned int lanes_slab_bytes(const tfc_tables* t, int64_t elems, bool worst) {
  const long long carry = 80;
  long long bytes = 2 * elems;
  if (t->any_escape && worst) bytes = (elems * (t->max_abs_prec + 1 + 65) + 7) / 8 + 8;
  bytes = (bytes + carry + 15) & ~15ll;
  return static_cast<unsigned int>(std::min<long long>(bytes, 0xFFFFFFF0ll));
}

// Records value / index of the first range error next to its position (stream-ordered, so the
// inputs are still alive whatever the caller does after the encode call returns).  One thread per job.
template <typename Src>
struct EncErrJobs {
  int64_t elems;
  int ntab, n;
  struct { unsigned long long* status; Src src; const int32_t* index; } job[EncLaneJobs<Src>::kMax];
};
template <typename Src>
__global__ void enc_error_kernel(const EncErrJobs<Src> jobs) {
  if (static_cast<int>(threadIdx.x) >= jobs.n) return;
  unsigned long long* status = jobs.job[threadIdx.x].status;
  const Src src = jobs.job[threadIdx.x].src;
  const int32_t* index = jobs.job[threadIdx.x].index;
  const unsigned long long pos = status[0];
  if (pos == ~0ull || status[3] != 0ull) return;
  int t = static_cast<int>((pos % static_cast<unsigned long long>(jobs.elems)) % static_cast<unsigned long long>(jobs.ntab));
  long long ix = 0;
  if (index) {
    ix = index[pos];
    t = (ix < 0 || ix >= jobs.ntab) ? 0 : static_cast<int>(ix);
  }
  status[1] = static_cast<unsigned long long>(static_cast<long long>(src.load(static_cast<int64_t>(pos), t)));
  status[2] = static_cast<unsigned long long>(ix);
  status[3] = 1ull;
}

// Builds the reference's range-error text for the element at `pos`.
int range_error_text(const tfc_tables* t, bool has_index, int32_t idx, int64_t channel,
                     int32_t value) {
  const int64_t ntab = static_cast<int64_t>(t->rows.size());
  if (has_index && (idx < 0 || idx >= ntab))
    return fail("index=%d not in range [0, %lld)", idx, static_cast<long long>(ntab));
  const int2 row = t->rows[has_index ? idx : channel];
  return fail("value=%d not in range [0, %d)", value, row.y - 2);
}

int encoder_error(tfc_encoder* e, const unsigned long long* host_status) {
  e->poisoned = true;
  const unsigned long long pos = host_status[0];
  const int64_t ch = static_cast<int64_t>((pos % static_cast<unsigned long long>(std::max<int64_t>(e->elems_last, 1))) %
                                          e->tables->rows.size());
  return range_error_text(e->tables, e->indexed_last, static_cast<int32_t>(static_cast<long long>(host_status[2])),
                          ch, static_cast<int32_t>(static_cast<long long>(host_status[1])));
}

// Which image the pipelined decoder's chain runs on — 1 "full" (the lane-per-stream kernels' image, one bit per quotient
// value), 2 "pairs" (the compact image, TFC_PDEC_STEP_H: half the bitmaps, ~10 % more cycles per row), 0: full where it
// fits and its chain waves all find a CU, else pairs — and the chain waves per workgroup (0: by launch size).
// tfc_set_pipe_format / TFC_PIPE_FORMAT, TFC_PIPE_WAVES: an A/B and test switch.
std::atomic<int>& pipe_format_value() {
  static std::atomic<int> v{[] {
    const char* e = std::getenv("TFC_PIPE_FORMAT");
    if (!e) return 0;
    if (std::strcmp(e, "full") == 0) return 1;
    if (std::strcmp(e, "pairs") == 0) return 2;
    return 0;
  }()};
  return v;
}
std::atomic<int>& pipe_waves_value() {
  static std::atomic<int> v{[] {
    const char* e = std::getenv("TFC_PIPE_WAVES");
    return e ? std::max(0, std::atoi(e)) : 0;
  }()};
  return v;
}
inline int pipe_format() { return pipe_format_value().load(std::memory_order_relaxed); }
inline int pipe_waves_env() { return pipe_waves_value().load(std::memory_order_relaxed); }

// The pipelined kernels of range_pipe.h in front of the lane-per-stream kernels (TFC_PIPE=0: the latter alone —
// an A/B switch for measurements, not a product setting).
inline bool pipe_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("TFC_PIPE");
    return !e || std::atoi(e) != 0;
  }();
  return on;
}
// TFC_PIPE_NOFALLBACK=1 (tools/pipe_probe.py): the lane-per-stream kernel is NOT launched behind the pipelined ones, so
// that a job they gave up on shows as wrong output instead of being covered up
inline bool pipe_fallback_launch() {
  static const bool on = [] {
    const char* e = std::getenv("TFC_PIPE_NOFALLBACK");
    return !e || std::atoi(e) == 0;
  }();
  return on;
}
std::atomic<long long> g_pipe_launches{0};

// TFC_PIPE_OVERLAP (default 2): how the encoder's chain kernel is launched relative to the expansion that feeds it.
//   0  behind it, on the caller's stream (the chain starts when every call word is in memory);
//   1  on a stream of the library's own, enqueued behind the expansion;  2  the same, enqueued in front of it.
// On its own stream the chain consumes the call words tile by tile while the expansion is still producing them
// (range_pipe.h: `done` flags); whichever way the two kernels end up scheduled, the result is the same — a chain
// that went first and is not fed within TFC_PIPE_POLL_MS (default 250) gives its job to the lane-per-stream kernel.
inline int pipe_overlap() {
  static const int v = [] {
    const char* e = std::getenv("TFC_PIPE_OVERLAP");
    return e ? std::atoi(e) : 2;
  }();
  return v;
}
inline long long pipe_poll_ticks() {
  static const long long v = [] {
    const char* e = std::getenv("TFC_PIPE_POLL_MS");
    return static_cast<long long>(e ? std::atoi(e) : 250) * 100000;     // wall_clock64(): 100 MHz
  }();
  return v;
}

// The library's own stream next to a caller's stream (one per caller stream and device, made on first use, never
// destroyed), with the two events that tie a launch on it into the caller's order.  It is a HIGH-priority stream
// (unless the caller's is): HIP multiplexes the streams of one priority onto a few hardware queues and kernels that
// share a queue run one after the other — a different priority is a different queue, and the chain is the critical
// path of an encode call.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
inline int side_stream(hipStream_t st, SideStream* out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SideStream> streams;
  int dev = 0;
  TFC_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto f = streams.find({dev, st});
  if (f == streams.end()) {
    SideStream ss;
    int least = 0, greatest = 0, mine = 0;
    TFC_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (hipStreamGetPriority(st, &mine) != hipSuccess) {
      (void)hipGetLastError();
      mine = least;
    }
    const int prio = (mine == greatest && least != greatest) ? least : greatest;
    TFC_HIP(hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, prio));
    TFC_HIP(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
    TFC_HIP(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming));
    f = streams.emplace(std::make_pair(dev, st), ss).first;
  }
  *out = f->second;
  return 0;
}
// Temporaries of one pipelined launch (call words / raw rows: ~150 / ~125 MB per 512-stream job of BASELINE config 2) are
// bounded: more jobs than fit go out as several launches, one behind the other.  6 GB, at most an eighth of the device's
// memory; TFC_PIPE_TEMP_MB overrides.  Measured (round 5, bench.py `saturation`, profiles/r05_notes.md): a launch takes
// the same ~14 ms up to ~40 jobs, but more per launch does not help — with 24 GB, 64 jobs in ONE launch took 12.4 ms of
// encode (the expansion, 0.19 ms per job, is what bounds it beyond ~20 jobs) + 31 ms of decode (512 chain waves: two per
// CU behind one 149 KB table image, and the parse beside them finds no CU) against 12.0 + 28.6 ms as 39 + 25 jobs.
// A launch whose temporaries cannot be allocated runs on the lane-per-stream kernels, which need none.
// (per device ordinal: a process may drive several devices of different memory sizes)
struct PipeDeviceInfo {
  size_t temp_bytes = 0;
  int cus = 0;
};
inline const PipeDeviceInfo& pipe_device_info() {
  constexpr int kMaxDevices = 64;
  static PipeDeviceInfo info[kMaxDevices];
  static std::once_flag once[kMaxDevices];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = std::min(std::max(dev, 0), kMaxDevices - 1);
  std::call_once(once[dev], [dev] {
    PipeDeviceInfo& d = info[dev];
    d.cus = 256;
    if (hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      d.cus = 256;
    }
    if (const char* e = std::getenv("TFC_PIPE_TEMP_MB")) {
      d.temp_bytes = static_cast<size_t>(std::max(1, std::atoi(e))) << 20;
      return;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      d.temp_bytes = size_t{2} << 30;
      return;
    }
    d.temp_bytes = std::min<size_t>(size_t{6} << 30, total_b / 8);
  });
  return info[dev];
}
inline size_t pipe_temp_bytes() { return pipe_device_info().temp_bytes; }

// Lane-per-stream family: n handles (same tables, same stream count) coded by one launch per
// kMaxLaneJobs of them; no counting pass, no read-back unless a handle wants its range errors now.
// Where the geometry allows, the pipelined kernels (range_pipe.h: parallel expansion into call words, then the
// chain) take the launch and the lane-per-stream kernel behind them only codes the jobs they gave up on.
template <typename Src>
int encode_lanes_many(tfc_encoder* const* es, int n, const Src* srcs, const int32_t* const* indexes,
                      int64_t elems, hipStream_t st, bool worst_case_slab) {
  constexpr int kMaxLaneJobs = EncLaneJobs<Src>::kMax;
  const tfc_tables* t = es[0]->tables;
  const int64_t streams = es[0]->streams;
  const bool indexed = indexes && indexes[0];
  LaneArgs la;
  la.image = t->d_lane_image.as<uint32_t>();
  la.bytes = t->lane_enc_bytes;
  la.ntab = static_cast<int>(t->rows.size());
  la.precision = t->lane_precision;
  la.cap = lanes_slab_bytes(t, elems, worst_case_slab);
  la.defer = lanes_escape_defer();
  la.guard = nullptr;
  const bool speculative = t->any_escape && !worst_case_slab;
  std::vector<DevBuf> backups(speculative ? n : 0);     // pre-call coder states of the handles that can retry
  la.lds_image = (t->lane_enc_bytes + 1023) & ~1023;
  using WaveLds = EncWaveLds<typename Src::raw_type>;
  la.lds_wave = indexed ? WaveLds::kBytes : WaveLds::kIndex;
  const int block = std::min(lanes_block(streams * n), 64 * ((160 * 1024 - la.lds_image) / la.lds_wave));
  const int lds_bytes = la.lds_image + (block / 64) * la.lds_wave;
  const void* fn = indexed ? reinterpret_cast<const void*>(&enc_lanes_kernel<true, Src>)
                           : reinterpret_cast<const void*>(&enc_lanes_kernel<false, Src>);
  TFC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  // pipelined launch plan: groups of 64 streams, tiles of kPipeTile symbols; rows (coder calls) per stream: one per
  // symbol, and room for escape codes when the tables have escape rows (half more than the symbols)
  PipeEncArgs pa;
  pa.nt = static_cast<int>(ceil_div(elems, kPipeTile));
  const int64_t rows64 = elems + (t->any_escape ? elems / 2 + 64 : 0);
  pa.rows = static_cast<int>(std::min<int64_t>((rows64 + 31) / 32 * 32, int64_t{1} << 30));
  pa.groups_per_job = static_cast<int>(ceil_div(streams, 64));
  pa.poll_ticks = pipe_poll_ticks();
  const size_t group_bytes = static_cast<size_t>(pa.rows) * 256 + (sizeof(unsigned int) * 65 * pa.nt) + 64 * (sizeof(uint4) + sizeof(uint2));
  const size_t job_bytes = group_bytes * pa.groups_per_job;
  const size_t kPipeTempBytes = pipe_temp_bytes();
  const bool pipe = pipe_enabled() && rows64 + 2048 < (int64_t{1} << 30) && job_bytes <= kPipeTempBytes && t->host.size() <= 65536 &&
                    static_cast<int64_t>(pa.nt) * pa.groups_per_job * kMaxLaneJobs < (int64_t{1} << 31);
  // A launch's chain workgroups must all be resident at once, next to an expansion that needs CUs of its own: a chain
  // workgroup (PipeEncChainLds::kGroups groups) takes a CU's whole LDS, so at most half the CUs go to chains — beyond
  // that the chains starve the expansion they wait for (and time out into the fallback).
  const int cus_e = pipe_device_info().cus;
  const size_t resident_jobs = std::max<size_t>(1, static_cast<size_t>(cus_e / 2) * PipeEncChainLds::kGroups / std::max(1, pa.groups_per_job));
  const int per_launch = !pipe ? kMaxLaneJobs
                               : static_cast<int>(std::max<size_t>(1, std::min<size_t>(std::min<size_t>(kMaxLaneJobs, resident_jobs),
                                                                                      kPipeTempBytes / std::max<size_t>(job_bytes, 1))));
  for (int g0 = 0; g0 < n; g0 += per_launch) {
    const int gn = std::min(per_launch, n - g0);
    EncLaneJobs<Src> jobs;
    EncErrJobs<Src> errs;
    jobs.streams = streams;
    jobs.elems = elems;
    jobs.blocks_per_job = static_cast<int>(ceil_div(streams, block));
    jobs.n = gn;
    errs.elems = elems;
    errs.ntab = la.ntab;
    errs.n = gn;
    for (int k = 0; k < gn; ++k) {
      tfc_encoder* e = es[g0 + k];
      e->elems_last = elems;
      e->indexed_last = indexed;
      if (speculative && !e->deferred) {
        TFC_HIP(backups[g0 + k].alloc(sizeof(uint4) * streams, st));
        TFC_HIP(hipMemcpyAsync(backups[g0 + k].p, e->state.p, sizeof(uint4) * streams, hipMemcpyDeviceToDevice, st));
      }
      EncChunk ch;
      ch.stride = la.cap;
      ch.data_bytes = static_cast<size_t>(la.cap) * streams;
      TFC_HIP(ch.data.alloc(ch.data_bytes + sizeof(unsigned int) * streams, st));
      ch.len_p = reinterpret_cast<unsigned int*>(ch.data.as<uint8_t>() + ch.data_bytes);
      EncLaneJob<Src>& J = jobs.job[k];
      J.src = srcs[g0 + k];
      J.index = indexed ? indexes[g0 + k] : nullptr;
      J.state = e->state.as<uint4>();
      J.chunk = ch.data.as<uint8_t>();
      J.chunk_len = ch.len_p;
      J.first_error = e->status.as<unsigned long long>();
      J.overflow_flag = e->oflag.as<unsigned int>();
      errs.job[k].status = J.first_error;
      errs.job[k].src = J.src;
      errs.job[k].index = J.index;
      e->chunks.push_back(std::move(ch));
    }
    {
      KernelTimer timer("enc_kernel", st);
      DevBuf temp;
      bool piped = pipe;
      la.guard = nullptr;
      if (pipe) do {
        const size_t groups = static_cast<size_t>(pa.groups_per_job) * gn;
        // (room behind the last group: the one-wave chain requests an iteration's rows ahead)
        const size_t calls_bytes = (groups * 64 * pa.rows + 64 * PipeEncChainLds::kRows) * sizeof(unsigned int);
        const size_t stage_bytes = groups * 64 * (sizeof(uint4) + sizeof(uint2));
        const size_t status_bytes = groups * pa.nt * 64 * sizeof(unsigned int);
        const size_t done_bytes = (groups * pa.nt * sizeof(unsigned int) + 255) & ~size_t{255};
        if (temp.alloc(calls_bytes + stage_bytes + status_bytes + done_bytes + 512, st) != hipSuccess) {
          // no room for the call words (a smaller or busier device): the lane-per-stream kernel codes this launch
          (void)hipGetLastError();
          temp.p = nullptr;
          piped = false;
          break;
        }
        uint8_t* base = temp.as<uint8_t>();
        pa.calls = reinterpret_cast<unsigned int*>(base);
        pa.stage_state = reinterpret_cast<uint4*>(base + calls_bytes);
        pa.stage_out = reinterpret_cast<uint2*>(base + calls_bytes + groups * 64 * sizeof(uint4));
        pa.status = reinterpret_cast<unsigned int*>(base + calls_bytes + stage_bytes);
        pa.done = reinterpret_cast<unsigned int*>(base + calls_bytes + stage_bytes + status_bytes);
        pa.fallback = reinterpret_cast<unsigned int*>(base + calls_bytes + stage_bytes + status_bytes + done_bytes);
        pa.started = pa.fallback + 64;          // (behind the 64 jobs' flags)
        pa.groups = static_cast<int>(groups);
        g_pipe_launches.fetch_add(1, std::memory_order_relaxed);
        pa.fast16 = t->d_fast.as<uint16_t>();
        pa.rows_fast = t->d_rows_fast.as<int2>();
        pa.ntab = la.ntab;
        pa.tab_entries = t->host.size() * 2 <= static_cast<size_t>(kExpandTabBytes) ? static_cast<int>(t->host.size()) : 0;
        pa.cap = la.cap;
        // nothing known about any tile, no tile released, no fallback
        TFC_HIP(hipMemsetAsync(pa.status, 0, status_bytes + done_bytes + 512, st));
        PipeChainJobs cj;
        cj.streams = streams;
        for (int k = 0; k < gn; ++k)
          cj.job[k] = PipeChainJob{jobs.job[k].state, jobs.job[k].chunk, jobs.job[k].chunk_len, jobs.job[k].overflow_flag};
        // A large launch: the chain (workgroups of PipeEncChainLds::kGroups groups, helper waves: range_pipe.h) on the library's own stream
        // next to the expansion.  A small one — a model step's few groups, next to other steps' convolutions — one-wave
        // chain workgroups behind the expansion on the caller's stream: measured on bmshj2018 with steps in flight, the
        // second stream, its events and the large workgroups cost more (45 against 36 ms per step) than the overlap of
        // two short kernels gives.
        const bool large = groups >= 64;
        const int overlap = large ? pipe_overlap() : 0;
        SideStream side;
        if (overlap) {
          if (side_stream(st, &side)) return 1;
          TFC_HIP(hipEventRecord(side.fork, st));
          TFC_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
        }
        const hipStream_t cst = overlap ? side.stream : st;
        const dim3 xgrid(static_cast<unsigned>(groups * pa.nt));
        auto expand = [&] {
          KernelTimer t2("enc_expand", st);
          const bool tlds = pa.tab_entries != 0;
          if (indexed && tlds) hipLaunchKernelGGL((enc_expand_kernel<true, Src, true>), xgrid, dim3(kExpandThreads), 0, st, jobs, pa);
          else if (indexed) hipLaunchKernelGGL((enc_expand_kernel<true, Src, false>), xgrid, dim3(kExpandThreads), 0, st, jobs, pa);
          else if (tlds) hipLaunchKernelGGL((enc_expand_kernel<false, Src, true>), xgrid, dim3(kExpandThreads), 0, st, jobs, pa);
          else hipLaunchKernelGGL((enc_expand_kernel<false, Src, false>), xgrid, dim3(kExpandThreads), 0, st, jobs, pa);
        };
        const int cgroups = PipeEncChainLds::kGroups;
        const unsigned cblocks = static_cast<unsigned>(ceil_div(static_cast<int64_t>(groups), cgroups));
        const int clds = cgroups * PipeEncChainLds::kGroup;
        if (large)
          TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, clds));
        auto chain = [&] {
          KernelTimer t2("enc_chain", cst);
          if (!large) {
            hipLaunchKernelGGL(enc_chain_direct_kernel, dim3(static_cast<unsigned>(groups)), dim3(64), 0, cst, cj, pa);
            return;
          }
          hipLaunchKernelGGL(enc_chain_kernel, dim3(cblocks), dim3(256 * cgroups), clds, cst, cj, pa);
          // (the expansion behind the chain's workgroups, not in their way)
          if (overlap == 2)
            hipLaunchKernelGGL(enc_gate_kernel, dim3(1), dim3(1), 0, st, pa.started, cblocks, static_cast<long long>(20000));
        };
        if (overlap == 2) { chain(); expand(); }
        else { expand(); chain(); }
        if (overlap) {
          TFC_HIP(hipEventRecord(side.join, side.stream));
          TFC_HIP(hipStreamWaitEvent(st, side.join, 0));
        }
        hipLaunchKernelGGL(enc_commit_kernel, dim3(static_cast<unsigned>(groups)), dim3(64), 0, st, cj, pa);
        la.guard = pa.fallback;
      } while (false);
      const dim3 grid(static_cast<unsigned>(jobs.blocks_per_job * gn));
      if (piped && !pipe_fallback_launch()) {}
      else if (indexed) hipLaunchKernelGGL((enc_lanes_kernel<true, Src>), grid, dim3(block), lds_bytes, st, jobs, la);
      else hipLaunchKernelGGL((enc_lanes_kernel<false, Src>), grid, dim3(block), lds_bytes, st, jobs, la);
      // `temp` goes back to the library's cache in stream order, behind these launches
    }
    hipLaunchKernelGGL((enc_error_kernel<Src>), dim3(1), dim3(64), 0, st, errs);
    TFC_HIP(hipGetLastError());
  }
  bool any_eager = false;
  for (int k = 0; k < n; ++k) any_eager |= !es[k]->deferred;
  if (any_eager) {
    TFC_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < n; ++k) {
      tfc_encoder* e = es[k];
      if (e->deferred) continue;
      unsigned long long host_status[4];
      TFC_HIP(hipMemcpy(host_status, e->status.p, sizeof(host_status), hipMemcpyDeviceToHost));
      if (host_status[0] != ~0ull) return encoder_error(e, host_status);
      if (!speculative) continue;
      unsigned int oflag = 0;
      TFC_HIP(hipMemcpy(&oflag, e->oflag.p, sizeof(oflag), hipMemcpyDeviceToHost));
      if (oflag) {
        // a stream outgrew the speculative slab: back to the pre-call state, code again with the bound
        // that cannot be exceeded
        TFC_HIP(hipMemcpyAsync(e->state.p, backups[k].p, sizeof(uint4) * streams, hipMemcpyDeviceToDevice, st));
        TFC_HIP(hipMemsetAsync(e->oflag.p, 0, sizeof(unsigned int), st));
        e->chunks.pop_back();
        const int32_t* ix = indexed ? indexes[k] : nullptr;
        if (encode_lanes_many(&es[k], 1, &srcs[k], &ix, elems, st, true)) return 1;
      }
    }
  }
  return 0;
}

// Deferred-error handles, wave-per-stream families: the counting pass's first error goes to the handle's
// status word on the device (the value / index behind it are recorded by enc_error_kernel).
__global__ void enc_defer_kernel(const unsigned long long* count_status, unsigned long long* status) {
  if (count_status[0] != ~0ull && status[0] == ~0ull) status[0] = count_status[0];
}

// Slab bytes for a wave-per-stream call sized WITHOUT the counting pass's result: what the call needs
// when no symbol takes an escape code (2 bytes per coder call + 4, rounded per stream), a quarter more
// when the tables have escape rows (an escape adds 2 log2|overflow| + 3 calls; the tables' own tail mass
// is 2^-8 ... 2^-7 of the symbols), and room for digits earlier calls held back.
size_t speculative_slab_bytes(const tfc_tables* t, int64_t streams, int64_t elems) {
  const size_t per = ((2 * static_cast<size_t>(elems) + 4 + 15) / 16) * 16 + 32;
  size_t bytes = per * static_cast<size_t>(streams);
  if (t->any_escape) bytes += bytes / 4;
  bytes += 64u << 10;
  // test hook (tests/test_pipeline_gpu.py): TFC_SPECULATIVE_SLAB_DIV=n shrinks the slab so that the
  // "outgrown" path can be exercised with ordinary data
  if (const char* e = std::getenv("TFC_SPECULATIVE_SLAB_DIV")) {
    const long n = std::strtol(e, nullptr, 10);
    if (n > 1) bytes = std::max<size_t>(bytes / static_cast<size_t>(n), 256);
  }
  return bytes;
}

// ---- fused quantise / dequantise for the lane family ----------------------------------------------
// The lane kernels' hand-scheduled blocks take plain int32 symbols in channel mode; their generic steps (which
// convert per element) are ~2.3x slower.  For bottleneck values the conversion therefore runs as its own
// elementwise pass — HBM-bound, ~25 us for the 25 M elements of config 2 against milliseconds of coding — into a
// stream-ordered temporary, and the blocks code int32: continuous_batched.py:370-380 / :416-422 at the speed of
// int32 channel mode (+ the pass).
// grid = (ceil(elems / 1024), streams): four elements per thread, 256 apart (every access of a wave is one
// contiguous run); 32-bit index arithmetic — the table of an element is its position in the stream modulo the
// table count, stepped by 256 mod ntab instead of divided per element
template <typename T>
__global__ void __launch_bounds__(256) lanes_quantize_kernel(tfc::SymQuant<T> src, unsigned int elems, unsigned int ntab,
                                                             int32_t* out) {
  const unsigned int j0 = blockIdx.x * 1024u + threadIdx.x;
  const long long base = static_cast<long long>(blockIdx.y) * elems;
  const unsigned int step = 256u % ntab;
  unsigned int t = j0 % ntab;
#pragma unroll
  for (unsigned int k = 0; k < 4u; ++k) {
    const unsigned int j = j0 + 256u * k;
    if (j < elems) out[base + j] = src.quant(src.y[base + j], static_cast<int>(t));
    t += step;
    t -= t >= ntab ? ntab : 0u;
  }
}
template <typename Dst>
__global__ void __launch_bounds__(256) lanes_dequantize_kernel(Dst dst, const int32_t* sym, unsigned int elems,
                                                               unsigned int ntab) {
  const unsigned int j0 = blockIdx.x * 1024u + threadIdx.x;
  const long long base = static_cast<long long>(blockIdx.y) * elems;
  const unsigned int step = 256u % ntab;
  unsigned int t = j0 % ntab;
#pragma unroll
  for (unsigned int k = 0; k < 4u; ++k) {
    const unsigned int j = j0 + 256u * k;
    if (j < elems) dst.store(base + j, static_cast<int>(t), sym[base + j]);
    t += step;
    t -= t >= ntab ? ntab : 0u;
  }
}

template <typename Src> struct is_plain_symbols : std::false_type {};
template <> struct is_plain_symbols<tfc::SymInt32> : std::true_type {};

// channel mode, bottleneck values: quantise pass, then the int32 blocks
template <typename Src>
int encode_lanes_prequantized(tfc_encoder* const* es, int n, const Src* srcs, int64_t elems, hipStream_t st) {
  const long long per = static_cast<long long>(es[0]->streams) * elems;
  const int ntab = static_cast<int>(es[0]->tables->rows.size());
  DevBuf tmp;
  TFC_HIP(tmp.alloc(sizeof(int32_t) * static_cast<size_t>(per) * n, st));
  std::vector<tfc::SymInt32> q(n);
  for (int k = 0; k < n; ++k) {
    int32_t* out = tmp.as<int32_t>() + static_cast<long long>(k) * per;
    hipLaunchKernelGGL((lanes_quantize_kernel<typename Src::raw_type>),
                       dim3(static_cast<unsigned>(ceil_div(elems, 1024)), static_cast<unsigned>(es[0]->streams)),
                       dim3(256), 0, st, srcs[k], static_cast<unsigned int>(elems), static_cast<unsigned int>(ntab), out);
    q[k] = tfc::SymInt32{out};
  }
  TFC_HIP(hipGetLastError());
  return encode_lanes_many(es, n, q.data(), static_cast<const int32_t* const*>(nullptr), elems, st, false);
  // `tmp` goes back to the pool in stream order, behind the coding launch
}

int encode_precheck(tfc_encoder* e, int64_t elems) {
  if (e->finalized) return fail("encoder handle was already finalized");
  if (e->poisoned) return fail("encoder handle met a range error in an earlier call");
  if (elems < 0) return fail("negative element count");
  return 0;
}

template <typename Src>
int run_encode(tfc_encoder* e, const int32_t* index, int64_t elems, const Src& src, hipStream_t st) {
  if (encode_precheck(e, elems)) return 1;
  e->touch(st);
  if (e->streams == 0 || elems == 0) return 0;
  const tfc_tables* t = e->tables;
  if (t->rows.empty()) return fail("index=0 not in range [0, 0)");
  if (e->family < 0) e->family = select_family(t, e->mode, e->streams, elems, e->fast);
  if (e->family == kLanes && elems >= (int64_t{1} << 29)) return fail("encode call too large for this handle");
  if (e->family == kLanes) {
    if constexpr (!is_plain_symbols<Src>::value) {
      // (the pipelined kernels quantise in their expansion pass)
      if (!index && !pipe_enabled()) return encode_lanes_prequantized(&e, 1, &src, elems, st);
    }
    return encode_lanes_many(&e, 1, &src, &index, elems, st, false);
  }
  e->elems_last = elems;
  e->indexed_last = index != nullptr;

  EncChunk ch;
  TFC_HIP(ch.len.alloc(sizeof(unsigned int) * e->streams, st));
  ch.len_p = ch.len.as<unsigned int>();
  EncParams p;
  p.tab = view_of(t);
  p.index = index;
  p.streams = e->streams;
  p.elems = elems;
  p.calls = nullptr;
  p.first_error = e->status.as<unsigned long long>();
  p.state = e->state.as<uint4>();
  p.chunk = nullptr;
  p.chunk_off = nullptr;
  p.chunk_len = ch.len.as<unsigned int>();
  p.overflow_flag = e->oflag.as<unsigned int>();
  p.guard = nullptr;
  p.cap_total = 0;
  unsigned long long host_status[4] = {~0ull, 0ull, 0ull, 0ull};

  // wave-per-stream families: validation + exact output bound first (one read-back)
  DevBuf calls, cstat;
  TFC_HIP(calls.alloc(sizeof(unsigned long long) * e->streams, st));
  TFC_HIP(cstat.alloc(sizeof(unsigned long long) * 3, st));
  TFC_HIP(hipMemsetAsync(calls.p, 0, sizeof(unsigned long long) * e->streams, st));
  // cstat[0] = first error position, [1] = total capacity, [2] = coder calls of all streams
  TFC_HIP(hipMemsetAsync(cstat.p, 0xFF, sizeof(unsigned long long), st));
  TFC_HIP(hipMemsetAsync(cstat.as<unsigned long long>() + 1, 0, 2 * sizeof(unsigned long long), st));
  TFC_HIP(ch.off.alloc(sizeof(long long) * (e->streams + 1), st));
  p.calls = calls.as<unsigned long long>();
  p.first_error = cstat.as<unsigned long long>();
  p.chunk_off = ch.off.as<long long>();

  const int64_t tiles = ceil_div(elems, kCountTile);
  if (e->streams * tiles >= (int64_t{1} << 31)) return fail("encode call too large for one launch");
  const size_t count_lds = p.tab.ntab <= kCountLdsRows ? sizeof(int) * p.tab.ntab : 0;
  hipLaunchKernelGGL((enc_count_kernel<Src>), dim3(static_cast<unsigned>(e->streams * tiles)),
                     dim3(256), count_lds, st, p, src);
  hipLaunchKernelGGL(enc_offsets_kernel, dim3(1), dim3(1024), 0, st, p.calls, p.state,
                     e->family == kFast ? 1 : 0, e->streams, ch.off.as<long long>(),
                     cstat.as<unsigned long long>() + 1);
  unsigned long long count_status[3] = {~0ull, 0ull, 0ull};
  if (e->deferred) {
    // no read-back: the slab is sized from the geometry alone and the coding pass checks the counting
    // pass's verdict on the device (EncParams::guard); an error or an outgrown slab surfaces at
    // tfc_encoder_status / tfc_encoder_finalize, as for the lane kernels
    count_status[1] = speculative_slab_bytes(t, e->streams, elems);
    p.guard = cstat.as<unsigned long long>();
    p.cap_total = count_status[1];
    hipLaunchKernelGGL(enc_defer_kernel, dim3(1), dim3(1), 0, st, cstat.as<unsigned long long>(),
                       e->status.as<unsigned long long>());
    EncErrJobs<Src> errs;
    errs.elems = elems;
    errs.ntab = p.tab.ntab;
    errs.n = 1;
    errs.job[0].status = e->status.as<unsigned long long>();
    errs.job[0].src = src;
    errs.job[0].index = index;
    hipLaunchKernelGGL((enc_error_kernel<Src>), dim3(1), dim3(64), 0, st, errs);
  } else {
  TFC_HIP(hipMemcpyAsync(count_status, cstat.p, sizeof(count_status), hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (count_status[0] != ~0ull) {
    // nothing was appended; fetch the offending element for the message
    TFC_HIP(hipMemcpyAsync(e->status.p, count_status, sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    EncErrJobs<Src> errs;
    errs.elems = elems;
    errs.ntab = p.tab.ntab;
    errs.n = 1;
    errs.job[0].status = e->status.as<unsigned long long>();
    errs.job[0].src = src;
    errs.job[0].index = index;
    hipLaunchKernelGGL((enc_error_kernel<Src>), dim3(1), dim3(64), 0, st, errs);
    TFC_HIP(hipMemcpyAsync(host_status, e->status.p, sizeof(host_status), hipMemcpyDeviceToHost, st));
    const unsigned long long clear[4] = {~0ull, 0ull, 0ull, 0ull};
    TFC_HIP(hipStreamSynchronize(st));
    TFC_HIP(hipMemcpyAsync(e->status.p, clear, sizeof(clear), hipMemcpyHostToDevice, st));
    TFC_HIP(hipStreamSynchronize(st));
    const int rc = encoder_error(e, host_status);
    e->poisoned = false;      // the validation pass runs before anything is appended
    return rc;
  }

  }
  TFC_HIP(ch.data.alloc(count_status[1], st));
  ch.data_bytes = count_status[1];
  p.chunk = ch.data.as<uint8_t>();
  const size_t lds = table_lds_bytes(t);
  const unsigned blocks = static_cast<unsigned>(ceil_div(e->streams, kWavesPerBlock));
  if (e->family == kFast) {
    KernelTimer timer("enc_kernel", st);
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_fast_kernel<Src>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_request(e->fast_lds))));
    hipLaunchKernelGGL((enc_fast_kernel<Src>),
                       dim3(static_cast<unsigned>(ceil_div(e->streams, e->fast_waves))),
                       dim3(64 * e->fast_waves), lds_request(e->fast_lds), st, p, src);
  } else {
    KernelTimer timer("enc_kernel", st);
    if (lds) {
      TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_kernel<true, Src>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL((enc_kernel<true, Src>), dim3(blocks), dim3(kBlock), lds, st, p, src);
    } else {
      hipLaunchKernelGGL((enc_kernel<false, Src>), dim3(blocks), dim3(kBlock), 0, st, p, src);
    }
  }
  TFC_HIP(hipGetLastError());
  e->chunks.push_back(std::move(ch));
  return 0;
}

}  // namespace

extern "C" int tfc_encoder_create(const tfc_tables* tables, int64_t streams, void* stream,
                                  tfc_encoder** out) {
  *out = nullptr;
  if (!tables) return fail("tables is null");
  if (streams < 0) return fail("negative stream count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::unique_ptr<tfc_encoder> e(new tfc_encoder);
  e->tables = tables;
  e->streams = streams;
  {
    // LDS plan of enc_fast_kernel: tables + row directory + one call ring per wave.
    const size_t fixed = sizeof(uint16_t) * ((tables->host.size() + 3) & ~size_t{3}) +
                         sizeof(int2) * tables->rows.size();
    const size_t ring = sizeof(unsigned int) * kRingWords;
    if (!tables->rows.empty() && fixed + ring <= 160 * 1024) {
      e->fast = true;
      const size_t fit = (160 * 1024 - fixed) / ring;
      const size_t want = static_cast<size_t>(waves_wanted(streams));
      e->fast_waves = static_cast<int>(std::min(fit, want));
      e->fast_lds = fixed + ring * e->fast_waves;
    }
  }
  TFC_HIP(e->ctl.alloc(64 + sizeof(uint4) * std::max<int64_t>(streams, 1), st));
  e->status.p = e->ctl.p;
  e->oflag.p = e->ctl.as<uint8_t>() + 32;
  e->state.p = e->ctl.as<uint8_t>() + 64;
  hipLaunchKernelGGL(fill_state_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(1, ceil_div(streams, 256)))),
                     dim3(256), 0, st, e->state.as<uint4>(), streams, make_uint4(0u, 0xFFFFFFFFu, 0u, 0u),
                     e->status.as<unsigned long long>(), e->oflag.as<unsigned int>());
  *out = e.release();
  return 0;
}

// n handles with ONE allocation and ONE initialisation launch (the per-handle driver calls and small
// kernels are a measurable part of a step once the coding kernels of a group share a launch).
extern "C" int tfc_encoder_create_many(const tfc_tables* tables, int64_t streams, int n, void* stream,
                                       tfc_encoder** out) {
  if (!tables) return fail("tables is null");
  if (streams < 0 || n < 0) return fail("negative count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int k = 0; k < n; ++k) out[k] = nullptr;
  if (n == 0) return 0;
  const long long ctl_bytes = 64 + static_cast<long long>(sizeof(uint4)) * std::max<int64_t>(streams, 1);
  auto group = std::make_shared<DevBuf>();
  TFC_HIP(group->alloc(static_cast<size_t>(ctl_bytes) * n, st));
  std::vector<std::unique_ptr<tfc_encoder>> made;
  for (int k = 0; k < n; ++k) {
    std::unique_ptr<tfc_encoder> e(new tfc_encoder);
    e->tables = tables;
    e->streams = streams;
    const size_t fixed = sizeof(uint16_t) * ((tables->host.size() + 3) & ~size_t{3}) + sizeof(int2) * tables->rows.size();
    const size_t ring = sizeof(unsigned int) * kRingWords;
    if (!tables->rows.empty() && fixed + ring <= 160 * 1024) {
      e->fast = true;
      const size_t fit = (160 * 1024 - fixed) / ring;
      const size_t want = static_cast<size_t>(waves_wanted(streams));
      e->fast_waves = static_cast<int>(std::min(fit, want));
      e->fast_lds = fixed + ring * e->fast_waves;
    }
    e->ctl_group = group;
    uint8_t* c = group->as<uint8_t>() + k * ctl_bytes;
    e->status.p = c;
    e->oflag.p = c + 32;
    e->state.p = c + 64;
    made.push_back(std::move(e));
  }
  hipLaunchKernelGGL(fill_state_many_kernel,
                     dim3(static_cast<unsigned>(std::max<int64_t>(1, ceil_div(streams, 256))), static_cast<unsigned>(n)),
                     dim3(256), 0, st, group->as<uint8_t>(), ctl_bytes, streams, make_uint4(0u, 0xFFFFFFFFu, 0u, 0u));
  TFC_HIP(hipGetLastError());
  for (int k = 0; k < n; ++k) out[k] = made[k].release();
  return 0;
}

// EntropyEncodeFinalize of n handles without host synchronisation, in three launches where every handle
// holds exactly one piece from the lane-per-stream kernels (what tfc_encoder_encode_many leaves behind);
// otherwise handle by handle.
extern "C" int tfc_encoder_finalize_device_many(int n, tfc_encoder* const* es, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  bool batch = n <= kMaxFinalizeJobs;
  for (int k = 0; k < n; ++k) es[k]->touch(st);
  for (int k = 0; k < n && batch; ++k) {
    const tfc_encoder* e = es[k];
    if (e->finalized || e->family != kLanes || e->chunks.size() != 1 || e->streams != es[0]->streams ||
        e->streams == 0 || e->chunks[0].stride != es[0]->chunks[0].stride)
      batch = false;
  }
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (tfc_encoder_finalize_device(es[k], stream)) return 1;
    return 0;
  }
  const int64_t streams = es[0]->streams;
  FinalizeJobs f;
  f.streams = streams;
  f.n = n;
  const size_t cap = es[0]->chunks[0].data_bytes + 4 * static_cast<size_t>(streams) + 64;     // pieces + Finalize bytes + slack
  f.length_off = static_cast<long long>((sizeof(Tail) * streams + 15) & ~size_t{15});
  f.offsets_off = f.length_off + static_cast<long long>(sizeof(long long)) * streams;
  f.blob_off = (f.offsets_off + static_cast<long long>(sizeof(long long)) * (streams + 1) + 15) & ~15ll;
  f.job_bytes = (f.blob_off + static_cast<long long>(cap) + 255) & ~255ll;
  auto group = std::make_shared<DevBuf>();
  TFC_HIP(group->alloc(static_cast<size_t>(f.job_bytes) * n, st));
  f.base = group->as<uint8_t>();
  for (int k = 0; k < n; ++k) {
    EncChunk& c = es[k]->chunks[0];
    f.job[k].state = es[k]->state.as<uint4>();
    f.job[k].chunk = ChunkRef{c.data.as<uint8_t>(), c.off.as<long long>(), c.len_p, c.stride};
  }
  hipLaunchKernelGGL(enc_tail_many_kernel, dim3(static_cast<unsigned>(ceil_div(streams, 256)), static_cast<unsigned>(n)),
                     dim3(256), 0, st, f);
  hipLaunchKernelGGL(scan_lengths_many_kernel, dim3(static_cast<unsigned>(n)), dim3(1024), 0, st, f);
  hipLaunchKernelGGL(enc_pack_many_kernel,
                     dim3(static_cast<unsigned>(ceil_div(streams, kWavesPerBlock)), static_cast<unsigned>(n)),
                     dim3(kBlock), 0, st, f);
  TFC_HIP(hipGetLastError());
  for (int k = 0; k < n; ++k) {
    tfc_encoder* e = es[k];
    e->result_group = group;
    e->offsets.p = f.base + k * f.job_bytes + f.offsets_off;
    e->blob.p = f.base + k * f.job_bytes + f.blob_off;
    e->blob_capacity = static_cast<int64_t>(cap);
    e->chunks.clear();       // stream-ordered frees behind the pack launch
    e->finalized = true;
  }
  return 0;
}

// (measurement aid, not part of the ABI: tools/chain_clock_probe.py)
extern "C" int tfc_debug_pipe_clocks(unsigned long long* out8) {
  TFC_HIP(hipDeviceSynchronize());
  TFC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(tfc::g_pipe_clock), 8 * sizeof(unsigned long long)));
  return 0;
}
extern "C" int tfc_debug_enc_clocks(unsigned long long* out4) {
  TFC_HIP(hipDeviceSynchronize());
  TFC_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(tfc::g_enc_clock), 4 * sizeof(unsigned long long)));
  return 0;
}

extern "C" int tfc_pipe_counters(int64_t* launches, int64_t* fallback_blocks) {
  if (launches) *launches = g_pipe_launches.load();
  if (fallback_blocks) {
    unsigned long long v = 0;
    TFC_HIP(hipDeviceSynchronize());
    TFC_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(tfc::g_pipe_fallback_blocks), sizeof(v)));
    *fallback_blocks = static_cast<int64_t>(v);
  }
  return 0;
}

extern "C" int tfc_set_pipe_format(int format, int waves_per_workgroup) {
  if (format < 0 || format > 2 || waves_per_workgroup < 0 || waves_per_workgroup > 8) {
    tfc::fail("tfc_set_pipe_format: format 0 ... 2, waves per workgroup 0 ... 8");
    return -1;
  }
  pipe_waves_value().store(waves_per_workgroup);
  return pipe_format_value().exchange(format);
}

extern "C" int tfc_encoder_set_mode(tfc_encoder* e, int mode) {
  if (mode != TFC_MODE_AUTO && mode != TFC_MODE_LATENCY && mode != TFC_MODE_THROUGHPUT)
    return fail("unknown mode %d", mode);
  if (e->family >= 0) return fail("the mode of a handle is fixed by its first coding call");
  e->mode = mode;
  return 0;
}

extern "C" int tfc_encoder_set_deferred_errors(tfc_encoder* e, int on) {
  e->deferred = on != 0;
  return 0;
}

extern "C" int tfc_encoder_encode(tfc_encoder* e, const int32_t* value, const int32_t* index,
                                  int64_t elems, void* stream) {
  return run_encode(e, index, elems, SymInt32{value}, static_cast<hipStream_t>(stream));
}

extern "C" int tfc_encoder_encode_many(int n, tfc_encoder* const* es, const int32_t* const* values,
                                       const int32_t* const* indexes, int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) {
    tfc_encoder* e = es[k];
    if (encode_precheck(e, elems)) return 1;
    e->touch(st);
    if (e->tables != es[0]->tables || e->streams != es[0]->streams)
      return fail("tfc_encoder_encode_many: handles must share tables and stream count");
    if ((indexes && indexes[k]) != (indexes && indexes[0]))
      return fail("tfc_encoder_encode_many: either all jobs carry an index or none");
    if (e->streams == 0 || e->tables->rows.empty()) batch = false;
    if (batch && e->family < 0) e->family = select_family(e->tables, e->mode, e->streams * n, elems, e->fast);
    if (e->family != kLanes) batch = false;
  }
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (tfc_encoder_encode(es[k], values[k], indexes ? indexes[k] : nullptr, elems, stream)) return 1;
    return 0;
  }
  std::vector<SymInt32> srcs(n);
  for (int k = 0; k < n; ++k) srcs[k] = SymInt32{values[k]};
  return encode_lanes_many(es, n, srcs.data(), indexes, elems, st, false);
}

namespace {

int dispatch_quantized(tfc_encoder* e, const void* y, int dtype, const float* qoffset,
                       const int32_t* index, const int32_t* cdf_offset, int64_t elems,
                       hipStream_t st) {
  switch (dtype) {
    case 0: return run_encode(e, index, elems, SymQuant<float>{static_cast<const float*>(y), qoffset, cdf_offset}, st);
    case 1: return run_encode(e, index, elems, SymQuant<__hip_bfloat16>{static_cast<const __hip_bfloat16*>(y), qoffset, cdf_offset}, st);
    case 2: return run_encode(e, index, elems, SymQuant<__half>{static_cast<const __half*>(y), qoffset, cdf_offset}, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

}  // namespace

extern "C" int tfc_encoder_encode_quantized(tfc_encoder* e, const void* y, int dtype,
                                            const float* qoffset, const int32_t* cdf_offset,
                                            int64_t channels, int64_t elems, void* stream) {
  if (channels != static_cast<int64_t>(e->tables->rows.size()))
    return fail("channel count %lld does not match table count %lld",
                static_cast<long long>(channels), static_cast<long long>(e->tables->rows.size()));
  return dispatch_quantized(e, y, dtype, qoffset, nullptr, cdf_offset, elems,
                            static_cast<hipStream_t>(stream));
}

extern "C" int tfc_encoder_encode_quantized_indexed(tfc_encoder* e, const void* y, int dtype,
                                                    const int32_t* index,
                                                    const int32_t* cdf_offset, int64_t elems,
                                                    void* stream) {
  if (!index) return fail("index is null");
  return dispatch_quantized(e, y, dtype, nullptr, index, cdf_offset, elems,
                            static_cast<hipStream_t>(stream));
}

namespace {
template <typename T>
int encode_quantized_many(int n, tfc_encoder* const* es, const void* const* ys, const float* qoffset,
                          const int32_t* cdf_offset, int64_t elems, hipStream_t st) {
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) {
    tfc_encoder* e = es[k];
    if (encode_precheck(e, elems)) return 1;
    e->touch(st);
    if (e->tables != es[0]->tables || e->streams != es[0]->streams)
      return fail("tfc_encoder_encode_quantized_many: handles must share tables and stream count");
    if (e->streams == 0 || e->tables->rows.empty()) batch = false;
    if (batch && e->family < 0) e->family = select_family(e->tables, e->mode, e->streams * n, elems, e->fast);
    if (e->family != kLanes) batch = false;
  }
  std::vector<tfc::SymQuant<T>> srcs(n);
  for (int k = 0; k < n; ++k) srcs[k] = tfc::SymQuant<T>{static_cast<const T*>(ys[k]), qoffset, cdf_offset};
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (run_encode(es[k], nullptr, elems, srcs[k], st)) return 1;
    return 0;
  }
  for (int g0 = 0; g0 < n; g0 += kMaxFinalizeJobs) {      // bounded temporaries: <= 64 batches of symbols at a time
    const int gn = std::min(kMaxFinalizeJobs, n - g0);
    for (int k = 0; k < gn; ++k) {
      es[g0 + k]->elems_last = elems;
      es[g0 + k]->indexed_last = false;
    }
    if (pipe_enabled()) {
      if (encode_lanes_many(es + g0, gn, srcs.data() + g0, static_cast<const int32_t* const*>(nullptr), elems, st, false)) return 1;
    } else if (encode_lanes_prequantized(es + g0, gn, srcs.data() + g0, elems, st)) {
      return 1;
    }
  }
  return 0;
}
}  // namespace

// EntropyEncodeChannel with the quantise prologue for n independent handles (same tables, same geometry) as one
// coding launch — tfc_encoder_encode_quantized x tfc_encoder_encode_many.
extern "C" int tfc_encoder_encode_quantized_many(int n, tfc_encoder* const* es, const void* const* ys, int dtype,
                                                 const float* qoffset, const int32_t* cdf_offset,
                                                 int64_t channels, int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  if (channels != static_cast<int64_t>(es[0]->tables->rows.size()))
    return fail("channel count %lld does not match table count %lld",
                static_cast<long long>(channels), static_cast<long long>(es[0]->tables->rows.size()));
  switch (dtype) {
    case 0: return encode_quantized_many<float>(n, es, ys, qoffset, cdf_offset, elems, st);
    case 1: return encode_quantized_many<__hip_bfloat16>(n, es, ys, qoffset, cdf_offset, elems, st);
    case 2: return encode_quantized_many<__half>(n, es, ys, qoffset, cdf_offset, elems, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

namespace {
template <typename T>
int encode_quantized_indexed_many(int n, tfc_encoder* const* es, const void* const* ys, const int32_t* const* indexes,
                                  const int32_t* cdf_offset, int64_t elems, hipStream_t st) {
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) {
    tfc_encoder* e = es[k];
    if (!indexes[k]) return fail("index is null");
    if (encode_precheck(e, elems)) return 1;
    e->touch(st);
    if (e->tables != es[0]->tables || e->streams != es[0]->streams)
      return fail("tfc_encoder_encode_quantized_indexed_many: handles must share tables and stream count");
    if (e->streams == 0 || e->tables->rows.empty()) batch = false;
    if (batch && e->family < 0) e->family = select_family(e->tables, e->mode, e->streams * n, elems, e->fast);
    if (e->family != kLanes) batch = false;
  }
  std::vector<tfc::SymQuant<T>> srcs(n);
  for (int k = 0; k < n; ++k) srcs[k] = tfc::SymQuant<T>{static_cast<const T*>(ys[k]), nullptr, cdf_offset};
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (run_encode(es[k], indexes[k], elems, srcs[k], st)) return 1;
    return 0;
  }
  return encode_lanes_many(es, n, srcs.data(), indexes, elems, st, false);
}
}  // namespace

// EntropyEncodeIndex with the quantise prologue for n independent handles (same tables, same geometry) as one
// coding launch — tfc_encoder_encode_quantized_indexed x tfc_encoder_encode_many: the main latents of several
// batches of a hyperprior model (continuous_indexed.py:355-386) behind one launch.
extern "C" int tfc_encoder_encode_quantized_indexed_many(int n, tfc_encoder* const* es, const void* const* ys, int dtype,
                                                         const int32_t* const* indexes, const int32_t* cdf_offset,
                                                         int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  if (!indexes) return fail("index is null");
  switch (dtype) {
    case 0: return encode_quantized_indexed_many<float>(n, es, ys, indexes, cdf_offset, elems, st);
    case 1: return encode_quantized_indexed_many<__hip_bfloat16>(n, es, ys, indexes, cdf_offset, elems, st);
    case 2: return encode_quantized_indexed_many<__half>(n, es, ys, indexes, cdf_offset, elems, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

namespace {

// Tail + lengths + offsets on the device; `exact` = synchronise to size the blob exactly, otherwise
// the blob gets the slabs' total capacity and nothing is read back.
int finalize_impl(tfc_encoder* e, hipStream_t st, bool exact) {
  e->touch(st);
  if (e->finalized) return 0;
  const int64_t n = e->streams;
  TFC_HIP(e->offsets_own.alloc(sizeof(long long) * (n + 1), st));
  e->offsets.p = e->offsets_own.p;
  if (n == 0) {
    TFC_HIP(hipMemsetAsync(e->offsets.p, 0, sizeof(long long), st));
    TFC_HIP(e->blob_own.alloc(0, st));
    e->blob.p = e->blob_own.p;
    e->blob_capacity = 0;
    e->total = 0;
    e->total_known = true;
    e->finalized = true;
    return 0;
  }
  std::vector<ChunkRef> refs;
  size_t capacity = 4 * static_cast<size_t>(n) + 64;   // Finalize bytes (<= 2 per stream) + slack
  for (auto& c : e->chunks) {
    refs.push_back(ChunkRef{c.data.as<uint8_t>(), c.off.as<long long>(), c.len_p, c.stride});
    capacity += c.data_bytes;
  }
  DevBuf d_refs, tail, length;
  ChunkList list;
  list.more = nullptr;
  list.n = static_cast<int>(refs.size());
  if (refs.size() <= static_cast<size_t>(kInlineChunks)) {
    for (size_t i = 0; i < refs.size(); ++i) list.inline_refs[i] = refs[i];
  } else {
    TFC_HIP(d_refs.alloc(sizeof(ChunkRef) * refs.size(), st));
    TFC_HIP(hipMemcpyAsync(d_refs.p, refs.data(), sizeof(ChunkRef) * refs.size(), hipMemcpyHostToDevice, st));
    TFC_HIP(hipStreamSynchronize(st));               // `refs` goes away with this frame
    list.more = d_refs.as<ChunkRef>();
  }
  TFC_HIP(tail.alloc((sizeof(Tail) + sizeof(long long)) * n + 16, st));
  long long* const length_p = reinterpret_cast<long long*>(tail.as<uint8_t>() + ((sizeof(Tail) * n + 15) & ~size_t{15}));
  const unsigned tb = static_cast<unsigned>(ceil_div(n, 256));
  hipLaunchKernelGGL(enc_tail_kernel, dim3(tb), dim3(256), 0, st, e->state.as<uint4>(), n,
                     list, static_cast<int>(refs.size()), e->family != kGeneric ? 1 : 0,
                     tail.as<Tail>(), length_p);
  hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, st, length_p, n,
                     e->offsets.as<long long>());
  if (exact) {
    long long total = 0;
    unsigned int oflag = 0;
    unsigned long long host_status[4];
    TFC_HIP(hipMemcpyAsync(&total, e->offsets.as<long long>() + n, sizeof(long long),
                           hipMemcpyDeviceToHost, st));
    TFC_HIP(hipMemcpyAsync(&oflag, e->oflag.p, sizeof(oflag), hipMemcpyDeviceToHost, st));
    TFC_HIP(hipMemcpyAsync(host_status, e->status.p, sizeof(host_status), hipMemcpyDeviceToHost, st));
    TFC_HIP(hipStreamSynchronize(st));
    if (host_status[0] != ~0ull) return encoder_error(e, host_status);
    if (oflag) return fail("a stream outgrew its output slab (more than 16 bits per symbol on average with "
                           "deferred errors on: encode with tfc_encoder_set_deferred_errors(e, 0))");
    capacity = static_cast<size_t>(total) + 64;
    e->total = total;
    e->total_known = true;
  }
  TFC_HIP(e->blob_own.alloc(capacity, st));
  e->blob.p = e->blob_own.p;
  e->blob_capacity = static_cast<int64_t>(capacity);
  hipLaunchKernelGGL(enc_pack_kernel, dim3(static_cast<unsigned>(ceil_div(n, kWavesPerBlock))),
                     dim3(kBlock), 0, st, n, list, static_cast<int>(refs.size()),
                     tail.as<Tail>(), e->offsets.as<long long>(), e->blob.as<uint8_t>());
  TFC_HIP(hipGetLastError());
  e->chunks.clear();     // stream-ordered frees: no need to wait for the pack kernel here
  e->finalized = true;
  return 0;
}

}  // namespace

extern "C" int tfc_encoder_finalize(tfc_encoder* e, void* stream, int64_t* total_bytes) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (e->finalized && !e->total_known) {
    const int rc = tfc_encoder_status(e, stream, total_bytes);
    return rc;
  }
  if (finalize_impl(e, st, true)) return 1;
  *total_bytes = e->total;
  return 0;
}

extern "C" int tfc_encoder_finalize_device(tfc_encoder* e, void* stream) {
  return finalize_impl(e, static_cast<hipStream_t>(stream), false);
}

extern "C" int tfc_encoder_status(tfc_encoder* e, void* stream, int64_t* total_bytes) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  e->touch(st);
  unsigned int oflag = 0;
  unsigned long long host_status[4];
  long long total = 0;
  TFC_HIP(hipMemcpyAsync(&oflag, e->oflag.p, sizeof(oflag), hipMemcpyDeviceToHost, st));
  TFC_HIP(hipMemcpyAsync(host_status, e->status.p, sizeof(host_status), hipMemcpyDeviceToHost, st));
  if (e->finalized && !e->total_known)
    TFC_HIP(hipMemcpyAsync(&total, e->offsets.as<long long>() + e->streams, sizeof(long long),
                           hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (host_status[0] != ~0ull) return encoder_error(e, host_status);
  if (oflag) return fail("a stream outgrew its output slab (more than 16 bits per symbol on average with "
                         "deferred errors on: encode with tfc_encoder_set_deferred_errors(e, 0))");
  if (e->finalized && !e->total_known) {
    e->total = total;
    e->total_known = true;
  }
  if (total_bytes) *total_bytes = e->finalized ? e->total : -1;
  return 0;
}

extern "C" int tfc_encoder_result(const tfc_encoder* e, const uint8_t** blob, const int64_t** offsets) {
  if (!e->finalized) return fail("encoder handle is not finalized");
  *blob = e->blob.as<uint8_t>();
  *offsets = e->offsets.as<int64_t>();
  return 0;
}

extern "C" int tfc_encoder_capacity(const tfc_encoder* e, int64_t* bytes) {
  if (!e->finalized) return fail("encoder handle is not finalized");
  *bytes = e->blob_capacity;
  return 0;
}

extern "C" int tfc_encoder_read(const tfc_encoder* e, uint8_t* blob_dst, int64_t* offsets_dst,
                                int dst_on_device, void* stream) {
  if (!e->finalized) return fail("encoder handle is not finalized");
  if (!e->total_known) return fail("tfc_encoder_status must read the size of a device-finalized handle first");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const hipMemcpyKind k = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (offsets_dst)
    TFC_HIP(hipMemcpyAsync(offsets_dst, e->offsets.p, sizeof(int64_t) * (e->streams + 1), k, st));
  if (blob_dst && e->total)
    TFC_HIP(hipMemcpyAsync(blob_dst, e->blob.p, static_cast<size_t>(e->total), k, st));
  if (!dst_on_device) TFC_HIP(hipStreamSynchronize(st));
  return 0;
}

extern "C" void tfc_encoder_destroy(tfc_encoder* e) { delete e; }

// ===========================================================================
// Host side: decoder
// ===========================================================================

struct tfc_decoder {
  const tfc_tables* tables = nullptr;
  int64_t streams = 0;
  int mode = TFC_MODE_AUTO;
  int family = -1;
  DevBuf blob, offsets, ctl;               // blob / offsets: owned copies of host input only
  std::shared_ptr<DevBuf> ctl_group;       // create_many: ctl is a slice of a shared allocation
  DevView state, status;                   // uint4 [streams]; u64 first index error (views of ctl)
  const uint8_t* blob_p = nullptr;         // device bytes the kernels read (owned or borrowed)
  const long long* off_p = nullptr;
  void touch(hipStream_t s) {              // released in the order of the stream that used the handle last
    blob.touch(s);
    offsets.touch(s);
    ctl.touch(s);
    if (ctl_group) ctl_group->touch(s);
  }
};

extern "C" int tfc_decoder_create(const tfc_tables* tables, const uint8_t* blob,
                                  const int64_t* offsets, int64_t streams, int src_on_device,
                                  void* stream, tfc_decoder** out) {
  *out = nullptr;
  if (!tables) return fail("tables is null");
  if (streams < 0) return fail("negative stream count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::unique_ptr<tfc_decoder> d(new tfc_decoder);
  d->tables = tables;
  d->streams = streams;
  if (src_on_device) {
    // Borrowed, like the reference's decoder, which reads its source in place and requires the
    // caller to keep it alive (cc/lib/range_coder.h:74-77): no copy, no size read-back, no sync.
    d->blob_p = blob;
    d->off_p = reinterpret_cast<const long long*>(offsets);
  } else {
    const int64_t total = offsets[streams];
    TFC_HIP(d->offsets.alloc(sizeof(int64_t) * (streams + 1), st));
    TFC_HIP(hipMemcpyAsync(d->offsets.p, offsets, sizeof(int64_t) * (streams + 1), hipMemcpyHostToDevice, st));
    TFC_HIP(d->blob.alloc(static_cast<size_t>(total), st));
    if (total) TFC_HIP(hipMemcpyAsync(d->blob.p, blob, static_cast<size_t>(total), hipMemcpyHostToDevice, st));
    d->blob_p = d->blob.as<uint8_t>();
    d->off_p = d->offsets.as<long long>();
  }
  TFC_HIP(d->ctl.alloc(16 + sizeof(uint4) * std::max<int64_t>(streams, 1), st));
  d->status.p = d->ctl.p;
  d->state.p = d->ctl.as<uint8_t>() + 16;
  hipLaunchKernelGGL(dec_open_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(1, ceil_div(streams, 256)))),
                     dim3(256), 0, st, d->blob_p, d->off_p, streams, d->state.as<uint4>(),
                     d->status.as<unsigned long long>());
  if (!src_on_device) TFC_HIP(hipStreamSynchronize(st));  // host buffers may go away
  *out = d.release();
  return 0;
}

// n decoders on the device-resident strings of n finalized encoders (borrowed, like tfc_decoder_create
// with src_on_device): one allocation, one launch.
extern "C" int tfc_decoder_create_many(const tfc_tables* tables, int n, tfc_encoder* const* from, void* stream,
                                       tfc_decoder** out) {
  if (!tables) return fail("tables is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int k = 0; k < n; ++k) out[k] = nullptr;
  if (n <= 0) return 0;
  for (int k = 0; k < n; ++k) {
    if (!from[k]->finalized) return fail("encoder handle is not finalized");
    if (from[k]->streams != from[0]->streams) return fail("tfc_decoder_create_many: handles must share the stream count");
  }
  const int64_t streams = from[0]->streams;
  std::vector<std::unique_ptr<tfc_decoder>> made;
  for (int g0 = 0; g0 < n; g0 += kMaxDecoderJobs) {
    const int gn = std::min(kMaxDecoderJobs, n - g0);
    DecoderJobs f;
    f.streams = streams;
    f.n = gn;
    f.ctl_bytes = 16 + static_cast<long long>(sizeof(uint4)) * std::max<int64_t>(streams, 1);
    auto group = std::make_shared<DevBuf>();
    TFC_HIP(group->alloc(static_cast<size_t>(f.ctl_bytes) * gn, st));
    f.ctl = group->as<uint8_t>();
    for (int k = 0; k < gn; ++k) {
      std::unique_ptr<tfc_decoder> d(new tfc_decoder);
      d->tables = tables;
      d->streams = streams;
      d->blob_p = from[g0 + k]->blob.as<uint8_t>();
      d->off_p = from[g0 + k]->offsets.as<long long>();
      d->ctl_group = group;
      d->status.p = f.ctl + k * f.ctl_bytes;
      d->state.p = f.ctl + k * f.ctl_bytes + 16;
      f.job[k].blob = d->blob_p;
      f.job[k].off = d->off_p;
      f.job[k].ok = nullptr;
      made.push_back(std::move(d));
    }
    hipLaunchKernelGGL(dec_open_many_kernel,
                       dim3(static_cast<unsigned>(std::max<int64_t>(1, ceil_div(streams, 256))), static_cast<unsigned>(gn)),
                       dim3(256), 0, st, f);
  }
  TFC_HIP(hipGetLastError());
  for (int k = 0; k < n; ++k) out[k] = made[k].release();
  return 0;
}

// EntropyDecodeFinalize of n handles, no synchronisation: ok DEV uint8 [n, streams].
extern "C" int tfc_decoder_finalize_device_many(int n, tfc_decoder* const* ds, uint8_t* ok, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  for (int k = 0; k < n; ++k) ds[k]->touch(st);
  const int64_t streams = ds[0]->streams;
  for (int g0 = 0; g0 < n; g0 += kMaxDecoderJobs) {
    const int gn = std::min(kMaxDecoderJobs, n - g0);
    // handles of one create_many group sit at a fixed stride; anything else goes handle by handle
    bool strided = streams > 0 && ds[g0]->ctl_group != nullptr;
    const long long ctl_bytes = 16 + static_cast<long long>(sizeof(uint4)) * std::max<int64_t>(streams, 1);
    for (int k = 0; k < gn && strided; ++k)
      if (ds[g0 + k]->ctl_group != ds[g0]->ctl_group || ds[g0 + k]->streams != streams ||
          ds[g0 + k]->status.as<uint8_t>() != ds[g0]->status.as<uint8_t>() + k * ctl_bytes)
        strided = false;
    if (!strided) {
      for (int k = 0; k < gn; ++k)
        if (tfc_decoder_finalize_device(ds[g0 + k], ok + (g0 + k) * ds[g0 + k]->streams, stream)) return 1;
      continue;
    }
    DecoderJobs f;
    f.streams = streams;
    f.n = gn;
    f.ctl_bytes = ctl_bytes;
    f.ctl = ds[g0]->status.as<uint8_t>();
    for (int k = 0; k < gn; ++k) {
      f.job[k].blob = ds[g0 + k]->blob_p;
      f.job[k].off = ds[g0 + k]->off_p;
      f.job[k].ok = ok + (g0 + k) * streams;
    }
    hipLaunchKernelGGL(dec_close_many_kernel,
                       dim3(static_cast<unsigned>(ceil_div(streams, 256)), static_cast<unsigned>(gn)), dim3(256), 0, st, f);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_decoder_set_mode(tfc_decoder* d, int mode) {
  if (mode != TFC_MODE_AUTO && mode != TFC_MODE_LATENCY && mode != TFC_MODE_THROUGHPUT)
    return fail("unknown mode %d", mode);
  d->mode = mode;      // the decoder state has one encoding: the mode may change between calls
  d->family = -1;
  return 0;
}

namespace {

// One wave per stream (dec_fast_kernel where the tables qualify, else dec_kernel) for ONE handle; `job_guard` (or null): a
// device flag without which the launch does nothing.
template <typename Dst>
int launch_wave_decoder(tfc_decoder* d, const int32_t* index, int64_t elems, const Dst& dst, hipStream_t st,
                        const unsigned int* job_guard) {
  const tfc_tables* t = d->tables;
  DecParams p;
  p.tab = view_of(t);
  p.index = index;
  p.streams = d->streams;
  p.elems = elems;
  p.blob = d->blob_p;
  p.off = d->off_p;
  p.state = d->state.as<uint4>();
  p.first_error = d->status.as<unsigned long long>();
  p.only_flagged = nullptr;
  p.job_guard = job_guard;
  p.blocks_after_escape = 64;
  const size_t lds = table_lds_bytes(t);
  const unsigned blocks = static_cast<unsigned>(ceil_div(d->streams, kWavesPerBlock));
  const size_t fast_lds = sizeof(int32_t) * ((t->dec_words + 3) & ~3) + sizeof(int4) * t->rows.size();
  const bool fast_ok = t->dec_fast_ok && fast_lds <= 160 * 1024;
  if (fast_ok) {
    const int waves = static_cast<int>(waves_wanted(d->streams));
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_fast_kernel<Dst>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_request(fast_lds))));
    hipLaunchKernelGGL((dec_fast_kernel<Dst>),
                       dim3(static_cast<unsigned>(ceil_div(d->streams, waves))), dim3(64 * waves),
                       lds_request(fast_lds), st, p, dst);
  } else if (lds) {
    TFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_kernel<true, Dst>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((dec_kernel<true, Dst>), dim3(blocks), dim3(kBlock), lds, st, p, dst);
  } else {
    hipLaunchKernelGGL((dec_kernel<false, Dst>), dim3(blocks), dim3(kBlock), 0, st, p, dst);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

// Lane-per-stream family: n handles (same tables, same stream count) decoded by one launch per
// kMaxLaneJobs of them.  Where the geometry allows, the pipelined kernels of range_pipe.h take the launch (chain into
// raw rows, then the parallel parse into `dsts`) and the lane-per-stream kernel behind them only decodes the jobs
// they gave up on.
template <typename Dst>
int decode_lanes_many(tfc_decoder* const* ds, int n, const Dst* dsts, const int32_t* const* indexes,
                      int64_t elems, hipStream_t st) {
  constexpr int kMaxLaneJobs = DecLaneJobs<Dst>::kMax;
  const tfc_tables* t = ds[0]->tables;
  const int64_t streams = ds[0]->streams;
  const bool indexed = indexes && indexes[0];
  LaneArgs la;
  la.image = t->d_lane_image.as<uint32_t>();
  la.bytes = t->lane_dec_bytes;
  la.ntab = static_cast<int>(t->rows.size());
  la.precision = t->lane_precision;
  la.cap = 0;
  la.defer = lanes_escape_defer();
  la.guard = nullptr;
  la.lds_image = (t->lane_dec_bytes + 1023) & ~1023;
  using WaveLds = DecWaveLds<typename Dst::elem>;
  la.lds_wave = indexed ? WaveLds::kBytes : WaveLds::kIndex;
  // the lane-per-stream kernel (the pipelined kernels' fallback, or the launch's decoder) needs its image + a wave's
  // staging in a CU's LDS; tables whose image is larger (bls2017's 192 x 128 symbols: 176 KB) have the wave-per-stream
  // kernels as the fallback, handle by handle
  const bool lanes_fit = la.lds_image + la.lds_wave <= 160 * 1024;
  const int block = lanes_fit ? std::min(lanes_block(streams * n), 64 * ((160 * 1024 - la.lds_image) / la.lds_wave)) : 64;
  const int lds_bytes = la.lds_image + (block / 64) * la.lds_wave;
  const void* fn = indexed ? reinterpret_cast<const void*>(&dec_lanes_kernel<true, Dst>)
                           : reinterpret_cast<const void*>(&dec_lanes_kernel<false, Dst>);
  if (lanes_fit) TFC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  // pipelined launch plan: groups of 64 streams; rows = steps of the chain (one per element, plus the bits of
  // escape codes: a quarter more is planned for when the tables have escape rows)
  PipeDecArgs pa;
  LaneArgs pla = la;
  pa.groups_per_job = static_cast<int>(ceil_div(streams, 64));
  // (+ the blocks lanes sit out or spend finishing an escape code before the wave's tail pass, and the tail pass)
  const int64_t rows64 = ((elems + (t->any_escape ? elems / 4 + 64 + 24 * kPipeBlock : 2 * kPipeBlock)) + kPipeBlock - 1) / kPipeBlock * kPipeBlock;
  pa.rows = static_cast<int>(std::min<int64_t>(rows64, (int64_t{1} << 30)));
  const size_t raw_bytes = static_cast<size_t>(pa.rows) * 128;      // 16-bit entries, 64 lanes
  const size_t rec_bytes = (static_cast<size_t>(pa.rows) / kPipeBlock + 1) * 256;
  const size_t group_bytes = raw_bytes + rec_bytes + 64 * sizeof(uint4) + sizeof(unsigned int);
  const size_t job_bytes = group_bytes * pa.groups_per_job + (indexed ? 2 * static_cast<size_t>(streams) * elems : 0);
  pla.lds_wave = (indexed ? PipeDecLds::kBytes : PipeDecLds::kRows) + PipeDecLds::kStage;
  const size_t kPipeTempBytes = pipe_temp_bytes();
  // Which image the chain runs on, and how many chain waves share a workgroup's copy of it.  A launch's chain
  // workgroups must all be resident at once (a second round doubles the launch's time and the parse next to the chain
  // gives up on groups that do not move), one wave per SIMD at most: the full image where it fits and hosts the call's
  // waves (BASELINE config 2: 150 KB, one wave per CU = 32 batches per launch), else the compact one (92 KB for config 2:
  // four waves per CU, 128 batches per launch; 115 KB for bls2017's tables, whose full image does not fit at all).
  const int cus_d = pipe_device_info().cus;
  const int64_t waves_call = static_cast<int64_t>(pa.groups_per_job) * std::min(n, kMaxLaneJobs);
  auto waves_fit = [&](int image_bytes) { return std::max(0, (160 * 1024 - image_bytes) / pla.lds_wave); };
  const int pair_image = (t->pair_dec_bytes + 1023) & ~1023;
  const int fit_full = waves_fit(pla.lds_image), fit_pairs = t->pairs_ok ? waves_fit(pair_image) : 0;
  bool pairs = false;
  switch (pipe_format()) {
    case 1: pairs = false; break;
    case 2: pairs = fit_pairs >= 1; break;
    // (beyond three quarters of the CUs under full-image chain workgroups — each fills its CU's LDS — the parse next to the
    // chain finds no room and runs behind it: 32 batches of config 2 decode in 10.1 ms on the full image, 9.1 on the
    // compact one, whose workgroups leave 50 KB of every CU to two parse workgroups)
    default: pairs = fit_pairs >= 1 && (fit_full < 1 || 4 * waves_call > 3 * static_cast<int64_t>(cus_d) * std::min(fit_full, 4));
  }
  if (pairs) {
    pla.image = t->d_pair_image.as<uint32_t>();
    pla.bytes = t->pair_dec_bytes;
    pla.lds_image = pair_image;
  }
  const int fit = pairs ? fit_pairs : fit_full;
  // waves per workgroup: as few as host the call on the chip's CUs (a chain wave wants a SIMD to itself); under
  // tfc_set_chip_shared — other kernels beside the coder's — four, so that the chains hold few CUs' LDS
  int per = static_cast<int>(std::min<int64_t>(8, std::max<int64_t>(1, ceil_div(waves_call, static_cast<int64_t>(cus_d)))));
  if (g_chip_shared.load(std::memory_order_relaxed) != 0) per = std::max(per, static_cast<int>(std::min<int64_t>(4, pa.groups_per_job)));
  if (pipe_waves_env() > 0) per = pipe_waves_env();
  per = std::min(per, std::min(fit, pa.groups_per_job));
  const int pblock = 64 * std::max(per, 0);
  // (precision 16: a quotient can equal the "no escape symbol" mark of dec_chain_kernel's directory, 0xFFFF)
  const bool pipe = pipe_enabled() && pblock >= 64 && rows64 < (int64_t{1} << 30) && job_bytes <= kPipeTempBytes &&
                    (!indexed || la.ntab < 4096) && t->lane_precision <= 15 &&
                    (static_cast<int64_t>(pa.rows) / kParseRows + 1) * pa.groups_per_job * kMaxLaneJobs < (int64_t{1} << 31);
  const int plds = pla.lds_image + (pblock / 64) * pla.lds_wave;
  const size_t chain_wgs_per_job = static_cast<size_t>(ceil_div(pa.groups_per_job, std::max(1, pblock / 64)));
  const size_t resident_wgs = static_cast<size_t>(cus_d) * std::max<size_t>(1, (160 * 1024) / std::max(1, plds));
  const size_t resident_jobs = std::max<size_t>(1, resident_wgs / std::max<size_t>(1, chain_wgs_per_job));
  const int per_launch = !pipe ? kMaxLaneJobs
                               : static_cast<int>(std::max<size_t>(1, std::min<size_t>(std::min<size_t>(kMaxLaneJobs, resident_jobs),
                                                                                      kPipeTempBytes / std::max<size_t>(job_bytes, 1))));
  const void* pfn = nullptr;
  if (pipe) {
    pfn = pairs ? (indexed ? reinterpret_cast<const void*>(&dec_chain_kernel<true, true>) : reinterpret_cast<const void*>(&dec_chain_kernel<false, true>))
                : (indexed ? reinterpret_cast<const void*>(&dec_chain_kernel<true, false>) : reinterpret_cast<const void*>(&dec_chain_kernel<false, false>));
    TFC_HIP(hipFuncSetAttribute(pfn, hipFuncAttributeMaxDynamicSharedMemorySize, plds));
  }
  if (!pipe && !lanes_fit) {
    // neither the pipelined kernels nor the lane-per-stream kernel can take the call: one wave per stream
    for (int k = 0; k < n; ++k) {
      ds[k]->family = kFast;
      KernelTimer timer("dec_kernel", st);
      if (launch_wave_decoder(ds[k], indexed ? indexes[k] : nullptr, elems, dsts[k], st, nullptr)) return 1;
    }
    return 0;
  }
  const PipePairMap pm{t->d_pair_adjust.as<int>()};
  for (int g0 = 0; g0 < n; g0 += per_launch) {
    const int gn = std::min(per_launch, n - g0);
    DecLaneJobs<Dst> jobs;
    jobs.streams = streams;
    jobs.elems = elems;
    jobs.blocks_per_job = static_cast<int>(ceil_div(streams, block));
    jobs.n = gn;
    for (int k = 0; k < gn; ++k) {
      tfc_decoder* d = ds[g0 + k];
      d->family = kLanes;
      DecLaneJob<Dst>& J = jobs.job[k];
      J.dst = dsts[g0 + k];
      J.index = indexed ? indexes[g0 + k] : nullptr;
      J.blob = d->blob_p;
      J.off = d->off_p;
      J.state = d->state.as<uint4>();
      J.first_error = d->status.as<unsigned long long>();
    }
    KernelTimer timer("dec_kernel", st);
    DevBuf temp;
    bool piped = pipe;
    la.guard = nullptr;
    if (pipe) do {
      const size_t groups = static_cast<size_t>(pa.groups_per_job) * gn;
      const size_t tiles = static_cast<size_t>(pa.rows) / kParseRows + 1;
      const size_t raw_all = raw_bytes * groups, rec_all = rec_bytes * groups, st_all = 64 * sizeof(uint4) * groups;
      const size_t kend_all = (sizeof(unsigned int) * groups + 255) & ~size_t{255};
      const size_t addr_all = indexed ? ((2 * static_cast<size_t>(streams) * elems * gn + 255) & ~size_t{255}) : 0;
      // zeroed per launch: fallback flags (64 jobs), chain workgroups started, rows released per group, tiles taken
      const size_t flags_all = (512 + sizeof(unsigned int) * groups * (1 + tiles) + 255) & ~size_t{255};
      if (temp.alloc(raw_all + rec_all + st_all + kend_all + addr_all + flags_all, st) != hipSuccess) {
        // no room for the raw rows: the lane-per-stream kernel decodes this launch
        (void)hipGetLastError();
        temp.p = nullptr;
        piped = false;
        break;
      }
      uint8_t* base = temp.as<uint8_t>();
      pa.raw = reinterpret_cast<unsigned short*>(base);
      pa.posrec = reinterpret_cast<unsigned int*>(base + raw_all);
      pa.state_out = reinterpret_cast<uint4*>(base + raw_all + rec_all);
      pa.kend = reinterpret_cast<unsigned int*>(base + raw_all + rec_all + st_all);
      unsigned short* rowaddr = reinterpret_cast<unsigned short*>(base + raw_all + rec_all + st_all + kend_all);
      pa.rowaddr = rowaddr;
      pa.fallback = reinterpret_cast<unsigned int*>(base + raw_all + rec_all + st_all + kend_all + addr_all);
      pa.started = pa.fallback + 64;
      pa.progress = pa.fallback + 128;
      pa.tile_done = pa.progress + groups;
      pa.groups = static_cast<int>(groups);
      pa.poll_ticks = 200000;         // 2 ms: the chain releases rows every ~0.2 ms
      // (kend of a group the chain gave up on stays unset: the parse skips the whole job under its fallback flag)
      TFC_HIP(hipMemsetAsync(pa.fallback, 0, flags_all, st));
      g_pipe_launches.fetch_add(1, std::memory_order_relaxed);
      PipeDecJobs cj;
      cj.streams = streams;
      cj.elems = elems;
      cj.blocks_per_job = static_cast<int>(ceil_div(pa.groups_per_job, pblock / 64));
      cj.n = gn;
      for (int k = 0; k < gn; ++k) cj.job[k] = PipeDecJob{jobs.job[k].blob, jobs.job[k].off, jobs.job[k].state};
      if (indexed) {
        PipeRowJobs rj;
        rj.per_job = streams * elems;
        rj.ntab = la.ntab;
        rj.n = gn;
        for (int k = 0; k < gn; ++k) { rj.job[k].index = jobs.job[k].index; rj.job[k].first_error = jobs.job[k].first_error; }
        hipLaunchKernelGGL(dec_rows_kernel, dim3(static_cast<unsigned>(ceil_div(rj.per_job, 1024)), static_cast<unsigned>(gn)),
                           dim3(256), 0, st, rj, rowaddr);
      }
      // The chain on the library's own stream, the parse next to it on the caller's: tiles of raw rows are turned
      // into elements as the chain releases them, and a second pass behind the chain takes what the first left
      // (the last rows, tiles it did not get in time) and commits the successor states.
      // (a large launch only: see encode_lanes_many)
      const int overlap = groups >= 64 ? pipe_overlap() : 0;
      SideStream side;
      if (overlap) {
        if (side_stream(st, &side)) return 1;
        TFC_HIP(hipEventRecord(side.fork, st));
        TFC_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
      }
      const hipStream_t cst = overlap ? side.stream : st;
      const dim3 cgrid(static_cast<unsigned>(cj.blocks_per_job * gn));
      const dim3 pgrid(static_cast<unsigned>(groups * tiles));
      // The pass next to the chain takes the tiles every stream is sure to reach (a row per element at least); the rows
      // beyond — escape codes' bit rows past the last element's place, a fraction of a percent — and the planned-for
      // capacity behind them (a quarter more) are the second pass's: workgroups of tiles that never fill would only be
      // dispatched behind the last real ones, wait for the chain's end and leave (16 000 of them in the 20-batch launch).
      const size_t ctiles = std::min<size_t>(tiles, static_cast<size_t>(ceil_div(elems, static_cast<int64_t>(kParseRows))));
      const dim3 pgrid_next(static_cast<unsigned>(groups * ctiles));
      {
        KernelTimer t2("dec_chain", cst);
        if (pairs && indexed) hipLaunchKernelGGL((dec_chain_kernel<true, true>), cgrid, dim3(pblock), plds, cst, cj, pla, pa);
        else if (pairs) hipLaunchKernelGGL((dec_chain_kernel<false, true>), cgrid, dim3(pblock), plds, cst, cj, pla, pa);
        else if (indexed) hipLaunchKernelGGL((dec_chain_kernel<true, false>), cgrid, dim3(pblock), plds, cst, cj, pla, pa);
        else hipLaunchKernelGGL((dec_chain_kernel<false, false>), cgrid, dim3(pblock), plds, cst, cj, pla, pa);
      }
      auto parse = [&](int concurrent) {
        pa.concurrent = concurrent;
        const dim3 g = concurrent ? pgrid_next : pgrid;
        const DecRow* dd = t->d_dec_dir.as<DecRow>();
        if (pairs && indexed) hipLaunchKernelGGL((dec_parse_kernel<true, true, Dst>), g, dim3(256), 0, st, jobs, pa, dd, la.ntab, pm);
        else if (pairs) hipLaunchKernelGGL((dec_parse_kernel<false, true, Dst>), g, dim3(256), 0, st, jobs, pa, dd, la.ntab, pm);
        else if (indexed) hipLaunchKernelGGL((dec_parse_kernel<true, false, Dst>), g, dim3(256), 0, st, jobs, pa, dd, la.ntab, pm);
        else hipLaunchKernelGGL((dec_parse_kernel<false, false, Dst>), g, dim3(256), 0, st, jobs, pa, dd, la.ntab, pm);
      };
      if (overlap) {
        KernelTimer t2("dec_parse_next", st);      // (next to the chain: as long as the chain, by construction)
        // (the chain's workgroups first: where one takes a CU's whole LDS it cannot be placed next to parse workgroups)
        hipLaunchKernelGGL(enc_gate_kernel, dim3(1), dim3(1), 0, st, pa.started, cgrid.x, static_cast<long long>(20000));
        parse(1);
      }
      if (overlap) {
        TFC_HIP(hipEventRecord(side.join, side.stream));
        TFC_HIP(hipStreamWaitEvent(st, side.join, 0));
      }
      {
        KernelTimer t2("dec_parse", st);           // behind the chain: what the first pass left
        parse(0);
      }
      la.guard = pa.fallback;
    } while (false);
    const dim3 grid(static_cast<unsigned>(jobs.blocks_per_job * gn));
    if (piped && !pipe_fallback_launch()) {}
    else if (!lanes_fit) {
      // (the tables' lane image does not fit a CU) what the pipelined kernels gave up on — or, without temporaries, the
      // whole launch — on the wave-per-stream kernels, handle by handle under its fallback flag
      for (int k = 0; k < gn; ++k)
        if (launch_wave_decoder(ds[g0 + k], jobs.job[k].index, elems, dsts[g0 + k], st, piped ? pa.fallback + k : nullptr)) return 1;
    }
    else if (indexed) hipLaunchKernelGGL((dec_lanes_kernel<true, Dst>), grid, dim3(block), lds_bytes, st, jobs, la);
    else hipLaunchKernelGGL((dec_lanes_kernel<false, Dst>), grid, dim3(block), lds_bytes, st, jobs, la);
  }
  TFC_HIP(hipGetLastError());
  return 0;
}

// channel mode, bottleneck outputs: the int32 blocks into a stream-ordered temporary, then the dequantise pass
template <typename Dst>
int decode_lanes_dequantized(tfc_decoder* const* ds, int n, const Dst* dsts, int64_t elems, hipStream_t st) {
  const long long per = static_cast<long long>(ds[0]->streams) * elems;
  const int ntab = static_cast<int>(ds[0]->tables->rows.size());
  DevBuf tmp;
  TFC_HIP(tmp.alloc(sizeof(int32_t) * static_cast<size_t>(per) * n, st));
  std::vector<OutInt32> outs(n);
  for (int k = 0; k < n; ++k) outs[k] = OutInt32{tmp.as<int32_t>() + static_cast<long long>(k) * per};
  if (decode_lanes_many(ds, n, outs.data(), static_cast<const int32_t* const*>(nullptr), elems, st)) return 1;
  for (int k = 0; k < n; ++k)
    hipLaunchKernelGGL((lanes_dequantize_kernel<Dst>),
                       dim3(static_cast<unsigned>(ceil_div(elems, 1024)), static_cast<unsigned>(ds[0]->streams)), dim3(256), 0,
                       st, dsts[k], outs[k].out, static_cast<unsigned int>(elems), static_cast<unsigned int>(ntab));
  TFC_HIP(hipGetLastError());
  return 0;
}

// Family of a decode call; lanes only when the tables' image plus one wave's staging fits the CU — the lane-per-stream
// kernels' own image, or (round 6) the compact image of the pipelined decoder, whose fallback is then the wave-per-stream
// kernel (decode_lanes_many).
template <typename Elem>
int decoder_family(const tfc_decoder* d, int64_t streams_in_launch, int64_t elems, bool indexed, bool fast_ok) {
  const tfc_tables* t = d->tables;
  int family = select_family(t, d->mode, streams_in_launch, elems, fast_ok);
  using WaveLds = DecWaveLds<Elem>;
  const int need = ((t->lane_dec_bytes + 1023) & ~1023) + (indexed ? WaveLds::kBytes : WaveLds::kIndex);
  const int need_pairs = ((t->pair_dec_bytes + 1023) & ~1023) + (indexed ? PipeDecLds::kBytes : PipeDecLds::kRows) + PipeDecLds::kStage;
  const bool compact_ok = t->pairs_ok && pipe_enabled() && pipe_format() != 1 && need_pairs <= 160 * 1024;
  if (family == kLanes && need > 160 * 1024 && !compact_ok) family = fast_ok ? kFast : kGeneric;
  return family;
}

template <typename Dst>
int run_decode(tfc_decoder* d, const int32_t* index, int64_t elems, const Dst& dst,
               hipStream_t st) {
  if (elems < 0) return fail("negative element count");
  d->touch(st);
  if (d->streams == 0 || elems == 0) return 0;
  const tfc_tables* t = d->tables;
  if (t->rows.empty()) return fail("index=0 not in range [0, 0)");
  const size_t fast_lds = sizeof(int32_t) * ((t->dec_words + 3) & ~3) + sizeof(int4) * t->rows.size();
  const bool fast_ok = t->dec_fast_ok && fast_lds <= 160 * 1024;
  const int family = decoder_family<typename Dst::elem>(d, d->streams, elems, index != nullptr, fast_ok);
  d->family = family;
  if (family == kLanes) {
    if constexpr (!std::is_same<Dst, OutInt32>::value) {
      // (the pipelined kernels dequantise in their parse pass)
      if (!index && !pipe_enabled()) return decode_lanes_dequantized(&d, 1, &dst, elems, st);
    }
    return decode_lanes_many(&d, 1, &dst, &index, elems, st);
  }
  KernelTimer timer("dec_kernel", st);
  return launch_wave_decoder(d, index, elems, dst, st, nullptr);
}

}  // namespace

extern "C" int tfc_decoder_decode_many(int n, tfc_decoder* const* ds, const int32_t* const* indexes,
                                       int32_t* const* outs, int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  if (elems < 0) return fail("negative element count");
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) ds[k]->touch(st);
  for (int k = 0; k < n; ++k) {
    const tfc_decoder* d = ds[k];
    if (d->tables != ds[0]->tables || d->streams != ds[0]->streams)
      return fail("tfc_decoder_decode_many: handles must share tables and stream count");
    if ((indexes && indexes[k]) != (indexes && indexes[0]))
      return fail("tfc_decoder_decode_many: either all jobs carry an index or none");
    if (d->streams == 0 || d->tables->rows.empty()) batch = false;
    if (batch && decoder_family<int32_t>(d, d->streams * n, elems, indexes && indexes[0], true) != kLanes) batch = false;
  }
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (tfc_decoder_decode(ds[k], indexes ? indexes[k] : nullptr, outs[k], elems, stream)) return 1;
    return 0;
  }
  std::vector<OutInt32> dsts(n);
  for (int k = 0; k < n; ++k) dsts[k] = OutInt32{outs[k]};
  return decode_lanes_many(ds, n, dsts.data(), indexes, elems, st);
}

extern "C" int tfc_decoder_decode(tfc_decoder* d, const int32_t* index, int32_t* out,
                                  int64_t elems, void* stream) {
  return run_decode(d, index, elems, OutInt32{out}, static_cast<hipStream_t>(stream));
}

extern "C" int tfc_decoder_decode_dequantized(tfc_decoder* d, const int32_t* index, void* y,
                                              int dtype, const float* qoffset,
                                              const int32_t* cdf_offset, int64_t channels,
                                              int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!index && channels != static_cast<int64_t>(d->tables->rows.size()))
    return fail("channel count %lld does not match table count %lld",
                static_cast<long long>(channels), static_cast<long long>(d->tables->rows.size()));
  if (index && qoffset) return fail("qoffset is not supported in index mode");
  switch (dtype) {
    case 0: return run_decode(d, index, elems, OutDequant<float>{static_cast<float*>(y), qoffset, cdf_offset}, st);
    case 1: return run_decode(d, index, elems, OutDequant<__hip_bfloat16>{static_cast<__hip_bfloat16*>(y), qoffset, cdf_offset}, st);
    case 2: return run_decode(d, index, elems, OutDequant<__half>{static_cast<__half*>(y), qoffset, cdf_offset}, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

namespace {
template <typename T>
int decode_dequantized_many(int n, tfc_decoder* const* ds, void* const* ys, const float* qoffset,
                            const int32_t* cdf_offset, int64_t elems, hipStream_t st) {
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) ds[k]->touch(st);
  for (int k = 0; k < n; ++k) {
    const tfc_decoder* d = ds[k];
    if (d->tables != ds[0]->tables || d->streams != ds[0]->streams)
      return fail("tfc_decoder_decode_dequantized_many: handles must share tables and stream count");
    if (d->streams == 0 || d->tables->rows.empty()) batch = false;
    if (batch && decoder_family<int32_t>(d, d->streams * n, elems, false, true) != kLanes) batch = false;
  }
  std::vector<OutDequant<T>> dsts(n);
  for (int k = 0; k < n; ++k) dsts[k] = OutDequant<T>{static_cast<T*>(ys[k]), qoffset, cdf_offset};
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (run_decode(ds[k], nullptr, elems, dsts[k], st)) return 1;
    return 0;
  }
  for (int g0 = 0; g0 < n; g0 += kMaxFinalizeJobs) {
    const int gn = std::min(kMaxFinalizeJobs, n - g0);
    if (pipe_enabled()) {
      if (decode_lanes_many(ds + g0, gn, dsts.data() + g0, static_cast<const int32_t* const*>(nullptr), elems, st)) return 1;
    } else if (decode_lanes_dequantized(ds + g0, gn, dsts.data() + g0, elems, st)) {
      return 1;
    }
  }
  return 0;
}
}  // namespace

// EntropyDecodeChannel with the dequantise epilogue for n independent handles as one coding launch.
extern "C" int tfc_decoder_decode_dequantized_many(int n, tfc_decoder* const* ds, void* const* ys, int dtype,
                                                   const float* qoffset, const int32_t* cdf_offset,
                                                   int64_t channels, int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  if (elems < 0) return fail("negative element count");
  if (channels != static_cast<int64_t>(ds[0]->tables->rows.size()))
    return fail("channel count %lld does not match table count %lld",
                static_cast<long long>(channels), static_cast<long long>(ds[0]->tables->rows.size()));
  switch (dtype) {
    case 0: return decode_dequantized_many<float>(n, ds, ys, qoffset, cdf_offset, elems, st);
    case 1: return decode_dequantized_many<__hip_bfloat16>(n, ds, ys, qoffset, cdf_offset, elems, st);
    case 2: return decode_dequantized_many<__half>(n, ds, ys, qoffset, cdf_offset, elems, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

namespace {
template <typename T>
int decode_dequantized_indexed_many(int n, tfc_decoder* const* ds, const int32_t* const* indexes, void* const* ys,
                                    const int32_t* cdf_offset, int64_t elems, hipStream_t st) {
  bool batch = elems > 0 && elems < (int64_t{1} << 29);
  for (int k = 0; k < n; ++k) ds[k]->touch(st);
  for (int k = 0; k < n; ++k) {
    const tfc_decoder* d = ds[k];
    if (!indexes[k]) return fail("index is null");
    if (d->tables != ds[0]->tables || d->streams != ds[0]->streams)
      return fail("tfc_decoder_decode_dequantized_indexed_many: handles must share tables and stream count");
    if (d->streams == 0 || d->tables->rows.empty()) batch = false;
    if (batch && decoder_family<int32_t>(d, d->streams * n, elems, true, true) != kLanes) batch = false;
  }
  std::vector<OutDequant<T>> dsts(n);
  for (int k = 0; k < n; ++k) dsts[k] = OutDequant<T>{static_cast<T*>(ys[k]), nullptr, cdf_offset};
  if (!batch) {
    for (int k = 0; k < n; ++k)
      if (run_decode(ds[k], indexes[k], elems, dsts[k], st)) return 1;
    return 0;
  }
  return decode_lanes_many(ds, n, dsts.data(), indexes, elems, st);
}
}  // namespace

// EntropyDecodeIndex with the dequantise epilogue for n independent handles as one coding launch.
extern "C" int tfc_decoder_decode_dequantized_indexed_many(int n, tfc_decoder* const* ds, const int32_t* const* indexes,
                                                           void* const* ys, int dtype, const int32_t* cdf_offset,
                                                           int64_t elems, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return 0;
  if (elems < 0) return fail("negative element count");
  if (!indexes) return fail("index is null");
  switch (dtype) {
    case 0: return decode_dequantized_indexed_many<float>(n, ds, indexes, ys, cdf_offset, elems, st);
    case 1: return decode_dequantized_indexed_many<__hip_bfloat16>(n, ds, indexes, ys, cdf_offset, elems, st);
    case 2: return decode_dequantized_indexed_many<__half>(n, ds, indexes, ys, cdf_offset, elems, st);
    default: return fail("unsupported dtype code %d", dtype);
  }
}

extern "C" int tfc_decoder_finalize_device(tfc_decoder* d, uint8_t* ok_dev, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  d->touch(st);
  const int64_t n = d->streams;
  if (n)
    hipLaunchKernelGGL(dec_close_kernel, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256),
                       0, st, d->state.as<uint4>(), d->off_p, n, ok_dev);
  TFC_HIP(hipGetLastError());
  return 0;
}

extern "C" int tfc_decoder_status(tfc_decoder* d, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  d->touch(st);
  unsigned long long first_error = ~0ull;
  TFC_HIP(hipMemcpyAsync(&first_error, d->status.p, sizeof(first_error), hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (first_error != ~0ull)
    return fail("index=<element %llu> not in range [0, %lld)", first_error,
                static_cast<long long>(d->tables->rows.size()));
  return 0;
}

extern "C" int tfc_decoder_finalize(tfc_decoder* d, uint8_t* ok, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n = d->streams;
  DevBuf d_ok;
  TFC_HIP(d_ok.alloc(std::max<int64_t>(n, 1), st));
  if (tfc_decoder_finalize_device(d, d_ok.as<uint8_t>(), stream)) return 1;
  if (n) TFC_HIP(hipMemcpyAsync(ok, d_ok.p, n, hipMemcpyDeviceToHost, st));
  return tfc_decoder_status(d, stream);
}

extern "C" void tfc_decoder_destroy(tfc_decoder* d) { delete d; }

// ===========================================================================
// Deprecated single-stream ops: RangeEncode / RangeDecode
// (cc/kernels/range_coding_kernels.cc:60-379, range_coding_kernels_util.cc:34-91)
// ===========================================================================

namespace tfc {

// Merged broadcast geometry: data index -> cdf row offset.
struct Broadcast {
  int nd;
  long long shape[6];       // merged data shape
  long long cdf_stride[6];  // cdf elements per step along the axis (0 when broadcast)
  long long width;          // cdf entries per row
};

__device__ inline long long cdf_row_of(const Broadcast& b, long long k) {
  long long off = 0;
  for (int i = b.nd - 1; i >= 0; --i) {
    const long long q = k / b.shape[i];
    off += (k - q * b.shape[i]) * b.cdf_stride[i];
    k = q;
  }
  return off;
}

// RangeEncoder::Finalize (cc/lib/range_coder.cc:266-307) behind the digits already stored:
// returns the stream length.  One lane.
__device__ inline unsigned int finalize_bytes(const EncoderState& st, const DigitSink& o, uint8_t* out) {
  unsigned int n = o.nbytes;
  uint8_t* dst = out + n;
  if (st.pend_digit != 0) {
    dst[0] = (st.pend_digit >> 8) & 0xFF; ++n;
    if ((st.pend_digit & 0xFF) != 0) { dst[1] = st.pend_digit & 0xFF; ++n; }
  } else if (st.base != 0) {
    const unsigned int top = st.base + st.span_m1;
    const unsigned int r24 = ((st.base - 1) >> 24) + 1;
    if (r24 <= (top >> 24)) {
      dst[0] = r24 & 0xFF; ++n;
    } else {
      const unsigned int r16 = ((st.base - 1) >> 16) + 1;
      dst[0] = (r16 >> 8) & 0xFF; ++n;
      if ((r16 & 0xFF) != 0) { dst[1] = r16 & 0xFF; ++n; }
    }
  }
  return n;
}

struct LegacyEncParams {
  const int16_t* data;
  const int32_t* cdf;
  Broadcast geo;
  long long total;
  int precision;
  int check;                         // debug_level > 0
  uint8_t* out;
  unsigned int cap;
  unsigned int* out_len;
  unsigned long long* first_error;
};

__global__ void __launch_bounds__(64) legacy_enc_kernel(LegacyEncParams p) {
  const int lane = threadIdx.x;
  EncoderState st{0u, 0xFFFFFFFFu, 0u, 0u};
  DigitSink o;
  o.dst = p.out;
  o.cap = p.cap;
  o.nbytes = 0;
  o.n = 0;
  o.reg = 0;
  o.overflow = 0;
  const int sh = 16 - p.precision;
  bool failed = false;
  for (long long k0 = 0; k0 < p.total && !failed; k0 += 64) {
    const long long k = k0 + lane;
    int lo = 0, hi = 0;
    bool bad = false;
    if (k < p.total) {
      const long long row = cdf_row_of(p.geo, k);
      long long v = p.data[k];
      if (v < 0 || p.geo.width <= v + 1) {
        bad = true;
        v = 0;
      }
      lo = p.cdf[row + v] << sh;
      hi = p.cdf[row + v + 1] << sh;
    }
    const unsigned long long badmask = __ballot(bad);
    int cnt = static_cast<int>(min<long long>(64, p.total - k0));
    if (badmask != 0) {
      // debug_level 1 reports the first offender; debug_level 0 leaves it
      // undefined in the reference (DCHECK only) — we stop there too.
      const int first = __builtin_ctzll(badmask);
      if (lane == 0) atomicMin(p.first_error, static_cast<unsigned long long>(k0 + first));
      cnt = first;
      failed = true;
    }
    for (int n = 0; n < cnt; ++n) {
      const unsigned int l = __builtin_amdgcn_readlane(lo, n);
      const unsigned int h = __builtin_amdgcn_readlane(hi, n);
      enc_update(st, l, h, o, lane);
    }
  }
  sink_flush(o, lane);
  if (lane == 0) *p.out_len = finalize_bytes(st, o, p.out);
}

struct LegacyDecParams {
  const uint8_t* bytes;
  long long nbytes;
  const int32_t* cdf;
  Broadcast geo;
  long long total;
  int precision;
  int16_t* out;
};

__global__ void __launch_bounds__(64) legacy_dec_kernel(LegacyDecParams p) {
  const int lane = threadIdx.x;
  DecoderState st{0u, 0xFFFFFFFFu, 0u};
  DigitWindow w;
  w.src = p.bytes;
  w.len = p.nbytes;
  w.pulls = 0;
  w.base = 0;
  window_load(w, lane);
  st.window = window_pull(w, lane) << 16;
  st.window |= window_pull(w, lane);
  for (long long k0 = 0; k0 < p.total; k0 += 64) {
    const long long k = k0 + lane;
    long long row = 0;
    if (k < p.total) row = cdf_row_of(p.geo, k);
    const int cnt = static_cast<int>(min<long long>(64, p.total - k0));
    int outv = 0;
    for (int n = 0; n < cnt; ++n) {
      const unsigned int rlo = __builtin_amdgcn_readlane(static_cast<int>(row & 0xFFFFFFFFll), n);
      const unsigned int rhi = __builtin_amdgcn_readlane(static_cast<int>(row >> 32), n);
      const long long r = (static_cast<long long>(rhi) << 32) | rlo;
      const int32_t* base = p.cdf + r;
      auto T = [&](int i) -> int32_t { return base[i]; };
      const int sym = dec_symbol(T, st, 0, static_cast<int>(p.geo.width), p.precision, w, lane);
      outv = tfc_writelane(sym, n, outv);
    }
    if (k < p.total) p.out[k] = static_cast<int16_t>(outv);
  }
}

// ---------------------------------------------------------------------------
// Deprecated UnboundedIndexRangeEncode / Decode
// (cc/kernels/unbounded_index_range_coding_kernels.cc:185-249, 307-367): ONE stream for the
// whole tensor, so one wave; the per-element table work (row, clamping, overflow value, the
// two cdf entries) is done 64 elements at a time across the lanes, the interval updates
// are the serial part.  Out-of-range values: the row's last symbol, then the digit count
// (unary in units of the largest digit) and the digits of the overflow value, least
// significant first, each a uniform `overflow_width`-bit symbol.
// ---------------------------------------------------------------------------
struct UnboundedParams {
  const int32_t* data;        // encode
  int32_t* out_values;        // decode
  const int32_t* index;
  const int32_t* cdf;
  const int32_t* cdf_size;
  const int32_t* offset;
  long long total, rows, width;
  int precision, overflow_width;
  uint8_t* out;               // encode
  unsigned int cap;
  unsigned int* out_len;
  const uint8_t* bytes;       // decode
  long long nbytes;
};

__global__ void __launch_bounds__(64) unbounded_enc_kernel(UnboundedParams p) {
  const int lane = threadIdx.x;
  EncoderState st{0u, 0xFFFFFFFFu, 0u, 0u};
  DigitSink o;
  o.dst = p.out; o.cap = p.cap; o.nbytes = 0; o.n = 0; o.reg = 0; o.overflow = 0;
  const int sh = 16 - p.precision, osh = 16 - p.overflow_width;
  const unsigned int max_overflow = (1u << p.overflow_width) - 1u;
  for (long long k0 = 0; k0 < p.total; k0 += 64) {
    const long long k = k0 + lane;
    int lo = 0, hi = 0, clamped = 0;
    unsigned int ovf = 0;
    if (k < p.total) {
      const long long row = min<long long>(max(p.index[k], 0), p.rows - 1);   // debug_level 0: DCHECK only
      const int max_value = p.cdf_size[row] - 2;
      int value = p.data[k] - p.offset[row];
      if (value < 0) {
        ovf = static_cast<unsigned int>(-2 * value - 1);
        value = max_value;
      } else if (value >= max_value) {
        ovf = static_cast<unsigned int>(2 * (value - max_value));
        value = max_value;
      }
      clamped = value == max_value;
      lo = p.cdf[row * p.width + value] << sh;
      hi = p.cdf[row * p.width + value + 1] << sh;
    }
    const int cnt = static_cast<int>(min<long long>(64, p.total - k0));
    for (int n = 0; n < cnt; ++n) {
      enc_update(st, __builtin_amdgcn_readlane(lo, n), __builtin_amdgcn_readlane(hi, n), o, lane);
      if (__builtin_amdgcn_readlane(clamped, n)) {
        const unsigned int v = __builtin_amdgcn_readlane(static_cast<int>(ovf), n);
        int widths = 0;
        while (widths * p.overflow_width < 32 && (v >> (widths * p.overflow_width)) != 0) ++widths;
        unsigned int val = static_cast<unsigned int>(widths);
        while (val >= max_overflow) {
          enc_update(st, max_overflow << osh, (max_overflow + 1u) << osh, o, lane);
          val -= max_overflow;
        }
        enc_update(st, val << osh, (val + 1u) << osh, o, lane);
        for (int j = 0; j < widths; ++j) {
          const unsigned int d = (v >> (j * p.overflow_width)) & max_overflow;
          enc_update(st, d << osh, (d + 1u) << osh, o, lane);
        }
      }
    }
  }
  sink_flush(o, lane);
  if (lane == 0) *p.out_len = o.overflow ? 0xFFFFFFFFu : finalize_bytes(st, o, p.out);
}

// One uniform symbol of `width` bits: cdf = 0, 1, ..., 2^width at precision `width`; the
// reference's search (first k with target <= span * k) in closed form.
__device__ inline unsigned int dec_uniform(DecoderState& st, int width, DigitWindow& w, int lane) {
  const unsigned long long span = static_cast<unsigned long long>(st.span_m1) + 1;
  const unsigned long long target =
      (static_cast<unsigned long long>(static_cast<unsigned int>(st.window - st.base)) + 1) << width;
  unsigned long long sym = (target - 1) / span;
  const unsigned long long last = (1ull << width) - 1;
  if (sym > last) sym = last;                      // damaged input
  dec_narrow(st, static_cast<unsigned int>(sym), static_cast<unsigned int>(sym) + 1u, width, w, lane);
  return static_cast<unsigned int>(sym);
}

__global__ void __launch_bounds__(64) unbounded_dec_kernel(UnboundedParams p) {
  const int lane = threadIdx.x;
  DecoderState st{0u, 0xFFFFFFFFu, 0u};
  DigitWindow w;
  w.src = p.bytes; w.len = p.nbytes; w.pulls = 0; w.base = 0;
  window_load(w, lane);
  st.window = window_pull(w, lane) << 16;
  st.window |= window_pull(w, lane);
  const unsigned int max_overflow = (1u << p.overflow_width) - 1u;
  for (long long k0 = 0; k0 < p.total; k0 += 64) {
    const long long k = k0 + lane;
    long long row = 0;
    int ncdf = 3, off = 0;
    if (k < p.total) {
      row = min<long long>(max(p.index[k], 0), p.rows - 1);
      ncdf = p.cdf_size[row];
      off = p.offset[row];
    }
    const int cnt = static_cast<int>(min<long long>(64, p.total - k0));
    int outv = 0;
    for (int n = 0; n < cnt; ++n) {
      const long long r = __builtin_amdgcn_readlane(static_cast<int>(row), n);
      const int nc = __builtin_amdgcn_readlane(ncdf, n);
      const int32_t* base = p.cdf + r * p.width;
      auto T = [&](int i) -> int32_t { return base[i]; };
      int value = dec_symbol(T, st, 0, nc, p.precision, w, lane);
      const int max_value = nc - 2;
      if (value == max_value) {
        int widths = 0;
        unsigned int val;
        do {
          val = dec_uniform(st, p.overflow_width, w, lane);
          widths += static_cast<int>(val);
        } while (val == max_overflow && widths < 64);
        unsigned int ovf = 0;
        for (int j = 0; j < widths; ++j) {
          const unsigned int d = dec_uniform(st, p.overflow_width, w, lane);
          if (j * p.overflow_width < 32) ovf |= d << (j * p.overflow_width);
        }
        value = static_cast<int>(ovf >> 1);
        if (ovf & 1u) value = -value - 1; else value += max_value;
      }
      outv = tfc_writelane(value, n, outv);
    }
    if (k < p.total) p.out_values[k] = outv + off;
  }
}

// CheckArgumentValues (unbounded_index_range_coding_kernels.cc:54-113): first offending
// index position / cdf_size row / cdf row (min), bit 0 ends wrong, bit 1 not monotonic.
__global__ void unbounded_check_kernel(const int32_t* index, long long total, const int32_t* cdf,
                                       long long rows, long long width, const int32_t* cdf_size,
                                       int precision, unsigned long long* first) {
  const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (i < total && (index[i] < 0 || rows <= index[i])) atomicMin(first, static_cast<unsigned long long>(i));
  if (i < rows) {
    const int n = cdf_size[i];
    if (n < 3 || width < n) {
      atomicMin(first + 1, static_cast<unsigned long long>(i));
    } else {
      const int32_t* s = cdf + i * width;
      if (s[0] != 0 || s[n - 1] != (1 << precision)) atomicMin(first + 2, static_cast<unsigned long long>(i));
      for (int j = 0; j + 1 < n; ++j)
        if (s[j + 1] <= s[j]) { atomicMin(first + 3, static_cast<unsigned long long>(i)); break; }
    }
  }
}

// cdf[..., 0] == 0, cdf[..., -1] == 1 << precision, strictly increasing
// (CheckCdfValues, range_coding_kernels.cc:149-173).  flag: bit0 ends wrong,
// bit1 not monotonic; bad_row = first offending row (min).
__global__ void check_cdf_kernel(const int32_t* cdf, long long rows, long long width,
                                 int precision, unsigned int* flag, unsigned long long* bad_row) {
  const long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (r >= rows) return;
  const int32_t* s = cdf + r * width;
  if (s[0] != 0 || s[width - 1] != (1 << precision)) {
    atomicOr(flag, 1u);
    atomicMin(bad_row, static_cast<unsigned long long>(r));
  }
  for (long long j = 0; j + 1 < width; ++j)
    if (s[j + 1] <= s[j]) { atomicOr(flag, 2u); break; }
}

}  // namespace tfc

namespace {

std::string shape_str(const int64_t* s, int n) {
  std::string r = "[";
  for (int i = 0; i < n; ++i) r += (i ? "," : "") + std::to_string(s[i]);
  return r + "]";
}

// MergeAxes (range_coding_kernels_util.cc:34-91) + the stride table the
// kernels use instead of BroadcastRange's incremental displacement.
int make_broadcast(const int64_t* data_shape, int nd, const int64_t* cdf_shape, int nc,
                   Broadcast* out) {
  if (nc != nd + 1)
    return fail("`cdf` should have one more axis than `data`: data shape=%s, cdf shape=%s",
                shape_str(data_shape, nd).c_str(), shape_str(cdf_shape, nc).c_str());
  if (cdf_shape[nc - 1] <= 1)
    return fail("The last dimension of `cdf` should be > 1: %s", shape_str(cdf_shape, nc).c_str());
  std::vector<int64_t> md(1, 1), mc(1, 1);
  for (int j = 0; j < nd; ++j) {
    if (data_shape[j] != cdf_shape[j] && cdf_shape[j] != 1)
      return fail("Cannot broadcast shape %s to %s", shape_str(cdf_shape, nc).c_str(),
                  shape_str(data_shape, nd).c_str());
    const bool was_b = mc.back() == 1;
    const bool is_b = cdf_shape[j] == 1;
    if (was_b == is_b || data_shape[j] <= 1 || md.back() <= 1) {
      md.back() *= data_shape[j];
      mc.back() *= cdf_shape[j];
    } else {
      md.push_back(data_shape[j]);
      mc.push_back(cdf_shape[j]);
    }
  }
  if (md.size() > 6)
    return fail("Irregular broadcast pattern: %s, %s", shape_str(data_shape, nd).c_str(),
                shape_str(cdf_shape, nc).c_str());
  out->nd = static_cast<int>(md.size());
  out->width = cdf_shape[nc - 1];
  long long stride = out->width;
  for (int i = out->nd - 1; i >= 0; --i) {
    out->shape[i] = md[i];
    out->cdf_stride[i] = mc[i] <= 1 ? 0 : stride;
    stride *= mc[i];
  }
  return 0;
}

int check_cdf_values(const int32_t* cdf, const int64_t* cdf_shape, int nc, int precision,
                     hipStream_t st) {
  const long long width = cdf_shape[nc - 1];
  if (width <= 2) return fail("CDF size should be > 2: %lld", width);
  long long rows = 1;
  for (int i = 0; i + 1 < nc; ++i) rows *= cdf_shape[i];
  if (rows == 0) return 0;
  DevBuf flag;
  TFC_HIP(flag.alloc(16, st));
  const unsigned long long init[2] = {0ull, ~0ull};
  TFC_HIP(hipMemcpyAsync(flag.p, init, 16, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(check_cdf_kernel, dim3(static_cast<unsigned>(ceil_div(rows, 256))), dim3(256),
                     0, st, cdf, rows, width, precision, flag.as<unsigned int>(),
                     flag.as<unsigned long long>() + 1);
  unsigned long long h[2];
  TFC_HIP(hipMemcpyAsync(h, flag.p, 16, hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  const unsigned int f = static_cast<unsigned int>(h[0] & 0xFFFFFFFFu);
  if (f & 1u) {
    int32_t ends[2] = {0, 0};
    (void)hipMemcpy(&ends[0], cdf + h[1] * width, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ends[1], cdf + h[1] * width + width - 1, 4, hipMemcpyDeviceToHost);
    return fail("CDF should start from 0 and end at %d: cdf[0]=%d, cdf[^1]=%d", 1 << precision,
                ends[0], ends[1]);
  }
  if (f & 2u) return fail("CDF is not monotonic");
  return 0;
}

}  // namespace

extern "C" int tfc_range_encode(const int16_t* data, const int64_t* data_shape, int nd,
                                const int32_t* cdf, const int64_t* cdf_shape, int nc,
                                int precision, int debug_level, void* stream, uint8_t** out,
                                int64_t* out_len) {
  *out = nullptr;
  *out_len = 0;
  if (!(0 < precision && precision <= 16)) return fail("`precision` must be in [1, 16]: %d", precision);
  if (debug_level != 0 && debug_level != 1) return fail("`debug_level` must be 0 or 1: %d", debug_level);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Broadcast geo;
  if (nc != nd + 1 || cdf_shape[nc - 1] <= 1) return make_broadcast(data_shape, nd, cdf_shape, nc, &geo);
  if (debug_level > 0 && check_cdf_values(cdf, cdf_shape, nc, precision, st)) return 1;
  if (make_broadcast(data_shape, nd, cdf_shape, nc, &geo)) return 1;
  long long total = 1;
  for (int i = 0; i < nd; ++i) total *= data_shape[i];
  if (2 * total + 16 >= (1ll << 32)) return fail("`data` too large for a single code stream");
  DevBuf buf, meta;
  const unsigned int cap = static_cast<unsigned int>(2 * total + 16);
  TFC_HIP(buf.alloc(cap, st));
  TFC_HIP(meta.alloc(16, st));
  const unsigned long long init[2] = {~0ull, 0ull};
  TFC_HIP(hipMemcpyAsync(meta.p, init, 16, hipMemcpyHostToDevice, st));
  LegacyEncParams p;
  p.data = data;
  p.cdf = cdf;
  p.geo = geo;
  p.total = total;
  p.precision = precision;
  p.check = debug_level;
  p.out = buf.as<uint8_t>();
  p.cap = cap - 4;
  p.first_error = meta.as<unsigned long long>();
  p.out_len = reinterpret_cast<unsigned int*>(meta.as<unsigned long long>() + 1);
  hipLaunchKernelGGL(legacy_enc_kernel, dim3(1), dim3(64), 0, st, p);
  TFC_HIP(hipGetLastError());
  unsigned long long h[2];
  TFC_HIP(hipMemcpyAsync(h, meta.p, 16, hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (h[0] != ~0ull) {
    int16_t v = 0;
    (void)hipMemcpy(&v, data + h[0], 2, hipMemcpyDeviceToHost);
    return fail("'data' value not in [0, %lld): value=%d", static_cast<long long>(geo.width - 1), v);
  }
  const unsigned int n = static_cast<unsigned int>(h[1] & 0xFFFFFFFFu);
  uint8_t* host = static_cast<uint8_t*>(std::malloc(n ? n : 1));
  if (n) TFC_HIP(hipMemcpy(host, buf.p, n, hipMemcpyDeviceToHost));
  *out = host;
  *out_len = n;
  return 0;
}

extern "C" int tfc_range_decode(const uint8_t* encoded, int64_t encoded_len,
                                const int64_t* out_shape, int nd, const int32_t* cdf,
                                const int64_t* cdf_shape, int nc, int precision, int debug_level,
                                void* stream, int16_t* out) {
  if (!(0 < precision && precision <= 16)) return fail("`precision` must be in [1, 16]: %d", precision);
  if (debug_level != 0 && debug_level != 1) return fail("`debug_level` must be 0 or 1: %d", debug_level);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Broadcast geo;
  if (nc != nd + 1 || cdf_shape[nc - 1] <= 1) return make_broadcast(out_shape, nd, cdf_shape, nc, &geo);
  if (debug_level > 0 && check_cdf_values(cdf, cdf_shape, nc, precision, st)) return 1;
  if (make_broadcast(out_shape, nd, cdf_shape, nc, &geo)) return 1;
  long long total = 1;
  for (int i = 0; i < nd; ++i) total *= out_shape[i];
  if (total == 0) return 0;
  DevBuf bytes;
  TFC_HIP(bytes.alloc(static_cast<size_t>(encoded_len), st));
  if (encoded_len)
    TFC_HIP(hipMemcpyAsync(bytes.p, encoded, static_cast<size_t>(encoded_len), hipMemcpyHostToDevice, st));
  LegacyDecParams p;
  p.bytes = bytes.as<uint8_t>();
  p.nbytes = encoded_len;
  p.cdf = cdf;
  p.geo = geo;
  p.total = total;
  p.precision = precision;
  p.out = out;
  hipLaunchKernelGGL(legacy_dec_kernel, dim3(1), dim3(64), 0, st, p);
  TFC_HIP(hipGetLastError());
  TFC_HIP(hipStreamSynchronize(st));  // `encoded` is a host buffer the caller may free
  return 0;
}

// ===========================================================================
// Deprecated UnboundedIndexRangeEncode / Decode
// ===========================================================================

namespace {

int unbounded_validate(const char* who, const int32_t* index, int64_t total, const int32_t* cdf, int64_t rows,
                       int64_t width, const int32_t* cdf_size, int precision, int overflow_width,
                       int debug_level, hipStream_t st) {
  if (!(0 < precision && precision <= 16)) return fail("`precision` must be in [1, 16]: %d", precision);
  if (!(0 < overflow_width && overflow_width <= 16))
    return fail("`overflow_width` must be in [1, 16]: %d", overflow_width);
  if (debug_level != 0 && debug_level != 1) return fail("`debug_level` must be 0 or 1: %d", debug_level);
  if (width < 3) return fail("'cdf' should be 2-D and cdf.dim_size(1) >= 3: [%lld,%lld]",
                             static_cast<long long>(rows), static_cast<long long>(width));
  if (rows < 1) return fail("%s: 'cdf' has no rows", who);
  if (debug_level == 0) return 0;
  DevBuf first;
  TFC_HIP(first.alloc(4 * sizeof(unsigned long long), st));
  TFC_HIP(hipMemsetAsync(first.p, 0xFF, 4 * sizeof(unsigned long long), st));
  const long long n = std::max<long long>(total, rows);
  hipLaunchKernelGGL(unbounded_check_kernel, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, st,
                     index, total, cdf, rows, width, cdf_size, precision, first.as<unsigned long long>());
  unsigned long long h[4];
  TFC_HIP(hipMemcpyAsync(h, first.p, sizeof(h), hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (h[0] != ~0ull) {
    int32_t v = 0;
    (void)hipMemcpy(&v, index + h[0], 4, hipMemcpyDeviceToHost);
    return fail("'index' has a value not in [0, %lld): value=%d", static_cast<long long>(rows), v);
  }
  if (h[1] != ~0ull) {
    int32_t v = 0;
    (void)hipMemcpy(&v, cdf_size + h[1], 4, hipMemcpyDeviceToHost);
    return fail("'cdf_size' has a value not in [3, %lld]: value=%d", static_cast<long long>(width), v);
  }
  if (h[2] != ~0ull) {
    int32_t n0 = 0, ends[2] = {0, 0};
    (void)hipMemcpy(&n0, cdf_size + h[2], 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ends[0], cdf + h[2] * width, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ends[1], cdf + h[2] * width + n0 - 1, 4, hipMemcpyDeviceToHost);
    return fail("Each cdf should start from 0 and end at %d: cdf[0]=%d, cdf[^1]=%d", 1 << precision, ends[0],
                ends[1]);
  }
  if (h[3] != ~0ull) return fail("CDF is not monotonic");
  return 0;
}

}  // namespace

extern "C" int tfc_unbounded_index_range_encode(const int32_t* data, const int32_t* index, int64_t total,
                                                const int32_t* cdf, int64_t rows, int64_t width,
                                                const int32_t* cdf_size, const int32_t* offset, int precision,
                                                int overflow_width, int debug_level, void* stream,
                                                uint8_t** out, int64_t* out_len) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  *out = nullptr;
  *out_len = 0;
  if (int rc = unbounded_validate("tfc_unbounded_index_range_encode", index, total, cdf, rows, width, cdf_size,
                                  precision, overflow_width, debug_level, st))
    return rc;
  // worst case per element: the symbol, ceil(32 / w) digits and their count in unary — 2 bytes per call
  const long long digits = (32 + overflow_width - 1) / overflow_width;
  const unsigned long long cap64 = 2ull * total * (2 + 2 * digits) + 16;
  if (cap64 >= (1ull << 32)) return fail("tfc_unbounded_index_range_encode: tensor too large for one stream");
  DevBuf buf, len;
  TFC_HIP(buf.alloc(cap64, st));
  TFC_HIP(len.alloc(sizeof(unsigned int), st));
  UnboundedParams p{};
  p.data = data; p.index = index; p.cdf = cdf; p.cdf_size = cdf_size; p.offset = offset;
  p.total = total; p.rows = rows; p.width = width; p.precision = precision; p.overflow_width = overflow_width;
  p.out = buf.as<uint8_t>(); p.cap = static_cast<unsigned int>(cap64 - 8); p.out_len = len.as<unsigned int>();
  hipLaunchKernelGGL(unbounded_enc_kernel, dim3(1), dim3(64), 0, st, p);
  unsigned int n = 0;
  TFC_HIP(hipMemcpyAsync(&n, len.p, sizeof(n), hipMemcpyDeviceToHost, st));
  TFC_HIP(hipStreamSynchronize(st));
  if (n == 0xFFFFFFFFu) return fail("internal error: unbounded encoder ran out of output space");
  uint8_t* host = static_cast<uint8_t*>(std::malloc(std::max<size_t>(n, 1)));
  if (!host) return fail("out of host memory");
  if (n) TFC_HIP(hipMemcpy(host, buf.p, n, hipMemcpyDeviceToHost));
  *out = host;
  *out_len = n;
  return 0;
}

extern "C" int tfc_unbounded_index_range_decode(const uint8_t* encoded, int64_t encoded_len, const int32_t* index,
                                                int64_t total, const int32_t* cdf, int64_t rows, int64_t width,
                                                const int32_t* cdf_size, const int32_t* offset, int precision,
                                                int overflow_width, int debug_level, int32_t* out, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = unbounded_validate("tfc_unbounded_index_range_decode", index, total, cdf, rows, width, cdf_size,
                                  precision, overflow_width, debug_level, st))
    return rc;
  if (total == 0) return 0;
  DevBuf bytes;
  TFC_HIP(bytes.alloc(static_cast<size_t>(encoded_len), st));
  if (encoded_len) TFC_HIP(hipMemcpyAsync(bytes.p, encoded, static_cast<size_t>(encoded_len), hipMemcpyHostToDevice, st));
  UnboundedParams p{};
  p.out_values = out; p.index = index; p.cdf = cdf; p.cdf_size = cdf_size; p.offset = offset;
  p.total = total; p.rows = rows; p.width = width; p.precision = precision; p.overflow_width = overflow_width;
  p.bytes = bytes.as<uint8_t>(); p.nbytes = encoded_len;
  hipLaunchKernelGGL(unbounded_dec_kernel, dim3(1), dim3(64), 0, st, p);
  TFC_HIP(hipGetLastError());
  TFC_HIP(hipStreamSynchronize(st));      // the host copy of `encoded` may go away
  return 0;
}
