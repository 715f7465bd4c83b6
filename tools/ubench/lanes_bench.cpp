// Stand-alone timing harness for the range coder through the C ABI (no Python, no torch: starts in a
// second on the GPU box).  Builds the C2 tables (192 discretised Gaussians, precision 12, escape rows),
// draws in-range symbols, and times encode / decode kernels alone and with D independent steps in flight
// on D HIP streams from one host thread.
//   hipcc -O2 -std=c++17 tools/ubench/lanes_bench.cpp -Iinclude -Lcompression_amd -ltfc_hip \
//         -Wl,-rpath,$PWD/compression_amd -o /tmp/lanes_bench
//   /tmp/lanes_bench [mode=2] [streams=512] [elems=49152] [depths=1,8,16,32]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "tfc_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define TF(x) do { if (x) { printf("tfc error at %d: %s\n", __LINE__, tfc_last_error()); exit(1); } } while (0)

static double ndtr(double x) { return 0.5 * std::erfc(-x / std::sqrt(2.0)); }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int mode = argc > 1 ? atoi(argv[1]) : 2;
  const int64_t streams = argc > 2 ? atoll(argv[2]) : 512;
  const int64_t elems = argc > 3 ? atoll(argv[3]) : 49152;
  std::vector<int> depths;
  {
    std::string d = argc > 4 ? argv[4] : "1,8,16,32";
    for (size_t p = 0; p < d.size();) {
      size_t q = d.find(',', p);
      if (q == std::string::npos) q = d.size();
      depths.push_back(atoi(d.substr(p, q - p).c_str()));
      p = q + 1;
    }
  }
  const int ntab = 192, prec = 12;
  // tables through the product's own PmfToQuantizedCdf
  std::vector<int32_t> lookup;
  std::vector<std::vector<int32_t>> cdfs;
  for (int c = 0; c < ntab; ++c) {
    const double sigma = 0.25 * std::pow(2.0, c / 24.0);
    const double half = 2.8856349124267573 * sigma;       // -ndtri(2^-9)
    const int lo = (int)std::floor(-half), hi = (int)std::ceil(half);
    std::vector<float> pmf;
    double sum = 0;
    for (int x = lo; x <= hi; ++x) {
      const double p = ndtr((x + 0.5) / sigma) - ndtr((x - 0.5) / sigma);
      pmf.push_back((float)p);
      sum += p;
    }
    pmf.push_back((float)std::max(1.0 - sum, 0.0));
    float* dp;
    int32_t* dc;
    CK(hipMalloc(&dp, pmf.size() * 4));
    CK(hipMalloc(&dc, (pmf.size() + 1) * 4));
    CK(hipMemcpy(dp, pmf.data(), pmf.size() * 4, hipMemcpyHostToDevice));
    TF(tfc_pmf_to_quantized_cdf(dp, 1, (int64_t)pmf.size(), prec, dc, nullptr));
    std::vector<int32_t> cdf(pmf.size() + 1);
    CK(hipMemcpy(cdf.data(), dc, cdf.size() * 4, hipMemcpyDeviceToHost));
    CK(hipFree(dp));
    CK(hipFree(dc));
    lookup.push_back(-prec);
    lookup.insert(lookup.end(), cdf.begin(), cdf.end());
    cdfs.push_back(cdf);
  }
  tfc_tables* tables;
  TF(tfc_tables_create(lookup.data(), 1, 1, (int64_t)lookup.size(), nullptr, &tables));

  const int maxd = *std::max_element(depths.begin(), depths.end());
  const int nslots = std::min(maxd, 4);
  std::vector<int32_t*> d_val(nslots), d_out;
  std::vector<int32_t> h(streams * elems);
  for (int k = 0; k < nslots; ++k) {
    std::mt19937 rng(1234 + k);
    for (int64_t s = 0; s < streams; ++s)
      for (int64_t j = 0; j < elems; ++j) {
        const std::vector<int32_t>& cdf = cdfs[j % ntab];
        const int u = (int)(rng() & 4095u);
        int sym = (int)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) - 1;
        sym = std::min(sym, (int)cdf.size() - 3);
        h[s * elems + j] = std::max(sym, 0);
      }
    CK(hipMalloc(&d_val[k], h.size() * 4));
    CK(hipMemcpy(d_val[k], h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const int ngroups_max = getenv("NGROUPS") ? atoi(getenv("NGROUPS")) : 2;
  d_out.resize(maxd * ngroups_max);
  for (auto& o : d_out) CK(hipMalloc(&o, h.size() * 4));
  uint8_t* d_ok;
  CK(hipMalloc(&d_ok, (size_t)maxd * streams));
  std::vector<hipStream_t> st(std::max(maxd, 8));
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  auto step = [&](int slot, int k, tfc_encoder** pe, tfc_decoder** pd, hipEvent_t* ev) {
    tfc_encoder* e;
    tfc_decoder* d;
    TF(tfc_encoder_create(tables, streams, st[k], &e));
    TF(tfc_encoder_set_mode(e, mode));
    TF(tfc_encoder_set_deferred_errors(e, 1));
    if (ev) CK(hipEventRecord(ev[0], st[k]));
    TF(tfc_encoder_encode(e, d_val[slot], nullptr, elems, st[k]));
    if (ev) CK(hipEventRecord(ev[1], st[k]));
    TF(tfc_encoder_finalize_device(e, st[k]));
    const uint8_t* blob;
    const int64_t* offs;
    TF(tfc_encoder_result(e, &blob, &offs));
    TF(tfc_decoder_create(tables, blob, offs, streams, 1, st[k], &d));
    TF(tfc_decoder_set_mode(d, mode));
    if (ev) CK(hipEventRecord(ev[2], st[k]));
    TF(tfc_decoder_decode(d, nullptr, d_out[k], elems, st[k]));
    if (ev) CK(hipEventRecord(ev[3], st[k]));
    *pe = e;
    *pd = d;
  };

  // alone, with events
  hipEvent_t ev[4];
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int rep = 0; rep < 3; ++rep) {
    tfc_encoder* e;
    tfc_decoder* d;
    step(0, 0, &e, &d, ev);
    CK(hipDeviceSynchronize());
    float enc, mid, dec;
    CK(hipEventElapsedTime(&enc, ev[0], ev[1]));
    CK(hipEventElapsedTime(&mid, ev[1], ev[2]));
    CK(hipEventElapsedTime(&dec, ev[2], ev[3]));
    int64_t total = 0;
    TF(tfc_encoder_status(e, st[0], &total));
    std::vector<int32_t> back(h.size());
    CK(hipMemcpy(back.data(), d_out[0], h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h.data(), d_val[0], h.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(back.data(), h.data(), h.size() * 4) == 0;
    if (rep == 2)
      printf("mode %d alone: encode %.3f ms  finalize+open %.3f ms  decode %.3f ms  bytes %lld (%.3f bits/sym) roundtrip %s\n",
             mode, enc, mid, dec, (long long)total, 8.0 * total / (streams * elems), same ? "exact" : "WRONG");
    tfc_decoder_destroy(d);
    tfc_encoder_destroy(e);
  }
  // `depth` independent 512-stream steps per launch (tfc_encoder_encode_many / tfc_decoder_decode_many),
  // `groups` such groups in flight on different streams
  const int groups = getenv("NGROUPS") ? atoi(getenv("NGROUPS")) : 2;
  for (int depth : depths) {
    if (depth < 2) continue;
    double best = 1e30, best_host = 0;
    for (int rep = 0; rep < 3; ++rep) {
      const int rounds = 2;
      std::vector<tfc_encoder*> es;
      std::vector<tfc_decoder*> ds;
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < rounds * groups; ++r) {
        hipStream_t s = st[r % groups];
        std::vector<tfc_encoder*> ge(depth);
        std::vector<tfc_decoder*> gd(depth);
        std::vector<const int32_t*> vals(depth);
        std::vector<int32_t*> outs(depth);
        TF(tfc_encoder_create_many(tables, streams, depth, s, ge.data()));
        for (int k = 0; k < depth; ++k) {
          TF(tfc_encoder_set_mode(ge[k], mode));
          TF(tfc_encoder_set_deferred_errors(ge[k], 1));
          vals[k] = d_val[k % nslots];
          outs[k] = d_out[(r % groups) * depth + k];
        }
        TF(tfc_encoder_encode_many(depth, ge.data(), vals.data(), nullptr, elems, s));
        TF(tfc_encoder_finalize_device_many(depth, ge.data(), s));
        TF(tfc_decoder_create_many(tables, depth, ge.data(), s, gd.data()));
        for (int k = 0; k < depth; ++k) TF(tfc_decoder_set_mode(gd[k], mode));
        TF(tfc_decoder_decode_many(depth, gd.data(), nullptr, outs.data(), elems, s));
        TF(tfc_decoder_finalize_device_many(depth, gd.data(), d_ok, s));
        es.insert(es.end(), ge.begin(), ge.end());
        ds.insert(ds.end(), gd.begin(), gd.end());
      }
      const double th = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      CK(hipDeviceSynchronize());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const int steps = rounds * groups * depth;
      if (dt / steps < best) {
        best = dt / steps;
        best_host = th / steps;
      }
      // check the last group's round trip
      std::vector<int32_t> back(streams * elems), want(streams * elems);
      CK(hipMemcpy(back.data(), d_out[((rounds * groups - 1) % groups) * depth + depth - 1], back.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(want.data(), d_val[(depth - 1) % nslots], want.size() * 4, hipMemcpyDeviceToHost));
      if (memcmp(back.data(), want.data(), back.size() * 4) != 0) printf("ROUND TRIP MISMATCH at depth %d\n", depth);
      for (size_t k = 0; k < es.size(); ++k) { tfc_decoder_destroy(ds[k]); tfc_encoder_destroy(es[k]); }
    }
    printf("mode %d, %3d steps per launch x %d groups in flight: %.3f ms/step (host enqueue %.3f)  %.2f Gsym/s round trip\n",
           mode, depth, groups, 1e3 * best, 1e3 * best_host, streams * elems / best / 1e9);
  }
  return 0;
}
