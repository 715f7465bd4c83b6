// GDN / IGDN forward entry point (kernels: gdn_common.h).
#include "gdn_common.h"
#include "gdn_params.h"

#include <cmath>
#include <memory>

namespace {
int gdn_forward_any(const void* x, void* y, int dtype, int64_t pixels, int64_t channels, const float* beta,
                    const float* gamma, int inverse, int rectify, int alpha_mode, int eps_mode, bool general,
                    float alpha, float epsilon, void* stream, const tfc_gdn_params* prepared = nullptr) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("tfc_gdn_forward: dtype must be 0 (float32) or 1 (bfloat16)");
  if (!general) {
    if (alpha_mode != 1 && alpha_mode != 2) return fail("tfc_gdn_forward: alpha must be 1 or 2");
    if (eps_mode != 0 && eps_mode != 1) return fail("tfc_gdn_forward: epsilon must be 1 or 0.5");
  } else {
    if (!(alpha > 0.f) || !(epsilon > 0.f) || !std::isfinite(alpha) || !std::isfinite(epsilon))
      return fail("tfc_gdn_forward_general: alpha and epsilon must be positive and finite (got %g, %g)",
                  static_cast<double>(alpha), static_cast<double>(epsilon));
  }
  if (channels <= 0 || channels % 32 != 0 || channels > 256)
    return fail("tfc_gdn_forward: channels must be a multiple of 32, at most 256 (got %lld)",
                static_cast<long long>(channels));
  if (pixels == 0) return 0;
  GdnParams p{};
  p.x = x; p.y = y; p.beta = beta; p.gamma = gamma;
  p.prepared = prepared ? prepared->image.p : nullptr;
  p.pixels = pixels; p.C = static_cast<int>(channels);
  p.inverse = inverse; p.rectify = rectify; p.alpha2 = alpha_mode == 2; p.eps_half = eps_mode == 1;
  if (general) {
    p.gen = 1;
    p.alpha = alpha;
    p.eps_s = inverse ? epsilon : -epsilon;
    // tf.pow of a negative base: real for an integer exponent, NaN otherwise
    auto negative_base = [](float e) {
      const bool integral = std::floor(e) == e && e < 16777216.f;
      return !integral ? std::nanf("") : (std::fmod(e, 2.f) == 0.f ? 1.f : -1.f);
    };
    p.negf = negative_base(alpha);
    p.negf_e = negative_base(epsilon);
  }
  p.tiles = ceil_div(pixels, 32);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (channels / 32) {
    case 1: return launch_gdn<1, MODE_FWD>(p, dtype, st);
    case 2: return launch_gdn<2, MODE_FWD>(p, dtype, st);
    case 3: return launch_gdn<3, MODE_FWD>(p, dtype, st);
    case 4: return launch_gdn<4, MODE_FWD>(p, dtype, st);
    case 5: return launch_gdn<5, MODE_FWD>(p, dtype, st);
    case 6: return launch_gdn<6, MODE_FWD>(p, dtype, st);
    case 7: return launch_gdn<7, MODE_FWD>(p, dtype, st);
    default: return launch_gdn<8, MODE_FWD>(p, dtype, st);
  }
}
}  // namespace

extern "C" int tfc_gdn_forward(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                               const float* beta, const float* gamma, int inverse, int rectify,
                               int alpha_mode, int eps_mode, void* stream) {
  return gdn_forward_any(x, y, dtype, pixels, channels, beta, gamma, inverse, rectify, alpha_mode, eps_mode, false,
                         0.f, 0.f, stream);
}

extern "C" int tfc_gdn_forward_general(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                                       const float* beta, const float* gamma, int inverse, int rectify,
                                       float alpha, float epsilon, void* stream) {
  return gdn_forward_any(x, y, dtype, pixels, channels, beta, gamma, inverse, rectify, 1, 0, true, alpha, epsilon,
                         stream);
}

extern "C" int tfc_gdn_params_create(const float* beta, const float* gamma, int64_t channels, int dtype,
                                     void* stream, tfc_gdn_params** out) {
  using namespace tfc;
  *out = nullptr;
  if (dtype != 0 && dtype != 1) return fail("tfc_gdn_params_create: dtype must be 0 (float32) or 1 (bfloat16)");
  if (channels <= 0 || channels % 32 != 0 || channels > 256 || (dtype == 0 && channels > 192))
    return fail("tfc_gdn_params_create: channels must be a multiple of 32, at most 256 (192 for float32), got %lld",
                static_cast<long long>(channels));
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::unique_ptr<tfc_gdn_params> p(new tfc_gdn_params);
  p->channels = channels;
  p->dtype = dtype;
  const int KT = static_cast<int>(channels / 32);
  if (dtype == 1) {
    const size_t lds = sizeof(bf16x8) * KT * (KT * 2) * 64 + sizeof(float) * KT * 32;
    TFC_HIP(p->image.alloc(lds, st));
    const int n = KT * KT * 2 * 64;
    hipLaunchKernelGGL(gdn_prep_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gamma, beta,
                       static_cast<int>(channels), 0, p->image.as<bf16x8>());
  } else {
    const size_t lds = sizeof(f32x4) * KT * KT * 4 * 64 + sizeof(float) * KT * 32;
    TFC_HIP(p->image.alloc(lds, st));
    const int n = KT * KT * 4 * 64;
    hipLaunchKernelGGL(gdn_prep_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gamma, beta,
                       static_cast<int>(channels), 0, p->image.as<f32x4>());
  }
  TFC_HIP(hipGetLastError());
  *out = p.release();
  return 0;
}

extern "C" void tfc_gdn_params_destroy(tfc_gdn_params* p) { delete p; }

extern "C" int tfc_gdn_forward_prepared(const tfc_gdn_params* params, const void* x, void* y, int64_t pixels,
                                        int inverse, int rectify, int alpha_mode, int eps_mode, void* stream) {
  if (!params) return tfc::fail("tfc_gdn_forward_prepared: params is null");
  const_cast<tfc_gdn_params*>(params)->image.touch(static_cast<hipStream_t>(stream));
  return gdn_forward_any(x, y, params->dtype, pixels, params->channels, nullptr, nullptr, inverse, rectify,
                         alpha_mode, eps_mode, false, 0.f, 0.f, stream, params);
}
