// GDN / IGDN forward entry point (kernels: gdn_common.h).
#include "gdn_common.h"

extern "C" int tfc_gdn_forward(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                               const float* beta, const float* gamma, int inverse, int rectify,
                               int alpha_mode, int eps_mode, void* stream) {
  using namespace tfc;
  if (dtype != 0 && dtype != 1) return fail("tfc_gdn_forward: dtype must be 0 (float32) or 1 (bfloat16)");
  if (alpha_mode != 1 && alpha_mode != 2) return fail("tfc_gdn_forward: alpha must be 1 or 2");
  if (eps_mode != 0 && eps_mode != 1) return fail("tfc_gdn_forward: epsilon must be 1 or 0.5");
  if (channels <= 0 || channels % 32 != 0 || channels > 256)
    return fail("tfc_gdn_forward: channels must be a multiple of 32, at most 256 (got %lld)",
                static_cast<long long>(channels));
  if (pixels == 0) return 0;
  GdnParams p{};
  p.x = x; p.y = y; p.beta = beta; p.gamma = gamma;
  p.pixels = pixels; p.C = static_cast<int>(channels);
  p.inverse = inverse; p.rectify = rectify; p.alpha2 = alpha_mode == 2; p.eps_half = eps_mode == 1;
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

