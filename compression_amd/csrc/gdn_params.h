// Prepared GDN parameters (tfc_gdn_params_create): shared by gdn.hip, which builds and runs them, and
// signal_conv.hip, whose third-generation kernel applies them as the convolution's activation.
#pragma once
#include "common.h"

// Fragment-ordered image of (gamma, beta): for bfloat16, KT x KS x 64 A fragments of 8 bf16 (fragment (t, s) of lane l
// = gamma[16 s + 4 h + (e & 3) + 8 (e >> 2)][32 t + i], h = l >> 5, i = l & 31: gdn_common.h, gdn_prep_bf16_kernel)
// followed by beta[C] as float.
struct tfc_gdn_params {
  tfc::DevBuf image;
  int64_t channels = 0;
  int dtype = 0;
};
