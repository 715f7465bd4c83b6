// Laplace-mixture tail of the training-time likelihood (python/entropy_models/continuous_base.py:298-334):
//   probs = (1 - m) P(v) + m Q(v),        Q = NoisyLaplace(0, 1): the Laplace mass of [v - .5, v + .5]
//   log_prob = probs < 1e-10 ? log m + log Q(v) : log max(probs, 1e-10)
// shared by the fused bits kernels (factorized_bits.hip, noisy_normal_bits.hip).  m is a host float in (0, 1).
//
// Q in closed form instead of the reference's difference of two cumulatives (uniform_noise.py:117-156 over a
// Laplace base), which cancels in float32 once |v| > ~15 and returns log 0 beyond ~17:
//   |v| >= .5 :  Q = sinh(.5) exp(-|v|)           |v| < .5 :  Q = 1 - exp(-.5) cosh(v)
#pragma once
#include <hip/hip_runtime.h>

namespace tfc {

struct LaplaceUnit {
  float q, dq;      // mass of the unit interval around v and its derivative
  float lq, dlq;    // log of it and its derivative
};

__device__ inline LaplaceUnit laplace_unit(float v) {
  constexpr float kLogSinhHalf = -0.65182232595f;      // log(sinh(.5))
  constexpr float kHalfExpMinusHalf = 0.30326532986f;  // exp(-.5) / 2
  const float a = fabsf(v);
  LaplaceUnit r;
  if (a >= 0.5f) {
    r.lq = kLogSinhHalf - a;
    r.q = __expf(r.lq);
    r.dlq = v > 0.f ? -1.f : 1.f;
    r.dq = r.dlq * r.q;
  } else {
    const float ep = __expf(v), em = __expf(-v);
    r.q = 1.f - kHalfExpMinusHalf * (ep + em);         // in [.316, .394]
    r.dq = -kHalfExpMinusHalf * (ep - em);
    r.lq = __logf(r.q);
    r.dlq = r.dq * __builtin_amdgcn_rcpf(r.q);
  }
  return r;
}

// log of the mixture, given log P(v)
__device__ inline float tail_mix(float log_p, float v, float m) {
  const LaplaceUnit l = laplace_unit(v);
  const float probs = fmaf(1.f - m, __expf(log_p), m * l.q);
  return probs < 1e-10f ? __logf(m) + l.lq : __logf(probs);
}

// d tail_mix = prior * dP + direct * dv   (dP: the differential of the prior's PROBABILITY, not its log)
struct TailGrad { float prior, direct; };
__device__ inline TailGrad tail_mix_grad(float p, float v, float m) {
  const LaplaceUnit l = laplace_unit(v);
  const float probs = fmaf(1.f - m, p, m * l.q);
  TailGrad g;
  if (probs < 1e-10f) {              // the tf.where branch that carries the gradient there
    g.prior = 0.f;
    g.direct = l.dlq;
  } else {
    const float inv = 1.f / probs;
    g.prior = (1.f - m) * inv;
    g.direct = m * l.dq * inv;
  }
  return g;
}

inline bool tail_mass_ok(float m) { return m > 0.f && m < 1.f; }

}  // namespace tfc
