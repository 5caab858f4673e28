// Arithmetic of the rational-quadratic spline shared by every kernel that evaluates one (the fused 2-D chains
// nf_rqs2d_h64.hip / nf_rqs2d_x3.hip through nf_rqs2d_common.h, the general path's transform kernels and the
// spline epilogue of the last GEMM through nf_rqs_eval.h).  distrax.RationalQuadraticSpline semantics as restated
// in SURVEY.md section 8a row c2 (src/maf_toy_densities.py:84-87, src/maf_mnist.py:63-66).  Not part of the ABI.
//
// Where a float32 spline loses its digits (measured with the NumPy restatement, one stage at a time in float64,
// tests/tools/rqs_precision_ablation.py; 8 layers, K = 12 on [-8, 8], scaled log_prob error): NOT in the conditioner
// (float32 there: 1e-7 rms) but in the KNOTS -- the softmax normalisation and the running sum put ~1e-7 * 16 of
// absolute error into every knot: 8e-7 rms, 1e-5 worst, i.e. all of the chain's float32 error.  Knots computed
// accurately and rounded ONCE to float32 cost 9e-8.  So the bin sizes' sum, the normalisation and the running sum
// are carried in float64 (gfx950 issues v_fma_f64 / v_add_f64 / v_mul_f64 / v_rndne_f64 / v_ldexp_f64 at the float32
// rate, conversions at 6 cycles, v_rcp_f64 at 17: profiles/microbench/valu_f64_issue_cost.hip; ~60 instructions per
// axis), the reciprocals get one Newton step and the log-det takes three logarithms instead of one logarithm of a
// five-factor product.  The bins' exponentials stay on the hardware exp2 (NF_RQS_XEXP = 0): measured on the GPU
// (tests/tools/rqs_error_stats.py, configs[2], 16,384 rows per density) float64 exponentials (NF_RQS_XEXP = 1, 14
// instructions per bin instead of 3) bring the scaled log_prob error from 1.88e-7 to 1.70e-7 rms -- v_exp_f32 is much
// better than the 1 ulp of noise per bin that costs 3.8e-7 in the NumPy emulation -- and cost 8 % of the fused
// kernel's time (1.01 -> 1.09 ms at 2^20 rows).
#pragma once

#include <hip/hip_runtime.h>

// 1: the knots' exponentials in float64 (exp_neg_f64, 14 instructions per bin); 0: hardware exp2 (3 per bin).
#ifndef NF_RQS_XEXP
#define NF_RQS_XEXP 0
#endif

namespace nf {
namespace rqsmath {

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
// 1/b: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_nr(float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  return fmaf(r, fmaf(-b, r, 1.f), r);
}
__device__ __forceinline__ float fast_div(float a, float b) { return a * rcp_nr(b); }

// log(1 + exp(a)) for a <= 0 with a RELATIVE error of ~2e-7 down to exp(a) -> 0.  The plain form log(1.f + e)
// loses e's digits in the sum (absolute error 6e-8: a slope of softplus + min_knot_slope = 1e-4 + 5e-5 comes out
// 4e-4 off, and so does its log in the log-det -- seen on out-of-range inputs of deep chains, where the conditioner
// sees |x| ~ 30 and emits slope logits of -10).  u = fl(1 + e) drops the low bits of e: they come back as the
// first-order term (e - (u - 1)) / u ~ (e - (u - 1)) (1 - e); log(u) itself is the hardware's, which is accurate
// RELATIVE to its result right down to u - 1 = 2^-24 (profiles/microbench/hw_log_near_one.hip: worst 1.1e-7 in every
// binade, so no series is needed for small e; for u == 1 the correction term alone is e).
__device__ __forceinline__ float log1p_exp(float a) {
  const float e = __builtin_amdgcn_exp2f(a * 1.44269504f);
  const float u = 1.f + e;
  return fmaf(e - (u - 1.f), 1.f - e, __builtin_amdgcn_logf(u) * 0.693147181f);
}

// softplus(v) = max(v, 0) + log(1 + exp(-|v|))
__device__ __forceinline__ float softplus_f(float v) { return fmaxf(v, 0.f) + log1p_exp(-fabsf(v)); }

// exp(t), t <= 0, as float64 with ~5e-9 relative error: 2^n * Taylor_7(f ln 2), t log2(e) = n + f, |f| <= 1/2.
__device__ __forceinline__ double exp_neg_f64(float t) {
  const double pq = (double)t * 1.4426950408889634;
  const double n = __builtin_rint(pq);
  const double x = (pq - n) * 0.6931471805599453;
  double y = 1.0 / 5040.0;
  y = __builtin_fma(y, x, 1.0 / 720.0);
  y = __builtin_fma(y, x, 1.0 / 120.0);
  y = __builtin_fma(y, x, 1.0 / 24.0);
  y = __builtin_fma(y, x, 1.0 / 6.0);
  y = __builtin_fma(y, x, 0.5);
  y = __builtin_fma(y, x, 1.0);
  y = __builtin_fma(y, x, 1.0);
  return __builtin_ldexp(y, (int)n);
}

// bin-size weight exp(lg - mx) as float64
template <int XEXP>
__device__ __forceinline__ double bin_exp(float lg, float mx) {
  if (XEXP) return exp_neg_f64(lg - mx);
  return (double)__builtin_amdgcn_exp2f(fmaf(lg, 1.44269504f, -mx * 1.44269504f));
}

// (range - K * min_bin) / sum: v_rcp_f64 + two Newton steps (sum is in [1, K])
template <class C>
__device__ __forceinline__ double bin_scale(double sum, int K, const C& c) {
  double r = __builtin_amdgcn_rcp(sum);
  r = __builtin_fma(r, __builtin_fma(-sum, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-sum, r, 1.0), r);
  return (((double)c.range_max - (double)c.range_min) - (double)K * (double)c.min_bin) * r;
}

// Knots of one axis from its K bin-size logits: softmax * (R - K*min_bin) + min_bin, running sum from
// range_min, both ends pinned (distrax: padded cumsum).  C: anything with range_min, range_max, min_bin.
// size[k] is bin k's width rounded from float64 on its own -- NOT knot[k+1] - knot[k]: a bin squeezed to
// min_bin_size = 1e-4 next to a knot at |x| ~ 5 would come out of that difference 0.5 % wrong (float32 spacing
// 4.8e-7 there), and so would delta = height / width and 2 log(delta) in the log-det (1e-2 absolute; seen on a
// sample whose layers park it at the range edge, where NumPy float32 is 2.5e-4 off for the same reason).
template <int K, int XEXP, class C>
__device__ __forceinline__ void knots_f64(const float (&lg)[K], const C& c, float (&knot)[K + 1], float (&size)[K]) {
  float mx = lg[0];
#pragma unroll
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
  double e[K];
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) { e[k] = bin_exp<XEXP>(lg[k], mx); sum += e[k]; }
  const double scale = bin_scale(sum, K, c);
  const double mb = (double)c.min_bin;
  double at = (double)c.range_min;
  knot[0] = c.range_min;
#pragma unroll
  for (int k = 0; k < K - 1; ++k) {
    const double w = __builtin_fma(e[k], scale, mb);
    size[k] = (float)w;
    at += w;
    knot[k + 1] = (float)at;
  }
  size[K - 1] = (float)((double)c.range_max - at);
  knot[K] = c.range_max;
}

// The rational-quadratic map on the bin with lower-left knot (xl, yl), width bw and height bh, the bin's two
// slope logits (raw_l, raw_r) and the boundary slope logits (raw_first, raw_last) for the linear tails.
// C: range_min, range_max, slope_offset, min_slope.  WAVE_TAILS: the (rare) tails sit behind a wave-uniform
// test -- only for callers whose whole wave is active here.
template <int INV, bool WAVE_TAILS, class C>
__device__ __forceinline__ void rqs_map(float xl, float bw, float yl, float bh, float raw_l, float raw_r,
                                        float raw_first, float raw_last, float v, const C& c, float* out,
                                        float* logdet) {
  const float sl = softplus_f(raw_l + c.slope_offset) + c.min_slope;
  const float sr = softplus_f(raw_r + c.slope_offset) + c.min_slope;
  const float rbw = rcp_nr(bw);
  const float delta = bh * rbw;
  const float st = sr + sl - 2.f * delta;
  constexpr float kLn2 = 0.693147181f;
  float res, ldv;
  if (!INV) {
    float z = (v - xl) * rbw;
    z = fminf(fmaxf(z, 0.f), 1.f);
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float num = bh * (delta * zz + sl * z1mz);
    const float den = delta + st * z1mz;
    res = fmaf(num, rcp_nr(den), yl);
    const float inner = sr * zz + 2.f * delta * z1mz + sl * omz * omz;
    ldv = (2.f * (__builtin_amdgcn_logf(delta) - __builtin_amdgcn_logf(den)) + __builtin_amdgcn_logf(inner)) * kLn2;
  } else {
    float u = (v - yl) * rcp_nr(bh);
    u = fminf(fmaxf(u, 0.f), 1.f);
    const float qc = -delta * u;
    const float qb = sl - st * u;
    const float qa = delta - qb;
    float z = -2.f * qc * rcp_nr(qb + __builtin_amdgcn_sqrtf(qb * qb - 4.f * qa * qc));
    z = fminf(fmaxf(z, 0.f), 1.f);
    res = fmaf(bw, z, xl);
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float den = delta + st * z1mz;
    const float inner = sr * zz + 2.f * delta * z1mz + sl * omz * omz;
    ldv = (2.f * (__builtin_amdgcn_logf(den) - __builtin_amdgcn_logf(delta)) - __builtin_amdgcn_logf(inner)) * kLn2;
  }
  // linear tails outside [range_min, range_max]
  const bool below = v <= c.range_min, above = v >= c.range_max;
  if (WAVE_TAILS ? (bool)__any(below || above) : (below || above)) {
    const float s_first = softplus_f(raw_first + c.slope_offset) + c.min_slope;
    const float s_last = softplus_f(raw_last + c.slope_offset) + c.min_slope;
    if (!INV) {
      if (below) { res = (v - c.range_min) * s_first + c.range_min; ldv = fast_log(s_first); }
      if (above) { res = (v - c.range_max) * s_last + c.range_max; ldv = fast_log(s_last); }
    } else {
      if (below) { res = fast_div(v - c.range_min, s_first) + c.range_min; ldv = -fast_log(s_first); }
      if (above) { res = fast_div(v - c.range_max, s_last) + c.range_max; ldv = -fast_log(s_last); }
    }
  }
  // jnp.clip keeps NaN, fmaxf / fminf above do not: a NaN coordinate stays NaN (integer test: the
  // kernels are built with -fno-honor-nans)
  const bool v_nan = (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u;
  *out = v_nan ? v : res;
  *logdet = v_nan ? v : ldv;
}

}  // namespace rqsmath
}  // namespace nf
