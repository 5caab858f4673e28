// Device code shared by the two fused spline-coupling kernels (nf_rqs2d_h64.hip: exact-float32
// MFMA; nf_rqs2d_x3.hip: split-bf16 MFMA): kernel parameters, the accumulator-order helpers and
// the rational-quadratic spline itself (distrax.RationalQuadraticSpline semantics as restated in
// SURVEY.md section 8a row c2).  Not part of the ABI.
#pragma once

#include "nf_internal.h"

namespace nf {
namespace rqs2d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// half 0: widths [0,K) then slopes; half 1: heights [K,2K) then slopes; -1 = unused row
static inline int param_of_slot(int half, int q, int K) {
  if (q < K) return half == 0 ? q : K + q;
  if (q < 2 * K + 1) return 2 * K + (q - K);
  return -1;
}

struct Params {
  const float* wimg;
  ChainIO io;
  int first_mask_parity;
  float range_min, range_max, min_bin, min_slope, slope_offset;
  // waves of a workgroup that take sample groups (the others only help staging the layer images): small
  // batches spread over more workgroups instead of filling one.  0 = all of them.
  int active_waves = 0;
};

__device__ __forceinline__ float swap_halves(float v) {
  // value of lane l^32
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

__device__ __forceinline__ f32x16 lds_f32x16(const float* p) {
  const float4* q = reinterpret_cast<const float4*>(p);
  float4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 v;
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
  return v;
}

__device__ __forceinline__ float exp_f32(float x) {
  const float kLog2eHi = 1.44269502e+00f, kLog2eLo = 1.92596299e-08f, kLn2 = 6.93147182e-01f;
  const float hi = x * kLog2eHi;
  const float lo = fmaf(x, kLog2eLo, fmaf(x, kLog2eHi, -hi));
  const float e = __builtin_amdgcn_exp2f(hi);
  return fmaf(e, lo * kLn2, e);
}

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32, ~1 ulp each).  The
// spline's intermediate quantities are O(1) and enter log_prob additively, so 1-2 ulp per
// operation is far inside the float32 noise of the 8-layer chain (tests/test_parity_gpu.py).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// softplus(v) = max(v, 0) + log(1 + exp(-|v|))
__device__ __forceinline__ float softplus_f(float v) {
  return fmaxf(v, 0.f) + fast_log(1.f + fast_exp(-fabsf(v)));
}

// One spline evaluation.  lg[K]: this half's bin-size logits (widths on half 0, heights on
// half 1); sg[K+1]: slope logits (both halves).  v: the value to transform (same in both
// halves).  Returns the transformed value and log|det| (identical bits in both halves).
template <int K, int INV>
__device__ __forceinline__ void spline(const float (&lg)[K], const float (&sg)[K + 1], float v, int h,
                                       const Params& p, float* out, float* logdet) {
  const float total = p.range_max - p.range_min;
  // softmax -> bin sizes -> knots on this half's axis
  float mx = lg[0];
#pragma unroll
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
  float e[K];
  float sum = 0.f;
  const float mxl = -mx * 1.44269504f;
#pragma unroll
  for (int k = 0; k < K; ++k) { e[k] = __builtin_amdgcn_exp2f(fmaf(lg[k], 1.44269504f, mxl)); sum += e[k]; }
  const float scale = fast_div(total - (float)K * p.min_bin, sum);
  float knot[K + 1];
  knot[0] = p.range_min;
#pragma unroll
  for (int k = 0; k < K - 1; ++k) knot[k + 1] = knot[k] + fmaf(e[k], scale, p.min_bin);
  knot[K] = p.range_max;
  // bin: searched on the x axis (half 0) when going forward, on the y axis (half 1) inversely
  int kb = 0;
#pragma unroll
  for (int k = 1; k < K; ++k) kb += (v >= knot[k]) ? 1 : 0;
  const int kb_other = __float_as_int(swap_halves(__int_as_float(kb)));
  kb = (h == INV) ? kb : kb_other;
  float lo = knot[0], hi = knot[1], sl = sg[0], sr = sg[1];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool c = kb >= k;
    lo = c ? knot[k] : lo;
    hi = c ? knot[k + 1] : hi;
    sl = c ? sg[k] : sl;
    sr = c ? sg[k + 1] : sr;
  }
  const float lo_o = swap_halves(lo), hi_o = swap_halves(hi);
  const float xl = h == 0 ? lo : lo_o, xr = h == 0 ? hi : hi_o;
  const float yl = h == 0 ? lo_o : lo, yr = h == 0 ? hi_o : hi;
  sl = softplus_f(sl + p.slope_offset) + p.min_slope;
  sr = softplus_f(sr + p.slope_offset) + p.min_slope;
  const float bw = xr - xl, bh = yr - yl;
  const float rbw = __builtin_amdgcn_rcpf(bw);
  const float delta = bh * rbw;
  const float st = sr + sl - 2.f * delta;
  float res, ldv;
  if (!INV) {
    float z = (v - xl) * rbw;
    z = fminf(fmaxf(z, 0.f), 1.f);
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float num = bh * (delta * zz + sl * z1mz);
    const float den = delta + st * z1mz;
    const float rden = __builtin_amdgcn_rcpf(den);
    res = fmaf(num, rden, yl);
    ldv = fast_log(delta * delta * (sr * zz + 2.f * delta * z1mz + sl * omz * omz) * rden * rden);
  } else {
    float u = fast_div(v - yl, bh);
    u = fminf(fmaxf(u, 0.f), 1.f);
    const float qc = -delta * u;
    const float qb = sl - st * u;
    const float qa = delta - qb;
    float z = fast_div(-2.f * qc, qb + __builtin_amdgcn_sqrtf(qb * qb - 4.f * qa * qc));
    z = fminf(fmaxf(z, 0.f), 1.f);
    res = bw * z + xl;
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float den = delta + st * z1mz;
    const float rden = __builtin_amdgcn_rcpf(den);
    ldv = -fast_log(delta * delta * (sr * zz + 2.f * delta * z1mz + sl * omz * omz) * rden * rden);
  }
  // linear tails outside [range_min, range_max]: rare, so behind a wave-uniform test
  const bool below = v <= p.range_min, above = v >= p.range_max;
  if (__any(below || above)) {
    const float s_first = softplus_f(sg[0] + p.slope_offset) + p.min_slope;
    const float s_last = softplus_f(sg[K] + p.slope_offset) + p.min_slope;
    if (!INV) {
      if (below) { res = (v - p.range_min) * s_first + p.range_min; ldv = fast_log(s_first); }
      if (above) { res = (v - p.range_max) * s_last + p.range_max; ldv = fast_log(s_last); }
    } else {
      if (below) { res = fast_div(v - p.range_min, s_first) + p.range_min; ldv = -fast_log(s_first); }
      if (above) { res = fast_div(v - p.range_max, s_last) + p.range_max; ldv = -fast_log(s_last); }
    }
  }
  // jnp.clip keeps NaN, fmaxf / fminf above do not: a NaN coordinate stays NaN (integer test: the
  // kernels are built with -fno-honor-nans)
  const bool v_nan = (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u;
  *out = v_nan ? v : res;
  *logdet = v_nan ? v : ldv;
}

}  // namespace rqs2d
}  // namespace nf
