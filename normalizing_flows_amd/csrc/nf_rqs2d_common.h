// Device code shared by the two fused spline-coupling kernels (nf_rqs2d_h64.hip: exact-float32
// MFMA; nf_rqs2d_x3.hip: split-bf16 MFMA): kernel parameters, the accumulator-order helpers and
// the rational-quadratic spline itself (distrax.RationalQuadraticSpline semantics as restated in
// SURVEY.md section 8a row c2).  Not part of the ABI.
#pragma once

#include "nf_internal.h"
#include "nf_rqs_math.h"

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
};

__device__ __forceinline__ float swap_halves(float v) {
  // value of lane l^32
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

__device__ __forceinline__ float add_halves(float v) {
  // sum of lane l and lane l^32, bit-identical in both lanes
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
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

using rqsmath::knots_f64;
using rqsmath::softplus_f;

// The bin that holds v on the search axis (x going forward: half 0; y inversely: half 1), its lower-left knot
// and its size on both axes, identical in both half-waves.  knot[], size[]: this half's axis.
template <int K, int INV>
__device__ __forceinline__ int spline_bin(const float (&knot)[K + 1], const float (&size)[K], float v, int h, float* xl,
                                          float* bw, float* yl, float* bh) {
  int kb = 0;
#pragma unroll
  for (int k = 1; k < K; ++k) kb += (v >= knot[k]) ? 1 : 0;
  const int kb_other = __float_as_int(swap_halves(__int_as_float(kb)));
  kb = (h == INV) ? kb : kb_other;
  float lo = knot[0], sz = size[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool c = kb >= k;
    lo = c ? knot[k] : lo;
    sz = c ? size[k] : sz;
  }
  const float lo_o = swap_halves(lo), sz_o = swap_halves(sz);
  *xl = h == 0 ? lo : lo_o; *bw = h == 0 ? sz : sz_o;
  *yl = h == 0 ? lo_o : lo; *bh = h == 0 ? sz_o : sz;
  return kb;
}

// the map itself: rqsmath::rqs_map, tails behind a wave-uniform test (every lane of a live wave is here)
template <int INV>
__device__ __forceinline__ void spline_map(float xl, float bw, float yl, float bh, float raw_l, float raw_r,
                                           float raw_first, float raw_last, float v, const Params& p, float* out,
                                           float* logdet) {
  rqsmath::rqs_map<INV, true>(xl, bw, yl, bh, raw_l, raw_r, raw_first, raw_last, v, p, out, logdet);
}

// One spline evaluation with every slope logit at hand (nf_rqs2d_x3.hip).  lg[K]: this half's bin-size
// logits (widths on half 0, heights on half 1); sg[K+1]: slope logits (both halves).  v: the value to
// transform (same in both halves).  Returns the transformed value and log|det| (identical bits in both halves).
template <int K, int INV>
__device__ __forceinline__ void spline(const float (&lg)[K], const float (&sg)[K + 1], float v, int h,
                                       const Params& p, float* out, float* logdet) {
  float knot[K + 1], size[K];
  knots_f64<K, NF_RQS_XEXP>(lg, p, knot, size);
  float xl, bw, yl, bh;
  const int kb = spline_bin<K, INV>(knot, size, v, h, &xl, &bw, &yl, &bh);
  float sl = sg[0], sr = sg[1];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool c = kb >= k;
    sl = c ? sg[k] : sl;
    sr = c ? sg[k + 1] : sr;
  }
  spline_map<INV>(xl, bw, yl, bh, sl, sr, sg[0], sg[K], v, p, out, logdet);
}

}  // namespace rqs2d
}  // namespace nf
