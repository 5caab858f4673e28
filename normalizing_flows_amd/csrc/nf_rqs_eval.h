// The register form of the rational-quadratic spline (K bins known at compile time) and the tile epilogue
// that applies it to a 128 x 128 GEMM result held in accumulators (SURVEY's K5), shared by
// nf_generic.hip (float32 GEMM) and nf_gemm_x3.hip (split-bf16 GEMM).  Not part of the ABI.
#pragma once

#include "nf_internal.h"

namespace nf {
namespace generic {

typedef float f32x16_rqs __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
__device__ __forceinline__ float softplus_fast(float v) {
  return fmaxf(v, 0.f) + flog(1.f + fexp2(-fabsf(v) * 1.44269504f));
}

template <int K, int INV>
__device__ __forceinline__ void rqs_eval_reg(const float* __restrict__ p, const RqsConst& c, float v, float* out,
                                             float* logdet) {
  constexpr float kL2E = 1.44269504f;
  const float total = c.range_max - c.range_min;
  float lw[K], lh[K];
  float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { lw[k] = p[k]; lh[k] = p[K + k]; mw = fmaxf(mw, lw[k]); mh = fmaxf(mh, lh[k]); }
  float sw = 0.f, sh = 0.f;
  const float mwl = -mw * kL2E, mhl = -mh * kL2E;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    lw[k] = fexp2(fmaf(lw[k], kL2E, mwl)); sw += lw[k];
    lh[k] = fexp2(fmaf(lh[k], kL2E, mhl)); sh += lh[k];
  }
  const float scw = (total - (float)K * c.min_bin) * __builtin_amdgcn_rcpf(sw);
  const float sch = (total - (float)K * c.min_bin) * __builtin_amdgcn_rcpf(sh);
  // walk the knots, keep the bin that contains v on the search axis
  float xl = c.range_min, yl = c.range_min;
  float bxl = xl, byl = yl, bxr = 0.f, byr = 0.f, rsl = p[2 * K], rsr = p[2 * K + 1];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float xr = k == K - 1 ? c.range_max : xl + fmaf(lw[k], scw, c.min_bin);
    const float yr = k == K - 1 ? c.range_max : yl + fmaf(lh[k], sch, c.min_bin);
    if (k == 0) { bxr = xr; byr = yr; }
    else {
      const bool hit = v >= (INV ? yl : xl);   // the last bin whose left knot is <= v
      bxl = hit ? xl : bxl; bxr = hit ? xr : bxr;
      byl = hit ? yl : byl; byr = hit ? yr : byr;
      rsl = hit ? p[2 * K + k] : rsl;
      rsr = hit ? p[2 * K + k + 1] : rsr;
    }
    xl = xr; yl = yr;
  }
  const float sl = softplus_fast(rsl + c.slope_offset) + c.min_slope;
  const float sr = softplus_fast(rsr + c.slope_offset) + c.min_slope;
  const float bw = bxr - bxl, bh = byr - byl;
  const float rbw = __builtin_amdgcn_rcpf(bw);
  const float delta = bh * rbw;
  const float st = sr + sl - 2.f * delta;
  float res, ldv;
  if (!INV) {
    float z = (v - bxl) * rbw;
    z = fminf(fmaxf(z, 0.f), 1.f);
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float rden = __builtin_amdgcn_rcpf(delta + st * z1mz);
    res = fmaf(bh * (delta * zz + sl * z1mz), rden, byl);
    ldv = flog(delta * delta * (sr * zz + 2.f * delta * z1mz + sl * omz * omz) * rden * rden);
  } else {
    float u = (v - byl) * __builtin_amdgcn_rcpf(bh);
    u = fminf(fmaxf(u, 0.f), 1.f);
    const float qc = -delta * u, qb = sl - st * u, qa = delta - qb;
    float z = -2.f * qc * __builtin_amdgcn_rcpf(qb + __builtin_amdgcn_sqrtf(qb * qb - 4.f * qa * qc));
    z = fminf(fmaxf(z, 0.f), 1.f);
    res = fmaf(bw, z, bxl);
    const float zz = z * z, z1mz = z - zz, omz = 1.f - z;
    const float rden = __builtin_amdgcn_rcpf(delta + st * z1mz);
    ldv = -flog(delta * delta * (sr * zz + 2.f * delta * z1mz + sl * omz * omz) * rden * rden);
  }
  if (v <= c.range_min || v >= c.range_max) {  // linear tails
    const bool below = v <= c.range_min;
    const float edge = below ? c.range_min : c.range_max;
    const float s = softplus_fast(p[below ? 2 * K : 3 * K] + c.slope_offset) + c.min_slope;
    res = INV ? (v - edge) * __builtin_amdgcn_rcpf(s) + edge : (v - edge) * s + edge;
    ldv = INV ? -flog(s) : flog(s);
  }
  // jnp.clip keeps NaN, fmaxf / fminf above do not: a NaN coordinate stays NaN (integer test: the
  // kernels are built with -fno-honor-nans)
  const bool v_nan = (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u;
  *out = v_nan ? v : res;
  *logdet = v_nan ? v : ldv;
}

constexpr int kCtStride = 129;  // floats per row of the staged result tile (odd: conflict-free column reads)

// acc[t][u]: the wave's 2 x 2 tiles (rows wm*64 + t*32 + ..., columns wn*64 + u*32 + lane&31) of a 128 x 128
// block whose columns hold PPB = 128 / (3K+1) whole transformed dims.  Ct: LDS, 64 x kCtStride floats;
// ldl: LDS, 64 x 8 floats.  The block goes through Ct in two 64-row halves; every thread takes
// (row, dim) items, applies the spline to y in place; a row's log-dets of the block's dims are added
// in dim order into ldpart[nb][row].  Must be entered with LDS free (after a barrier).
template <int K, int INV>
__device__ __forceinline__ void rqs_tile_epilogue(const f32x16_rqs (&acc)[2][2], const float* __restrict__ bias, int n0,
                                                  int m0, int nb, int M, int D, int pass_parity,
                                                  float* __restrict__ y, float* __restrict__ ldpart, float* Ct,
                                                  float* ldl, const RqsConst& c) {
  constexpr int P = 3 * K + 1, PPB = 128 / P;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int Dt = pass_parity == 0 ? D / 2 : (D + 1) / 2;
  const int first_t = pass_parity == 0 ? 1 : 0;
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = wn * 64 + u * 32 + i;
          const float bv = bias[n0 + col];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Ct[(t * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * kCtStride + col] = acc[t][u][r] + bv;
        }
    }
    __syncthreads();
    for (int item = tid; item < 64 * PPB; item += 256) {
      const int r = item & 63, px = item >> 6;
      const int t_global = nb * PPB + px;
      const long long row = (long long)m0 + half * 64 + r;
      float l = 0.f;
      if (t_global < Dt && row < M) {
        float* yp = y + row * D + first_t + 2 * t_global;
        float res;
        rqs_eval_reg<K, INV>(Ct + r * kCtStride + px * P, c, *yp, &res, &l);
        *yp = res;
      }
      ldl[r * 8 + px] = l;
    }
    __syncthreads();
    if (tid < 64) {
      const long long row = (long long)m0 + half * 64 + tid;
      if (row < M) {
        float sum = 0.f;
#pragma unroll
        for (int px = 0; px < PPB; ++px) sum += ldl[tid * 8 + px];
        ldpart[(size_t)nb * M + row] = sum;
      }
    }
    __syncthreads();
  }
}

}  // namespace generic
}  // namespace nf
