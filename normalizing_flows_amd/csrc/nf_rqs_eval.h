// The register form of the rational-quadratic spline (K bins known at compile time) and the tile epilogue
// that applies it to a 128 x 128 GEMM result held in accumulators (SURVEY's K5), shared by
// nf_generic.hip (float32 GEMM) and nf_gemm_x3.hip (split-bf16 GEMM).  Not part of the ABI.
#pragma once

#include "nf_internal.h"
#include "nf_rqs_math.h"

namespace nf {
namespace generic {

typedef float f32x16_rqs __attribute__((ext_vector_type(16)));

// One (row, transformed dim) with the 3K+1 raw parameters at p [widths K | heights K | slopes K+1]: the search
// axis first (its knots, the bin), then the other axis' knots of that bin -- one axis' float64 bin weights live
// at a time.  Arithmetic: nf_rqs_math.h.
template <int K, int INV>
__device__ __forceinline__ void rqs_eval_reg(const float* __restrict__ p, const RqsConst& c, float v, float* out,
                                             float* logdet) {
  float lg[K], knot[K + 1], size[K];
  // search axis: x (widths) going forward, y (heights) inversely
#pragma unroll
  for (int k = 0; k < K; ++k) lg[k] = p[(INV ? K : 0) + k];
  rqsmath::knots_f64<K, NF_RQS_XEXP>(lg, c, knot, size);
  int kb = 0;
#pragma unroll
  for (int k = 1; k < K; ++k) kb += (v >= knot[k]) ? 1 : 0;
  float slo = knot[0], ssz = size[0], raw_l = p[2 * K], raw_r = p[2 * K + 1];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = kb >= k;
    slo = hit ? knot[k] : slo;
    ssz = hit ? size[k] : ssz;
    raw_l = hit ? p[2 * K + k] : raw_l;
    raw_r = hit ? p[2 * K + k + 1] : raw_r;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) lg[k] = p[(INV ? 0 : K) + k];
  rqsmath::knots_f64<K, NF_RQS_XEXP>(lg, c, knot, size);
  float olo = knot[0], osz = size[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = kb >= k;
    olo = hit ? knot[k] : olo;
    osz = hit ? size[k] : osz;
  }
  const float xl = INV ? olo : slo, bw = INV ? osz : ssz, yl = INV ? slo : olo, bh = INV ? ssz : osz;
  rqsmath::rqs_map<INV, false>(xl, bw, yl, bh, raw_l, raw_r, p[2 * K], p[3 * K], v, c, out, logdet);
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
