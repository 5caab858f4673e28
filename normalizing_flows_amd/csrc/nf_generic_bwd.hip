// General path, training: d(NLL)/d(weights) for the masked coupling kinds (affine and
// rational-quadratic spline) of any event size and conditioner widths -- what jax.grad(loss_fn)
// computes in src/real_nvp_toy_densities.py:164-173, src/maf_toy_densities.py:167-176 and
// src/maf_mnist.py:159-168.  SURVEY.md 8f rank 2 ("backward of K1 / K4").
//
// The forward walk keeps every layer's input, hidden activations and conditioner output (2-D
// flows at batch 2^20: 5.6 GB; the MNIST shape at 2^16: 19 GB -- memory is what a 288 GB part has
// to spend; up to 64 GiB, beyond that the batch must be split by the caller).  Going back through
// the chain, every layer
//   1. finds a_1 .. a_n and theta on the tape (the first build recomputed them: three more GEMMs
//      per layer)
//   2. has nothing to invert: unlike the affine map, a spline's float32 inverse can be off by
//      1e-4 where |dx/dy| is large, and the parameter gradients inherit that error (measured when
//      the first build re-derived the inputs from the output)
//   3. transform backward: d theta and dL/dx of the transformed dims from dL/dy and dL/d(logdet)
//   4. MLP backward: [dW_j; db_j] += [a_{j-1}, 1]^T g (rows on the MFMA k dimension, fixed-order
//      partial sums), g <- (g W_j^T) * act'(a_{j-1}) (the forward GEMM kernels on a transposed
//      copy of W_j, the activation mask in their epilogue), and finally dL/dx of the pass-through
//      dims += g W_1^T.
// Gradients are accumulated in the plan's own weight layout (masked rows / pruned columns are
// structural zeros there) and scattered to the canonical layout at the end.
//
// The spline's partial derivatives are not hand-derived: the rational-quadratic formula is
// evaluated on dual numbers carrying the seven partials (x, the bin's two knots on each axis, its
// two slopes), forward mode; softmax / cumulative-sum / softplus are reversed by hand.
#include "nf_internal.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- small kernels ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void base_grad_kernel(const float* __restrict__ z, float* __restrict__ g,
                                                        long long total, int base, float inv_n) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256)
    g[q] = base == NF_BASE_STD_NORMAL ? z[q] * inv_n : 0.f;  // -d/dz log N(z; 0, 1) = z
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                  long long total) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256)
    dst[q] += src[q];
}

// every Linear's W [K][N] -> WT [N][K] in one launch: blockIdx.y picks the Linear from a device table
struct TransposeJob {
  unsigned long long w_off, wT_off;
  int K, N;
};
__global__ __launch_bounds__(256) void transpose_all_kernel(const float* __restrict__ weights, float* __restrict__ wT,
                                                            const TransposeJob* __restrict__ jobs) {
  // 32 x 32 tiles through LDS: reads run along n, writes along k (the element-wise form read with stride N:
  // 0.7 TB/s on the MNIST shape's 351 MB)
  __shared__ float tile[64][65];
  const TransposeJob job = jobs[blockIdx.y];
  const int K = job.K, N = job.N;
  const float* W = weights + job.w_off;
  float* WT = wT + job.wT_off;
  if (((K | N) & 3) == 0 && ((job.w_off | job.wT_off) & 3) == 0) {
    // 64 x 64 tiles, 16-byte loads along n and 16-byte stores along k (the 32 x 32 form below moves 128-byte pieces:
    // 3.3 TB/s on the MNIST shape's 372 MB)
    const int tk = (K + 63) / 64, tn = (N + 63) / 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 quads x 16 rows
    for (long long t = blockIdx.x; t < (long long)tk * tn; t += gridDim.x) {
      const int k0 = (int)(t / tn) * 64, n0 = (int)(t % tn) * 64;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + 16 * r, n = n0 + 4 * tx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && n < N) v = *reinterpret_cast<const float4*>(W + (size_t)k * N + n);   // (N % 4 == 0: whole quads)
        float* d = &tile[ty + 16 * r][4 * tx];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + 16 * r, k = k0 + 4 * tx;
        if (n < N && k < K)
          *reinterpret_cast<float4*>(WT + (size_t)n * K + k) =
              make_float4(tile[4 * tx][ty + 16 * r], tile[4 * tx + 1][ty + 16 * r], tile[4 * tx + 2][ty + 16 * r],
                          tile[4 * tx + 3][ty + 16 * r]);
      }
      __syncthreads();
    }
    return;
  }
  const int tk = (K + 31) / 32, tn = (N + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (long long t = blockIdx.x; t < (long long)tk * tn; t += gridDim.x) {
    const int k0 = (int)(t / tn) * 32, n0 = (int)(t % tn) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + ty + 8 * r, n = n0 + tx;
      tile[ty + 8 * r][tx] = (k < K && n < N) ? W[(size_t)k * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + ty + 8 * r, k = k0 + tx;
      if (n < N && k < K) WT[(size_t)n * K + k] = tile[tx][ty + 8 * r];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gather_weights_kernel(const float* __restrict__ canon,
                                                             const int* __restrict__ index, float* __restrict__ w,
                                                             long long total) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256)
    w[q] = index[q] >= 0 ? canon[index[q]] : 0.f;
}

__global__ __launch_bounds__(256) void scatter_weights_kernel(const float* __restrict__ w, const int* __restrict__ index,
                                                              float* __restrict__ canon, long long total) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256)
    if (index[q] >= 0) canon[index[q]] = w[q];
}

__global__ __launch_bounds__(256) void scatter_grad_kernel(const float* __restrict__ g, const int* __restrict__ index,
                                                           float* __restrict__ canon_grad, long long total) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256)
    if (index[q] >= 0) canon_grad[index[q]] = g[q];
}

// ---- masked affine: y_t = x_t exp(ls) + shift on the transformed dims, logdet = n_t ls -----------
// v holds the layer's INPUT.  th[row] = (shift, ls).  G: dL/dy in, dL/dx (transformed dims) out.
__global__ __launch_bounds__(256) void affine_masked_bwd_kernel(const float* __restrict__ v, float* __restrict__ G,
                                                                const float* __restrict__ th,
                                                                float* __restrict__ dth, long long rows, int D,
                                                                int pass_parity, float gld) {
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float e = __expf(th[2 * r + 1]);
    float gs = 0.f, gl = 0.f;
    int nt = 0;
    for (int d = 0; d < D; ++d)
      if ((d & 1) != pass_parity) {
        const float gy = G[r * D + d];
        gs += gy;
        gl += gy * v[r * D + d] * e;
        G[r * D + d] = gy * e;
        ++nt;
      }
    dth[2 * r + 0] = gs;
    dth[2 * r + 1] = gl + gld * nt;
  }
}

// ---- spline ------------------------------------------------------------------------------------
struct D7 {
  float v, d[7];
};
__device__ __forceinline__ D7 d7_var(float v, int which) {
  D7 r;
  r.v = v;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = j == which ? 1.f : 0.f;
  return r;
}
__device__ __forceinline__ D7 operator+(const D7& a, const D7& b) {
  D7 r;
  r.v = a.v + b.v;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = a.d[j] + b.d[j];
  return r;
}
__device__ __forceinline__ D7 operator-(const D7& a, const D7& b) {
  D7 r;
  r.v = a.v - b.v;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = a.d[j] - b.d[j];
  return r;
}
__device__ __forceinline__ D7 operator*(const D7& a, const D7& b) {
  D7 r;
  r.v = a.v * b.v;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = a.d[j] * b.v + a.v * b.d[j];
  return r;
}
__device__ __forceinline__ D7 operator/(const D7& a, const D7& b) {
  D7 r;
  const float inv = 1.f / b.v;
  r.v = a.v * inv;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = (a.d[j] - r.v * b.d[j]) * inv;
  return r;
}
__device__ __forceinline__ D7 d7_scale(const D7& a, float c) {
  D7 r;
  r.v = a.v * c;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = a.d[j] * c;
  return r;
}
__device__ __forceinline__ D7 d7_log(const D7& a) {
  D7 r;
  const float inv = 1.f / a.v;
  r.v = logf(a.v);
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = a.d[j] * inv;
  return r;
}
__device__ __forceinline__ D7 d7_one_minus(const D7& a) {
  D7 r;
  r.v = 1.f - a.v;
#pragma unroll
  for (int j = 0; j < 7; ++j) r.d[j] = -a.d[j];
  return r;
}

__device__ __forceinline__ float softplus_bwd(float v) { return fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v))); }
__device__ __forceinline__ float sigmoid_bwd(float v) { return 1.f / (1.f + __expf(-v)); }

// One (row, transformed dim).  p: the 3K+1 raw parameters, gp: their gradient (written, all of it).
// x: the layer's input value; gy = dL/dy; gld = dL/d(logdet).  Returns dL/dx.
template <int KMAX>
__device__ float rqs_bwd_one(const float* __restrict__ p, float* __restrict__ gp, const RqsConst& c, float x,
                             float gy, float gld) {
  const int K = c.K;
  const float total = c.range_max - c.range_min;
  const float cs = total - (float)K * c.min_bin;
  // softmax of the width / height logits
  float sw[KMAX], sh[KMAX];
  float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) { mw = fmaxf(mw, p[k]); mh = fmaxf(mh, p[K + k]); }
  float sumw = 0.f, sumh = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      sw[k] = __expf(p[k] - mw);
      sh[k] = __expf(p[K + k] - mh);
      sumw += sw[k];
      sumh += sh[k];
    }
  const float iw = 1.f / sumw, ih = 1.f / sumh;
  // find the bin on the x axis while accumulating both axes' knots
  int kb = 0;
  float xl = c.range_min, yl = c.range_min, xr = c.range_max, yr = c.range_max;
  {
    float kx = c.range_min, ky = c.range_min;
    bool found = false;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        sw[k] *= iw;
        sh[k] *= ih;
        const float nx = k == K - 1 ? c.range_max : kx + fmaf(sw[k], cs, c.min_bin);
        const float ny = k == K - 1 ? c.range_max : ky + fmaf(sh[k], cs, c.min_bin);
        if (!found && (x < nx || k == K - 1)) {
          found = true;
          kb = k;
          xl = kx; xr = nx; yl = ky; yr = ny;
        }
        kx = nx;
        ky = ny;
      }
  }
  const bool below = x <= c.range_min, above = x >= c.range_max;
  float g_xl = 0.f, g_xr = 0.f, g_yl = 0.f, g_yr = 0.f, g_sl = 0.f, g_sr = 0.f, gx;
  int ksl = kb, ksr = kb + 1;
  if (below || above) {
    const int ks = below ? 0 : K;
    const float raw = p[2 * K + ks] + c.slope_offset;
    const float s = softplus_bwd(raw) + c.min_slope;
    gx = gy * s;
    g_sl = gy * (x - (below ? c.range_min : c.range_max)) + gld / s;
    ksl = ks;
    ksr = -1;
  } else {
    const float rl = p[2 * K + kb] + c.slope_offset, rr = p[2 * K + kb + 1] + c.slope_offset;
    const D7 X = d7_var(x, 0), XL = d7_var(xl, 1), XR = d7_var(xr, 2), YL = d7_var(yl, 3), YR = d7_var(yr, 4);
    const D7 SL = d7_var(softplus_bwd(rl) + c.min_slope, 5), SR = d7_var(softplus_bwd(rr) + c.min_slope, 6);
    const D7 w = XR - XL, hgt = YR - YL;
    const D7 delta = hgt / w;
    const D7 z = (X - XL) / w;
    const D7 zz = z * z;
    const D7 z1mz = z - zz;
    const D7 omz = d7_one_minus(z);
    const D7 st = SR + SL - d7_scale(delta, 2.f);
    const D7 den = delta + st * z1mz;
    const D7 y = YL + hgt * (delta * zz + SL * z1mz) / den;
    const D7 ld = d7_scale(d7_log(delta), 2.f) + d7_log(SR * zz + d7_scale(delta * z1mz, 2.f) + SL * omz * omz) -
                  d7_scale(d7_log(den), 2.f);
    float g7[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) g7[j] = gy * y.d[j] + gld * ld.d[j];
    gx = g7[0];
    g_xl = g7[1]; g_xr = g7[2]; g_yl = g7[3]; g_yr = g7[4];
    g_sl = g7[5]; g_sr = g7[6];
  }
  // slopes: s = softplus(raw + offset) + min_slope
#pragma unroll
  for (int k = 0; k <= KMAX; ++k)
    if (k <= K) {
      float g = 0.f;
      if (k == ksl) g = g_sl;
      else if (k == ksr) g = g_sr;
      gp[2 * K + k] = g != 0.f ? g * sigmoid_bwd(p[2 * K + k] + c.slope_offset) : 0.f;
    }
  // knots -> bin sizes: knot[k] = min + sum_{j<k} size_j for 1 <= k <= K-1 (knot 0 and K are constants),
  // so d/d size_j = sum of the knot gradients with index > j: only knots kb and kb+1 carry any
  const float gxl = kb >= 1 ? g_xl : 0.f, gxr = kb + 1 <= K - 1 ? g_xr : 0.f;
  const float gyl = kb >= 1 ? g_yl : 0.f, gyr = kb + 1 <= K - 1 ? g_yr : 0.f;
  float dotw = 0.f, doth = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      const float gw = (k < kb ? gxl : 0.f) + (k <= kb ? gxr : 0.f);
      const float gh = (k < kb ? gyl : 0.f) + (k <= kb ? gyr : 0.f);
      dotw += gw * sw[k];
      doth += gh * sh[k];
    }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      const float gw = (k < kb ? gxl : 0.f) + (k <= kb ? gxr : 0.f);
      const float gh = (k < kb ? gyl : 0.f) + (k <= kb ? gyr : 0.f);
      gp[k] = cs * sw[k] * (gw - dotw);
      gp[K + k] = cs * sh[k] * (gh - doth);
    }
  return gx;
}

// One (row, transformed dim) item per thread and pass.  An item's 3K+1 parameters are contiguous, a thread
// reading its own would touch 64 different cache lines per instruction: the workgroup's 256 items' parameters
// (one contiguous block) go through LDS with coalesced loads, the gradient goes back the same way (a lane's
// values are then a stride-(3K+1) LDS access: odd stride, conflict-free).
template <int KMAX>
__global__ __launch_bounds__(256) void rqs_masked_bwd_kernel(const float* __restrict__ v, float* __restrict__ G,
                                                             const float* __restrict__ th, float* __restrict__ dth,
                                                             long long rows, int D, int pass_parity, float gld,
                                                             int ipp, RqsConst c) {
  extern __shared__ __attribute__((aligned(16))) float sm_bwd[];
  const int Dt = pass_parity == 0 ? D / 2 : (D + 1) / 2;
  const int P = 3 * c.K + 1;
  const long long total = rows * Dt;
  float* sp = sm_bwd;             // [ipp][P] parameters (ipp <= 256 items per pass: what fits the LDS)
  float* sg = sm_bwd + ipp * P;   // [ipp][P] their gradient
  for (long long q0 = (long long)blockIdx.x * ipp; q0 < total; q0 += (long long)gridDim.x * ipp) {
    const int items = (int)(total - q0 < ipp ? total - q0 : ipp);
    const int nfl = items * P;
    const float* src = th + (size_t)q0 * P;
    for (int e = threadIdx.x; e < nfl; e += 256) sp[e] = src[e];
    __syncthreads();
    if ((int)threadIdx.x < items) {
      const long long q = q0 + threadIdx.x;
      const long long r = q / Dt;
      const int t = (int)(q - r * Dt);
      const int d = 2 * t + (pass_parity == 0 ? 1 : 0);
      G[r * D + d] = rqs_bwd_one<KMAX>(sp + threadIdx.x * P, sg + threadIdx.x * P, c, v[r * D + d], G[r * D + d], gld);
    }
    __syncthreads();
    float* dst = dth + (size_t)q0 * P;
    for (int e = threadIdx.x; e < nfl; e += 256) dst[e] = sg[e];
    __syncthreads();
  }
}

// ---- weight and bias gradient: [dW; db] (+)= [A, 1]^T B, rows on the MFMA k dimension ----------
// A [rows, K] is extended by a column of ones (index K), so the bias gradient (column sums of B) is
// row K of the same product.  One workgroup: a 32 x 32 tile of the (K+1) x N result over one chunk
// of rows, split between its 4 waves (8 row pairs in flight per wave) and added through LDS in a
// fixed order.
// grad_w != null (only with ONE row chunk): the tile is added straight into the gradient -- every element
// belongs to exactly one workgroup -- and the separate reduction launch disappears (small batches).
__global__ __launch_bounds__(256) void tn_partial_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ B, int ldb,
                                                         float* __restrict__ partial, long long rows,
                                                         long long chunk_rows, int K, int N, int tiles_n,
                                                         float* __restrict__ grad_w, float* __restrict__ grad_b,
                                                         int overwrite) {
  __shared__ float red[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int k0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
  const long long c0 = (long long)blockIdx.y * chunk_rows;
  const long long c1 = std::min(rows, c0 + chunk_rows);
  const long long per = ((c1 - c0 + 3) / 4 + 1) & ~1LL;  // even: a wave's 2-row steps stay aligned
  const long long m0 = c0 + wave * per, m1 = std::min(c1, m0 + per);
  const int kcol = k0 + i;
  const bool kin = kcol < K, kone = kcol == K, nin = n0 + i < N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (long long m = m0; m < m1; m += 16) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long row = m + 2 * u + h;
      const bool ok = row < m1;
      a[u] = ok ? (kin ? A[row * lda + kcol] : (kone ? 1.f : 0.f)) : 0.f;
      b[u] = (ok && nin) ? B[row * ldb + n0 + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.y * (K + 1) * N;
  for (int q = threadIdx.x; q < 1024; q += 256) {
    const int r = q >> 6, l = q & 63;
    const int kk = k0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), nn = n0 + (l & 31);
    if (kk <= K && nn < N) {
      const float t = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
      if (grad_w == nullptr) out[(size_t)kk * N + nn] = t;
      else if (kk < K) grad_w[(size_t)kk * N + nn] = overwrite ? t : grad_w[(size_t)kk * N + nn] + t;
      else grad_b[nn] = overwrite ? t : grad_b[nn] + t;
    }
  }
}

// The same product for FEW rows and a WIDE result (the last Linear of the 784-D flow at batch 128: [513, 9800] over 128
// rows).  tn_partial_kernel's 32 x 32 tiles read 32 KB of operands per 4 KB of result with one 4-byte-per-lane load per
// MFMA operand: 164 MB of L2 -> CU traffic and 2,600 wave-loads per CU, 37 us per launch against 8 us of MFMA time, a
// quarter of the training step at batch 128 (EXPERIMENTS 3.8).  Here: one workgroup = 64 x 128 of the result, eight
// waves with one 32 x 32 tile each; 32 rows of both operands at a time are staged through LDS with 16-byte loads, MFMA
// operands are 4-byte LDS reads (lane = column: conflict-free).  66 MB of L2 -> CU traffic, a quarter of the
// vector-memory instructions.  Every result element belongs to one lane: added straight into the gradient.
// 24 KB of LDS, so that three of these workgroups fit on a CU BESIDE one of gemm_splitk_kernel's (80 KB): Plan::backward
// runs the two on different streams, and with 64-row stages (48 KB) the data-gradient GEMM waited for LDS and took 50 us
// instead of 20 (training step 1.757 ms against 1.663).
constexpr int kTnChunk = 32;
__global__ __launch_bounds__(512) void tn_few_rows_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, int rows, int K, int N,
                                                          int tiles_k, int tiles_n, float* __restrict__ grad_w,
                                                          float* __restrict__ grad_b, int overwrite) {
  __shared__ __attribute__((aligned(16))) float As[kTnChunk][64];
  __shared__ __attribute__((aligned(16))) float Bs[kTnChunk][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // Workgroup -> block, XCD-aware: workgroups are dealt to the eight XCDs round-robin (blockIdx.x % 8), and the
  // tiles_k workgroups that share a 128-column block of B (64 KB of d theta per 64 rows) should find it in ONE L2:
  // on XCD x, slot s = blockIdx.x / 8 walks column block 8 (s / tiles_k) + x with the k block s % tiles_k fastest.
  // (Dealt out k-major, every column block was fetched from HBM by all nine k blocks: 62 MB per launch, rocprofv3
  // FETCH_SIZE, against 25 MB of operands + result.)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nblock = 8 * (slot / tiles_k) + xcd;
  if (nblock >= tiles_n) return;   // (the grid is rounded up to whole rounds of eight column blocks; no barrier reached yet)
  const int kb = (slot % tiles_k) * 64, nb = nblock * 128;
  const int kt = 32 * (wave & 1), nt = 32 * (wave >> 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int m0 = 0; m0 < rows; m0 += kTnChunk) {
    if (m0) __syncthreads();
    // A: 64 rows x 16 quads, B: 64 rows x 32 quads; a quad past the edge is assembled element by element
    // (column K of [A, 1] is the column of ones that makes row K of the product the bias gradient)
    for (int q = tid; q < kTnChunk * 16; q += 512) {
      const int r = q >> 4, c = (q & 15) * 4;
      const int row = m0 + r, col = kb + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows) {
        const float* src = A + (size_t)row * lda + col;
        if (col + 3 < K) v = *reinterpret_cast<const float4*>(src);
        else {
          v.x = col < K ? src[0] : (col == K ? 1.f : 0.f);
          v.y = col + 1 < K ? src[1] : (col + 1 == K ? 1.f : 0.f);
          v.z = col + 2 < K ? src[2] : (col + 2 == K ? 1.f : 0.f);
          v.w = col + 3 < K ? src[3] : (col + 3 == K ? 1.f : 0.f);
        }
      }
      *reinterpret_cast<float4*>(&As[r][c]) = v;
    }
    for (int q = tid; q < kTnChunk * 32; q += 512) {
      const int r = q >> 5, c = (q & 31) * 4;
      const int row = m0 + r, col = nb + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows) {
        const float* src = B + (size_t)row * ldb + col;
        if (col + 3 < N) v = *reinterpret_cast<const float4*>(src);
        else {
          if (col < N) v.x = src[0];
          if (col + 1 < N) v.y = src[1];
          if (col + 2 < N) v.z = src[2];
        }
      }
      *reinterpret_cast<float4*>(&Bs[r][c]) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int u = 0; u < kTnChunk / 2; ++u)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * u + h][kt + i], Bs[2 * u + h][nt + i], acc, 0, 0, 0);
  }
  const int nn = nb + nt + i;
  if (nn < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = kb + kt + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (kk < K) grad_w[(size_t)kk * N + nn] = overwrite ? acc[r] : grad_w[(size_t)kk * N + nn] + acc[r];
      else if (kk == K) grad_b[nn] = overwrite ? acc[r] : grad_b[nn] + acc[r];
    }
  }
}

// 64 outputs per workgroup; its four waves each add every fourth chunk slab (in order), LDS joins the
// four partial sums in a fixed order.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, int nchunks, size_t kn,
                                                        size_t n, float* __restrict__ grad_w,
                                                        float* __restrict__ grad_b) {
  __shared__ float red[4][64];
  const size_t count = kn + n;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t q = (size_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (q < count)
    for (int c = grp; c < nchunks; c += 4) s += partial[(size_t)c * count + q];
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && q < count) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (q < kn) grad_w[q] += t;
    else grad_b[q - kn] += t;
  }
}

static int grid_for(long long total, int cap = 8192) {
  return (int)std::max<long long>(1, std::min<long long>(cap, (total + 255) / 256));
}

// grad_w[K*N] += A^T B, grad_b[N] += column sums of B.  overwrite (only honoured on the one-chunk routes, which the
// caller has checked with tn_one_chunk()): = instead of +=, for a gradient buffer that was not zeroed.
static bool tn_one_chunk(long long rows) { return rows < 512; }   // below: want <= rows / 256 <= 1
static int accumulate_tn(const float* A, int lda, const float* B, int ldb, long long rows, int K, int N,
                         float* grad_w, float* grad_b, float* partials, size_t partial_floats, hipStream_t stream,
                         bool overwrite = false) {
  const size_t count = (size_t)(K + 1) * N;
  const int tiles_k = (K + 1 + 31) / 32, tiles_n = (N + 31) / 32;
  // enough row chunks to fill the chip even when the result has few tiles (64 x 64: four)
  long long want = std::max<long long>(1, 2048 / ((long long)tiles_k * tiles_n));
  want = std::min<long long>(want, std::max<long long>(1, rows / 256));
  int nchunks = (int)std::min<long long>(256, want);
  nchunks = (int)std::max<size_t>(1, std::min<size_t>((size_t)nchunks, partial_floats / count));
  long long chunk_rows = (rows + nchunks - 1) / nchunks;
  chunk_rows = (chunk_rows + 1) & ~1LL;
  nchunks = (int)((rows + chunk_rows - 1) / chunk_rows);
  const int wide_k = (K + 1 + 63) / 64, wide_n = (N + 127) / 128;
  const bool quads = (lda & 3) == 0 && (ldb & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (nchunks == 1 && rows <= 256 && quads && (long long)wide_k * wide_n >= 256 && !switches().no_tn_few_rows) {
    hipLaunchKernelGGL(tn_few_rows_kernel, dim3(8 * wide_k * ((wide_n + 7) / 8)), dim3(512), 0, stream, A, lda, B, ldb,
                       (int)rows, K, N, wide_k, wide_n, grad_w, grad_b, overwrite ? 1 : 0);
  } else if (nchunks == 1) {
    hipLaunchKernelGGL(tn_partial_kernel, dim3(tiles_k * tiles_n, 1), dim3(256), 0, stream, A, lda, B, ldb, partials,
                       rows, chunk_rows, K, N, tiles_n, grad_w, grad_b, overwrite ? 1 : 0);
  } else {
    if (overwrite) { set_error("weight gradient: overwrite asked for on the chunked route"); return NF_ERR_STATE; }
    hipLaunchKernelGGL(tn_partial_kernel, dim3(tiles_k * tiles_n, nchunks), dim3(256), 0, stream, A, lda, B, ldb,
                       partials, rows, chunk_rows, K, N, tiles_n, (float*)nullptr, (float*)nullptr, 0);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, partials, nchunks,
                       (size_t)K * N, (size_t)N, grad_w, grad_b);
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ---- host ----------------------------------------------------------------------------------
int Plan::prepare_backward() {
  Backward& b = bwd;
  if (desc.kind != NF_KIND_AFFINE_MASKED && desc.kind != NF_KIND_RQS_MASKED) {
    set_error("the general backward pass covers the masked coupling kinds");
    return NF_ERR_UNSUPPORTED;
  }
  if (desc.kind == NF_KIND_RQS_MASKED && desc.rqs_bins > 32) {
    set_error("the spline backward kernel is built for up to 32 bins");
    return NF_ERR_UNSUPPORTED;
  }
  hipFree(b.d_grad); hipFree(b.d_wT); hipFree(b.d_canon_index); hipFree(b.d_zero); hipFree(b.d_partials);
  hipFree(b.d_small_partials);
  b.d_grad = b.d_wT = b.d_zero = b.d_partials = b.d_small_partials = nullptr;
  b.d_canon_index = nullptr;
  b.ready = false;
  b.n_weights = host_weights.size();
  NF_HIP(hipMalloc(&b.d_grad, b.n_weights * sizeof(float)));
  NF_HIP(hipMemset(b.d_grad, 0, b.n_weights * sizeof(float)));   // (a walk that overwrites skips the alignment gaps)
  NF_HIP(hipMalloc(&b.d_canon_index, b.n_weights * sizeof(int)));
  NF_HIP(hipMemcpy(b.d_canon_index, canon_index.data(), b.n_weights * sizeof(int), hipMemcpyHostToDevice));
  {
    std::vector<unsigned> bits((b.n_weights + 31) / 32, 0u);
    for (size_t e = 0; e < b.n_weights; ++e)
      if (canon_index[e] >= 0) bits[e >> 5] |= 1u << (e & 31);
    hipFree(b.d_live_bits);
    b.d_live_bits = nullptr;
    NF_HIP(hipMalloc(&b.d_live_bits, bits.size() * sizeof(unsigned)));
    NF_HIP(hipMemcpy(b.d_live_bits, bits.data(), bits.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  }
  size_t wt = 0, biggest = 0;
  int widest = desc.event_dim;
  b.wT_off.clear();
  for (const Linear& L : linears) {
    b.wT_off.push_back(wt);
    wt += ((size_t)L.K * L.N + 3) / 4 * 4;
    biggest = std::max(biggest, (size_t)(L.K + 1) * L.N);
    widest = std::max(widest, std::max(L.K, L.N));
  }
  NF_HIP(hipMalloc(&b.d_wT, wt * sizeof(float)));
  NF_HIP(hipMalloc(&b.d_zero, (size_t)widest * sizeof(float)));
  NF_HIP(hipMemset(b.d_zero, 0, (size_t)widest * sizeof(float)));
  // up to 256 row-chunk slabs of the largest Linear, at most 512 MiB
  b.partial_floats = std::max(biggest, std::min<size_t>(biggest * 256, ((size_t)512 << 20) / sizeof(float)));
  NF_HIP(hipMalloc(&b.d_partials, b.partial_floats * sizeof(float)));
  {  // narrow conditioners: one slab of a layer's weight span per workgroup
    const int nlin = desc.num_hidden + 1;
    size_t span = 0;
    bool any = false;
    for (int l = 0; l < desc.num_layers; ++l) {
      const Linear* Ls = &linears[(size_t)l * nlin];
      if (!mlp_small_bwd_applicable(Ls, nlin, desc.event_dim)) continue;
      any = true;
      span = std::max(span, Ls[nlin - 1].b_off + Ls[nlin - 1].N - Ls[0].w_off);
    }
    if (any) NF_HIP(hipMalloc(&b.d_small_partials, span * mlp_small_bwd_max_blocks() * sizeof(float)));
  }
  std::vector<TransposeJob> jobs(linears.size());
  for (size_t q = 0; q < linears.size(); ++q)
    jobs[q] = TransposeJob{(unsigned long long)linears[q].w_off, (unsigned long long)b.wT_off[q], linears[q].K,
                           linears[q].N};
  hipFree(b.d_jobs);
  b.d_jobs = nullptr;
  NF_HIP(hipMalloc(&b.d_jobs, jobs.size() * sizeof(TransposeJob)));
  NF_HIP(hipMemcpy(b.d_jobs, jobs.data(), jobs.size() * sizeof(TransposeJob), hipMemcpyHostToDevice));
  b.transpose_blocks = grid_for((long long)biggest, 256);
  hipLaunchKernelGGL(transpose_all_kernel, dim3(b.transpose_blocks, (unsigned)linears.size()), dim3(256), 0, 0,
                     d_weights, b.d_wT, static_cast<const TransposeJob*>(b.d_jobs));
  NF_HIP(hipGetLastError());
  NF_HIP(hipDeviceSynchronize());
  b.ready = true;
  return NF_OK;
}

int Plan::refresh_from_canonical(const float* d_canon, hipStream_t stream) {
  Backward& b = bwd;
  hipLaunchKernelGGL(gather_weights_kernel, dim3(grid_for((long long)b.n_weights, 2048)), dim3(256), 0, stream,
                     d_canon, b.d_canon_index, d_weights, (long long)b.n_weights);
  NF_HIP(hipGetLastError());
  return refresh_after_step(stream);
}

int Plan::refresh_after_step(hipStream_t stream) {
  Backward& b = bwd;
  hipLaunchKernelGGL(transpose_all_kernel, dim3(b.transpose_blocks, (unsigned)linears.size()), dim3(256), 0, stream,
                     d_weights, b.d_wT, static_cast<const TransposeJob*>(b.d_jobs));
  NF_HIP(hipGetLastError());
  k5_stale = true;  // the training walk does not use the fused last Linear: re-laid when an evaluation asks for it
  w1c_stale = true; // nor the compact first Linear
  wsmr_stale = true; // nor the small-batch copy of the last Linears
  return NF_OK;
}

int Plan::scatter_weights_to_canonical(float* d_canon, hipStream_t stream) {
  Backward& b = bwd;
  hipLaunchKernelGGL(scatter_weights_kernel, dim3(grid_for((long long)b.n_weights, 2048)), dim3(256), 0, stream,
                     d_weights, b.d_canon_index, d_canon, (long long)b.n_weights);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int Plan::ensure_tape(long long rows) {
  Backward& b = bwd;
  if (rows <= b.tape_rows) return NF_OK;
  const size_t bytes = sizeof(float) * (size_t)desc.num_layers * rows * desc.event_dim;
  if (bytes > ((size_t)64 << 30)) {
    set_error("training batch needs %zu bytes of layer inputs, more than the 64 GiB this build allows", bytes);
    return NF_ERR_UNSUPPORTED;
  }
  size_t widths = 0;
  for (const Linear& L : linears) widths += (size_t)L.N;
  const size_t act_bytes = sizeof(float) * widths * (size_t)rows;
  if (bytes + act_bytes > ((size_t)64 << 30)) {
    set_error("training batch needs %zu bytes of recorded activations, more than the 64 GiB this build allows: "
              "split the batch", bytes + act_bytes);
    return NF_ERR_UNSUPPORTED;
  }
  hipFree(b.d_tape);
  hipFree(b.d_acts);
  b.d_tape = b.d_acts = nullptr;
  b.tape_rows = 0;
  NF_HIP(hipMalloc(&b.d_tape, bytes));
  NF_HIP(hipMalloc(&b.d_acts, act_bytes));
  b.act_off.clear();
  size_t off = 0;
  for (const Linear& L : linears) {
    b.act_off.push_back(off);
    off += (size_t)L.N * rows;
  }
  b.tape_rows = rows;
  return NF_OK;
}

int Plan::backward(const float* v, float* g, long long rows, long long rows_global, float* grad_canon,
                   size_t weight_count, hipStream_t stream) {
  Backward& b = bwd;
  if (!b.ready) { set_error("Plan::prepare_backward has not run"); return NF_ERR_STATE; }
  if (rows > b.tape_rows) { set_error("Plan::backward without a recorded forward walk"); return NF_ERR_STATE; }
  const nf_flow_desc& d = desc;
  const int D = d.event_dim, nh = d.num_hidden, nlin = nh + 1;
  const int act = d.activation == NF_ACT_RELU ? 1 : 2;
  const RqsConst rc = make_rqs_const(d);
  const float inv_n = 1.f / (float)rows_global, gld = -inv_n;
  // scratch per row: d theta and two gradient buffers
  const int gwidth = std::max(act_width, D);
  const size_t per_row = ((size_t)param_width + 2 * (size_t)gwidth) * sizeof(float);
  long long cap = (long long)(((size_t)4 << 30) / per_row);
  cap = std::max<long long>(128, cap / 128 * 128);
  const long long want = std::min(rows, cap);
  if (want > b.scratch_rows) {
    hipFree(b.d_scratch);
    b.d_scratch = nullptr;
    b.scratch_rows = 0;
    NF_HIP(hipMalloc(&b.d_scratch, per_row * (size_t)want));
    b.scratch_rows = want;
  }
  // Small batches: a layer's weight-gradient products (a third of the step at batch 128 of the 784-D flow, and nothing
  // in the chain waits for them) go to a second stream once the layer's data gradients are out; each layer keeps its own
  // d theta and hidden gradients for that.
  const bool narrow_bwd = b.d_small_partials != nullptr && mlp_small_bwd_applicable(&linears[0], nlin, D);
  const bool overlap = rows <= 512 && rows <= want && !narrow_bwd && !switches().no_bwd_overlap;
  const size_t layer_floats = (size_t)param_width + (size_t)nh * gwidth;   // per row
  if (overlap) {
    if (b.side == nullptr) NF_HIP(hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking));
    while ((int)b.ev.size() < d.num_layers + 1) {
      hipEvent_t e;
      NF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      b.ev.push_back(e);
    }
    if (rows > b.layer_grad_rows) {
      NF_HIP(hipStreamSynchronize(b.side));
      hipFree(b.d_layer_grads);
      b.d_layer_grads = nullptr;
      b.layer_grad_rows = 0;
      NF_HIP(hipMalloc(&b.d_layer_grads, layer_floats * (size_t)rows * d.num_layers * sizeof(float)));
      b.layer_grad_rows = rows;
    }
  }
  // one row chunk and every product on a one-chunk route: each gradient entry is written exactly once, so the
  // products store instead of adding and the 186 MB memset goes (the alignment gaps were zeroed when the buffer was made)
  const bool fresh = overlap && tn_one_chunk(rows);
  if (!fresh) NF_HIP(hipMemsetAsync(b.d_grad, 0, b.n_weights * sizeof(float), stream));
  if (grad_canon) NF_HIP(hipMemsetAsync(grad_canon, 0, weight_count * sizeof(float), stream));
  hipLaunchKernelGGL(base_grad_kernel, dim3(grid_for(rows * D)), dim3(256), 0, stream, v, g, rows * D, d.base, inv_n);

  const long long R = b.scratch_rows;
  float* const dth0 = b.d_scratch;                             // [R][param_width]
  float* ga = dth0 + (size_t)R * param_width;                  // [R][gwidth]
  float* gb = ga + (size_t)R * gwidth;
  for (long long r0 = 0; r0 < rows; r0 += R) {
    const long long n = std::min(R, rows - r0);
    float* gc = g + r0 * D;
    for (int l = 0; l < d.num_layers; ++l) {  // the layer log_prob applied last comes first
      const int par = (d.first_mask_parity + l) & 1;
      const Linear* Ls = &linears[(size_t)l * nlin];
      const float* vc = b.d_tape + ((size_t)l * b.tape_rows + r0) * D;  // this layer's input
      // the recorded outputs of this layer's Linears: hidden activations, then theta
      auto recorded = [&](int j) { return b.d_acts + b.act_off[(size_t)l * nlin + j] + (size_t)r0 * Ls[j].N; };
      const float* th = recorded(nlin - 1);
      float* const lg = overlap ? b.d_layer_grads + (size_t)l * layer_floats * b.layer_grad_rows : nullptr;
      float* const dth = overlap ? lg : dth0;
      // 3. transform backward
      if (d.kind == NF_KIND_AFFINE_MASKED) {
        hipLaunchKernelGGL(affine_masked_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, vc, gc, th, dth, n, D,
                           par, gld);
      } else {
        const int Dt = par == 0 ? D / 2 : (D + 1) / 2;
        const int P = 3 * rc.K + 1;
        const int ipp = std::min<int>(256, (int)((size_t)150 * 1024 / ((size_t)2 * P * sizeof(float))));  // K = 32: 197
        const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(4096, (n * Dt + ipp - 1) / ipp)));
        const size_t lds = (size_t)2 * ipp * P * sizeof(float);
        NF_DYN_LDS(160 * 1024, &rqs_masked_bwd_kernel<4>);
        NF_DYN_LDS(160 * 1024, &rqs_masked_bwd_kernel<8>);
        NF_DYN_LDS(160 * 1024, &rqs_masked_bwd_kernel<12>);
        NF_DYN_LDS(160 * 1024, &rqs_masked_bwd_kernel<32>);
        if (rc.K <= 4) hipLaunchKernelGGL((rqs_masked_bwd_kernel<4>), grid, dim3(256), lds, stream, vc, gc, th, dth, n, D, par, gld, ipp, rc);
        else if (rc.K <= 8) hipLaunchKernelGGL((rqs_masked_bwd_kernel<8>), grid, dim3(256), lds, stream, vc, gc, th, dth, n, D, par, gld, ipp, rc);
        else if (rc.K <= 12) hipLaunchKernelGGL((rqs_masked_bwd_kernel<12>), grid, dim3(256), lds, stream, vc, gc, th, dth, n, D, par, gld, ipp, rc);
        else hipLaunchKernelGGL((rqs_masked_bwd_kernel<32>), grid, dim3(256), lds, stream, vc, gc, th, dth, n, D, par, gld, ipp, rc);
      }
      NF_HIP(hipGetLastError());
      // 4. MLP backward
      if (b.d_small_partials != nullptr && mlp_small_bwd_applicable(Ls, nlin, D)) {
        const float* acts[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j + 1 < nlin; ++j) acts[j] = recorded(j);
        const float* wts[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < nlin; ++j) wts[j] = b.d_wT + b.wT_off[(size_t)l * nlin + j];
        NF_TRY(launch_mlp_small_bwd(vc, D, Ls, nlin, d_weights, wts, acts, dth, act, gc, D, b.d_small_partials, b.d_grad,
                                    n, stream, r0 == 0));
        continue;
      }
      const float* gcur = dth;
      const float* gkept[NF_MAX_HIDDEN + 1] = {nullptr};   // overlap: the gradient that entered Linear j
      for (int j = nlin - 1; j >= 0; --j) {
        const Linear& L = Ls[j];
        const float* ain = j == 0 ? vc : recorded(j - 1);
        const int lain = j == 0 ? D : Ls[j - 1].N;
        if (overlap) gkept[j] = gcur;
        else
          NF_TRY(accumulate_tn(ain, lain, gcur, L.N, n, L.K, L.N, b.d_grad + L.w_off, b.d_grad + L.b_off, b.d_partials,
                               b.partial_floats, stream));
        float* gprev = (gcur == ga) ? gb : ga;
        // overlap: the hidden gradients of this layer stay (slot j - 1 behind d theta); the last one, d x, is consumed
        // by add_kernel right away and may share
        if (overlap && j > 0) gprev = lg + (size_t)b.layer_grad_rows * ((size_t)param_width + (size_t)(j - 1) * gwidth);
        if (j == 0 && !switches().gemm_simple && gemm_small_m16_applicable(gcur, L.N, (int)n, L.K, L.N)) {
          // small batches: d x += g W_1^T in the GEMM's own epilogue (no separate add)
          NF_TRY(launch_gemm(gcur, L.N, b.d_wT + b.wT_off[(size_t)l * nlin + j], b.d_zero, gc, D, (int)n, L.K, L.N, 5, stream));
          gcur = gprev;
          continue;
        }
        // j > 0: times act'(a_{j-1}) in the GEMM's epilogue (codes 3 / 4, masked by the sign of `ain`)
        NF_TRY(launch_gemm(gcur, L.N, b.d_wT + b.wT_off[(size_t)l * nlin + j], b.d_zero, gprev, L.K, (int)n, L.K,
                           L.N, j > 0 ? act + 2 : 0, stream, j > 0 ? ain : nullptr));
        if (j == 0) {
          // rows of W_1 that belong to transformed dims are structural zeros: nothing flows there
          hipLaunchKernelGGL(add_kernel, dim3(grid_for(n * D)), dim3(256), 0, stream, gc, gprev, n * D);
        }
        gcur = gprev;
      }
      if (overlap) {
        NF_HIP(hipEventRecord(b.ev[l], stream));
        NF_HIP(hipStreamWaitEvent(b.side, b.ev[l], 0));
        for (int j = nlin - 1; j >= 0; --j) {
          const Linear& L = Ls[j];
          const float* ain = j == 0 ? vc : recorded(j - 1);
          const int lain = j == 0 ? D : Ls[j - 1].N;
          NF_TRY(accumulate_tn(ain, lain, gkept[j], L.N, n, L.K, L.N, b.d_grad + L.w_off, b.d_grad + L.b_off, b.d_partials,
                               b.partial_floats, b.side, fresh));
        }
      }
      NF_HIP(hipGetLastError());
    }
  }
  if (overlap) {   // the gradient is whole once the side stream is through
    NF_HIP(hipEventRecord(b.ev[d.num_layers], b.side));
    NF_HIP(hipStreamWaitEvent(stream, b.ev[d.num_layers], 0));
  }
  if (grad_canon)
    hipLaunchKernelGGL(scatter_grad_kernel, dim3(grid_for((long long)b.n_weights, 2048)), dim3(256), 0, stream, b.d_grad,
                       b.d_canon_index, grad_canon, (long long)b.n_weights);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
