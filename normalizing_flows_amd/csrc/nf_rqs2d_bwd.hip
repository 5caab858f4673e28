// Backward pass of the fused spline-coupling chain (variant C, D = 2, conditioner 2 -> 64 -> 64 -> 2*(3K+1),
// K in {4, 6, 8, 10, 12}): d(NLL)/d(weights), i.e. the jax.grad(loss_fn) inside the reference's `update`
// (src/maf_toy_densities.py:152-176, the layer itself :84-99), SURVEY.md section 8f rank 2.
//
// Modelled on nf_affine2d_bwd.hip: ONE launch walks the chain backwards (layer 0, the one log_prob applied last,
// first); per 32-sample tile and layer a wave recomputes the conditioner and keeps every intermediate in registers or
// in its own LDS tile -- no activation, no [rows, 3K+1] parameter tensor and no parameter gradient touches HBM.  What
// does: the layer's INPUT, 8 B per sample and layer, recorded by the forward kernel (rqs2d::chain_kernel's `tape`).
// An affine layer can be undone exactly enough from its output (nf_affine2d_bwd.hip does); the float32 inverse of a
// spline is off by up to 1e-4 where |dx/dy| is large and the parameter gradients inherit it.
// Before this kernel these flows trained through the general plan, which records every hidden activation and the
// parameter tensor: 15.8 GB of HBM traffic per step at 2^20 rows against 21 MB algorithmic, 8.5 ms
// (profiles/r03/train_c_2p20_pmc_summary.json).
//
// Per tile and layer (lane = sample i of the tile, half-wave h; accumulator register (m, r) of lane (i, h) is hidden
// unit 32 m + 8 (r >> 2) + 4 h + (r & 3) of sample i):
//   1. a2 = W2^T relu(w1 c + b1) + b2                     64 MFMA 32x32x2 (as the forward kernel)
//   2. h2 = relu(a2) -> the wave's transposed LDS tile; theta tile (2K bin-size logits + boundary slope logits)
//      = A3 h2, B operand = the accumulator register by register    32 MFMA
//   3. knots (float64 normalisation, nf_rqs_math.h), bin search, the bin's two slope logits gathered from LDS column
//      pairs -- all as the forward kernel; then the spline's gradient, HAND-DERIVED (no dual numbers: the general
//      plan's kernel carries seven partials through the formula, 8x its flops): with z the position in the bin,
//      u = z (1 - z), delta = height / width, den = delta + (sl + sr - 2 delta) u and
//      inner = sr z^2 + 2 delta u + sl (1 - z)^2,
//        y = yl + height (delta z^2 + sl u) / den,   logdet = 2 log delta + log inner - 2 log den,
//      the partials wrt (z, delta, sl, sr) are closed forms in (z, u, den, inner); chain rule to (x, knot pair per
//      axis, slope logits); softmax + running sum reversed per axis in the half-wave that owns it (half 0 widths,
//      half 1 heights) with sum_{j<bin} softmax_j read off the knot itself.
//   4. dW3 (64 x (3K+1)) += h2 (x) dtheta summed over the tile's samples: v_mfma_f32_4x4x1 with the sample index as
//      the one k step (sample_sums_n): dtheta rows from LDS as the A operand, the transposed tile as B -- lane u of
//      the result is hidden unit u.  db3 = row sums of the same LDS rows.
//   5. g2 = (W3 dtheta) * [a2 > 0]: the 2K bin-size columns as K MFMA k steps whose B operand is the lane's own
//      dtheta registers (width s in half 0, height s in half 1: the image's k order is made for it), the two slope
//      columns of the sample's bin gathered like in the forward walk      2K MFMA + 64 v_fma
//   6. g1 = W2 g2                                          64 MFMA (k in accumulator order)
//   7. first Linear: gz = g1 * [a1 > 0], d loss / d c
//   8. dW2 += h1 (x) g2 with the SAMPLE index on the MFMA's k dimension                      64 MFMA
//      db2, (db1, dW1): sample sums of g2 / gz against the rows (1, c)
// Every wave keeps its partial gradient (dW2 64 + dW3 40 + 8 registers) across all its tiles; workgroups' partials
// are added in a fixed order by gather_kernel, scaled by 1/rows_global and scattered into the canonical layout.
#include "nf_rqs2d_common.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace nf {
namespace rqs2d_bwd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using rqs2d::add_halves;
using rqs2d::lds_f32x16;
using rqs2d::swap_halves;

constexpr int kReluShift = 64;
static inline int unit_of(int m, int r, int h) { return 32 * m + 8 * (r >> 2) + 4 * h + (r & 3); }

bool supported_bins(int K) { return rqs2d::supported_bins(K); }

void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                int cond_dim, int K, float* img) {
  const int P = 3 * K + 1;
  const int dT = 1 - cond_dim;  // the transformed dim
  memset(img, 0, sizeof(float) * kImageFloats);
  // the operands the recompute shares with the forward kernel: taken from ITS image, so the two cannot drift apart
  std::vector<float> fwd(rqs2d::kLayerFloats);
  rqs2d::pack_layer(W1, b1, W2, b2, W3, b3, cond_dim, K, fwd.data());
  memcpy(img + oA2, fwd.data() + rqs2d::kA2, sizeof(float) * 4096);
  memcpy(img + oA3, fwd.data() + rqs2d::kA3, sizeof(float) * 2048);
  memcpy(img + oWS, fwd.data() + rqs2d::kWS, sizeof(float) * 2 * 32 * rqs2d::kSlopePairs * 2);
  memcpy(img + oW1B1, fwd.data() + rqs2d::kW1B1, sizeof(float) * 128);
  memcpy(img + oB2, fwd.data() + rqs2d::kB2, sizeof(float) * 64);
  memcpy(img + oB3, fwd.data() + rqs2d::kB3, sizeof(float) * 32);
  memcpy(img + oBS, fwd.data() + rqs2d::kBS, sizeof(float) * 24);
  // g1 = W2 g2: A[row = unit k of the first hidden layer][k step = unit j in accumulator order]
  for (int m = 0; m < 2; ++m)
    for (int r = 0; r < 16; ++r)
      for (int lane = 0; lane < 64; ++lane)
        for (int mo = 0; mo < 2; ++mo)
          img[oA2T + ((m * 16 + r) * 64 + lane) * 2 + mo] = W2[(32 * mo + (lane & 31)) * 64 + unit_of(m, r, lane >> 5)];
  // g2pre = W3 dtheta: A[row = hidden unit][k step s, k = half: width s | height s]
  for (int s = 0; s < K; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int m = 0; m < 2; ++m) {
        const int u = 32 * m + (lane & 31);
        const int prm = (lane >> 5) == 0 ? s : K + s;
        img[oA3T + (s * 64 + lane) * 2 + m] = W3[(size_t)u * (2 * P) + dT * P + prm];
      }
  for (int h = 0; h < 2; ++h)
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) {
        const int u = unit_of(m, r, h), q = (h * 2 + m) * 16 + r;
        img[oW1acc + q] = W1[cond_dim * 64 + u];
        img[oB1acc + q] = b1[u];
      }
}

// canonical (per layer: W1[2][64], b1[64], W2[64][64], b2[64], W3[64][2P], b3[2P]) -> index into a workgroup's
// partial block (-1: structurally zero), and the factor that undoes the 2^-64 carried by h1 in dW2
void build_gather_map(int cond_dim, int K, int* map, float* scale) {
  const int P = 3 * K + 1, n = canon_floats(K);
  const int dT = 1 - cond_dim;
  for (int e = 0; e < n; ++e) { map[e] = -1; scale[e] = 1.f; }
  const int cW1 = 0, cb1 = 128, cW2 = 192, cb2 = 192 + 4096, cW3 = cb2 + 64, cb3 = cW3 + 64 * 2 * P;
  for (int mk = 0; mk < 2; ++mk)
    for (int mj = 0; mj < 2; ++mj)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int k = unit_of(mk, r, lane >> 5), j = 32 * mj + (lane & 31);
          map[cW2 + k * 64 + j] = pW2 + ((mk * 2 + mj) * 16 + r) * 64 + lane;
          scale[cW2 + k * 64 + j] = ldexpf(1.f, kReluShift);
        }
  for (int u = 0; u < 64; ++u) {
    map[cb2 + u] = pB2 + u;
    map[cb1 + u] = pB1 + u;
    map[cW1 + cond_dim * 64 + u] = pW1 + u;   // the masked input's row has zero gradient
    for (int q = 0; q < P; ++q) map[cW3 + u * 2 * P + dT * P + q] = pW3 + u * 40 + q;  // the pass-through dim's
  }                                                                                    // columns are never read
  for (int q = 0; q < P; ++q) map[cb3 + dT * P + q] = pB3 + q;
}

struct Params {
  const float* img;       // images of the layers, layer 0 first
  const float2* tape;     // [num_layers][rows]: layer l's input
  const float2* z;        // [rows]: the chain's output (the base density's argument)
  float2* G;              // [rows]: d loss_i / d state
  float* partials;        // [num_layers][gridDim.x][kPartialFloats]
  long long rows;
  int num_layers;
  int parity0;            // layer l conditions on dim (parity0 + l) & 1
  int base;               // nf_base: standard normal -> G0 = z; uniform -> G0 = 0
  float range_min, range_max, min_bin, min_slope, slope_offset;
};

// written by this launch (G, by the previous layer): read past the CU's vector cache
__device__ __forceinline__ float2 load_state(const float2* q) {
  const unsigned long long u =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ float half_sum(float x) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

// Development: -DNF_RQSBWD_SKIP=<mask> builds a TIMING-ONLY kernel without some of its phases (wrong results, same remaining
// instruction stream): 1 recompute of a2, 2 the parameter tile, 4 knots + bin + slope gather + the spline's gradient,
// 8 the dW3 / db3 sums, 16 g2 = W3 dtheta (MFMAs + gather + mask), 32 g1 = W2 g2, 64 the first Linear, 128 dW2 + db2,
// 256 the (db1, dW1) sums.  profiles/scratch/rqsbwd_phase_timing.sh; prices at 2^20 rows in EXPERIMENTS.md (round 4).
#ifndef NF_RQSBWD_SKIP
#define NF_RQSBWD_SKIP 0
#endif
constexpr int kWaves = 8;                // one workgroup per CU, two waves per SIMD
constexpr int kGT = 32 * 65;             // one transposed tile [sample][unit], padded rows
constexpr int kAopRow = 36;              // floats between factor rows (16-byte reads of four rows hit different banks)
constexpr int kRowsAB = 24;              // dtheta rows, two batches through the same LDS: the 2K bin-size logits' (batch
                                         // A), then the K+1 slope logits' padded to 16 (batch B)
constexpr int kRows = kRowsAB + 4;       // + the constant group (1, c, 0, 0)
constexpr int kWaveScratch = kGT + kRows * kAopRow;  // 3,088 floats = 12.4 KB per wave

// D[g][r] of lane u += sum over the tile's 32 samples of row (4 g + r)[sample] * X[u][sample], NG groups of four
// factor rows at once (one read of X per sample serves them all), on v_mfma_f32_4x4x1 -- see nf_affine2d_bwd.hip's
// sample_sums and profiles/microbench/mfma_f32_4x4x1_layout.hip for the layout.
template <int NG>
__device__ __forceinline__ void sample_sums_n(f32x4 (&D)[NG], const float* __restrict__ frow,
                                              const float* __restrict__ xt, int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float4 a[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) a[g] = *reinterpret_cast<const float4*>(frow + g * 4 * kAopRow + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = xt[(4 * q + e) * 65 + lane];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float av = e == 0 ? a[g].x : (e == 1 ? a[g].y : (e == 2 ? a[g].z : a[g].w));
        D[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x, D[g], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ float sigmoid_f(float v) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504f));
}

template <int K>
__global__ __launch_bounds__(kWaves * 64) void chain_kernel(Params p) {
  // two LDS objects (see nf_affine2d_bwd.hip): the image, read-only while tiles are walked, and the waves' scratch
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float scratch[kWaves * kWaveScratch];
  constexpr int NGA = (2 * K) / 4;           // groups of four rows in batch A (K even)
  constexpr int NGB = (K + 1 + 3) / 4;       // ... in batch B
  static_assert((2 * K) % 4 == 0 && 2 * K <= kRowsAB && 4 * NGB <= kRowsAB, "row batches");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  float* gT = scratch + wave * kWaveScratch;
  float* aop = gT + kGT;
  float* crow = aop + (kRowsAB + 1) * kAopRow;              // row "c" of the constant group
  const float* f_ab = aop + (lane & 3) * kAopRow;           // A operands: rows of batch A / B, group g at + g*4*kAopRow
  const float* f_1c = aop + (kRowsAB + (lane & 3)) * kAopRow;  // (1, c, 0, 0)
  float* gTw = gT + i * 65 + 4 * h;                         // this lane's 32 units of sample i: + 32 m + 8 (r >> 2) + (r & 3)
  const float* L = lds;
  const float2* a2 = reinterpret_cast<const float2*>(L + oA2) + lane;
  const float2* a2t = reinterpret_cast<const float2*>(L + oA2T) + lane;
  const float4* a3 = reinterpret_cast<const float4*>(L + oA3) + lane;
  const float2* a3t = reinterpret_cast<const float2*>(L + oA3T) + lane;
  const float2* w1b1 = reinterpret_cast<const float2*>(L + oW1B1);
  const int hq = h * 32;

  const long long rows = p.rows;
  const long long tiles = (rows + 31) / 32;
  const long long tstride = (long long)gridDim.x * kWaves;
  const float cs_inv = 1.f / ((p.range_max - p.range_min) - (float)K * p.min_bin);

  for (int layer = 0; layer < p.num_layers; ++layer) {
    const int cond_dim = (p.parity0 + layer) & 1;
    const bool first = layer == 0;
    __syncthreads();  // every wave is done with the previous layer's image and with the scratch
    {
      const float4* src = reinterpret_cast<const float4*>(p.img + (size_t)layer * kImageFloats);
      float4* dst = reinterpret_cast<float4*>(lds);
      for (int q = tid; q < kImageFloats / 4; q += kWaves * 64) dst[q] = src[q];
    }
    if (lane < 32) {
      aop[(kRowsAB + 0) * kAopRow + lane] = 1.f;
      aop[(kRowsAB + 2) * kAopRow + lane] = 0.f;
      aop[(kRowsAB + 3) * kAopRow + lane] = 0.f;
    }
    __syncthreads();

    f32x16 dW2[2][2];
    f32x4 D3a[NGA], D3b[NGB];             // lane u: dW3[u][bin-size logit 4g + r], dW3[u][slope logit 4g + r]
    f32x4 Dy = {0.f, 0.f, 0.f, 0.f};      // lane u: (db2[u], ., ., .)
    f32x4 Dz = {0.f, 0.f, 0.f, 0.f};      // lane u: (db1[u], dW1[u], ., .)
    float db3a = 0.f, db3b = 0.f;         // lane q: db3 of bin-size logit q (q < 2K) / slope logit q (q <= K)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dW2[a][0][r] = 0.f; dW2[a][1][r] = 0.f; }
#pragma unroll
    for (int g = 0; g < NGA; ++g) D3a[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NGB; ++g) D3b[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float2* xsrc = p.tape + (size_t)layer * rows;
    const float2* Gsrc = first ? p.z : p.G;
    long long tile = (long long)blockIdx.x * kWaves + wave;
    float2 v = make_float2(0.f, 0.f), G = make_float2(0.f, 0.f);
    if (tile < tiles && tile * 32 + i < rows) { v = xsrc[tile * 32 + i]; G = load_state(Gsrc + tile * 32 + i); }
    asm volatile("; tile state loaded" : "+v"(v.x), "+v"(v.y), "+v"(G.x), "+v"(G.y));
    for (; tile < tiles; tile += tstride) {
      const long long idx = tile * 32 + i;
      const bool valid = idx < rows;
      float2 v_next = make_float2(0.f, 0.f), G_next = make_float2(0.f, 0.f);
      {
        const long long nidx = (tile + tstride) * 32 + i;
        if (tile + tstride < tiles && nidx < rows) { v_next = xsrc[nidx]; G_next = load_state(Gsrc + nidx); }
      }
      if (first && p.base != NF_BASE_STD_NORMAL) G = make_float2(0.f, 0.f);  // Uniform base: zero gradient inside the cube
      const float c = cond_dim ? v.y : v.x;        // the conditioning (pass-through) value
      const float t = cond_dim ? v.x : v.y;        // the transformed dim's input
      float Gc = cond_dim ? G.y : G.x;
      const float gy = cond_dim ? G.x : G.y;       // d loss / d y of the transformed dim

      // 1. a2 = W2^T relu(w1 c + b1) + b2
      f32x16 acc[2];
      {
        const f32x16 bias0 = lds_f32x16(L + oB2 + hq);
        const f32x16 bias1 = lds_f32x16(L + oB2 + hq + 16);
        const float2 a = a2[0];
        const float2 wb = w1b1[h];
        const float b = __builtin_amdgcn_fmed3f(fmaf(c, wb.x, wb.y), 0.f, 1.f);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, bias0, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, bias1, 0, 0, 0);
      }
#pragma unroll 8
      for (int s = 1; s < ((NF_RQSBWD_SKIP & 1) ? 1 : 32); ++s) {
        const float2 a = a2[s * 64];
        const float2 wb = w1b1[2 * s + h];
        const float b = __builtin_amdgcn_fmed3f(fmaf(c, wb.x, wb.y), 0.f, 1.f);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
      }

      // 2. h2 = relu(a2): into the transposed tile (B operand of the dW3 sums, and the relu mask of step 5) and,
      //    register by register, the B operand of the parameter tile
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[m][r] = fmaxf(acc[m][r], 0.f);
          gTw[32 * m + 8 * (r >> 2) + (r & 3)] = acc[m][r];
        }
      f32x16 out;
      {
        const f32x16 pb = lds_f32x16(L + oB3 + h * 16);
#pragma unroll
        for (int g = 0; g < ((NF_RQSBWD_SKIP & 2) ? 1 : 8); ++g) {
          const float4 a = a3[g * 64];
          const int m = g >> 2, r = (g & 3) * 4;
          out = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[m][r + 0], g == 0 ? pb : out, 0, 0, 0);
          out = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[m][r + 1], out, 0, 0, 0);
          out = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc[m][r + 2], out, 0, 0, 0);
          out = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc[m][r + 3], out, 0, 0, 0);
        }
      }

#if (NF_RQSBWD_SKIP & 4)
      const int kb = lane & 7;
      const float2* ws = reinterpret_cast<const float2*>(L + oWS) + h * (32 * rqs2d::kSlopePairs) + kb;
      float d_l = out[0] * gy, d_r = out[1];
      float dlg[K];
#pragma unroll
      for (int k = 0; k < K; ++k) dlg[k] = out[k] + t;
      const float Gt = gy;
      (void)cs_inv; (void)valid;
#else
      // 3. knots, bin, the bin's slope logits -- the forward kernel's own steps (nf_rqs2d_h64.hip)
      float lg[K], knot[K + 1], size[K];
#pragma unroll
      for (int k = 0; k < K; ++k) lg[k] = out[k];
      rqsmath::knots_f64<K, NF_RQS_XEXP>(lg, p, knot, size);
      float xl, bw, yl, bh;
      const int kb = rqs2d::spline_bin<K, 0>(knot, size, t, h, &xl, &bw, &yl, &bh);
      const float2* ws = reinterpret_cast<const float2*>(L + oWS) + h * (32 * rqs2d::kSlopePairs) + kb;
      float raw_l, raw_r;
      {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float2 w = ws[(m * 16 + r) * rqs2d::kSlopePairs];
            d0 = fmaf(acc[m][r], w.x, d0);
            d1 = fmaf(acc[m][r], w.y, d1);
          }
        const float2 bs = reinterpret_cast<const float2*>(L + oBS)[kb];
        raw_l = add_halves(d0) + bs.x;
        raw_r = add_halves(d1) + bs.y;
      }

      //    the spline's gradient.  Per sample: loss = gy * y - logdet (+ what later layers and the base add through gy)
      const bool below = t <= p.range_min, above = t >= p.range_max;
      const bool tail = below || above;
      const float gl = valid ? -1.f : 0.f;          // d loss / d logdet; rows past the batch contribute nothing
      float g_x, g_kl, g_kr, d_l, d_r;              // wrt x, this half's knot pair (left, right), the two slope logits
      {
        const float vl = raw_l + p.slope_offset, vr = raw_r + p.slope_offset;
        const float sl = rqsmath::softplus_f(vl) + p.min_slope, sr = rqsmath::softplus_f(vr) + p.min_slope;
        const float rbw = rqsmath::rcp_nr(bw);
        const float delta = bh * rbw;
        float z = (t - xl) * rbw;
        z = fminf(fmaxf(z, 0.f), 1.f);
        const float q = z * z, u = z - q, omz = 1.f - z, omz2 = omz * omz, tz = 1.f - 2.f * z, tu = 1.f - 2.f * u;
        const float st = sr + sl - 2.f * delta;
        const float A = fmaf(delta, q, sl * u);
        const float den = fmaf(st, u, delta);
        const float inner = fmaf(sr, q, fmaf(2.f * delta, u, sl * omz2));
        const float rden = rqsmath::rcp_nr(den), rin = rqsmath::rcp_nr(inner), rdel = rqsmath::rcp_nr(delta);
        const float hr2 = bh * rden * rden;          // height / den^2
        // partials of y (times gy) and of logdet (times gl)
        const float y_z = hr2 * delta * inner;
        const float l_z = 2.f * (fmaf(sr, z, fmaf(delta, tz, -sl * omz)) * rin - st * tz * rden);
        const float y_d = hr2 * fmaf(q, den, -A * tu);
        const float l_d = 2.f * (rdel + u * rin - tu * rden);
        const float y_sl = hr2 * u * (den - A), l_sl = fmaf(omz2, rin, -2.f * u * rden);
        const float y_sr = -hr2 * A * u, l_sr = fmaf(q, rin, -2.f * u * rden);
        const float t_z = fmaf(gy, y_z, gl * l_z), t_d = fmaf(gy, y_d, gl * l_d);
        float t_sl = fmaf(gy, y_sl, gl * l_sl), t_sr = fmaf(gy, y_sr, gl * l_sr);
        g_x = t_z * rbw;
        // width axis: (xl, w) -> knots (left | right); height axis: (yl, height)
        const float t_w = -(fmaf(t_z, z, t_d * delta)) * rbw;      // d / d width at fixed xl
        const float t_hh = fmaf(gy, A * rden, t_d * rbw);           // d / d height at fixed yl
        float kl = h == 0 ? (-g_x - t_w) : (gy - t_hh);
        float kr = h == 0 ? t_w : t_hh;
        if (tail) {
          // linear tails: y = (x - edge) s + edge, logdet = log s, s = the boundary slope; no knot takes part
          const float vb = (below ? out[rqs2d::kSlotSlopeFirst] : out[rqs2d::kSlotSlopeLast]) + p.slope_offset;
          const float s = rqsmath::softplus_f(vb) + p.min_slope;
          const float t_s = fmaf(gy, t - (below ? p.range_min : p.range_max), gl * rqsmath::rcp_nr(s)) * sigmoid_f(vb);
          g_x = gy * s;
          kl = 0.f; kr = 0.f;
          // the boundary logits are slope 0 (= pair 0's first) and slope K (= pair K-1's second); spline_bin put
          // kb = 0 below the range and K - 1 above it
          d_l = below ? t_s : 0.f;
          d_r = below ? 0.f : t_s;
        } else {
          d_l = t_sl * sigmoid_f(vl);
          d_r = t_sr * sigmoid_f(vr);
        }
        g_kl = kb >= 1 ? kl : 0.f;        // knot 0 and knot K are the range's ends: constants
        g_kr = kb + 1 <= K - 1 ? kr : 0.f;
      }
      //    this half's axis: bin sizes = softmax * cs + min_bin, knot k = range_min + sum_{j<k} size_j; the softmax's
      //    Jacobian needs sum_j g_j softmax_j = g_kl S(<kb) + g_kr S(<=kb), and S(<kb) = (knot_kb - range_min - kb
      //    min_bin) / cs can be read off the knot
      float dlg[K];
      {
        const float lo = h == 0 ? xl : yl, sz = h == 0 ? bw : bh;
        const float s_lt = (lo - p.range_min - (float)kb * p.min_bin) * cs_inv;
        const float s_le = s_lt + (sz - p.min_bin) * cs_inv;
        const float dot = fmaf(g_kl, s_lt, g_kr * s_le);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float gk = (k < kb ? g_kl : 0.f) + (k <= kb ? g_kr : 0.f);
          dlg[k] = (size[k] - p.min_bin) * (gk - dot);
        }
      }
      //    d loss / d c through the first Linear comes later; the transformed dim's gradient is done
      const float Gt = g_x;
#endif

      // 4. dW3 and db3.  Batch A: the 2K bin-size logits' rows (half 0 writes widths, half 1 heights)
#pragma unroll
      for (int k = 0; k < K; ++k) aop[(h * K + k) * kAopRow + i] = dlg[k];
      if (h == 0) crow[i] = c;
      if ((NF_RQSBWD_SKIP & 8) == 0) sample_sums_n<NGA>(D3a, f_ab, gT, lane);
      if ((NF_RQSBWD_SKIP & 8) == 0 && lane < 2 * K) {
        const float4* r4 = reinterpret_cast<const float4*>(aop + lane * kAopRow);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 a = r4[q]; s += (a.x + a.y) + (a.z + a.w); }
        db3a += s;
      }
      //    batch B: the K+1 slope logits' rows, nonzero in the bin's two (over batch A: a wave's LDS operations
      //    execute in order); half 0 writes rows 0 .. 7, half 1 rows 8 .. 15
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int k = h * 8 + k8;
        if (k < 4 * NGB) aop[k * kAopRow + i] = k == kb ? d_l : (k == kb + 1 ? d_r : 0.f);
      }
      if ((NF_RQSBWD_SKIP & 8) == 0) sample_sums_n<NGB>(D3b, f_ab, gT, lane);
      if ((NF_RQSBWD_SKIP & 8) == 0 && lane <= K) {
        const float4* r4 = reinterpret_cast<const float4*>(aop + lane * kAopRow);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 a = r4[q]; s += (a.x + a.y) + (a.z + a.w); }
        db3b += s;
      }

      // 5. g2 = (W3 dtheta) * [a2 > 0]
      f32x16 g2[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { g2[0][r] = 0.f; g2[1][r] = 0.f; }
#pragma unroll
      for (int s = 0; s < ((NF_RQSBWD_SKIP & 16) ? 1 : K); ++s) {
        const float2 a = a3t[s * 64];
        g2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, dlg[s], g2[0], 0, 0, 0);
        g2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, dlg[s], g2[1], 0, 0, 0);
      }
      if ((NF_RQSBWD_SKIP & 16) == 0)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 w = ws[(m * 16 + r) * rqs2d::kSlopePairs];
          const float hv = gTw[32 * m + 8 * (r >> 2) + (r & 3)];   // this lane's own h2 (the tile still holds it)
          float gfull = fmaf(w.x, d_l, fmaf(w.y, d_r, g2[m][r]));
          asm volatile("" : "+v"(gfull));  // keep the select a select (see nf_affine2d_bwd.hip)
          g2[m][r] = hv > 0.f ? gfull : 0.f;
        }

      // 6. g1 = W2 g2 (k steps in accumulator order)
      f32x16 g1[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { g1[0][r] = 0.f; g1[1][r] = 0.f; }
      if ((NF_RQSBWD_SKIP & 32) == 0)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 a = a2t[(m * 16 + r) * 64];
          g1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, g2[m][r], g1[0], 0, 0, 0);
          g1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, g2[m][r], g1[1], 0, 0, 0);
        }
      //    g2 transposed into LDS, over h2
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) gTw[32 * m + 8 * (r >> 2) + (r & 3)] = g2[m][r];

      // 7. the first Linear: gz = g1 * [w1 c + b1 > 0] -> d loss / d c
      float gcm = 0.f;
      if ((NF_RQSBWD_SKIP & 64) == 0)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float w1a = L[oW1acc + hq + m * 16 + r], b1a = L[oB1acc + hq + m * 16 + r];
          const float gz = fmaf(w1a, c, b1a) > 0.f ? g1[m][r] : 0.f;
          gcm = fmaf(gz, w1a, gcm);
          g1[m][r] = gz;
        }
      Gc += add_halves(gcm);

      // 8. dW2[k][j] += sum_i h1[k][i] g2[j][i]: samples on the MFMA's K dimension; between them db2
#pragma unroll 4
      for (int s = 0; s < ((NF_RQSBWD_SKIP & 128) ? 1 : 16); ++s) {
        const float ci = crow[2 * s + h];
        const float2 wb0 = w1b1[i], wb1 = w1b1[32 + i];
        const float a0 = __builtin_amdgcn_fmed3f(fmaf(ci, wb0.x, wb0.y), 0.f, 1.f);
        const float a1 = __builtin_amdgcn_fmed3f(fmaf(ci, wb1.x, wb1.y), 0.f, 1.f);
        const float b0 = gT[(2 * s + h) * 65 + i], b1v = gT[(2 * s + h) * 65 + 32 + i];
        dW2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, dW2[0][0], 0, 0, 0);
        Dy = __builtin_amdgcn_mfma_f32_4x4x1f32(f_1c[2 * s], gT[(2 * s) * 65 + lane], Dy, 0, 0, 0);
        dW2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1v, dW2[0][1], 0, 0, 0);
        dW2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, dW2[1][0], 0, 0, 0);
        Dy = __builtin_amdgcn_mfma_f32_4x4x1f32(f_1c[2 * s + 1], gT[(2 * s + 1) * 65 + lane], Dy, 0, 0, 0);
        dW2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1v, dW2[1][1], 0, 0, 0);
      }
      //    (db1, dW1)[u] += sum_i (1, c[i]) gz[u][i]  (gz over g2 in the tile)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) gTw[32 * m + 8 * (r >> 2) + (r & 3)] = g1[m][r];
      if ((NF_RQSBWD_SKIP & 256) == 0) {
        f32x4 one[1] = {Dz};
        sample_sums_n<1>(one, f_1c, gT, lane);
        Dz = one[0];
      }

      const float2 G_out = cond_dim ? make_float2(Gt, Gc) : make_float2(Gc, Gt);
      v = v_next;
      G = G_next;
      asm volatile("; next tile's state arrived" : "+v"(v.x), "+v"(v.y), "+v"(G.x), "+v"(G.y));
      if (h == 0 && valid) p.G[idx] = G_out;
    }

    // one partial gradient per workgroup: the waves lay theirs out in LDS (all of it dead by now), FOUR at a time --
    // three in the scratch, the fourth over the layer's image -- and they are added in wave order (two rounds; two
    // partials per round cost four rounds of ~3 us per layer)
    static_assert(3 * kPartialFloats <= kWaves * kWaveScratch && kPartialFloats <= kImageFloats, "staging of four partials");
    constexpr int kPer = (kPartialFloats + kWaves * 64 - 1) / (kWaves * 64);
    float total[kPer];
#pragma unroll
    for (int e = 0; e < kPer; ++e) total[e] = 0.f;
#pragma unroll
    for (int round = 0; round < kWaves / 4; ++round) {
      __syncthreads();
      if ((wave >> 2) == round) {
        float* P = (wave & 3) == 3 ? lds : scratch + (wave & 3) * kPartialFloats;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[pW2 + ((a * 2 + b) * 16 + r) * 64 + lane] = dW2[a][b][r];
        P[pB2 + lane] = Dy[0];
        P[pB1 + lane] = Dz[0];
        P[pW1 + lane] = Dz[1];
        // dW3 of unit `lane`: canonical column order of the transformed dim = widths | heights | slopes
#pragma unroll
        for (int g = 0; g < NGA; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[pW3 + lane * 40 + 4 * g + r] = D3a[g][r];
#pragma unroll
        for (int g = 0; g < NGB; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * g + r <= K) P[pW3 + lane * 40 + 2 * K + 4 * g + r] = D3b[g][r];
        if (lane < 2 * K) P[pB3 + lane] = db3a;
        if (lane <= K) P[pB3 + 2 * K + lane] = db3b;
        if (lane < 40 - (3 * K + 1)) {             // the unused tail of the 40-wide rows
          P[pB3 + 3 * K + 1 + lane] = 0.f;
        }
#pragma unroll
        for (int q = 3 * K + 1; q < 40; ++q) P[pW3 + lane * 40 + q] = 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < kPer; ++e) {
        const int q = tid + e * kWaves * 64;
        if (q < kPartialFloats)
          total[e] = (((total[e] + scratch[q]) + scratch[kPartialFloats + q]) + scratch[2 * kPartialFloats + q]) + lds[q];
      }
    }
    float* outp = p.partials + ((size_t)layer * gridDim.x + blockIdx.x) * kPartialFloats;
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const int q = tid + e * kWaves * 64;
      if (q < kPartialFloats) outp[q] = total[e];
    }
  }  // layer
}

// grad[e] = scale[e] / rows_global * sum over workgroups of partial[b][map[e]] (-1: 0).  Fixed order: chunk c adds
// workgroups c, c+4, c+8, ... (Kahan), then the four chunks are added in order (as nf_affine2d_bwd.hip's).
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ partials_all, int nparts, int canon,
                                                     const int* __restrict__ map0, const int* __restrict__ map1,
                                                     const float* __restrict__ scale0, const float* __restrict__ scale1,
                                                     int parity0, float* __restrict__ grad_all, float inv_rows) {
  const int layer = blockIdx.y;
  const float* __restrict__ partials = partials_all + (size_t)layer * nparts * kPartialFloats;
  const int cond = (parity0 + layer) & 1;
  const int* __restrict__ map = cond ? map1 : map0;
  const float* __restrict__ scale = cond ? scale1 : scale0;
  float* __restrict__ grad = grad_all + (size_t)layer * canon;
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int src = e < canon ? map[e] : -1;
  float s = 0.f, c = 0.f;
  if (src >= 0) {
    for (int w0 = chunk; w0 < nparts; w0 += 32) {
      float y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int w = w0 + 4 * u;
        y[u] = w < nparts ? partials[(size_t)w * kPartialFloats + src] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float yy = y[u] - c;
        const float t = s + yy;
        c = (t - s) - yy;
        s = t;
      }
    }
  }
  red[chunk][el] = s;
  __syncthreads();
  if (chunk == 0 && e < canon)
    grad[e] = src < 0 ? 0.f : (((red[0][el] + red[1][el]) + red[2][el]) + red[3][el]) * scale[e] * inv_rows;
}

int grid_blocks() { return 256; }

template <int K>
static int launch_k(const Params& p, int grid, hipStream_t stream) {
  const size_t lds_bytes = (size_t)kImageFloats * sizeof(float);   // + the kernel's static scratch
  // (the attribute bounds dynamic + static: 54.8 KB of image beside 98.8 KB of scratch is what 160 KB holds)
  NF_DYN_LDS((int)lds_bytes, &chain_kernel<K>);
  hipLaunchKernelGGL(chain_kernel<K>, dim3(grid), dim3(kWaves * 64), lds_bytes, stream, p);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int launch_chain(const float* images, const nf_flow_desc& d, const float* tape, const float* z, float* G,
                 float* partials, long long rows, hipStream_t stream, int* out_parts) {
  Params p;
  p.img = images;
  p.tape = reinterpret_cast<const float2*>(tape);
  p.z = reinterpret_cast<const float2*>(z);
  p.G = reinterpret_cast<float2*>(G);
  p.partials = partials;
  p.rows = rows;
  p.num_layers = d.num_layers;
  p.parity0 = d.first_mask_parity & 1;
  p.base = d.base;
  p.range_min = d.rqs_range_min;
  p.range_max = d.rqs_range_max;
  p.min_bin = d.rqs_min_bin_size;
  p.min_slope = d.rqs_min_knot_slope;
  p.slope_offset = logf(expf(1.f - d.rqs_min_knot_slope) - 1.f);
  const long long tiles = (rows + 31) / 32;
  // few tiles: one per workgroup before any workgroup gets two (every workgroup stages the layer images itself, in
  // parallel: at the reference's batch of 128 four CUs walk one tile each instead of one CU walking four)
  const int grid = (int)std::min<long long>(grid_blocks(), std::max<long long>(1, tiles <= grid_blocks() ? tiles : (tiles + kWaves - 1) / kWaves));
  if (out_parts) *out_parts = grid;
  switch (d.rqs_bins) {
    case 12: return launch_k<12>(p, grid, stream);
    case 10: return launch_k<10>(p, grid, stream);
    case 8: return launch_k<8>(p, grid, stream);
    case 6: return launch_k<6>(p, grid, stream);
    case 4: return launch_k<4>(p, grid, stream);
    default: set_error("rqs2d backward is built for 4, 6, 8, 10 and 12 bins"); return NF_ERR_UNSUPPORTED;
  }
}

int launch_gather(const float* partials, int nparts, const nf_flow_desc& d, const int* map0, const int* map1,
                  const float* scale0, const float* scale1, float* grad, long long rows_global, hipStream_t stream) {
  const int canon = canon_floats(d.rqs_bins);
  hipLaunchKernelGGL(gather_kernel, dim3((canon + 63) / 64, d.num_layers), dim3(256), 0, stream, partials, nparts, canon,
                     map0, map1, scale0, scale1, d.first_mask_parity & 1, grad, 1.0f / (float)rows_global);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace rqs2d_bwd
}  // namespace nf
