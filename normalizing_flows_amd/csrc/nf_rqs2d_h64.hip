// Fused chain of masked rational-quadratic-spline coupling layers, event_dim = 2,
// conditioner 2 -> 64 -> 64 -> 2*(3K+1) with relu (variant C of the reference:
// src/maf_toy_densities.py:55-125; distrax.RationalQuadraticSpline semantics as
// restated in SURVEY.md section 8a row c2: boundary_slopes='unconstrained',
// min_bin_size, min_knot_slope, linear tails).
//
// Same skeleton as nf_affine2d_h64.hip (first Linear generated in the MFMA B-operand
// layout, 64x64 Linear on v_mfma_f32_32x32x2_f32, accumulator = next operand), plus:
//   * the last Linear (64 -> 3K+1 parameters of the ONE transformed dim; the masked dim's
//     parameters are never read by MaskedCoupling and are not computed) is split by use:
//     the 2K bin-size logits + the two boundary slope logits are ONE more 32-row MFMA tile whose
//     B operand is relu(accumulator) taken register by register (weight image k-permuted to the
//     accumulator's order -- no lane movement, no LDS); of the K+1 slope logits a sample only
//     ever needs the two of its bin, so those two dot products are gathered after the bin search
//     (per lane: 32 LDS reads of a column pair + 64 v_fma on its 32 hidden units, one
//     v_permlane32_swap add).  The first build computed all 3K+1 (padded to 64 rows, K+1 slopes
//     in both half-waves): 128 MFMAs per tile and layer, 20 % of them on zero rows; now 96.
//   * half-wave 0 receives the K width logits of its sample, half-wave 1 the K height logits:
//     each half builds its own knots (float64 normalisation, nf_rqs2d_common.h), the half that
//     owns the search axis finds the bin, three v_permlane32_swap carry (bin index, knot pair)
//     across, both halves finish with identical bits;
//   * a layer's weight image is 28,864 B, so layers are streamed through a double buffer in
//     LDS: the workgroup walks the chain layer by layer with one barrier per layer while the
//     next layer's image is fetched (the flow state of every sample stays in registers).
#include "nf_rqs2d_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace nf {
namespace rqs2d {

constexpr int kReluShift = 64;  // see nf_affine2d_h64.hip

// row of the 32-row parameter tile that lands in half-wave `half`, accumulator slot q: the parameter it holds
static inline int tile_param(int half, int q, int K) {
  if (q < K) return half == 0 ? q : K + q;   // width / height logit q
  if (q == kSlotSlopeFirst) return 2 * K;     // slope logit 0 (lower tail)
  if (q == kSlotSlopeLast) return 3 * K;      // slope logit K (upper tail)
  return -1;
}

void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, int K, float* img) {
  const int P = 3 * K + 1;
  const int dT = 1 - cond_dim;  // the transformed dim
  memset(img, 0, sizeof(float) * kLayerFloats);
  for (int s = 0; s < 32; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int m = 0; m < 2; ++m) {
        const int k = 2 * s + (lane >> 5);
        const int u = 32 * m + (lane & 31);
        img[kA2 + (s * 64 + lane) * 2 + m] = ldexpf(W2[k * 64 + u], kReluShift);
      }
  // second pass: k-step s = (m, r) of the accumulator, lane (u, hk): hidden unit j of that step; output row u
  // of the tile lands in half hD, slot rr
  for (int m = 0; m < 2; ++m)
    for (int r = 0; r < 16; ++r)
      for (int lane = 0; lane < 64; ++lane) {
        const int hk = lane >> 5, u = lane & 31;
        const int j = 32 * m + 8 * (r >> 2) + 4 * hk + (r & 3);
        const int hD = (u >> 2) & 1, rr = (u & 3) + 4 * (u >> 3);
        const int prm = tile_param(hD, rr, K);
        const int s = m * 16 + r;
        img[kA3 + ((s >> 2) * 64 + lane) * 4 + (s & 3)] = prm >= 0 ? W3[(size_t)j * (2 * P) + dT * P + prm] : 0.f;
      }
  // slope columns as pairs (k, k+1), hidden units in the order a lane of half h holds them (slot = m*16 + r)
  for (int h = 0; h < 2; ++h)
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * m + 8 * (r >> 2) + 4 * h + (r & 3);
        for (int c = 0; c < K; ++c)
          for (int e = 0; e < 2; ++e)
            img[kWS + ((h * 32 + m * 16 + r) * kSlopePairs + c) * 2 + e] = W3[(size_t)j * (2 * P) + dT * P + 2 * K + c + e];
      }
  for (int c = 0; c < K; ++c)
    for (int e = 0; e < 2; ++e) img[kBS + 2 * c + e] = b3[dT * P + 2 * K + c + e];
  for (int k = 0; k < 64; ++k) {
    img[kW1B1 + 2 * k + 0] = ldexpf(W1[cond_dim * 64 + k], -kReluShift);
    img[kW1B1 + 2 * k + 1] = ldexpf(b1[k], -kReluShift);
  }
  for (int h = 0; h < 2; ++h) {
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) img[kB2 + (h * 2 + m) * 16 + r] = b2[32 * m + 8 * (r >> 2) + 4 * h + (r & 3)];
    for (int q = 0; q < 16; ++q) {
      const int prm = tile_param(h, q, K);
      img[kB3 + h * 16 + q] = prm >= 0 ? b3[dT * P + prm] : 0.f;
    }
  }
}

template <int T, int WAVES, int INV, int K>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void chain_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // two layer images
  const ChainIO& io = p.io;
  const int tid = threadIdx.x;
  const int nlay = io.layer_end - io.layer_begin;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int F4 = kLayerFloats / 4;               // float4s per image
  constexpr int PER = (F4 + NTHREADS - 1) / NTHREADS;  // float4s per thread per image

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long rows = io.rows;
  const long long groups = (rows + (T * 32 - 1)) / (T * 32);
  // Passes (same count for every block): full ones -- every wave of every block has a group -- then one that deals
  // the remaining groups evenly: aw_last live waves per block (waves 0 .. aw_last-1: one per SIMD first), so every
  // SIMD of the chip ends up with the same number of groups whenever 1024 divides them.  A batch smaller than one
  // full pass is the last pass alone: one group per workgroup before any workgroup gets two.
  const long long groups_per_pass = (long long)gridDim.x * WAVES;
  const long long full_passes = groups / groups_per_pass;
  const long long groups_left = groups - full_passes * groups_per_pass;
  const int aw_last = (int)((groups_left + gridDim.x - 1) / gridDim.x);
  const long long passes = full_passes + (groups_left > 0 ? 1 : 0);
  const float2* __restrict__ xin = reinterpret_cast<const float2*>(io.x);

  auto layer_of = [&](int li) { return io.descending ? io.layer_end - 1 - li : io.layer_begin + li; };
  // image staging: every thread carries PER (<= 5) float4 of the next layer's image in named
  // registers while the current layer computes (reads are clamped, writes guarded)
  static_assert(PER <= 5, "staging registers");
  float4 st0, st1, st2, st3, st4;
  st0 = st1 = st2 = st3 = st4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define NF_FETCH(li_)                                                                                  \
  {                                                                                                   \
    const float4* src_ = reinterpret_cast<const float4*>(p.wimg + (size_t)layer_of(li_) * kLayerFloats); \
    st0 = src_[tid < F4 ? tid : F4 - 1];                                                               \
    if (PER > 1) st1 = src_[tid + NTHREADS < F4 ? tid + NTHREADS : F4 - 1];                            \
    if (PER > 2) st2 = src_[tid + 2 * NTHREADS < F4 ? tid + 2 * NTHREADS : F4 - 1];                    \
    if (PER > 3) st3 = src_[tid + 3 * NTHREADS < F4 ? tid + 3 * NTHREADS : F4 - 1];                    \
    if (PER > 4) st4 = src_[tid + 4 * NTHREADS < F4 ? tid + 4 * NTHREADS : F4 - 1];                    \
  }
#define NF_STASH(buf_)                                                                \
  {                                                                                  \
    float4* dst_ = reinterpret_cast<float4*>(lds + (buf_) * kLayerFloats);             \
    if (tid < F4) dst_[tid] = st0;                                                   \
    if (PER > 1 && tid + NTHREADS < F4) dst_[tid + NTHREADS] = st1;                  \
    if (PER > 2 && tid + 2 * NTHREADS < F4) dst_[tid + 2 * NTHREADS] = st2;          \
    if (PER > 3 && tid + 3 * NTHREADS < F4) dst_[tid + 3 * NTHREADS] = st3;          \
    if (PER > 4 && tid + 4 * NTHREADS < F4) dst_[tid + 4 * NTHREADS] = st4;          \
  }
  NF_FETCH(0);
  NF_STASH(0);
  __syncthreads();

  double wave_sum = 0.0;
  long long step = 0;  // layers walked so far by this block: selects the LDS buffer
  for (long long pass = 0; pass < passes; ++pass) {
    // a wave without a group of its own walks the layers' barriers and staging only
    const bool last = pass >= full_passes;
    const long long g_mine = last ? full_passes * groups_per_pass + (long long)blockIdx.x * aw_last + wave
                                  : (pass * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool live = last ? (wave < aw_last && g_mine < groups) : true;
    const long long g = live ? g_mine : groups;
    const int c_first = (p.first_mask_parity + layer_of(0)) & 1;
    float xc[T], xt[T], ld[T];
    long long idx[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      idx[t] = g * (T * 32) + t * 32 + i;
      float2 v = make_float2(0.f, 0.f);
      if (idx[t] < rows) v = xin[idx[t]];
      xc[t] = c_first ? v.y : v.x;
      xt[t] = c_first ? v.x : v.y;
      ld[t] = (io.logdet_in != nullptr && idx[t] < rows) ? io.logdet_in[idx[t]] : 0.f;
    }

    for (int li = 0; li < nlay; ++li, ++step) {
      // request the image of the layer this block walks next (next pass wraps to its first)
      const bool more = li + 1 < nlay || pass + 1 < passes;
      if (more) NF_FETCH(li + 1 < nlay ? li + 1 : 0);

      if (live) {
      // training forward: this layer's input, 8 B per sample (rqs2d_bwd recomputes everything else from it)
      if (!INV && io.tape != nullptr) {
        const int cd = (c_first + li) & 1;   // the dim this layer conditions on: xc holds it, xt the transformed one
        float2* tp = reinterpret_cast<float2*>(io.tape) + (size_t)layer_of(li) * rows;
#pragma unroll
        for (int t = 0; t < T; ++t)
          if (h == 0 && idx[t] < rows) tp[idx[t]] = cd ? make_float2(xt[t], xc[t]) : make_float2(xc[t], xt[t]);
      }
      const float* L = lds + (step & 1) * kLayerFloats;
      const float2* a2 = reinterpret_cast<const float2*>(L + kA2) + lane;
      const float4* a3 = reinterpret_cast<const float4*>(L + kA3) + lane;
      const float2* w1b1 = reinterpret_cast<const float2*>(L + kW1B1) + h;

      f32x16 acc[T][2];
      {
        const f32x16 bias0 = lds_f32x16(L + kB2 + (h * 2 + 0) * 16);
        const f32x16 bias1 = lds_f32x16(L + kB2 + (h * 2 + 1) * 16);
        const float2 a = a2[0];
        const float2 wb = w1b1[0];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float b = __builtin_amdgcn_fmed3f(fmaf(xc[t], wb.x, wb.y), 0.f, 1.f);
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, bias0, 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, bias1, 0, 0, 0);
        }
      }
#pragma unroll 8
      for (int s = 1; s < 32; ++s) {
        const float2 a = a2[s * 64];
        const float2 wb = w1b1[2 * s];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float b = __builtin_amdgcn_fmed3f(fmaf(xc[t], wb.x, wb.y), 0.f, 1.f);
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[t][1], 0, 0, 0);
        }
      }

      // second hidden activation, in place: the B operand of the next pass and the input of the slope gather
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][m][r] = fmaxf(acc[t][m][r], 0.f);

      // bin-size logits + boundary slope logits: one 32-row tile, k order = accumulator order
      f32x16 out[T];
      {
        const f32x16 pb = lds_f32x16(L + kB3 + h * 16);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const float4 a = a3[g * 64];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int m = g >> 2, r = (g & 3) * 4;
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[t][m][r + 0], g == 0 ? pb : out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[t][m][r + 1], out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc[t][m][r + 2], out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc[t][m][r + 3], out[t], 0, 0, 0);
          }
        }
      }

#pragma unroll
      for (int t = 0; t < T; ++t) {
        float lg[K], knot[K + 1], size[K];
#pragma unroll
        for (int k = 0; k < K; ++k) lg[k] = out[t][k];
        knots_f64<K, NF_RQS_XEXP>(lg, p, knot, size);
        float xl, bw, yl, bh;
        const int kb = spline_bin<K, INV>(knot, size, xt[t], h, &xl, &bw, &yl, &bh);
        // slope logits kb and kb + 1: this lane's 32 hidden units against the column pair of its bin (one aligned
        // 8-byte LDS read per unit, immediate offsets), then the other half's 32
        const float2* ws = reinterpret_cast<const float2*>(L + kWS) + h * (32 * kSlopePairs) + kb;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float2 w = ws[(m * 16 + r) * kSlopePairs];
            d0 = fmaf(acc[t][m][r], w.x, d0);
            d1 = fmaf(acc[t][m][r], w.y, d1);
          }
        const float2 bs = reinterpret_cast<const float2*>(L + kBS)[kb];
        const float raw_l = add_halves(d0) + bs.x;
        const float raw_r = add_halves(d1) + bs.y;
        float nt, l;
        spline_map<INV>(xl, bw, yl, bh, raw_l, raw_r, out[t][kSlotSlopeFirst], out[t][kSlotSlopeLast], xt[t], p,
                        &nt, &l);
        ld[t] += l;
        xt[t] = xc[t];
        xc[t] = nt;
      }

      }

      if (more) NF_STASH((int)((step + 1) & 1));
      __syncthreads();
    }

    const int c_last = (c_first + nlay) & 1;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float v0 = c_last ? xt[t] : xc[t];
      const float v1 = c_last ? xc[t] : xt[t];
      float lp = ld[t];
      if (io.add_base) {
        if (io.base == NF_BASE_STD_NORMAL) {
          lp += -0.5f * (v0 * v0 + v1 * v1) - 1.8378770664093453f;
        } else {
          const bool inside = v0 >= 0.f && v0 <= 1.f && v1 >= 0.f && v1 <= 1.f;
          if (!inside) lp = -INFINITY;
        }
      }
      if (h == 0 && idx[t] < rows) {
        if (io.y) reinterpret_cast<float2*>(io.y)[idx[t]] = make_float2(v0, v1);
        if (io.logdet_out) io.logdet_out[idx[t]] = ld[t];
        if (io.logprob_out) io.logprob_out[idx[t]] = lp;
        wave_sum += (double)lp;
      }
    }
  }

  if (io.block_sums) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_sum += __shfl_down(wave_sum, off);
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds);
    if (lane == 0) red[wave] = wave_sum;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += red[w];
      io.block_sums[blockIdx.x] = s;
    }
  }
}

#undef NF_FETCH
#undef NF_STASH

int max_grid() { return 256; }
bool supported_bins(int K) { return K == 12 || K == 10 || K == 8 || K == 6 || K == 4; }

template <int T, int WAVES, int INV, int K>
static int launch_t(const Params& p, int grid, hipStream_t stream) {
  const size_t lds_bytes = 2 * (size_t)kLayerFloats * sizeof(float);
  NF_DYN_LDS(160 * 1024, &chain_kernel<T, WAVES, INV, K>);
  hipLaunchKernelGGL((chain_kernel<T, WAVES, INV, K>), dim3(grid), dim3(WAVES * 64), lds_bytes, stream, p);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

template <int T, int WAVES, int K>
static int launch_dir(const Params& p, int grid, hipStream_t stream) {
  return p.io.inverse_op ? launch_t<T, WAVES, 1, K>(p, grid, stream) : launch_t<T, WAVES, 0, K>(p, grid, stream);
}

template <int K>
static int launch_k(const Params& p, int grid, int T, int waves, hipStream_t stream) {
  if (waves == 8) return T == 2 ? launch_dir<2, 8, K>(p, grid, stream) : launch_dir<1, 8, K>(p, grid, stream);
  return launch_dir<1, 16, K>(p, grid, stream);
}

int launch(const float* wimg, const nf_flow_desc& d, const ChainIO& io, hipStream_t stream, int* out_grid) {
  Params p;
  p.wimg = wimg;
  p.io = io;
  p.first_mask_parity = d.first_mask_parity;
  p.range_min = d.rqs_range_min;
  p.range_max = d.rqs_range_max;
  p.min_bin = d.rqs_min_bin_size;
  p.min_slope = d.rqs_min_knot_slope;
  p.slope_offset = logf(expf(1.f - d.rqs_min_knot_slope) - 1.f);
  int T = 1, waves = 16;  // NF_RQS2D_TUNE="T,WAVES": development knob
  if (const int* e = switches().rqs2d_tune; e[0]) { T = e[0]; waves = e[1]; }
  if (waves != 8) { waves = 16; T = 1; }
  if (T != 2) T = 1;
  const long long groups = (io.rows + (T * 32 - 1)) / (T * 32);
  // few groups: one per workgroup first (every CU gets a wave before any CU gets two), then more waves each
  long long aw = (groups + max_grid() - 1) / max_grid();
  aw = aw < 1 ? 1 : (aw > waves ? waves : aw);
  long long want = (groups + aw - 1) / aw;
  // 8-wave workgroups of one tile per wave use 63.4 KB of LDS and <= 128 VGPRs: two of them share a CU
  const int cap = (waves == 8 && T == 1) ? 2 * max_grid() : max_grid();
  const int grid = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  if (out_grid) *out_grid = grid;
  switch (d.rqs_bins) {
    case 12: return launch_k<12>(p, grid, T, waves, stream);
    case 10: return launch_k<10>(p, grid, T, waves, stream);
    case 8: return launch_k<8>(p, grid, T, waves, stream);
    case 6: return launch_k<6>(p, grid, T, waves, stream);
    case 4: return launch_k<4>(p, grid, T, waves, stream);
    default: set_error("rqs2d kernel is built for 4, 6, 8, 10 and 12 bins"); return NF_ERR_UNSUPPORTED;
  }
}

}  // namespace rqs2d
}  // namespace nf
