// Fused chain of split-halves affine coupling layers, event_dim = 2, conditioner
// MLP 1 -> H -> 2 (variant A of the reference: src/real_nvp.py:71-171, chain
// order src/real_nvp.py:228-283, base density src/real_nvp.py:57-68).
//
// 12 FMA per sample per layer: there is no matrix work here, the kernel is bound by vector issue (HBM traffic
// is 12 B per sample).  What the vector ALU charges on gfx950 depends on the operand FORM
// (profiles/microbench/valu_f32_operands.hip, 8 waves per SIMD): v_fma / v_mul / v_add with VGPR operands 2.5-3.3
// cycles per wave64 instruction, the same opcodes with an SGPR operand 4.1-4.4, v_max 4.2, v_exp 8.2.  Hence:
//   * the walked layers' weights are staged ONCE per workgroup in LDS, four floats per hidden unit
//     (w0, b0, w_shift, w_logscale * log2 e): one 16-byte broadcast read per unit brings them into VGPRs -- no SGPR
//     operands, no moves, and nothing on the vector-memory counter, so
//   * the next samples of a lane are requested before the current ones are walked and that request stays in
//     flight across the whole walk (first build: weights through vector loads with a wait per layer pair that
//     also waited for the samples -- the vector ALU was busy half of the time, 184 us at 2^24 rows);
//   * the log-scale comes out of the conditioner in units of log 2 (weights and bias prescaled when staged), so
//     exp(log_scale) is ONE v_exp_f32 and the log-det is summed in those units and scaled by ln 2 once per sample
//     (the two-float product that makes exp2(x * log2 e) accurate cost 5 instructions per sample and layer;
//     prescaling rounds each weight once, the same order of error as log_scale's own float32 rounding);
//   * a lane carries two samples (one 16-byte load); the LDS reads of a layer are shared by them.
// Measured at 2^24 rows (profiles/bench_configs.py a): 184 us -> 84-96 us from box to box; 56,228 vector instructions
// per SIMD in 181,800 cycles (profiles/r02/config_a_pmc_summary.json): 3.2 cycles each, the table's price for the mix --
// per sample and layer 21 instructions (4 x (fma, mul, max, 2 fma) + exp + 4) priced at 79 cycles.
#include "nf_internal.h"

#include <math.h>
#include <stdlib.h>

#include <type_traits>

namespace nf {
namespace split2d {

struct Params {
  ChainIO io;
  int H;
  int first_flip;  // flip of layer 0
};

// LDS image of one layer: [H] x {w0, b0, C_shift, C_logscale}, then {B_shift, B_logscale, A_shift, A_logscale}
// (A, B, C: see layer(); everything of the log-scale output times log2 e)
__host__ __device__ inline int lds_layer_floats(int H) { return 4 * H + 4; }
constexpr int kMaxLdsFloats = 15 * 1024;  // 60 KB of weights per launch

int max_layers_per_launch(int H) { return kMaxLdsFloats / lds_layer_floats(H); }

// exp for the backward walk: 2^(x*log2 e) with the product carried in two floats (see nf_affine2d_h64.hip)
__device__ __forceinline__ float exp_f32(float x) {
  const float kLog2eHi = 1.44269502e+00f, kLog2eLo = 1.92596299e-08f, kLn2f = 6.93147182e-01f;
  const float hi = x * kLog2eHi;
  const float lo = fmaf(x, kLog2eLo, fmaf(x, kLog2eHi, -hi));
  const float e = __builtin_amdgcn_exp2f(hi);
  return fmaf(e, lo * kLn2f, e);
}

constexpr double kLog2e = 1.44269504088896341;
constexpr float kLn2 = 0.69314718055994531f;

// one coupling layer on S samples: c conditions, t is transformed, ld2 = log-det in units of log 2.
// The activation is written act(a) = alpha a + beta |a| (relu: 1/2, 1/2; leaky_relu 0.01: 0.505, 0.495), so
//   out_s = b_s + sum_k w_sk act(a_k),  a_k = w0_k c + b0_k
//         = [b_s + alpha sum_k w_sk b0_k] + [alpha sum_k w_sk w0_k] c + sum_k (beta w_sk) |a_k|  =  B_s + A_s c + sum_k C_sk |a_k|
// with A, B, C formed once per workgroup when the layer is staged (float64 sums, one rounding each): per hidden unit
// THREE v_fma with VGPR operands (the |a| is a free source modifier) instead of fma, mul, max, fma, fma -- 14 + 3
// vector instructions per sample and layer instead of 23, and none of the 4.2-cycle v_max (operand-form table above).
// Not the reference's order of roundings: an INACTIVE unit's contribution (0 for relu, 0.01 w a for leaky_relu) is the
// difference of two terms of size |w_sk a_k| / 2 here, so the error of an output is ~eps/2 * sum_k |w_sk| |a_k| where
// the reference's order has eps * |result|.  For weights like the scripts' initialisation (|w| ~ 0.1, b0 = 0) that is
// far below float32's rounding of the result itself (emulated in NumPy float32, tests/tools/split2d_abs_form_study.py:
// 8.5e-7 scaled log_prob error for both orders); for a trained flow with strongly negative pre-activations and large
// second-layer weights (b0 ~ -30, |w1| ~ 10, H = 16, 8 layers) the same emulation gives 1.3e-4 against 2.4e-6 (ADVICE
// r03).  So the form is chosen per launch from the weights themselves: FORM = 0, the folded form, while
// sum_k |w_sk| (|b0_k| + 8 |w0_k|) <= kFoldedFormBound for every staged layer and output (a worst-case error of
// 2.4e-7 per layer), FORM = 1, the reference's order (five instructions per unit instead of three), otherwise.
constexpr float kFoldedFormBound = 8.f;
template <int INV, int HT, int S, int FORM, int LEAKY>
__device__ __forceinline__ void layer(const float* __restrict__ w, int Hrt, const float (&c)[S], float (&t)[S],
                                      float (&ld2)[S]) {
  const int H = HT > 0 ? HT : Hrt;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w);
  const float4 lin = w4[H];  // FORM 0: {B_shift, B_logscale, A_shift, A_logscale}; FORM 1: {b_shift, b_logscale, ., .}
  float sh[S], ls[S];
#pragma unroll
  for (int u = 0; u < S; ++u) {
    sh[u] = FORM ? lin.x : fmaf(lin.z, c[u], lin.x);
    ls[u] = FORM ? lin.y : fmaf(lin.w, c[u], lin.y);
  }
#pragma unroll 4
  for (int k = 0; k < H; ++k) {
    const float4 wk = w4[k];  // {w0, b0, C_shift, C_logscale} (FORM 1: w_shift, w_logscale in the last two)
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const float a = fmaf(c[u], wk.x, wk.y);
      if (FORM) {   // src/real_nvp.py:161-171: act = leaky_relu(x @ W + b), then act @ W_last
        const float act = LEAKY ? fmaxf(a, 0.01f * a) : fmaxf(a, 0.f);
        sh[u] = fmaf(act, wk.z, sh[u]);
        ls[u] = fmaf(act, wk.w, ls[u]);
      } else {
        sh[u] = fmaf(fabsf(a), wk.z, sh[u]);
        ls[u] = fmaf(fabsf(a), wk.w, ls[u]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < S; ++u) {
    if (INV) {
      t[u] = (t[u] - sh[u]) * __builtin_amdgcn_exp2f(-ls[u]);  // src/real_nvp.py:129
      ld2[u] -= ls[u];                                          // src/real_nvp.py:157
    } else {
      t[u] = fmaf(t[u], __builtin_amdgcn_exp2f(ls[u]), sh[u]);  // src/real_nvp.py:96
      ld2[u] += ls[u];
    }
  }
}

__device__ __forceinline__ float finish(const ChainIO& io, float v0, float v1, float ld) {
  float lp = ld;
  if (io.add_base) {
    if (io.base == NF_BASE_STD_NORMAL) {
      lp += -0.5f * (v0 * v0 + v1 * v1) - 1.8378770664093453f;
    } else {
      const bool inside = v0 >= 0.f && v0 <= 1.f && v1 >= 0.f && v1 <= 1.f;
      if (!inside) lp = -INFINITY;
    }
  }
  return lp;
}

// S = 2 * V samples per lane: V 16-byte loads {x0, y0, x1, y1}
template <int LEAKY, int INV, int HT, int V>
__global__ __launch_bounds__(256) void chain_kernel(Params p, const float* __restrict__ wimg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int S = 2 * V;
  const ChainIO& io = p.io;
  const long long rows = io.rows;
  const long long groups = (rows + S - 1) / S;
  const int H = HT > 0 ? HT : p.H;
  const int nlay = io.layer_end - io.layer_begin;
  const int SL = lds_layer_floats(H);

  // stage the walked layers (global image per layer: w0[H], b0[H], w1[H][2], b1[2]) in the A / B / C form of layer()
  // (form 1: the weights as they are)
  auto stage = [&](int form) {
    const double alpha = form ? 1.0 : (LEAKY ? 0.505 : 0.5), beta = form ? 1.0 : (LEAKY ? 0.495 : 0.5);
    for (int idx = threadIdx.x; idx < nlay * SL; idx += 256) {
      const int li = idx / SL, r = idx - li * SL;
      const float* __restrict__ src = wimg + (size_t)(io.layer_begin + li) * layer_floats(H);
      const int k = r >> 2, e = r & 3;
      double v;
      if (k < H) {
        v = e == 0 ? (double)src[k] : e == 1 ? (double)src[H + k] : beta * (double)src[2 * H + 2 * k + (e - 2)];
      } else if (form) {
        v = e < 2 ? (double)src[4 * H + e] : 0.0;
      } else {
        const int o = e & 1;  // output: 0 shift, 1 log-scale
        double acc = 0.0;
        for (int q = 0; q < H; ++q) acc += (double)src[2 * H + 2 * q + o] * (double)(e < 2 ? src[H + q] : src[q]);
        v = e < 2 ? (double)src[4 * H + o] + alpha * acc : alpha * acc;
      }
      if ((k < H && e == 3) || (k == H && (e & 1))) v *= (double)kLog2e;
      lds[idx] = (float)v;
    }
  };
  // which form: every workgroup decides alike, from the weights (see layer())
  __shared__ int use_ref_form;
  if (threadIdx.x == 0) use_ref_form = 0;
  __syncthreads();
  for (int q = threadIdx.x; q < nlay * 2; q += 256) {
    const float* __restrict__ src = wimg + (size_t)(io.layer_begin + (q >> 1)) * layer_floats(H);
    float bound = 0.f;
    for (int k = 0; k < H; ++k) bound += fabsf(src[2 * H + 2 * k + (q & 1)]) * (fabsf(src[H + k]) + 8.f * fabsf(src[k]));
    if (!(bound <= kFoldedFormBound)) use_ref_form = 1;   // (NaN weights: the reference's order too)
  }
  __syncthreads();
  const bool ref_form = use_ref_form != 0;
  stage(ref_form ? 1 : 0);
  __syncthreads();

  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(io.x);
  // samples S*g .. S*g + S-1 of group g as V float4 {x, y, x, y}; a group that runs past the last row is read
  // sample by sample
  auto fetch = [&](long long g, float4 (&v)[V]) {
    if (S * g + S <= rows) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = x4[V * g + j];
    } else {
      const float2* __restrict__ x2 = reinterpret_cast<const float2*>(io.x);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const long long r0 = S * g + 2 * j;
        const float2 a = r0 < rows ? x2[r0] : make_float2(0.f, 0.f);
        const float2 b = r0 + 1 < rows ? x2[r0 + 1] : make_float2(0.f, 0.f);
        v[j] = make_float4(a.x, a.y, b.x, b.y);
      }
    }
  };

  double sum = 0.0;
  const long long gstride = (long long)gridDim.x * 256;
  auto walk = [&](auto form_tag) {
  constexpr int FORM = decltype(form_tag)::value;
  long long gn = (long long)blockIdx.x * 256 + threadIdx.x;
  float4 vn[V];
#pragma unroll
  for (int j = 0; j < V; ++j) vn[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gn < groups) fetch(gn, vn);
  while (gn < groups) {
    const long long g = gn;
    float d0[S], d1[S], ld2[S];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      d0[2 * j] = vn[j].x; d1[2 * j] = vn[j].y; d0[2 * j + 1] = vn[j].z; d1[2 * j + 1] = vn[j].w;
    }
    gn += gstride;
    if (gn < groups) fetch(gn, vn);  // stays in flight while this group's layers are walked
#pragma unroll
    for (int u = 0; u < S; ++u) ld2[u] = 0.f;

    // flips alternate, so layers are walked in pairs with fixed roles; an odd one first
    int li = 0;
    int l = io.descending ? io.layer_end - 1 : io.layer_begin;
    const int step = io.descending ? -1 : 1;
    const float* L = lds + (l - io.layer_begin) * SL;
    const int Lstep = step * SL;
    if (((l & 1) ^ p.first_flip) != 0) {  // flipped layer: dim 1 conditions, dim 0 is transformed
      layer<INV, HT, S, FORM, LEAKY>(L, H, d1, d0, ld2);
      L += Lstep;
      ++li;
    }
    for (; li + 1 < nlay; li += 2) {
      layer<INV, HT, S, FORM, LEAKY>(L, H, d0, d1, ld2);
      layer<INV, HT, S, FORM, LEAKY>(L + Lstep, H, d1, d0, ld2);
      L += 2 * Lstep;
    }
    if (li < nlay) layer<INV, HT, S, FORM, LEAKY>(L, H, d0, d1, ld2);

    const long long r0 = S * g;
    const bool full = r0 + S <= rows;
    float ld[S], lp[S];
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const float in = (io.logdet_in && r0 + u < rows) ? io.logdet_in[r0 + u] : 0.f;
      ld[u] = fmaf(ld2[u], kLn2, in);
      lp[u] = finish(io, d0[u], d1[u], ld[u]);
    }
    if (full) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (io.y) reinterpret_cast<float4*>(io.y)[V * g + j] = make_float4(d0[2 * j], d1[2 * j], d0[2 * j + 1], d1[2 * j + 1]);
        if (io.logdet_out) reinterpret_cast<float2*>(io.logdet_out)[V * g + j] = make_float2(ld[2 * j], ld[2 * j + 1]);
        if (io.logprob_out) reinterpret_cast<float2*>(io.logprob_out)[V * g + j] = make_float2(lp[2 * j], lp[2 * j + 1]);
      }
#pragma unroll
      for (int u = 0; u < S; ++u) sum += (double)lp[u];
    } else {
#pragma unroll
      for (int u = 0; u < S; ++u)
        if (r0 + u < rows) {
          if (io.y) reinterpret_cast<float2*>(io.y)[r0 + u] = make_float2(d0[u], d1[u]);
          if (io.logdet_out) io.logdet_out[r0 + u] = ld[u];
          if (io.logprob_out) io.logprob_out[r0 + u] = lp[u];
          sum += (double)lp[u];
        }
    }
  }
  };  // walk
  if (ref_form) walk(std::integral_constant<int, 1>());
  else walk(std::integral_constant<int, 0>());
  if (io.block_sums) {
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) io.block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// two resident rounds of 8 workgroups of 4 waves per CU (measured at 2^24 rows: 1024 / 2048 / 4096 workgroups
// 115 / 101 / 96 us with two samples per lane, 101 / 96 / 96 us with four)
int max_grid() { return 4096; }

template <int LEAKY, int INV, int V>
static void launch_h(const Params& p, const float* wimg, int grid, size_t lds_bytes, hipStream_t stream) {
  if (p.H == 4) hipLaunchKernelGGL((chain_kernel<LEAKY, INV, 4, V>), dim3(grid), dim3(256), lds_bytes, stream, p, wimg);
  else hipLaunchKernelGGL((chain_kernel<LEAKY, INV, 0, V>), dim3(grid), dim3(256), lds_bytes, stream, p, wimg);
}

template <int V>
static void launch_v(const Params& p, const float* wimg, bool leaky, int grid, size_t lds_bytes, hipStream_t stream) {
  if (leaky) {
    if (p.io.inverse_op) launch_h<1, 1, V>(p, wimg, grid, lds_bytes, stream);
    else launch_h<1, 0, V>(p, wimg, grid, lds_bytes, stream);
  } else {
    if (p.io.inverse_op) launch_h<0, 1, V>(p, wimg, grid, lds_bytes, stream);
    else launch_h<0, 0, V>(p, wimg, grid, lds_bytes, stream);
  }
}

int launch(const float* wimg, int H, int activation, int first_flip, const ChainIO& io,
           hipStream_t stream, int* out_grid) {
  if (H < 1 || H > kMaxHidden) {
    set_error("split2d kernel supports hidden width 1..%d, got %d", kMaxHidden, H);
    return NF_ERR_UNSUPPORTED;
  }
  const int nlay = io.layer_end - io.layer_begin;
  if (nlay > max_layers_per_launch(H)) {
    set_error("split2d kernel walks at most %d layers of width %d per launch, got %d", max_layers_per_launch(H), H, nlay);
    return NF_ERR_UNSUPPORTED;
  }
  Params p;
  p.io = io;
  p.H = H;
  p.first_flip = first_flip & 1;
  const bool leaky = activation == NF_ACT_LEAKY_RELU;
  const size_t lds_bytes = (size_t)nlay * lds_layer_floats(H) * sizeof(float);
  // four samples per lane (two 16-byte loads): a layer's LDS reads and a group's loop overhead (addresses, the
  // float64 block sum, stores) are shared by more samples.  With the 17-instruction layer the overhead is what is
  // left to share: 2 / 4 / 8 samples per lane measure 72.8 / 68.6-71.0 / 71.8-72.3 us at 2^24 rows
  // (profiles/scratch/sweep_split2d.sh; NF_SPLIT2D_SAMPLES, NF_SPLIT2D_GRID are the development knobs)
  // -- once the batch fills the chip twice over; below that two per lane keep more waves in flight
  int S = io.rows >= (1LL << 22) ? 4 : 2;
  if (switches().split2d_samples) S = switches().split2d_samples;
  const long long groups = (io.rows + S - 1) / S;
  long long want = (groups + 255) / 256;
  int cap = max_grid();
  if (switches().split2d_grid) cap = switches().split2d_grid;
  const int grid = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  if (out_grid) *out_grid = grid;
  if (S == 8) launch_v<4>(p, wimg, leaky, grid, lds_bytes, stream);
  else if (S == 4) launch_v<2>(p, wimg, leaky, grid, lds_bytes, stream);
  else launch_v<1>(p, wimg, leaky, grid, lds_bytes, stream);
  NF_HIP(hipGetLastError());
  return NF_OK;
}


// ---------------------------------------------------------------------------
// Backward of the chain (training step of src/real_nvp.py:295-306: jax.value_and_grad(loss) + p - lr * g).
// Like the masked flow's backward it stores no activations: the walk starts from z and rebuilds each layer's
// input with forward_nvp while the 12-FMA conditioner is recomputed.
//   loss_i = |z|^2/2 + log(2 pi) + sum(log_scale)      (log_prob = base - sum log_scale, :157-158)
// ONE launch walks up to LG layers with a sample's state (value, gradient) in registers and the 4H+2 gradient
// accumulators of every walked layer in registers too (LG = 8 / 4 / 2 for H = 4 / 8 / 16: ~140 of them); longer
// chains are several launches with the state carried through HBM.  The first build launched one kernel per layer
// (state through HBM every time: 32 B per sample and layer, 112 us per layer at 2^24 rows, at the HBM roofline)
// plus a gather per layer whose 4H+2 threads added the blocks' partials one dependent load at a time (84 us per
// layer whatever the batch: 90 % of the step at 2^17 rows).  Sums are in a fixed order throughout: a thread's rows
// in grid-stride order, lanes by a shuffle tree, the four waves, then the blocks (gather_kernel).
// ---------------------------------------------------------------------------
template <int H, int LEAKY, int LG, int FLIP0>
__global__ __launch_bounds__(256) void bwd_chain_kernel(const float* __restrict__ w, float2* __restrict__ v,
                                                        float2* __restrict__ G, float* __restrict__ partials,
                                                        long long rows, int layer_begin, int nl, int nl_total,
                                                        int first, int last) {
  constexpr int N = 4 * H + 2;
  float dw0[LG][H], db0[LG][H], dw1s[LG][H], dw1l[LG][H], db1s[LG], db1l[LG];
#pragma unroll
  for (int j = 0; j < LG; ++j) {
    db1s[j] = 0.f; db1l[j] = 0.f;
#pragma unroll
    for (int k = 0; k < H; ++k) { dw0[j][k] = 0.f; db0[j][k] = 0.f; dw1s[j][k] = 0.f; dw1l[j][k] = 0.f; }
  }
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float2 s = v[r];
    const float2 g = first ? s : G[r];
    float d[2] = {s.x, s.y}, gd[2] = {g.x, g.y};
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      if (j < nl) {
        constexpr int kC = 0;  // silence unused warnings in some instantiations
        (void)kC;
        const int ic = (FLIP0 ^ (j & 1)) ? 1 : 0;  // conditioning half (unchanged by the layer)
        const int it = 1 - ic;                      // transformed half: d[it] is the layer's OUTPUT
        const float* __restrict__ wl = w + (size_t)(layer_begin + j) * N;
        const float c = d[ic], x2 = d[it], G2 = gd[it];
        float Gc = gd[ic];
        float a[H], pre[H];
        float shift = wl[4 * H], ls = wl[4 * H + 1];
#pragma unroll
        for (int k = 0; k < H; ++k) {
          pre[k] = fmaf(c, wl[k], wl[H + k]);
          a[k] = LEAKY ? fmaxf(pre[k], 0.01f * pre[k]) : fmaxf(pre[k], 0.f);
          shift = fmaf(a[k], wl[2 * H + 2 * k], shift);
          ls = fmaf(a[k], wl[2 * H + 2 * k + 1], ls);
        }
        const float einv = exp_f32(-ls);
        const float y2 = fmaf(x2, exp_f32(ls), shift);   // the layer's input: forward_nvp (src/real_nvp.py:96)
        const float g_s = -G2 * einv;                     // x2 = (y2 - shift) e^-ls
        const float g_ls = fmaf(-G2, x2, 1.f);            // d x2 / d ls = -x2; +1 from the sum of log_scale
        db1s[j] += g_s;
        db1l[j] += g_ls;
#pragma unroll
        for (int k = 0; k < H; ++k) {
          dw1s[j][k] = fmaf(a[k], g_s, dw1s[j][k]);
          dw1l[j][k] = fmaf(a[k], g_ls, dw1l[j][k]);
          const float ga = fmaf(wl[2 * H + 2 * k], g_s, wl[2 * H + 2 * k + 1] * g_ls);
          const float gp = LEAKY ? (pre[k] >= 0.f ? ga : 0.01f * ga) : (pre[k] > 0.f ? ga : 0.f);
          dw0[j][k] = fmaf(gp, c, dw0[j][k]);
          db0[j][k] += gp;
          Gc = fmaf(gp, wl[k], Gc);
        }
        d[it] = y2;
        gd[it] = G2 * einv;
        gd[ic] = Gc;
      }
    }
    if (!last) {
      v[r] = make_float2(d[0], d[1]);
      G[r] = make_float2(gd[0], gd[1]);
    }
  }
  // block reduction in a fixed order: lanes (shuffle tree), then the four waves
  __shared__ float red[4][LG * N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto wsum = [](float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    return x;
  };
#pragma unroll
  for (int j = 0; j < LG; ++j) {
    if (j < nl) {
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float s0 = wsum(dw0[j][k]), s1 = wsum(db0[j][k]), s2 = wsum(dw1s[j][k]), s3 = wsum(dw1l[j][k]);
        if (lane == 0) {
          red[wave][j * N + k] = s0; red[wave][j * N + H + k] = s1;
          red[wave][j * N + 2 * H + 2 * k] = s2; red[wave][j * N + 2 * H + 2 * k + 1] = s3;
        }
      }
      const float t0 = wsum(db1s[j]), t1 = wsum(db1l[j]);
      if (lane == 0) { red[wave][j * N + 4 * H] = t0; red[wave][j * N + 4 * H + 1] = t1; }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nl * N; e += 256)
    partials[((size_t)blockIdx.x * nl_total + layer_begin) * N + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

// grad[e] = (1/rows_global) * sum over blocks of partial[b][e]: one wave per element, lane t adds blocks
// t, t + 64, ... in order (float64), the lanes are added by a shuffle tree -- a fixed order
__global__ __launch_bounds__(64) void bwd_gather_kernel(const float* __restrict__ partials, int nparts, int n,
                                                       float* __restrict__ grad, double inv_rows) {
  const int e = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nparts; b += 64) s += (double)partials[(size_t)b * n + e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (threadIdx.x == 0) grad[e] = (float)(s * inv_rows);
}

int bwd_max_blocks() { return 512; }
bool bwd_supported(int H) { return H == 4 || H == 8 || H == 16; }
int bwd_layers_per_launch(int H) { return H == 4 ? 8 : H == 8 ? 4 : 2; }

template <int H, int LG>
static void launch_bwd_h(const float* w, float* v, float* G, float* partials, long long rows, int layer_begin, int nl,
                         int nl_total, int flip0, int first, int last, int leaky, int grid, hipStream_t stream) {
  float2* v2 = reinterpret_cast<float2*>(v);
  float2* G2 = reinterpret_cast<float2*>(G);
#define NF_BWD_LAUNCH(LK, FL) \
  hipLaunchKernelGGL((bwd_chain_kernel<H, LK, LG, FL>), dim3(grid), dim3(256), 0, stream, w, v2, G2, partials, rows, \
                     layer_begin, nl, nl_total, first, last)
  if (leaky) { if (flip0) NF_BWD_LAUNCH(1, 1); else NF_BWD_LAUNCH(1, 0); }
  else { if (flip0) NF_BWD_LAUNCH(0, 1); else NF_BWD_LAUNCH(0, 0); }
#undef NF_BWD_LAUNCH
}

// Gradient of the NLL with respect to every layer's weights, into grad [num_layers][4H+2].  v: z on entry (the
// forward walk's output), G: scratch of the same size; partials: [bwd_max_blocks()][num_layers][4H+2].
int launch_bwd_chain(const float* wimg, int H, int num_layers, int activation, int first_flip, float* v, float* G,
                     float* partials, float* grad, long long rows, long long rows_global, hipStream_t stream) {
  if (!bwd_supported(H)) { set_error("split2d backward is built for hidden width 4, 8, 16"); return NF_ERR_UNSUPPORTED; }
  const long long want = (rows + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > bwd_max_blocks() ? bwd_max_blocks() : want));
  const int leaky = activation == NF_ACT_LEAKY_RELU;
  const int LG = bwd_layers_per_launch(H);
  // the layer log_prob applied last is walked first: layer 0, then 1, ...
  for (int lb = 0; lb < num_layers; lb += LG) {
    const int nl = num_layers - lb < LG ? num_layers - lb : LG;
    const int flip0 = ((lb & 1) ^ first_flip) & 1;
    const int first = lb == 0, last = lb + nl == num_layers;
    if (H == 4) launch_bwd_h<4, 8>(wimg, v, G, partials, rows, lb, nl, num_layers, flip0, first, last, leaky, grid, stream);
    else if (H == 8) launch_bwd_h<8, 4>(wimg, v, G, partials, rows, lb, nl, num_layers, flip0, first, last, leaky, grid, stream);
    else launch_bwd_h<16, 2>(wimg, v, G, partials, rows, lb, nl, num_layers, flip0, first, last, leaky, grid, stream);
    NF_HIP(hipGetLastError());
  }
  const int n = num_layers * layer_floats(H);
  hipLaunchKernelGGL(bwd_gather_kernel, dim3(n), dim3(64), 0, stream, partials, grid, n, grad, 1.0 / (double)rows_global);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace split2d

// ---------------------------------------------------------------------------
// fixed-order float64 sum of per-block partial sums -> one scalar
// ---------------------------------------------------------------------------
// host_out (may be null): the handle's pinned, device-mapped result slot.
// One workgroup.  Launched with 64 threads (n <= 256: every NLL step of the fused 2-D kernels): lane t adds parts
// t, t + 64, ... in order and a shuffle tree adds the lanes -- no LDS, no barrier; with 256 threads: the same per thread,
// then an LDS tree.  The scalar also goes to the handle's pinned host slot (system scope; the kernel's end releases it,
// the host reads it after the event that follows the launch), so the tail of a step is one launch and no 8-byte copy.
// Measured at BASELINE configs[0] (profiles/scratch/nll_step_b4k.py): 17.8 -> 17.4 us per step against the
// 256-thread tree with a system fence on one box, no difference on another (boxes differ by +-1 us on this step):
// the step is the two kernels' launch-to-launch latency, not their work.
__global__ __launch_bounds__(256) void sum_blocks_kernel(const double* __restrict__ part, int n,
                                                         double* __restrict__ out, double* __restrict__ host_out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  double total;
  if (blockDim.x == 64) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    total = s;
  } else {
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    total = red[0];
  }
  if (threadIdx.x == 0) {
    out[0] = total;
    if (host_out != nullptr) __hip_atomic_store(host_out, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int launch_sum_blocks(const double* block_sums, int n, double* out_sum, hipStream_t stream, double* host_out) {
  hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(n <= 256 ? 64 : 256), 0, stream, block_sums, n, out_sum, host_out);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace nf
