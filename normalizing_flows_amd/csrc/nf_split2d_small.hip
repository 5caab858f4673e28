// Fused chain of split-halves affine coupling layers, event_dim = 2, conditioner
// MLP 1 -> H -> 2 (variant A of the reference: src/real_nvp.py:71-171, chain
// order src/real_nvp.py:228-283, base density src/real_nvp.py:57-68).
//
// 12 FMA per sample per layer: there is no matrix work here.  One lane carries
// two samples (one 16-byte load), the 4H+2 weights of a layer are wave-uniform
// and come through the scalar cache (s_load), so the inner loop is pure VALU
// with SGPR operands: per sample and layer 5 VALU per hidden unit + a
// 6-instruction exp + 3.  Bound: VALU issue (HBM traffic is 12 B per sample),
// see DESIGN.md.
#include "nf_internal.h"

#include <math.h>

namespace nf {
namespace split2d {

struct Params {
  const float* wimg;  // [L][4H+2]: w0[H], b0[H], w1[H][2], b1[2]
  ChainIO io;
  int H;
  int first_flip;  // flip of layer 0
};

__device__ __forceinline__ float exp_f32(float x) {
  // 2^(x*log2 e) with the product carried in two floats (see nf_affine2d_h64.hip)
  const float kLog2eHi = 1.44269502e+00f, kLog2eLo = 1.92596299e-08f, kLn2 = 6.93147182e-01f;
  const float hi = x * kLog2eHi;
  const float lo = fmaf(x, kLog2eLo, fmaf(x, kLog2eHi, -hi));
  const float e = __builtin_amdgcn_exp2f(hi);
  return fmaf(e, lo * kLn2, e);
}

// one coupling layer on two samples: c* condition, t* are transformed
template <int LEAKY, int INV, int HT>
__device__ __forceinline__ void layer(const float* __restrict__ w, int Hrt, float ca, float cb, float& ta, float& tb,
                                      float& lda, float& ldb) {
  const int H = HT > 0 ? HT : Hrt;
  float sa = 0.f, la = 0.f, sb = 0.f, lb = 0.f;
#pragma unroll 4
  for (int k = 0; k < H; ++k) {
    const float w0 = w[k], b0 = w[H + k], ws = w[2 * H + 2 * k], wl = w[2 * H + 2 * k + 1];
    float a = fmaf(ca, w0, b0), b = fmaf(cb, w0, b0);
    if (LEAKY) {  // jax.nn.leaky_relu, slope 0.01 < 1: max(v, 0.01 v)  (src/real_nvp.py:167)
      a = fmaxf(a, 0.01f * a);
      b = fmaxf(b, 0.01f * b);
    } else {
      a = fmaxf(a, 0.f);
      b = fmaxf(b, 0.f);
    }
    sa = fmaf(a, ws, sa); la = fmaf(a, wl, la);
    sb = fmaf(b, ws, sb); lb = fmaf(b, wl, lb);
  }
  const float bs = w[4 * H], bl = w[4 * H + 1];
  sa += bs; la += bl; sb += bs; lb += bl;
  if (INV) {
    ta = (ta - sa) * exp_f32(-la);  // src/real_nvp.py:129
    tb = (tb - sb) * exp_f32(-lb);
    lda -= la;                      // src/real_nvp.py:157
    ldb -= lb;
  } else {
    ta = fmaf(ta, exp_f32(la), sa);  // src/real_nvp.py:96
    tb = fmaf(tb, exp_f32(lb), sb);
    lda += la;
    ldb += lb;
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

template <int LEAKY, int INV, int HT>
__global__ __launch_bounds__(256) void chain_kernel(Params p) {
  const ChainIO& io = p.io;
  const long long rows = io.rows;
  const long long pairs = (rows + 1) / 2;
  const int H = HT > 0 ? HT : p.H;
  const int stride = layer_floats(H);
  const int nlay = io.layer_end - io.layer_begin;
  double sum = 0.0;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < pairs;
       q += (long long)gridDim.x * 256) {
    const long long r0 = 2 * q;
    const bool two = r0 + 1 < rows;
    float4 v;
    if (two) {
      v = reinterpret_cast<const float4*>(io.x)[q];
    } else {
      const float2 s = reinterpret_cast<const float2*>(io.x)[r0];
      v = make_float4(s.x, s.y, 0.f, 0.f);
    }
    float lda = io.logdet_in ? io.logdet_in[r0] : 0.f;
    float ldb = (io.logdet_in && two) ? io.logdet_in[r0 + 1] : 0.f;
    // flips alternate, so layers are walked in pairs with fixed roles; an odd one first
    int li = 0;
    int l = io.descending ? io.layer_end - 1 : io.layer_begin;
    const int step = io.descending ? -1 : 1;
    if (((l & 1) ^ p.first_flip) != 0) {  // flipped layer: dim 1 conditions, dim 0 is transformed
      layer<LEAKY, INV, HT>(p.wimg + (size_t)l * stride, H, v.y, v.w, v.x, v.z, lda, ldb);
      l += step;
      ++li;
    }
    for (; li + 1 < nlay; li += 2) {
      layer<LEAKY, INV, HT>(p.wimg + (size_t)l * stride, H, v.x, v.z, v.y, v.w, lda, ldb);
      layer<LEAKY, INV, HT>(p.wimg + (size_t)(l + step) * stride, H, v.y, v.w, v.x, v.z, lda, ldb);
      l += 2 * step;
    }
    if (li < nlay) layer<LEAKY, INV, HT>(p.wimg + (size_t)l * stride, H, v.x, v.z, v.y, v.w, lda, ldb);

    const float lpa = finish(io, v.x, v.y, lda);
    const float lpb = finish(io, v.z, v.w, ldb);
    if (two) {
      if (io.y) reinterpret_cast<float4*>(io.y)[q] = v;
      if (io.logdet_out) reinterpret_cast<float2*>(io.logdet_out)[q] = make_float2(lda, ldb);
      if (io.logprob_out) reinterpret_cast<float2*>(io.logprob_out)[q] = make_float2(lpa, lpb);
      sum += (double)lpa + (double)lpb;
    } else {
      if (io.y) reinterpret_cast<float2*>(io.y)[r0] = make_float2(v.x, v.y);
      if (io.logdet_out) io.logdet_out[r0] = lda;
      if (io.logprob_out) io.logprob_out[r0] = lpa;
      sum += (double)lpa;
    }
  }
  if (io.block_sums) {
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) io.block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

int max_grid() { return 4096; }

template <int LEAKY, int INV>
static void launch_h(const Params& p, int grid, hipStream_t stream) {
  if (p.H == 4) hipLaunchKernelGGL((chain_kernel<LEAKY, INV, 4>), dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((chain_kernel<LEAKY, INV, 0>), dim3(grid), dim3(256), 0, stream, p);
}

int launch(const float* wimg, int H, int activation, int first_flip, const ChainIO& io,
           hipStream_t stream, int* out_grid) {
  if (H < 1 || H > kMaxHidden) {
    set_error("split2d kernel supports hidden width 1..%d, got %d", kMaxHidden, H);
    return NF_ERR_UNSUPPORTED;
  }
  Params p;
  p.wimg = wimg;
  p.io = io;
  p.H = H;
  p.first_flip = first_flip & 1;
  const bool leaky = activation == NF_ACT_LEAKY_RELU;
  const long long pairs = (io.rows + 1) / 2;
  long long want = (pairs + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > max_grid() ? max_grid() : want));
  if (out_grid) *out_grid = grid;
  if (leaky) { if (io.inverse_op) launch_h<1, 1>(p, grid, stream); else launch_h<1, 0>(p, grid, stream); }
  else { if (io.inverse_op) launch_h<0, 1>(p, grid, stream); else launch_h<0, 0>(p, grid, stream); }
  NF_HIP(hipGetLastError());
  return NF_OK;
}


// ---------------------------------------------------------------------------
// Backward of one split coupling layer (training step of src/real_nvp.py:295-306:
// jax.value_and_grad(loss) + p - lr * g).  Like the masked flow's backward it stores no
// activations: the walk starts from z and rebuilds each layer's input with forward_nvp while the
// 12-FMA conditioner is recomputed.  One launch per layer, state (v, G) in HBM, the 4H+2
// gradient accumulators of a thread in registers, fixed-order reduction to one partial per block.
// loss_i = |z|^2/2 + log(2 pi) + sum(log_scale)  (log_prob = base - sum log_scale, :157-158)
// ---------------------------------------------------------------------------
template <int H, int LEAKY>
__global__ __launch_bounds__(256) void bwd_layer_kernel(const float* __restrict__ w, float2* __restrict__ v,
                                                        float2* __restrict__ G, float* __restrict__ partials,
                                                        long long rows, int flip, int first) {
  float dw0[H], db0[H], dw1s[H], dw1l[H];
  float db1s = 0.f, db1l = 0.f;
#pragma unroll
  for (int k = 0; k < H; ++k) { dw0[k] = 0.f; db0[k] = 0.f; dw1s[k] = 0.f; dw1l[k] = 0.f; }
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float2 s = v[r];
    const float2 g = first ? s : G[r];
    const float c = flip ? s.y : s.x;      // conditioning half (unchanged by the layer)
    const float x2 = flip ? s.x : s.y;     // the layer's OUTPUT on the transformed half
    float Gc = flip ? g.y : g.x;
    const float G2 = flip ? g.x : g.y;
    float a[H], pre[H];
    float shift = w[4 * H], ls = w[4 * H + 1];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      pre[k] = fmaf(c, w[k], w[H + k]);
      a[k] = LEAKY ? fmaxf(pre[k], 0.01f * pre[k]) : fmaxf(pre[k], 0.f);
      shift = fmaf(a[k], w[2 * H + 2 * k], shift);
      ls = fmaf(a[k], w[2 * H + 2 * k + 1], ls);
    }
    const float einv = exp_f32(-ls);
    const float y2 = fmaf(x2, exp_f32(ls), shift);   // the layer's input: forward_nvp (src/real_nvp.py:96)
    const float g_s = -G2 * einv;                     // x2 = (y2 - shift) e^-ls
    const float g_ls = fmaf(-G2, x2, 1.f);            // d x2 / d ls = -x2; +1 from the sum of log_scale
    db1s += g_s;
    db1l += g_ls;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      dw1s[k] = fmaf(a[k], g_s, dw1s[k]);
      dw1l[k] = fmaf(a[k], g_ls, dw1l[k]);
      const float ga = fmaf(w[2 * H + 2 * k], g_s, w[2 * H + 2 * k + 1] * g_ls);
      const float gp = LEAKY ? (pre[k] >= 0.f ? ga : 0.01f * ga) : (pre[k] > 0.f ? ga : 0.f);
      dw0[k] = fmaf(gp, c, dw0[k]);
      db0[k] += gp;
      Gc = fmaf(gp, w[k], Gc);
    }
    const float Gy2 = G2 * einv;
    v[r] = flip ? make_float2(y2, c) : make_float2(c, y2);
    G[r] = flip ? make_float2(Gy2, Gc) : make_float2(Gc, Gy2);
  }
  // block reduction in a fixed order: lanes (shuffle tree), then the four waves
  __shared__ float red[4][4 * H + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto wsum = [](float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    return x;
  };
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const float s0 = wsum(dw0[k]), s1 = wsum(db0[k]), s2 = wsum(dw1s[k]), s3 = wsum(dw1l[k]);
    if (lane == 0) { red[wave][k] = s0; red[wave][H + k] = s1; red[wave][2 * H + 2 * k] = s2; red[wave][2 * H + 2 * k + 1] = s3; }
  }
  const float t0 = wsum(db1s), t1 = wsum(db1l);
  if (lane == 0) { red[wave][4 * H] = t0; red[wave][4 * H + 1] = t1; }
  __syncthreads();
  if (threadIdx.x < 4 * H + 2)
    partials[(size_t)blockIdx.x * (4 * H + 2) + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// grad[e] = (1/rows_global) * sum over blocks of partial[b][e], blocks added in order
__global__ __launch_bounds__(64) void bwd_gather_kernel(const float* __restrict__ partials, int nparts, int n,
                                                       float* __restrict__ grad, float inv_rows) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f, c = 0.f;
  for (int b = 0; b < nparts; ++b) {
    const float y = partials[(size_t)b * n + e] - c;
    const float t = s + y;
    c = (t - s) - y;
    s = t;
  }
  grad[e] = s * inv_rows;
}

int bwd_max_blocks() { return 512; }
bool bwd_supported(int H) { return H == 4 || H == 8 || H == 16; }

template <int H>
static void launch_bwd_h(const float* w, float* v, float* G, float* partials, long long rows, int flip, int first,
                         int leaky, int grid, hipStream_t stream) {
  if (leaky)
    hipLaunchKernelGGL((bwd_layer_kernel<H, 1>), dim3(grid), dim3(256), 0, stream, w, reinterpret_cast<float2*>(v),
                       reinterpret_cast<float2*>(G), partials, rows, flip, first);
  else
    hipLaunchKernelGGL((bwd_layer_kernel<H, 0>), dim3(grid), dim3(256), 0, stream, w, reinterpret_cast<float2*>(v),
                       reinterpret_cast<float2*>(G), partials, rows, flip, first);
}

int launch_bwd_layer(const float* w_layer, int H, int activation, float* v, float* G, float* partials, float* grad_layer,
                     long long rows, long long rows_global, int flip, int first, hipStream_t stream) {
  if (!bwd_supported(H)) { set_error("split2d backward is built for hidden width 4, 8, 16"); return NF_ERR_UNSUPPORTED; }
  const long long want = (rows + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > bwd_max_blocks() ? bwd_max_blocks() : want));
  const int leaky = activation == NF_ACT_LEAKY_RELU;
  if (H == 4) launch_bwd_h<4>(w_layer, v, G, partials, rows, flip, first, leaky, grid, stream);
  else if (H == 8) launch_bwd_h<8>(w_layer, v, G, partials, rows, flip, first, leaky, grid, stream);
  else launch_bwd_h<16>(w_layer, v, G, partials, rows, flip, first, leaky, grid, stream);
  NF_HIP(hipGetLastError());
  const int n = layer_floats(H);
  hipLaunchKernelGGL(bwd_gather_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, partials, grid, n, grad_layer,
                     1.0f / (float)rows_global);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace split2d

// ---------------------------------------------------------------------------
// fixed-order float64 sum of per-block partial sums -> one scalar
// ---------------------------------------------------------------------------
// host_out (may be null): the handle's pinned, device-mapped result slot -- written from the kernel, so that the
// NLL tail of a step is ONE launch and no 8-byte copy (at the reference's batch sizes the tail, not the chain
// kernel, sets the step time: 35 -> see profiles/bench_configs.py b4k).
__global__ __launch_bounds__(256) void sum_blocks_kernel(const double* __restrict__ part, int n,
                                                         double* __restrict__ out, double* __restrict__ host_out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0];
    if (host_out != nullptr) {
      __hip_atomic_store(host_out, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
  }
}

int launch_sum_blocks(const double* block_sums, int n, double* out_sum, hipStream_t stream, double* host_out) {
  hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, stream, block_sums, n, out_sum, host_out);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace nf
