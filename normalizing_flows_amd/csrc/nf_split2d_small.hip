// Fused chain of split-halves affine coupling layers, event_dim = 2, conditioner
// MLP 1 -> H -> 2 (variant A of the reference: src/real_nvp.py:71-171, chain
// order src/real_nvp.py:228-283, base density src/real_nvp.py:57-68).
//
// 12 FMA per sample per layer: there is no matrix work here.  One lane carries
// two samples (one 16-byte load), the 4H+2 weights of a layer are wave-uniform
// and come through the scalar cache (s_load), so the inner loop is pure VALU
// with SGPR operands.  Bound: VALU issue / HBM (12 B per sample), see DESIGN.md.
#include "nf_internal.h"

#include <math.h>

namespace nf {
namespace split2d {

struct Params {
  const float* wimg;  // [L][4H+2]
  ChainIO io;
  int H;
  int leaky;       // activation: 1 = leaky_relu(0.01), 0 = relu
  int first_flip;  // flip of layer 0
};

__device__ __forceinline__ float act_fn(float v, int leaky) {
  return leaky ? (v >= 0.f ? v : 0.01f * v) : fmaxf(v, 0.f);
}

// one sample through the layers this launch walks
__device__ __forceinline__ void walk(const Params& p, float& v0, float& v1, float& ld) {
  const ChainIO& io = p.io;
  const int nlay = io.layer_end - io.layer_begin;
  const int H = p.H;
  for (int li = 0; li < nlay; ++li) {
    const int l = io.descending ? io.layer_end - 1 - li : io.layer_begin + li;
    const int flip = (l & 1) ^ p.first_flip;
    const float* __restrict__ w = p.wimg + (size_t)l * layer_floats(H);
    const float c = flip ? v1 : v0;  // conditioning half (src/real_nvp.py:121-125)
    float t = flip ? v0 : v1;
    float shift = 0.f, ls = 0.f;
#pragma unroll 4
    for (int k = 0; k < H; ++k) {
      const float a = act_fn(fmaf(c, w[k], w[H + k]), p.leaky);
      shift = fmaf(a, w[2 * H + 2 * k + 0], shift);
      ls = fmaf(a, w[2 * H + 2 * k + 1], ls);
    }
    shift += w[4 * H + 0];
    ls += w[4 * H + 1];
    if (io.inverse_op) {
      t = (t - shift) * expf(-ls);  // src/real_nvp.py:129
      ld -= ls;                     // src/real_nvp.py:157
    } else {
      t = fmaf(t, expf(ls), shift);  // src/real_nvp.py:96
      ld += ls;
    }
    if (flip) v0 = t; else v1 = t;
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

__global__ __launch_bounds__(256) void chain_kernel(Params p) {
  const ChainIO& io = p.io;
  const long long rows = io.rows;
  const long long pairs = (rows + 1) / 2;
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
    walk(p, v.x, v.y, lda);
    walk(p, v.z, v.w, ldb);
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

int max_grid() { return 2048; }

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
  p.leaky = activation == NF_ACT_LEAKY_RELU;
  p.first_flip = first_flip & 1;
  const long long pairs = (io.rows + 1) / 2;
  long long want = (pairs + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > max_grid() ? max_grid() : want));
  if (out_grid) *out_grid = grid;
  hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), 0, stream, p);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace split2d

// ---------------------------------------------------------------------------
// fixed-order float64 sum of per-block partial sums -> one scalar
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_blocks_kernel(const double* __restrict__ part, int n,
                                                         double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

int launch_sum_blocks(const double* block_sums, int n, double* out_sum, hipStream_t stream) {
  hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, stream, block_sums, n, out_sum);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace nf
