// Optional fast mode of the fused masked-affine chain (nf_flow_set_option "hidden_matmul" =
// "bf16x3"): the 64x64 hidden Linear as SIX bf16 MFMAs on a three-way bf16 split of both
// operands instead of one exact-f32 MFMA.
//
//   x = x_h + x_m + x_l (each part a bf16, 24 significant bits in all);
//   W^T h ~= W_h h_h + W_h h_m + W_m h_h + W_h h_l + W_l h_h + W_m h_m       (float32 accumulate)
//
// The dropped terms are <= 2^-24 relative, i.e. the size of float32's own rounding of each product
// (oracle emulation: same number of out-of-tolerance samples as the float32 path, none; see
// DESIGN.md section 4).  Why it pays on gfx950: v_mfma_f32_32x32x2_f32 occupies the SIMD's vector
// issue for its whole 64 cycles, while v_mfma_f32_32x32x16_bf16 does 8x the work in 32 cycles and
// only blocks issue for 8 of them (profiles/microbench/mfma_bf16_valu_overlap.hip), so 48 bf16
// MFMAs (1536 cycles) + the ~330 VALU instructions that build and split the operand replace 64
// f32 MFMAs (4096 cycles) + ~150 VALU.
//
// The default remains the exact-f32 kernel (nf_affine2d_h64.hip); everything else -- first
// Linear in operand layout (clamp-bit relu), last Linear on the accumulator, streaming of layer
// images through an LDS double buffer -- is as in nf_affine2d_h64.hip / nf_rqs2d_h64.hip.
#include "nf_bf16x3.h"
#include "nf_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace nf {
namespace affine2d_x3 {

using bf16x3::bf16x8;
using bf16x3::f32x16;
using bf16x3::split3;
constexpr int kReluShift = 64;

void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, float* img) {
  memset(img, 0, sizeof(float) * kLayerFloats);
  unsigned short* a = reinterpret_cast<unsigned short*>(img + kA);
  for (int s = 0; s < 4; ++s)
    for (int m = 0; m < 2; ++m)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * (lane >> 5) + j, u = 32 * m + (lane & 31);
          unsigned short part[3];
          bf16x3::host_split3(ldexpf(W2[k * 64 + u], kReluShift), part);
          for (int p = 0; p < 3; ++p) a[((((p * 4 + s) * 2 + m) * 64) + lane) * 8 + j] = part[p];
        }
  for (int k = 0; k < 64; ++k) {
    img[kW1B1 + 2 * k + 0] = ldexpf(W1[cond_dim * 64 + k], -kReluShift);
    img[kW1B1 + 2 * k + 1] = ldexpf(b1[k], -kReluShift);
  }
  for (int h = 0; h < 2; ++h)
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) {
        const int u = 32 * m + 8 * (r >> 2) + 4 * h + (r & 3), q = (h * 2 + m) * 16 + r;
        img[kB2 + q] = b2[u];
        img[kW3 + 2 * q + 0] = W3[u * 2 + 0];
        img[kW3 + 2 * q + 1] = W3[u * 2 + 1];
      }
  img[kB3 + 0] = b3[0];
  img[kB3 + 1] = b3[1];
}

// ---- device ------------------------------------------------------------------------------
struct Params {
  const float* wimg;
  ChainIO io;
  int first_mask_parity;
};

__device__ __forceinline__ float add_halves(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x16 lds_f32x16(const float* p) {
  const float4* q = reinterpret_cast<const float4*>(p);
  float4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 v;
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
  return v;
}
__device__ __forceinline__ float exp_f32(float x) {
  const float kLog2eHi = 1.44269502e+00f, kLog2eLo = 1.92596299e-08f, kLn2 = 6.93147182e-01f;
  const float hi = x * kLog2eHi;
  const float lo = fmaf(x, kLog2eLo, fmaf(x, kLog2eHi, -hi));
  const float e = __builtin_amdgcn_exp2f(hi);
  return fmaf(e, lo * kLn2, e);
}

template <int T, int WAVES, int INV>
__global__ __launch_bounds__(WAVES * 64, WAVES >= 8 ? WAVES / 4 : 2) void chain_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // two layer images
  const ChainIO& io = p.io;
  const int tid = threadIdx.x;
  const int nlay = io.layer_end - io.layer_begin;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int F4 = kLayerFloats / 4;
  constexpr int PER = (F4 + NTHREADS - 1) / NTHREADS;
  static_assert(PER <= 7, "staging registers");

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long rows = io.rows;
  const long long groups = (rows + (T * 32 - 1)) / (T * 32);
  const long long groups_per_pass = (long long)gridDim.x * WAVES;
  const long long passes = (groups + groups_per_pass - 1) / groups_per_pass;
  const float2* __restrict__ xin = reinterpret_cast<const float2*>(io.x);

  auto layer_of = [&](int li) { return io.descending ? io.layer_end - 1 - li : io.layer_begin + li; };
  float4 st0, st1, st2, st3, st4, st5, st6;
  st0 = st1 = st2 = st3 = st4 = st5 = st6 = make_float4(0.f, 0.f, 0.f, 0.f);
#define NF_FETCH(li_)                                                                                  \
  {                                                                                                   \
    const float4* src_ = reinterpret_cast<const float4*>(p.wimg + (size_t)layer_of(li_) * kLayerFloats); \
    st0 = src_[tid < F4 ? tid : F4 - 1];                                                               \
    if (PER > 1) st1 = src_[tid + NTHREADS < F4 ? tid + NTHREADS : F4 - 1];                            \
    if (PER > 2) st2 = src_[tid + 2 * NTHREADS < F4 ? tid + 2 * NTHREADS : F4 - 1];                    \
    if (PER > 3) st3 = src_[tid + 3 * NTHREADS < F4 ? tid + 3 * NTHREADS : F4 - 1];                    \
    if (PER > 4) st4 = src_[tid + 4 * NTHREADS < F4 ? tid + 4 * NTHREADS : F4 - 1];                    \
    if (PER > 5) st5 = src_[tid + 5 * NTHREADS < F4 ? tid + 5 * NTHREADS : F4 - 1];                    \
    if (PER > 6) st6 = src_[tid + 6 * NTHREADS < F4 ? tid + 6 * NTHREADS : F4 - 1];                    \
  }
#define NF_STASH(buf_)                                                                \
  {                                                                                  \
    float4* dst_ = reinterpret_cast<float4*>(lds + (buf_) * kLayerFloats);             \
    if (tid < F4) dst_[tid] = st0;                                                   \
    if (PER > 1 && tid + NTHREADS < F4) dst_[tid + NTHREADS] = st1;                  \
    if (PER > 2 && tid + 2 * NTHREADS < F4) dst_[tid + 2 * NTHREADS] = st2;          \
    if (PER > 3 && tid + 3 * NTHREADS < F4) dst_[tid + 3 * NTHREADS] = st3;          \
    if (PER > 4 && tid + 4 * NTHREADS < F4) dst_[tid + 4 * NTHREADS] = st4;          \
    if (PER > 5 && tid + 5 * NTHREADS < F4) dst_[tid + 5 * NTHREADS] = st5;          \
    if (PER > 6 && tid + 6 * NTHREADS < F4) dst_[tid + 6 * NTHREADS] = st6;          \
  }
  NF_FETCH(0);
  NF_STASH(0);
  __syncthreads();

  double wave_sum = 0.0;
  long long step = 0;
  for (long long pass = 0; pass < passes; ++pass) {
    const long long g = (pass * gridDim.x + blockIdx.x) * WAVES + wave;
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
      const bool more = li + 1 < nlay || pass + 1 < passes;
      if (more) NF_FETCH(li + 1 < nlay ? li + 1 : 0);

      const float* L = lds + (step & 1) * kLayerFloats;
      const bf16x8* ap = reinterpret_cast<const bf16x8*>(L + kA) + lane;   // + ((p*4+s)*2+m)*64
      const float4* wb = reinterpret_cast<const float4*>(L + kW1B1) + 4 * h;  // + 8*s: (w1,b1) x 8 for k = 16s+8h..

      f32x16 acc[T][2];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        acc[t][0] = lds_f32x16(L + kB2 + (h * 2 + 0) * 16);
        acc[t][1] = lds_f32x16(L + kB2 + (h * 2 + 1) * 16);
      }
      // q = (k16-step s, tile t), t fastest.  The operand split of q+1 (52 VALU) is issued between
      // the 12 MFMAs of q: one MFMA occupies the vector issue for 8 of its 32 cycles, so up to six
      // VALU instructions hide behind each (profiles/microbench/mfma_valu_wave_roles.hip,
      // mfma_bf16_split_pipeline.hip).
      bf16x8 B[2][3];
      bf16x8 Ah[2], Am[2], Al[2];
      float w1[8], bb[8];
      auto make_B = [&](int q) {
        const int s = q / T, t = q % T;
        if (t == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 f = wb[8 * s + k];
            w1[2 * k] = f.x; bb[2 * k] = f.y; w1[2 * k + 1] = f.z; bb[2 * k + 1] = f.w;
          }
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(fmaf(xc[t], w1[j], bb[j]), 0.f, 1.f);
        split3(v, B[q & 1][0], B[q & 1][1], B[q & 1][2]);
      };
      make_B(0);
#pragma unroll
      for (int q = 0; q < 4 * T; ++q) {
        const int s = q / T, t = q % T;
        if (t == 0) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            Ah[m] = ap[((0 * 4 + s) * 2 + m) * 64];
            Am[m] = ap[((1 * 4 + s) * 2 + m) * 64];
            Al[m] = ap[((2 * 4 + s) * 2 + m) * 64];
          }
        }
        if (q + 1 < 4 * T) make_B(q + 1);
        const bf16x8 Bh = B[q & 1][0], Bm = B[q & 1][1], Bl = B[q & 1][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) bf16x3::mfma6(acc[t][m], Ah[m], Am[m], Al[m], Bh, Bm, Bl);
        if (q + 1 < 4 * T) {
#pragma unroll
          for (int g = 0; g < 12; ++g) {  // 52 VALU of the next split spread over these 12 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
      }

      const float4* w3 = reinterpret_cast<const float4*>(L + kW3 + h * 64);
      const float b3s = L[kB3 + 0], b3l = L[kB3 + 1];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float ps = 0.f, pl = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float4 w = w3[(m * 16 + r) / 2];
            const float v0 = fmaxf(acc[t][m][r], 0.f);
            const float v1 = fmaxf(acc[t][m][r + 1], 0.f);
            ps = fmaf(v0, w.x, ps);
            pl = fmaf(v0, w.y, pl);
            ps = fmaf(v1, w.z, ps);
            pl = fmaf(v1, w.w, pl);
          }
        const float shift = add_halves(ps) + b3s;
        const float ls = add_halves(pl) + b3l;
        float nt;
        if (INV) {
          nt = (xt[t] - shift) * exp_f32(-ls);
          ld[t] -= ls;
        } else {
          nt = fmaf(xt[t], exp_f32(ls), shift);
          ld[t] += ls;
        }
        xt[t] = xc[t];
        xc[t] = nt;
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
#undef NF_FETCH
#undef NF_STASH

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

template <int T, int WAVES, int INV>
static int launch_t(const Params& p, int grid, hipStream_t stream) {
  const size_t lds_bytes = 2 * (size_t)kLayerFloats * sizeof(float);
  NF_DYN_LDS(160 * 1024, &chain_kernel<T, WAVES, INV>);
  hipLaunchKernelGGL((chain_kernel<T, WAVES, INV>), dim3(grid), dim3(WAVES * 64), lds_bytes, stream, p);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

template <int T, int WAVES>
static int launch_dir(const Params& p, int grid, hipStream_t stream) {
  return p.io.inverse_op ? launch_t<T, WAVES, 1>(p, grid, stream) : launch_t<T, WAVES, 0>(p, grid, stream);
}

int launch(const float* wimg, int first_mask_parity, const ChainIO& io, hipStream_t stream, int* out_grid) {
  Params p;
  p.wimg = wimg;
  p.io = io;
  p.first_mask_parity = first_mask_parity;
  int T = 4, waves = 8;  // NF_AFFINE2D_X3_TUNE="T,WAVES": development knob (measured: profiles/x3_tune.py)
  if (const int* e = switches().affine2d_x3_tune; e[0]) { T = e[0]; waves = e[1]; }
  if (waves != 16 && waves != 4) waves = 8;
  if (T != 1 && T != 2 && T != 4) T = 2;
  if (waves == 16 && T != 1) T = 1;
  if (waves == 4 && T == 4) T = 2;  // 4 tiles at 256 registers spill
  // workgroups a CU holds at once: 8 waves -> 1 (register bound), 4 waves -> 2
  const int cap = waves == 4 ? kMaxGrid : 256;
  while (T > 1 && io.rows < (long long)cap * waves * 32 * T) T >>= 1;
  const long long groups = (io.rows + (T * 32 - 1)) / (T * 32);
  long long want = (groups + waves - 1) / waves;
  const int grid = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  if (out_grid) *out_grid = grid;
  if (waves == 16) return launch_dir<1, 16>(p, grid, stream);
  if (waves == 4) return T == 2 ? launch_dir<2, 4>(p, grid, stream) : launch_dir<1, 4>(p, grid, stream);
  if (T == 4) return launch_dir<4, 8>(p, grid, stream);
  return T == 2 ? launch_dir<2, 8>(p, grid, stream) : launch_dir<1, 8>(p, grid, stream);
}

}  // namespace affine2d_x3
}  // namespace nf
