// Fused spline-coupling chain with the optional "bf16x3" matmul (nf_flow_set_option
// "hidden_matmul" = "bf16x3"): nf_rqs2d_h64.hip with both of its MFMA passes -- the 64x64 hidden
// Linear and the 64 -> 3K+1 parameter Linear -- done as six v_mfma_f32_32x32x16_bf16 on three-way
// bf16 splits of both operands (see nf_affine2d_x3.hip for the arithmetic and why it pays).
//
// What carries over unchanged: first Linear generated in operand layout with the clamp-bit relu,
// parameter rows assigned to half-waves, the spline (nf_rqs2d_common.h), layer images streamed
// through an LDS double buffer.  What is specific:
//   * second pass: one 16-deep k step = eight consecutive accumulator registers of the first
//     pass (m, g): the lane already holds relu(acc[m][8g .. 8g+7]) for its sample, which is
//     exactly its B operand once the weight image's k order is permuted to
//     j = 32m + 16g + 8(r>>2) + 4h + (r&3).  No lane movement, no LDS; one operand split
//     (52 VALU) per k step and tile.
//   * with 32-cycle MFMAs the spline's ~600 VALU instructions are no longer serialised behind
//     the matrix pipe (the f32 MFMA blocks the vector issue for its whole 64 cycles, the bf16
//     one for 8 of 32): the kernel is bound by vector issue, not by the MFMA.
#include "nf_bf16x3.h"
#include "nf_rqs2d_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace nf {
namespace rqs2d_x3 {

using rqs2d::Params;
using rqs2d::f32x16;
using rqs2d::lds_f32x16;
using rqs2d::param_of_slot;
using rqs2d::spline;

using bf16x3::bf16x8;
using bf16x3::split3;
constexpr int kReluShift = 64;

// ---- host ------------------------------------------------------------------------------------
static void split_store(float w, unsigned short* base, size_t part_stride, size_t at) {
  unsigned short part[3];
  bf16x3::host_split3(w, part);
  base[at] = part[0];
  base[part_stride + at] = part[1];
  base[2 * part_stride + at] = part[2];
}

void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, int K, float* img) {
  const int P = 3 * K + 1;
  const int dT = 1 - cond_dim;  // the transformed dim
  memset(img, 0, sizeof(float) * kLayerFloats);
  constexpr size_t kPart = 4 * 2 * 64 * 8;  // bf16 per part
  unsigned short* a2 = reinterpret_cast<unsigned short*>(img + kA2);
  unsigned short* a3 = reinterpret_cast<unsigned short*>(img + kA3);
  for (int s = 0; s < 4; ++s)
    for (int m = 0; m < 2; ++m)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int hk = lane >> 5, u = lane & 31;
          const size_t at = (((size_t)s * 2 + m) * 64 + lane) * 8 + j;
          // pass 1: k = 16 s + 8 hk + j, output unit 32 m + u
          split_store(ldexpf(W2[(16 * s + 8 * hk + j) * 64 + 32 * m + u], kReluShift), a2, kPart, at);
          // pass 2: k step s = (mk, g) of the accumulator, hidden unit in accumulator order;
          // output row u of tile m lands in half hD, register rr of the result
          const int mk = s >> 1, g = s & 1;
          const int hid = 32 * mk + 16 * g + 8 * (j >> 2) + 4 * hk + (j & 3);
          const int hD = (u >> 2) & 1, rr = (u & 3) + 4 * (u >> 3);
          const int prm = param_of_slot(hD, 16 * m + rr, K);
          split_store(prm >= 0 ? W3[(size_t)hid * (2 * P) + dT * P + prm] : 0.f, a3, kPart, at);
        }
  for (int k = 0; k < 64; ++k) {
    img[kW1B1 + 2 * k + 0] = ldexpf(W1[cond_dim * 64 + k], -kReluShift);
    img[kW1B1 + 2 * k + 1] = ldexpf(b1[k], -kReluShift);
  }
  for (int h = 0; h < 2; ++h)
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) {
        img[kB2 + (h * 2 + m) * 16 + r] = b2[32 * m + 8 * (r >> 2) + 4 * h + (r & 3)];
        const int prm = param_of_slot(h, 16 * m + r, K);
        img[kB3 + (h * 2 + m) * 16 + r] = prm >= 0 ? b3[dT * P + prm] : 0.f;
      }
}

// ---- device ----------------------------------------------------------------------------------
// acc += A (3 parts) x B (3 parts)
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&A)[3], const bf16x8 (&B)[3]) {
  bf16x3::mfma6(acc, A[0], A[1], A[2], B[0], B[1], B[2]);
}

template <int T, int WAVES, int INV, int K>
__global__ __launch_bounds__(WAVES * 64, 1) void chain_kernel(Params p) {
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
#define NF_LD(n_, dst_) \
  if (PER > n_) dst_ = src_[tid + n_ * NTHREADS < F4 ? tid + n_ * NTHREADS : F4 - 1];
#define NF_ST(n_, src__) \
  if (PER > n_ && tid + n_ * NTHREADS < F4) dst_[tid + n_ * NTHREADS] = src__;
#define NF_FETCH(li_)                                                                                  \
  {                                                                                                   \
    const float4* src_ = reinterpret_cast<const float4*>(p.wimg + (size_t)layer_of(li_) * kLayerFloats); \
    NF_LD(0, st0) NF_LD(1, st1) NF_LD(2, st2) NF_LD(3, st3) NF_LD(4, st4) NF_LD(5, st5) NF_LD(6, st6)  \
  }
#define NF_STASH(buf_)                                                                \
  {                                                                                  \
    float4* dst_ = reinterpret_cast<float4*>(lds + (buf_) * kLayerFloats);             \
    NF_ST(0, st0) NF_ST(1, st1) NF_ST(2, st2) NF_ST(3, st3) NF_ST(4, st4) NF_ST(5, st5) NF_ST(6, st6) \
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
      const bf16x8* a2 = reinterpret_cast<const bf16x8*>(L + kA2) + lane;  // + ((part*4+s)*2+m)*64
      const bf16x8* a3 = reinterpret_cast<const bf16x8*>(L + kA3) + lane;
      const float4* wb = reinterpret_cast<const float4*>(L + kW1B1) + 4 * h;  // + 8 s

      // ---- pass 1: hidden Linear; q = (k step s, tile t), the operand split of q+1 issued
      //      between the MFMAs of q (nf_affine2d_x3.hip)
      f32x16 acc[T][2];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        acc[t][0] = lds_f32x16(L + kB2 + (h * 2 + 0) * 16);
        acc[t][1] = lds_f32x16(L + kB2 + (h * 2 + 1) * 16);
      }
      {
        bf16x8 B[2][3], A[2][3];
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
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int part = 0; part < 3; ++part) A[m][part] = a2[((part * 4 + s) * 2 + m) * 64];
          }
          if (q + 1 < 4 * T) make_B(q + 1);
          mfma6(acc[t][0], A[0], B[q & 1]);
          mfma6(acc[t][1], A[1], B[q & 1]);
          if (q + 1 < 4 * T) {
#pragma unroll
            for (int gq = 0; gq < 12; ++gq) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (gq < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
              else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          }
        }
      }

      // ---- pass 2: parameter Linear; k step (m, g) = accumulator registers acc[m][8g .. 8g+7]
      f32x16 out[T][2];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        out[t][0] = lds_f32x16(L + kB3 + (h * 2 + 0) * 16);
        out[t][1] = lds_f32x16(L + kB3 + (h * 2 + 1) * 16);
      }
      {
        bf16x8 B[2][3], A[2][3];
        auto make_B = [&](int q) {
          const int s = q / T, t = q % T;
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = fmaxf(acc[t][s >> 1][8 * (s & 1) + r], 0.f);
          split3(v, B[q & 1][0], B[q & 1][1], B[q & 1][2]);
        };
        make_B(0);
#pragma unroll
        for (int q = 0; q < 4 * T; ++q) {
          const int s = q / T, t = q % T;
          if (t == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int part = 0; part < 3; ++part) A[m][part] = a3[((part * 4 + s) * 2 + m) * 64];
          }
          if (q + 1 < 4 * T) make_B(q + 1);
          mfma6(out[t][0], A[0], B[q & 1]);
          mfma6(out[t][1], A[1], B[q & 1]);
          if (q + 1 < 4 * T) {
#pragma unroll
            for (int gq = 0; gq < 12; ++gq) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (gq < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
              else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          }
        }
      }

#pragma unroll
      for (int t = 0; t < T; ++t) {
        float slot[32], lg[K], sg[K + 1];
#pragma unroll
        for (int q = 0; q < 16; ++q) { slot[q] = out[t][0][q]; slot[16 + q] = out[t][1][q]; }
#pragma unroll
        for (int k = 0; k < K; ++k) lg[k] = slot[k];
#pragma unroll
        for (int k = 0; k <= K; ++k) sg[k] = slot[K + k];
        float nt, l;
        spline<K, INV>(lg, sg, xt[t], h, p, &nt, &l);
        ld[t] += l;
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
#undef NF_LD
#undef NF_ST
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
  int T = 2, waves = 8;  // NF_RQS2D_X3_TUNE="T,WAVES": development knob (measured: profiles/bench_configs.py c)
  if (const int* e = switches().rqs2d_x3_tune; e[0]) { T = e[0]; waves = e[1]; }
  if (waves != 8) { waves = 16; T = 1; }
  if (T != 2) T = 1;
  const int cap = 256;  // one workgroup per CU: two images are 100 KB of LDS
  if (T == 2 && io.rows < (long long)cap * waves * 64) T = 1;  // not enough rows for two tiles a wave
  const long long groups = (io.rows + (T * 32 - 1)) / (T * 32);
  long long want = (groups + waves - 1) / waves;
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

}  // namespace rqs2d_x3
}  // namespace nf
