// Candidate geometry for gemm_x3_kernel: block 128 x 256, four waves of 128 x 64 (4 x 2 MFMA tiles, 128
// accumulator registers): every B fragment is loaded by exactly one wave (straight from the L2-resident
// image), A (split on the fly, two LDS stages) is shared by all four.  Per MFMA: half the vector-memory
// bytes and half the split work of the 128 x 128 / 64 x 64 geometry.  Same workload as gemm_x3_probe.hip.
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I ../../normalizing_flows_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "nf_bf16x3.h"
#include "nf_internal.h"  // block_to_tile
using nf::bf16x3::bf16x8;
using nf::bf16x3::f32x16;
using nf::bf16x3::split_pair;
constexpr int XBM = 128, XBK = 32, kTileU4 = 1536;

template <int OCC>
__global__ __launch_bounds__(256, OCC) void wide(const float* __restrict__ A, int lda, const uint4* __restrict__ Wimg,
                                               float* __restrict__ Cm, int ldc, int M, int N, int K, int gx, int gy) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_u4[];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, i = lane & 31, h = lane >> 5;
  int mb, nb;
  nf::generic::block_to_tile(blockIdx.x, gx, gy, mb, nb);
  const int m0 = mb * XBM, n0 = nb * 256;
  const int nk = K / XBK;
  const int c = tid & 7, ar0 = tid >> 3;
  const float* aline_ptr = A + (size_t)(m0 + ar0) * lda + c * 4;
  // this wave's two column tiles: 128-column image 2 nb + (wn >> 1), tiles (wn & 1) * 2 + {0, 1}
  const uint4* wsrc = Wimg + (size_t)(2 * nb + (wn >> 1)) * nk * kTileU4 + (wn & 1) * 128 + lane;
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 ra0, ra1, ra2, ra3;
  uint4 b0[2][3], b1[2][3];
#define FETCH(kt_) { const float* p_ = aline_ptr + (kt_) * XBK; ra0 = *reinterpret_cast<const float4*>(p_); \
    ra1 = *reinterpret_cast<const float4*>(p_ + (size_t)32 * lda); ra2 = *reinterpret_cast<const float4*>(p_ + (size_t)64 * lda); \
    ra3 = *reinterpret_cast<const float4*>(p_ + (size_t)96 * lda); }
#define STASH1(r_, e_) { unsigned h0_, m0_, l0_, h1_, m1_, l1_; split_pair(r_.x, r_.y, h0_, m0_, l0_); split_pair(r_.z, r_.w, h1_, m1_, l1_); \
    uint2* d_ = sbase_ + (size_t)(32 * (e_)) * 4; d_[0] = make_uint2(h0_, h1_); d_[2 * 128 * 4] = make_uint2(m0_, m1_); d_[4 * 128 * 4] = make_uint2(l0_, l1_); }
#define STASH(stage_) { uint2* sbase_ = reinterpret_cast<uint2*>(lds_u4 + (stage_) * kTileU4) + ((size_t)((c >> 2) * 128 + ar0) * 4 + (c & 3)); \
    STASH1(ra0, 0) STASH1(ra1, 1) STASH1(ra2, 2) STASH1(ra3, 3) }
#define BLOAD(dst_, kt_, ks_) { const uint4* w_ = wsrc + (size_t)(kt_) * kTileU4 + (ks_) * 256; \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) dst_[t_][p_] = w_[p_ * 512 + t_ * 64]; }
#define STEP(stage_, ks_, B_) { const uint4* As_ = lds_u4 + (stage_) * kTileU4; \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { bf16x8 Af_[3]; \
      _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) Af_[p_] = __builtin_bit_cast(bf16x8, As_[(((p_ * 2 + (ks_)) * 128) + t_ * 32 + i) * 2 + h]); \
      _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) nf::bf16x3::mfma6(acc[t_][u_], Af_[0], Af_[1], Af_[2], __builtin_bit_cast(bf16x8, B_[u_][0]), \
                      __builtin_bit_cast(bf16x8, B_[u_][1]), __builtin_bit_cast(bf16x8, B_[u_][2])); } }
  BLOAD(b0, 0, 0);
  FETCH(0);
  STASH(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) FETCH(kt + 1);
    BLOAD(b1, kt, 1);
    STEP(cur, 0, b0);
    if (more) BLOAD(b0, kt + 1, 0);
    STEP(cur, 1, b1);
    if (more) STASH(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = n0 + wn * 64 + u * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        Cm[(size_t)row * ldc + col] = acc[t][u][r];
      }
    }
}

template <int OCC>
static void run(const char* what, const float* A, const uint4* W, float* C, int M, int N, int K) {
  const int gx = N / 256, gy = M / XBM;
  const size_t lds = 2 * kTileU4 * sizeof(uint4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wide<OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((wide<OCC>), dim3(gx * gy), dim3(256), lds, 0, A, K, W, C, N, M, N, K, gx, gy);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) best = ms < best ? ms : best;
  }
  const double mfma = (double)M / 32 * N / 32 * K / 16 * 6;
  printf("wide occ=%d %-30s %7.3f ms   %6.1f %% of the bf16 MFMA issue rate at 2.4 GHz  (%s)\n", OCC, what, best,
         100.0 * mfma * 32 / 1024 / 2.4e9 / (best * 1e-3), hipGetErrorString(hipGetLastError()));
}

int main() {
  const int M = 65536, K = 512, N = 10240;   // 40 blocks of 256 columns
  float* A; uint4* W; float* C;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&W, (size_t)(N / 128) * (K / XBK) * kTileU4 * 16);
  hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> ha((size_t)M * K);
  unsigned s = 12345u;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.3f) * 1.7f; }
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned short> hw((size_t)(N / 128) * (K / XBK) * kTileU4 * 8);
  for (size_t q = 0; q < hw.size(); ++q) {
    s = s * 1664525u + 1013904223u;
    const int part = (int)((q / 8 / 512) % 3);
    const float v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * (part == 0 ? 0.1f : part == 1 ? 4e-4f : 1.5e-6f);
    hw[q] = nf::bf16x3::host_bf16_rne(v);
  }
  hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  run<2>("2 workgroups per CU", A, W, C, M, N, K);
  run<1>("1 workgroup per CU", A, W, C, M, N, K);
  return 0;
}
