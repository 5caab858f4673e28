// One Linear of the general path at SMALL batches (M <= 256 rows: the reference trains and evaluates its image flow at
// batch 128, src/maf_mnist.py:159-171,228-230), in ONE launch: C[M, N] = act(A[M, K] W[K, N] + bias).
//
// What this replaces: gemm_splitk_kernel + gemm_splitk_reduce_kernel (nf_generic.hip) -- the 128 x 128 tile loop split
// over K with its partial slabs through HBM and a second launch to add them: 13.4 + 5.1 us for a [128, 512] x [512, 512]
// product and 24 + 7 us for the [128, 512] x [512, 9800] one (rocprofv3 kernel trace, gpurun_out/r04/mnist_small_kt:
// 65 % + 22 % of a 765 us log_prob call of the 8-layer MNIST-shaped flow at batch 128).  Those kernels are built for
// big tiles and long k loops; a 128-row product is neither: it is 67 MFLOP (0.4 us of the chip's f32 MFMA rate), so
// what a launch costs is its latency chain -- loads, a handful of MFMAs, the stores -- and how many CUs pull the
// weights at once.
//
// Here: one workgroup = one 32-row block x NT 32-column blocks, SIXTEEN waves that split the workgroup's (column block,
// k range) pairs between them: each wave loads its operands straight into MFMA operand registers (no LDS staging, no
// barrier in the k loop; A as one 16-byte load per lane and eight k, W as four 128-byte row segments per half-wave),
// runs K / (16 / NT) / 2 dependent v_mfma_f32_32x32x2_f32 and leaves its raw tile in LDS; the k groups' tiles are added
// in a fixed order (run-to-run identical results, like the slabs they replace), bias and activation applied, rows
// stored 128 bytes per half-wave.  [128, 512] x [512, 512]: 64 workgroups, 16 MFMAs per wave.
#include "nf_internal.h"

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSmallMWaves = 16;
constexpr int kSmallMChunk = 4;   // octets (8 k each) per software-pipeline stage: K = 512 over sixteen waves is ONE stage
                                  // (all of a wave's loads in flight at once; with 2 the launch took 11.4 us)

template <int NT>
__global__ __launch_bounds__(kSmallMWaves * 64) void gemm_small_m_kernel(const float* __restrict__ A, int lda,
                                                                        const float* __restrict__ W,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ Cm, int ldc, int M, int N,
                                                                        int K, int act, const float* __restrict__ aux) {
  constexpr int KG = kSmallMWaves / NT;   // k groups per column block
  __shared__ float red[kSmallMWaves][16][64];   // every wave's raw 32 x 32 tile (64 KB)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % NT, kg = wave / NT;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.y * 32;
  const int nblk = blockIdx.x * (NT * 32);
  const int n0 = nblk + nt * 32;
  // k in octets (8 consecutive k: lane (row, kk) holds A[row][8 o + 4 kk .. + 3], MFMA j of the octet multiplies k slots
  // (8 o + j | 8 o + 4 + j)); a wave's octets are contiguous
  const int octets = (K + 7) >> 3;
  const int per = (octets + KG - 1) / KG;
  const int o_begin = kg * per;
  const int o_end = min(octets, o_begin + per);
  const int row = min(m0 + i, M - 1);       // rows / columns past the edge are computed on clamped addresses, never stored
  const int col = min(n0 + i, N - 1);
  const float* ap = A + (size_t)row * lda + 4 * kk;
  const float* wp = W + col;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 a[2][kSmallMChunk];
  float b[2][kSmallMChunk][4];
  auto fetch = [&](int buf, int o0) {
#pragma unroll
    for (int c = 0; c < kSmallMChunk; ++c) {
      const int o = o0 + c;
      const int k = 8 * min(o, octets - 1) + 4 * kk;     // this lane's four k of the octet
      const bool live = o < o_end && k < K;              // (K % 4 == 0: an octet's second half may lie past K)
      const int kc = min(k, K - 4);
      float4 v = *reinterpret_cast<const float4*>(ap + (kc - 4 * kk));
      if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);    // a zero A operand: the k slot contributes nothing
      a[buf][c] = v;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[buf][c][j] = wp[(size_t)(kc + j) * N];
    }
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int c = 0; c < kSmallMChunk; ++c) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][c].x, b[buf][c][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][c].y, b[buf][c][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][c].z, b[buf][c][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][c].w, b[buf][c][3], acc, 0, 0, 0);
    }
  };
  // two stages per trip: the buffer index is then a compile-time constant (indexed at run time the operand arrays went
  // to scratch memory, 144 bytes per lane)
  if (o_begin < o_end) {
    fetch(0, o_begin);
    for (int o = o_begin; o < o_end; o += 2 * kSmallMChunk) {
      const bool second = o + kSmallMChunk < o_end;   // wave-uniform
      if (second) fetch(1, o + kSmallMChunk);
      multiply(0);
      if (o + 2 * kSmallMChunk < o_end) fetch(0, o + 2 * kSmallMChunk);
      if (second) multiply(1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // 1024 threads, NT * 1024 outputs: thread (r, l) of column block e adds the k groups' tiles in order
  const int r = tid >> 6, l = lane;
  const int orow = m0 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
#pragma unroll
  for (int e = 0; e < NT; ++e) {
    const int ocol = nblk + e * 32 + (l & 31);
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < KG; ++g) v += red[g * NT + e][r][l];
    if (orow < M && ocol < N) {
      v += bias[ocol];
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
      else if (act >= 3) v = aux[(size_t)orow * ldc + ocol] > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
      Cm[(size_t)orow * ldc + ocol] = v;
    }
  }
}

// The same product on 16 x 16 tiles (v_mfma_f32_16x16x4_f32): four times the workgroups of the 32 x 32 form -- [128, 512]
// x [512, 512] covers all 256 CUs instead of 64, a wave's share of K is 8 MFMAs of 32 cycles instead of 16 of 64, a
// workgroup's operands are 64 KB instead of 128 KB.  (The launch boundary itself is <= 2.9 us of these figures.)
// Lane (m = lane & 15, kq = lane >> 4) of a wave holds, per group of 16 k, A[row m][16 g + 4 kq .. + 3] (one 16-byte load)
// and W[16 g + 4 kq + j][column m] for j < 4; MFMA j of the group multiplies k slot kq with k = 16 g + 4 kq + j on both
// sides.  Result register r of lane l: row 4 (l >> 4) + r, column l & 15.
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kSmallM16Groups = 2;   // groups of 16 k per software-pipeline stage

__global__ __launch_bounds__(kSmallMWaves * 64) void gemm_small_m16_kernel(const float* __restrict__ A, int lda,
                                                                          const float* __restrict__ W,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ Cm, int ldc, int M, int N,
                                                                          int K, int act, const float* __restrict__ aux,
                                                                          float* __restrict__ a_copy) {
  __shared__ float red[kSmallMWaves][4][64];   // every wave's raw 16 x 16 tile (16 KB)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int groups = (K + 15) >> 4;
  const int per = (groups + kSmallMWaves - 1) / kSmallMWaves;
  const int g_begin = wave * per;
  const int g_end = min(groups, g_begin + per);
  const int row = min(m0 + i, M - 1);
  const int col = min(n0 + i, N - 1);
  const float* ap = A + (size_t)row * lda;
  const float* wp = W + col;
  const bool copy_a = a_copy != nullptr && blockIdx.x == 0 && m0 + i < M;

  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  float4 a[2][kSmallM16Groups];
  float b[2][kSmallM16Groups][4];
  auto fetch = [&](int buf, int g0) {
#pragma unroll
    for (int c = 0; c < kSmallM16Groups; ++c) {
      const int g = g0 + c;
      const int k = 16 * min(g, groups - 1) + 4 * kq;
      const bool live = g < g_end && k < K;
      const int kc = min(k, K - 4);
      float4 v = *reinterpret_cast<const float4*>(ap + kc);
      // training: the first column block also leaves A in the tape (every element of its rows passes through exactly
      // one live lane) -- the layer's input, without a copy launch
      if (copy_a && live) *reinterpret_cast<float4*>(a_copy + (size_t)row * lda + kc) = v;
      if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
      a[buf][c] = v;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[buf][c][j] = wp[(size_t)(kc + j) * N];
    }
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int c = 0; c < kSmallM16Groups; ++c) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][c].x, b[buf][c][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][c].y, b[buf][c][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][c].z, b[buf][c][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][c].w, b[buf][c][3], acc, 0, 0, 0);
    }
  };
  // two stages per trip so that the buffer index is a constant (a run-time index would put the operands in scratch)
  if (g_begin < g_end) {
    fetch(0, g_begin);
    for (int g = g_begin; g < g_end; g += 2 * kSmallM16Groups) {
      const bool second = g + kSmallM16Groups < g_end;   // wave-uniform
      if (second) fetch(1, g + kSmallM16Groups);
      multiply(0);
      if (g + 2 * kSmallM16Groups < g_end) fetch(0, g + 2 * kSmallM16Groups);
      if (second) multiply(1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (tid < 256) {   // 256 outputs: thread (r, l) adds the sixteen waves' tiles in order
    const int r = tid >> 6, l = tid & 63;
    const int orow = m0 + 4 * (l >> 4) + r, ocol = n0 + (l & 15);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kSmallMWaves; ++w) v += red[w][r][l];
    if (orow < M && ocol < N) {
      v += bias[ocol];
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
      else if (act == 5) v += Cm[(size_t)orow * ldc + ocol];   // C += A W (the backward walk's d x)
      else if (act >= 3) v = aux[(size_t)orow * ldc + ocol] > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
      Cm[(size_t)orow * ldc + ocol] = v;
    }
  }
}

// Shapes the kernel takes: few rows, a k range that sixteen waves can share without starving and short enough that
// one workgroup's share of W -- K x 32 NT floats -- does not sit on one CU's memory pipe (the data-gradient product of
// the MNIST shape's last Linear, K = 9800, keeps the split-K form), 16-byte rows of A -- and N <= 1024: every operand
// element is used by 32 MFMA rows or columns only (no LDS tile shares it further), which is the right trade for the
// hidden Linears (in a dependent chain outside the profiler, profiles/microbench/small_m_chain.hip: 3.7 us per
// [128, 512] x [512, 512] launch and 4.6 us at K = 784 on 16 x 16 tiles, 6.7 / 10.7 us on 32 x 32 tiles; slabs + reduction
// took 13.4 + 5.1 us in the kernel trace) and the wrong one for the MNIST shape's last Linear (N = 9800: 157 MB of
// L2 -> CU traffic; NF_SMALL_M_NT=8 routes it here: the 8-layer log_prob call at batch 128 takes 542 us against 369).
// Until round 4's last day both kernels indexed their operand double buffer with a run-time variable, which put the
// buffer in scratch memory (144 bytes per lane): 8.9 / 14.8 us per launch on 32 x 32 tiles, 28 us on 16 x 16 tiles.
bool gemm_small_m_applicable(const float* A, int lda, int M, int N, int K) {
  return !switches().no_small_m && M >= 1 && M <= 256 && K >= 64 && K <= 2048 && N >= 32 &&
         (N <= 1024 || switches().small_m_nt != 0) && (lda & 3) == 0 && (K & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

// The 16 x 16 form (one resident round of small tiles) is the one that can also copy A (a_copy) and add into C (act 5).
bool gemm_small_m16_applicable(const float* A, int lda, int M, int N, int K) {
  return gemm_small_m_applicable(A, lda, M, N, K) && (long long)((N + 15) / 16) * ((M + 15) / 16) <= 512 &&
         (switches().small_m_nt == 0 || switches().small_m_nt == 8);
}

int launch_gemm_small_m(const float* A, int lda, const float* W, const float* bias, float* Cm, int ldc, int M, int N,
                        int K, int act, hipStream_t stream, const float* aux, float* a_copy) {
  // small enough for one resident round of 16 x 16 tiles: all 256 CUs take part (NF_SMALL_M_NT set: the 32 x 32 forms)
  if (gemm_small_m16_applicable(A, lda, M, N, K)) {
    hipLaunchKernelGGL(gemm_small_m16_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(kSmallMWaves * 64), 0, stream, A, lda, W,
                       bias, Cm, ldc, M, N, K, act, aux, a_copy);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  if (a_copy != nullptr || act == 5) { set_error("this Linear's shape has no copying / accumulating form"); return NF_ERR_UNSUPPORTED; }
  const int gy = (M + 31) / 32;
  // wider column blocks once the grid fills the chip anyway: A's rows are then fetched by fewer workgroups
  int NT = 1;
  if ((long long)((N + 127) / 128) * gy >= 256) NT = 4;
  else if ((long long)((N + 63) / 64) * gy >= 256) NT = 2;
  if (switches().small_m_nt == 1 || switches().small_m_nt == 2 || switches().small_m_nt == 4) NT = switches().small_m_nt;
  const dim3 grid((N + 32 * NT - 1) / (32 * NT), gy);
  if (NT == 4)
    hipLaunchKernelGGL(gemm_small_m_kernel<4>, grid, dim3(kSmallMWaves * 64), 0, stream, A, lda, W, bias, Cm, ldc, M, N, K, act, aux);
  else if (NT == 2)
    hipLaunchKernelGGL(gemm_small_m_kernel<2>, grid, dim3(kSmallMWaves * 64), 0, stream, A, lda, W, bias, Cm, ldc, M, N, K, act, aux);
  else
    hipLaunchKernelGGL(gemm_small_m_kernel<1>, grid, dim3(kSmallMWaves * 64), 0, stream, A, lda, W, bias, Cm, ldc, M, N, K, act, aux);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
