// General path, optional "bf16x3" conditioner matmul (nf_flow_set_option "hidden_matmul"):
//   C[M,N] = act(A[M,K] @ W[K,N] + bias[N])
// with both operands split three ways into bf16 (x = hi + mid + lo, round-to-nearest at each
// level, 24 significant bits) and six v_mfma_f32_32x32x16_bf16 per 32x32x16 block
// (hh, hm, mh, hl, lh, mm; float32 accumulate; the dropped terms are <= 2^-25 relative).
// See nf_affine2d_x3.hip for the reasoning and DESIGN.md section 4 for the measurements.
//
// Layout.  Samples are the MFMA A operand (rows), output units the B operand (columns), so a
// lane's accumulator registers hold 32 consecutive output units of one sample row: coalesced
// float32 stores, as in gemm_fast_kernel.
//   * W is split and laid out ONCE on the host (pack_w_x3): per (128-column block, 32-deep k
//     tile) one contiguous 24 KiB image [part 3][k16 2][column tile 4][lane 64][8 bf16] that is
//     copied to LDS verbatim and read back with one ds_read_b128 per operand.
//   * A stays float32 row-major in HBM (it is the previous Linear's output, or the sample
//     matrix).  Each thread loads 2 x 8 consecutive k of one row (4 x float4; a row's 128-byte
//     line is shared by two lanes), splits them in registers (11 VALU per pair of values, hidden behind the
//     MFMAs of the other workgroup on the CU) and writes [part][k16][row 128][16 bf16] to LDS --
//     with a 16-deep k step the row-major bf16 tile IS the A operand order.
// 128x128 block tile, BK = 32, 4 waves of 64x64, one 48 KiB LDS stage + the next tile in
// registers, three workgroups per CU (144 KiB of LDS, <= 168 registers).
#include "nf_bf16x3.h"
#include "nf_internal.h"
#include "nf_rqs_eval.h"

#include <math.h>
#include <string.h>

namespace nf {
namespace generic {

using bf16x3::bf16x8;
using bf16x3::f32x16;
using bf16x3::split_pair;

constexpr int XBM = 128, XBN = 128, XBK = 32;
constexpr int kTileU4 = 3 * 2 * 4 * 64;  // uint4 (8 bf16) per W tile image = 1536 (24 KiB)

// ---- host ------------------------------------------------------------------------------------
size_t x3_image_halfwords(int K, int N) {
  const size_t nb = (N + XBN - 1) / XBN, nkt = (K + XBK - 1) / XBK;
  return nb * nkt * (size_t)kTileU4 * 8;
}

void pack_w_x3(const float* W, int K, int N, unsigned short* img) {
  const int nb = (N + XBN - 1) / XBN, nkt = (K + XBK - 1) / XBK;
  memset(img, 0, x3_image_halfwords(K, N) * sizeof(unsigned short));
  for (int b = 0; b < nb; ++b)
    for (int kt = 0; kt < nkt; ++kt) {
      unsigned short* tile = img + ((size_t)b * nkt + kt) * kTileU4 * 8;
      for (int ks = 0; ks < 2; ++ks)
        for (int nt = 0; nt < 4; ++nt)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int k = kt * XBK + ks * 16 + 8 * (lane >> 5) + j;
              const int n = b * XBN + nt * 32 + (lane & 31);
              if (k >= K || n >= N) continue;
              unsigned short part[3];
              bf16x3::host_split3(W[(size_t)k * N + n], part);
              for (int p = 0; p < 3; ++p) tile[((((p * 2 + ks) * 4 + nt) * 64) + lane) * 8 + j] = part[p];
            }
    }
}

// ---- device ----------------------------------------------------------------------------------
// The k loop shared by gemm_x3_kernel and gemm_x3_rqs_kernel: acc[t][u] = the wave's 2 x 2 tiles of
// A[m0.., :] @ W[:, 128 nb ..]; returns after a barrier (LDS free).
__device__ __forceinline__ void gemm_x3_mainloop(const float* __restrict__ A, int lda, const uint4* __restrict__ Wimg,
                                                 int M, int K, int m0, int nb, uint4* lds_u4, f32x16 (&acc)[2][2]) {
  uint4* As = lds_u4;             // [part 3][k16 2][row 128][2]   (2 uint4 = 16 bf16 per row)
  uint4* Ws = lds_u4 + kTileU4;   // [part 3][k16 2][ntile 4][lane 64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (K + XBK - 1) / XBK;

  // this thread's share of an A tile: row arow, k = 16 ks + 8 akq + 0..7 for ks = 0, 1 -- so that the
  // 64 lanes of one ds_write_b128 cover one contiguous KiB of the [k16][row][16] plane
  const int arow = tid >> 1, akq = tid & 1;
  const float* arow_ptr = A + (size_t)(m0 + arow) * lda;
  const bool arow_ok = m0 + arow < M;
  const uint4* wsrc = Wimg + (size_t)nb * nk * kTileU4 + tid;

  // the next tile, held in named registers (arrays captured by a lambda end up in scratch)
  float4 ra0, ra1, ra2, ra3;
  uint4 rw0, rw1, rw2, rw3, rw4, rw5;
#define NF_A_LOAD(e_, dst_)                                                                             \
  {                                                                                                     \
    const int gk_ = kt_ * XBK + ((e_) >> 1) * 16 + akq * 8 + ((e_) & 1) * 4;                            \
    dst_ = (arow_ok && gk_ < K) ? *reinterpret_cast<const float4*>(arow_ptr + gk_) : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
#define NF_FETCH(kt_expr)                                   \
  {                                                         \
    const int kt_ = (kt_expr);                              \
    NF_A_LOAD(0, ra0) NF_A_LOAD(1, ra1) NF_A_LOAD(2, ra2) NF_A_LOAD(3, ra3) \
    const uint4* w_ = wsrc + (size_t)kt_ * kTileU4;         \
    rw0 = w_[0]; rw1 = w_[256]; rw2 = w_[512]; rw3 = w_[768]; rw4 = w_[1024]; rw5 = w_[1280]; \
  }
#define NF_SPLIT4(r_, q_)                                          \
  split_pair(r_.x, r_.y, hi##q_##a, mid##q_##a, lo##q_##a);        \
  split_pair(r_.z, r_.w, hi##q_##b, mid##q_##b, lo##q_##b);
#define NF_STASH()                                                                  \
  {                                                                                 \
    unsigned hi0a, hi0b, hi1a, hi1b, hi2a, hi2b, hi3a, hi3b;                        \
    unsigned mid0a, mid0b, mid1a, mid1b, mid2a, mid2b, mid3a, mid3b;                \
    unsigned lo0a, lo0b, lo1a, lo1b, lo2a, lo2b, lo3a, lo3b;                        \
    NF_SPLIT4(ra0, 0) NF_SPLIT4(ra1, 1) NF_SPLIT4(ra2, 2) NF_SPLIT4(ra3, 3)         \
    uint4* dst_ = As + tid; /* = (row * 2 + akq) of k16 step 0; step 1 is 256 further */ \
    dst_[0] = make_uint4(hi0a, hi0b, hi1a, hi1b);                                   \
    dst_[256] = make_uint4(hi2a, hi2b, hi3a, hi3b);                                 \
    dst_[512] = make_uint4(mid0a, mid0b, mid1a, mid1b);                             \
    dst_[768] = make_uint4(mid2a, mid2b, mid3a, mid3b);                             \
    dst_[1024] = make_uint4(lo0a, lo0b, lo1a, lo1b);                                \
    dst_[1280] = make_uint4(lo2a, lo2b, lo3a, lo3b);                                \
    Ws[tid] = rw0; Ws[tid + 256] = rw1; Ws[tid + 512] = rw2;                        \
    Ws[tid + 768] = rw3; Ws[tid + 1024] = rw4; Ws[tid + 1280] = rw5;                \
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  NF_FETCH(0);
  NF_STASH();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) NF_FETCH(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 Af[2][3], Bf[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          Af[t][p] = __builtin_bit_cast(bf16x8, As[(((p * 2 + ks) * 128) + wm * 64 + t * 32 + i) * 2 + h]);
          Bf[t][p] = __builtin_bit_cast(bf16x8, Ws[((p * 2 + ks) * 4 + wn * 2 + t) * 64 + lane]);
        }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          bf16x3::mfma6(acc[t][u], Af[t][0], Af[t][1], Af[t][2], Bf[u][0], Bf[u][1], Bf[u][2]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      NF_STASH();
      __syncthreads();
    }
  }
#undef NF_A_LOAD
#undef NF_FETCH
#undef NF_SPLIT4
#undef NF_STASH
}

__global__ __launch_bounds__(256, 3) void gemm_x3_kernel(const float* __restrict__ A, int lda,
                                                         const uint4* __restrict__ Wimg,
                                                         const float* __restrict__ bias, float* __restrict__ Cm,
                                                         int ldc, int M, int N, int K, int act, int gx, int gy) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_u4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int id = blockIdx.x;
  const int nblk = gx * gy;
  if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);  // one XCD sees a contiguous run of tiles
  const int mb = id / gx, nb = id % gx;
  const int m0 = mb * XBM, n0 = nb * XBN;
  f32x16 acc[2][2];
  gemm_x3_mainloop(A, lda, Wimg, M, K, m0, nb, lds_u4, acc);

#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = n0 + wn * 64 + u * 32 + i;
      if (col >= N) continue;
      const float bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (row >= M) continue;
        float v = acc[t][u][r] + bv;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
        Cm[(size_t)row * ldc + col] = v;
      }
    }
}

// SURVEY's K5 under the mode: the last Linear of a spline layer (weights in Plan's K5 column layout,
// split image) with the spline in the epilogue (rqs_tile_epilogue, nf_rqs_eval.h).
template <int KB, int INV>
__global__ __launch_bounds__(256, 3) void gemm_x3_rqs_kernel(const float* __restrict__ A, int lda,
                                                             const uint4* __restrict__ Wimg,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             float* __restrict__ ldpart, int M, int K, int D,
                                                             int pass_parity, int gx, int gy, RqsConst c) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_u4[];
  int id = blockIdx.x;
  const int nblk = gx * gy;
  if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
  const int mb = id / gx, nb = id % gx;
  const int m0 = mb * XBM, n0 = nb * XBN;
  f32x16 acc[2][2];
  gemm_x3_mainloop(A, lda, Wimg, M, K, m0, nb, lds_u4, acc);
  float* smem = reinterpret_cast<float*>(lds_u4);
  static_assert(2 * kTileU4 * 16 >= (64 * kCtStride + 64 * 8) * 4, "epilogue tile must fit the GEMM's LDS stage");
  rqs_tile_epilogue<KB, INV>(acc, bias, n0, m0, nb, M, D, pass_parity, y, ldpart, smem, smem + 64 * kCtStride, c);
}

template <int KB>
static int launch_gemm_x3_rqs_k(const float* A, int lda, const unsigned short* wimg, const float* bias, float* y,
                                float* ldpart, int M, int Npad, int K, int D, int par, int inverse,
                                const RqsConst& rc, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds_bytes = 2 * (size_t)kTileU4 * sizeof(uint4);
  if (!attr_set) {
    NF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_rqs_kernel<KB, 0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    NF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_rqs_kernel<KB, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int gx = Npad / XBN, gy = (M + XBM - 1) / XBM;
  const uint4* w = reinterpret_cast<const uint4*>(wimg);
  if (inverse)
    hipLaunchKernelGGL((gemm_x3_rqs_kernel<KB, 1>), dim3(gx * gy), dim3(256), lds_bytes, stream, A, lda, w, bias, y,
                       ldpart, M, K, D, par, gx, gy, rc);
  else
    hipLaunchKernelGGL((gemm_x3_rqs_kernel<KB, 0>), dim3(gx * gy), dim3(256), lds_bytes, stream, A, lda, w, bias, y,
                       ldpart, M, K, D, par, gx, gy, rc);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ldpart [Npad / 128][M]: the caller adds the column blocks' log-dets in order (ld_reduce_kernel)
int launch_gemm_x3_rqs(const float* A, int lda, const unsigned short* wimg, const float* bias, float* y,
                       float* ldpart, int M, int Npad, int K, int D, int par, int inverse, const RqsConst& rc,
                       hipStream_t stream) {
  switch (rc.K) {
    case 6: return launch_gemm_x3_rqs_k<6>(A, lda, wimg, bias, y, ldpart, M, Npad, K, D, par, inverse, rc, stream);
    case 8: return launch_gemm_x3_rqs_k<8>(A, lda, wimg, bias, y, ldpart, M, Npad, K, D, par, inverse, rc, stream);
    case 10: return launch_gemm_x3_rqs_k<10>(A, lda, wimg, bias, y, ldpart, M, Npad, K, D, par, inverse, rc, stream);
    case 12: return launch_gemm_x3_rqs_k<12>(A, lda, wimg, bias, y, ldpart, M, Npad, K, D, par, inverse, rc, stream);
  }
  set_error("no fused spline epilogue for %d bins", rc.K);
  return NF_ERR_UNSUPPORTED;
}

bool gemm_x3_applicable(const float* A, int lda, int N, int K) {
  return (lda & 3) == 0 && (K & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && N >= 64 && K >= 32;
}

int launch_gemm_x3(const float* A, int lda, const unsigned short* wimg, const float* bias, float* Cm, int ldc,
                   int M, int N, int K, int act, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds_bytes = 2 * (size_t)kTileU4 * sizeof(uint4);
  if (!attr_set) {
    NF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int gx = (N + XBN - 1) / XBN, gy = (M + XBM - 1) / XBM;
  hipLaunchKernelGGL(gemm_x3_kernel, dim3(gx * gy), dim3(256), lds_bytes, stream, A, lda,
                     reinterpret_cast<const uint4*>(wimg), bias, Cm, ldc, M, N, K, act, gx, gy);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
