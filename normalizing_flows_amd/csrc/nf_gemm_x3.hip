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
//     tile) one contiguous 24 KiB image [part 3][k16 2][column tile 4][lane 64][8 bf16]: the B-operand
//     order, read by each wave straight into registers (1 KiB per instruction, L2-resident), never
//     through LDS.
//   * A stays float32 row-major in HBM (it is the previous Linear's output, or the sample
//     matrix).  Each thread loads 4 consecutive k of four rows (8 lanes cover a row's 128-byte line, a
//     wave's instruction 8 whole lines), splits them in registers (11 VALU per pair of values) and writes
//     [part][k16][row 128][16 bf16] to LDS -- with a 16-deep k step the row-major bf16 tile IS the A
//     operand order.
// 128x128 block tile, BK = 32, 4 waves of 64x64, two 24 KiB LDS stages of A (one barrier per k tile),
// two workgroups per CU (176 registers).  What was measured on the way (profiles/microbench/
// gemm_x3_probe.hip, gemm_x3_wide_probe.hip): W through LDS, W direct, 128x256 blocks with
// 128x64 wave tiles (half the vector-memory bytes and split work per MFMA), branch-free prefetch and the
// split interleaved with the MFMAs all land within 4 % of each other -- the MFMA stream itself runs at
// 71 % of its nominal rate on dense operands (88 % on zeros: clock/power), and that is what binds.  The wide
// geometry won in the probe (-9 %) and lost in this file (+8 % with the spline epilogue, +10 % plain: one
// workgroup per CU leaves prologue, store and epilogue without another workgroup's MFMAs beside them).
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
//   * W never touches LDS: the host image is already in B-operand order, so a wave reads its two column
//     tiles' fragments (1 KiB per instruction, L2-resident: a column block's images are shared by every
//     row block) straight into registers, one 16-deep k step ahead of their use.
//   * A goes through LDS (split on the fly, see the file header) in two stages: one barrier per k tile.
__device__ __forceinline__ void gemm_x3_mainloop(const float* __restrict__ A, int lda, const uint4* __restrict__ Wimg,
                                                 int M, int K, int m0, int nb, uint4* lds_u4, f32x16 (&acc)[2][2]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (K + XBK - 1) / XBK;

  // this thread's share of an A tile: rows (tid >> 3) + 32 e, e = 0..3, the 16-byte chunk c = tid & 7 of the
  // row's 128-byte line (k = 4c .. 4c+3 of the tile): a wave's load instruction covers 8 whole lines
  const int c = tid & 7, ar0 = tid >> 3;
  const uint4* wsrc = Wimg + (size_t)nb * nk * kTileU4 + wn * 128 + lane;  // + kt*kTileU4 + ((p*2+ks)*4 + t)*64

  float4 ra0, ra1, ra2, ra3;
  // Loads carry no branch and no select: rows past M and the prefetch past the last tile are clamped to
  // valid addresses, and a k tail re-reads the row's own last chunk -- the image's rows k >= K are zero.
  // (A predicated load is a branch; at its join the compiler has to assume the shorter queue and waits
  // for EVERY outstanding load, and the prefetch distance collapses to zero.)
  const float* arow0 = A + (size_t)min(m0 + ar0, M - 1) * lda + c * 4;
  const float* arow1 = A + (size_t)min(m0 + ar0 + 32, M - 1) * lda + c * 4;
  const float* arow2 = A + (size_t)min(m0 + ar0 + 64, M - 1) * lda + c * 4;
  const float* arow3 = A + (size_t)min(m0 + ar0 + 96, M - 1) * lda + c * 4;
  const int klast = (K - 4) - c * 4;  // the last float4 of a row this thread may start at, minus its chunk
#define NF_A_LOAD(ptr_, dst_) dst_ = *reinterpret_cast<const float4*>(ptr_ + min(kt_ * XBK, klast));
#define NF_FETCH(kt_expr)                                   \
  {                                                         \
    const int kt_ = (kt_expr);                              \
    NF_A_LOAD(arow0, ra0) NF_A_LOAD(arow1, ra1) NF_A_LOAD(arow2, ra2) NF_A_LOAD(arow3, ra3) \
  }
  // LDS A tile: [part 3][k16 2][row 128][16 bf16]; this thread's four values of a row are 8 bytes of it
#define NF_STASH1(r_, e_)                                                           \
  {                                                                                 \
    unsigned h0_, m0_, l0_, h1_, m1_, l1_;                                          \
    split_pair(r_.x, r_.y, h0_, m0_, l0_);                                          \
    split_pair(r_.z, r_.w, h1_, m1_, l1_);                                          \
    uint2* d_ = sbase_ + (size_t)(32 * (e_)) * 4;                                   \
    d_[0] = make_uint2(h0_, h1_);                                                   \
    d_[2 * 128 * 4] = make_uint2(m0_, m1_);                                         \
    d_[4 * 128 * 4] = make_uint2(l0_, l1_);                                         \
  }
#define NF_STASH(stage_)                                                            \
  {                                                                                 \
    uint2* sbase_ = reinterpret_cast<uint2*>(lds_u4 + (stage_) * kTileU4) + ((size_t)((c >> 2) * 128 + ar0) * 4 + (c & 3)); \
    NF_STASH1(ra0, 0) NF_STASH1(ra1, 1) NF_STASH1(ra2, 2) NF_STASH1(ra3, 3)         \
  }
#define NF_B_LOAD(dst_, kt_, ks_)                                                   \
  {                                                                                 \
    const uint4* w_ = wsrc + (size_t)(kt_) * kTileU4 + (ks_) * 256;                 \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) dst_[t_][p_] = w_[p_ * 512 + t_ * 64]; \
  }
#define NF_A_FRAGS(stage_, ks_)                                                     \
    const uint4* As_ = lds_u4 + (stage_) * kTileU4;                                 \
    bf16x8 Af[2][3];                                                                \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                              \
        Af[t_][p_] = __builtin_bit_cast(bf16x8, As_[(((p_ * 2 + (ks_)) * 128) + wm * 64 + t_ * 32 + i) * 2 + h]);
#define NF_STEP(stage_, ks_, B_)                                                    \
  {                                                                                 \
    NF_A_FRAGS(stage_, ks_)                                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                \
      _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_)                              \
        bf16x3::mfma6(acc[t_][u_], Af[t_][0], Af[t_][1], Af[t_][2], __builtin_bit_cast(bf16x8, B_[u_][0]), \
                      __builtin_bit_cast(bf16x8, B_[u_][1]), __builtin_bit_cast(bf16x8, B_[u_][2])); \
  }
  // the same step with the split of the next A tile spread over its MFMAs: per (t, u) six MFMAs and one
  // register's share of the split (two split_pair, three ds_write_b64), fenced so that the compiler keeps
  // the mix (a bf16 MFMA leaves the vector issue free for ~6 VALU instructions)
#define NF_STEP_STASH1(B_, t_, u_, r_, e_)                                              \
    bf16x3::mfma6(acc[t_][u_], Af[t_][0], Af[t_][1], Af[t_][2], __builtin_bit_cast(bf16x8, B_[u_][0]), \
                  __builtin_bit_cast(bf16x8, B_[u_][1]), __builtin_bit_cast(bf16x8, B_[u_][2])); \
    NF_STASH1(r_, e_)                                                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
    __builtin_amdgcn_sched_barrier(0);
#define NF_STEP_STASH(stage_, ks_, B_, next_stage_)                                 \
  {                                                                                 \
    NF_A_FRAGS(stage_, ks_)                                                         \
    uint2* sbase_ = reinterpret_cast<uint2*>(lds_u4 + (next_stage_) * kTileU4) + ((size_t)((c >> 2) * 128 + ar0) * 4 + (c & 3)); \
    __builtin_amdgcn_sched_barrier(0);                                              \
    NF_STEP_STASH1(B_, 0, 0, ra0, 0) NF_STEP_STASH1(B_, 0, 1, ra1, 1) NF_STEP_STASH1(B_, 1, 0, ra2, 2) NF_STEP_STASH1(B_, 1, 1, ra3, 3) \
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  uint4 b0[2][3], b1[2][3];  // the B fragments of k16 step 0 / 1 of a tile
  NF_B_LOAD(b0, 0, 0);
  NF_FETCH(0);
  NF_STASH(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int nxt = min(kt + 1, nk - 1);  // the last iteration re-loads and re-stashes its own tile (unused)
    NF_FETCH(nxt);
    NF_B_LOAD(b1, kt, 1);
    __builtin_amdgcn_sched_barrier(0);  // the loads stay ahead of the step that hides them
    NF_STEP(cur, 0, b0);
    NF_B_LOAD(b0, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    NF_STEP_STASH(cur, 1, b1, cur ^ 1);
    __syncthreads();
  }
#undef NF_A_LOAD
#undef NF_FETCH
#undef NF_STASH1
#undef NF_STASH
#undef NF_B_LOAD
#undef NF_STEP
#undef NF_A_FRAGS
#undef NF_STEP_STASH1
#undef NF_STEP_STASH
}

__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const float* __restrict__ A, int lda,
                                                         const uint4* __restrict__ Wimg,
                                                         const float* __restrict__ bias, float* __restrict__ Cm,
                                                         int ldc, int M, int N, int K, int act, int gx, int gy) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_u4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int mb, nb;
  block_to_tile(blockIdx.x, gx, gy, mb, nb);
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
__global__ __launch_bounds__(256, 2) void gemm_x3_rqs_kernel(const float* __restrict__ A, int lda,
                                                             const uint4* __restrict__ Wimg,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             float* __restrict__ ldpart, int M, int K, int D,
                                                             int pass_parity, int gx, int gy, RqsConst c) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_u4[];
  int mb, nb;
  block_to_tile(blockIdx.x, gx, gy, mb, nb);
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
    const size_t lds_bytes = 2 * (size_t)kTileU4 * sizeof(uint4);
  NF_DYN_LDS(64 * 1024, &gemm_x3_rqs_kernel<KB, 0>);
  NF_DYN_LDS(64 * 1024, &gemm_x3_rqs_kernel<KB, 1>);
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
    const size_t lds_bytes = 2 * (size_t)kTileU4 * sizeof(uint4);
  NF_DYN_LDS(64 * 1024, &gemm_x3_kernel);
  const int gx = (N + XBN - 1) / XBN, gy = (M + XBM - 1) / XBM;
  hipLaunchKernelGGL(gemm_x3_kernel, dim3(gx * gy), dim3(256), lds_bytes, stream, A, lda,
                     reinterpret_cast<const uint4*>(wimg), bias, Cm, ldc, M, N, K, act, gx, gy);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
