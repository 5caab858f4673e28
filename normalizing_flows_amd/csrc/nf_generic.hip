// General path: any event size, any conditioner widths, all three coupling kinds.
//
// One coupling layer = a chain of f32-MFMA GEMMs (conditioner MLP, bias +
// activation fused into the epilogue) followed by one transform kernel (affine or
// rational-quadratic spline) that updates the sample matrix and the running
// log-det in place.  Activations and conditioner outputs live in device scratch,
// the batch is walked in row chunks so the scratch stays bounded.
//
// This is the path of the reference's spline flows (src/maf_toy_densities.py:55-111,
// src/maf_mnist.py:34-89; distrax RationalQuadraticSpline semantics in SURVEY.md
// section 8a row c2) and of every affine configuration the two fused kernels do
// not cover.  Mask handling: the first Linear's rows of the transformed dims are
// zeroed on the host (x @ W' == where(mask, x, 0) @ W) and only the spline
// parameters of the transformed dims are computed (the others are never read by
// distrax.MaskedCoupling either: y = where(mask, x, y0)).
#include "nf_internal.h"
#include "nf_rqs_eval.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------
// C[M,N] = act(A[M,K] @ W[K,N] + bias[N]);  A row-major with leading dim lda,
// W row-major [K,N], C row-major with leading dim ldc.
// MFMA 32x32x2 f32: rows = samples, cols = output units.
// ------------------------------------------------------------------------
// BK (template): depth of one LDS stage: 16 in general; 64 when K <= 64, so that the narrow layers of the
// 2-D flows (and their backward products) need one stage and two barriers instead of four dependent rounds

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bias_act_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Cm, int ldc, int M, int N, int K, int act, const float* __restrict__ aux) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_PER = BM * BK / NT;  // floats of the A tile per thread
  constexpr int W_PER = BK * BN / NT;
  __shared__ float As[BM][BK + 1];
  __shared__ float Ws[BK][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float ra[A_PER], rw[W_PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < A_PER; ++e) {
      const int idx = tid + e * NT;  // over BM*BK, k fastest
      const int r = idx / BK, k = idx % BK;
      const int gr = m0 + r, gk = k0 + k;
      ra[e] = (gr < M && gk < K) ? A[(size_t)gr * lda + gk] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < W_PER; ++e) {
      const int idx = tid + e * NT;  // over BK*BN, n fastest
      const int k = idx / BN, c = idx % BN;
      const int gk = k0 + k, gc = n0 + c;
      rw[e] = (gk < K && gc < N) ? W[(size_t)gk * N + gc] : 0.f;
    }
  };

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < A_PER; ++e) {
      const int idx = tid + e * NT;
      As[idx / BK][idx % BK] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < W_PER; ++e) {
      const int idx = tid + e * NT;
      Ws[idx / BN][idx % BN] = rw[e];
    }
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
    const int steps = ((K - k0 < BK ? K - k0 : BK) + 1) / 2;  // a short K (first Linear of a 2-D flow: 2) stops early
    for (int s = 0; s < steps; ++s) {
      float a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = As[wm * (TM * 32) + t * 32 + i][2 * s + h];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = Ws[2 * s + h][wn * (TN * 32) + t * 32 + i];
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
  }

#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * (TN * 32) + u * 32 + i;
      if (col >= N) continue;
      const float bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (TM * 32) + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (row >= M) continue;
        float v = acc[t][u][r] + bv;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
        else if (act >= 3) v = aux[(size_t)row * ldc + col] > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
        Cm[(size_t)row * ldc + col] = v;
      }
    }
}

// ------------------------------------------------------------------------
// Same product for the aligned case (lda, N, K multiples of 4, 16-byte aligned bases):
// 128x128 block tile, BK = 32, float4 global loads, one 33 KB LDS stage with the next tile held in
// registers while the current one computes, 2x2 waves of 64x64, FOUR workgroups per CU: with the
// f32 MFMA blocking the vector issue a wave cannot overlap its own stash with its MFMAs, so the
// overlap has to come from other workgroups (double-buffered at two per CU: 73 % MFMA-busy).
// Blocks are renumbered so that each XCD's L2 sees a contiguous run of tiles.
// ------------------------------------------------------------------------
constexpr int FBM = 128, FBN = 128, FBK = 32;
constexpr int kFastLdsFloats = FBM * (FBK + 1) + FBK * FBN;  // one 33 KB stage: four workgroups per CU

// The k loop shared by gemm_fast_kernel and gemm_rqs_kernel: acc[t][u] = the wave's 2 x 2 tiles of
// A[m0.., :] @ W[:, n0..] (no bias).  Ends with a barrier: smem is free again.
__device__ __forceinline__ void gemm_fast_mainloop(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                   int M, int N, int K, int m0, int n0, float* smem,
                                                   f32x16 (&acc)[2][2]) {
  constexpr int kAs = FBM * (FBK + 1);  // A tile, then the W tile
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // this thread's share of a k-tile: A rows ar + 32 e (e < 4), k quad akq; W rows wk + 8 e, col quad wc
  const int ar = tid >> 3, akq = (tid & 7) * 4;
  const int wk = tid >> 5, wc = (tid & 31) * 4;
  float4 ra[4], rw[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gr = m0 + ar + 32 * e, gk = k0 + akq;
      ra[e] = (gr < M && gk < K) ? *reinterpret_cast<const float4*>(A + (size_t)gr * lda + gk)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
      const int kk = k0 + wk + 8 * e, gc = n0 + wc;
      rw[e] = (kk < K && gc < N) ? *reinterpret_cast<const float4*>(W + (size_t)kk * N + gc)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* a = smem + (ar + 32 * e) * (FBK + 1) + akq;
      a[0] = ra[e].x; a[1] = ra[e].y; a[2] = ra[e].z; a[3] = ra[e].w;
      *reinterpret_cast<float4*>(smem + kAs + (wk + 8 * e) * FBN + wc) = rw[e];
    }
  };
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (K + FBK - 1) / FBK;
  fetch(0);
  stash();
  __syncthreads();
  const float* as = smem + (wm * 64 + i) * (FBK + 1) + h;
  const float* ws = smem + kAs + h * FBN + wn * 64 + i;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch((kt + 1) * FBK);
#pragma unroll
    for (int s = 0; s < FBK / 2; ++s) {
      const float a0 = as[2 * s], a1 = as[32 * (FBK + 1) + 2 * s];
      const float b0 = ws[2 * s * FBN], b1 = ws[2 * s * FBN + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();  // every wave is done with this tile
    if (kt + 1 < nk) {
      stash();
      __syncthreads();
    }
  }
}

// The same k loop for the case every large GEMM of the path is in (N a multiple of 128; rows may be ragged, and so may
// the last k tile: K a multiple of 4): what differs is everything BUT the MFMAs.  gemm_fast_mainloop spends ~90 VALU instructions per k
// tile and wave beside its 64 MFMAs -- predicates, zero-fills and 64-bit address arithmetic of eight guarded loads,
// address adds for 64 scalar LDS operand reads -- and on gfx950 every one of them costs the MFMA pipe its cycles
// (DESIGN.md section 4).  Here
//   * loads are unconditional buffer loads (one descriptor per operand built from uniform values, the thread's share
//     as a 32-bit byte offset, the k tile's advance as the scalar offset): K and N need no guard, rows past M are
//     clamped to M - 1 (their products are computed and never stored) -- no branches, no zero-fills, no 64-bit
//     vector addresses (eight of those did not fit beside 64 accumulator + 32 prefetch registers at four workgroups
//     per CU: they were spilled and reloaded every k tile);
//   * the LDS tiles hold each lane's OPERAND PAIRS side by side -- A as [wm][k][i][{row i, row i + 32}], W as
//     [k][wn][i][{col i, col i + 32}] -- so a k step is two aligned 8-byte reads with immediate offsets instead of
//     four ds_read_b32 behind computed addresses;
//   * stores are laid out for the LDS's store banking (32 banks, 16-lane groups for 8-byte stores): A rows are
//     padded to 66 floats (a thread holds one operand pair at one k quad; sixteen lanes -- four pairs x four quads
//     of a half tile -- hit 32 distinct banks; the rows of k >= 16 are rotated by two lanes, which did the same for
//     the 32-wide form of the loop); a thread loads W as two float2 (columns c, c+1 and c+32, c+33) and stores them
//     as ONE 16-byte write of two whole pairs.  (A first version kept float4 W loads and scattered
//     them with 4-byte stores: 4-way bank conflicts, SQ_LDS_BANK_CONFLICT 338 M cycles per launch, 18 % slower
//     than the loop it replaced.)
// Same sums in the same order as gemm_fast_mainloop: results are bit-identical.
constexpr int kFullARow = 66;  // floats per k row of one wm half: 32 lanes x 2 + 2 pad
static_assert(2 * FBK * kFullARow + FBK * FBN <= kFastLdsFloats, "full-tile layout must fit the shared LDS stage");

// The k tile is handled as two HALVES of 16: one LDS image (its k < 16 and k >= 16 parts are the halves), but a half
// is fetched, stored and consumed on its own -- while the
// waves multiply half h, each of them stores the next half into the other part as soon as its own MFMAs are issued,
// and ONE barrier per half closes the step.  The 32-wide loop had a phase per k tile in which every wave of the
// workgroup stored and waited (barrier, stores, barrier); timing-only builds priced that bookkeeping at ~4 % of
// configs[4]; the half form recovers 0.55 % of it (35.85 -> 35.65 ms) with sixteen
// prefetch registers instead of 32.  Same sums in the same order.
__device__ __forceinline__ void gemm_full_mainloop(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                   int M, int N, int K, int m0, int n0, float* smem,
                                                   f32x16 (&acc)[2][2]) {
  constexpr int kAs = 2 * FBK * kFullARow;  // A tile, then the W tile
  constexpr int HK = FBK / 2;               // 16
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // this thread's share of a half: A rows (pw * 64 + pi) and (+ 32) -- one operand pair -- at k quad pq of the half;
  // W rows wk and wk + 8 of the half, the column pairs (wc, wc + 1) and (wc + 32, wc + 33) of half wh
  const int pq = (tid & 3) * 4, pi = (tid >> 2) & 31, pw = tid >> 7;
  const int wk = tid >> 5, wh = (tid >> 4) & 1, wc = wh * 64 + 2 * (tid & 15);
  const long long aleft = ((long long)(M - m0 - 1) * lda + K) * 4;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A + (size_t)m0 * lda), 0, aleft > 0x7fffffffLL ? 0x7fffffff : (int)aleft, 0x00020000);
  const long long wleft = ((long long)K * N - n0) * 4;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(W + n0), 0, wleft > 0x7fffffffLL ? 0x7fffffff : (int)wleft, 0x00020000);
  unsigned aoff[2], woff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    aoff[e] = ((unsigned)(min(m0 + pw * 64 + pi + 32 * e, M - 1) - m0) * (unsigned)lda + pq) * 4u;
    woff[e] = ((unsigned)(wk + 8 * e) * (unsigned)N + wc) * 4u;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra[2];
  unsigned rw[2][4];
  auto fetch = [&](int ht) {
    const int asoff = ht * (HK * 4), wsoff = ht * (HK * 4) * N;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ra[e] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[e], asoff, 0);
      rw[e][0] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, woff[e], wsoff, 0);
      rw[e][1] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, woff[e] + 128u, wsoff, 0);
      rw[e][2] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, woff[e] + 4u, wsoff, 0);
      rw[e][3] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, woff[e] + 132u, wsoff, 0);
    }
  };
  // LDS byte addresses of the pair's first k row in half 0 / half 1 (rows of k >= 16 are rotated by two lanes)
  const unsigned sa_h0 = (unsigned)(size_t)(smem + (pw * FBK + pq) * kFullARow + 2 * pi);
  const unsigned sa_h1 = (unsigned)(size_t)(smem + (pw * FBK + HK + pq) * kFullARow + 2 * ((pi + 2) & 31));
  float* sw = smem + kAs + wk * FBN + wh * 64 + 4 * (tid & 15);
#define NF_A_PAIR(base_, q_, lo_, hi_)                                                                                 \
  asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base_), "v"(lo_), "v"(hi_), "n"((q_)*kFullARow), \
               "n"((q_)*kFullARow + 1)                                                                                 \
               : "memory")
  const int ktail = K % HK;  // a multiple of 4
  auto stash = [&](int half, bool tail) {
    if (tail && pq >= ktail) {  // this thread's k quad lies past K: it fetched another row's leading floats
      ra[0] = u32x4{0u, 0u, 0u, 0u};
      ra[1] = u32x4{0u, 0u, 0u, 0u};
    }
    const unsigned sa = half ? sa_h1 : sa_h0;
    NF_A_PAIR(sa, 0, ra[0][0], ra[1][0]); NF_A_PAIR(sa, 1, ra[0][1], ra[1][1]);
    NF_A_PAIR(sa, 2, ra[0][2], ra[1][2]); NF_A_PAIR(sa, 3, ra[0][3], ra[1][3]);
    float* swh = sw + half * HK * FBN;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      *reinterpret_cast<float4*>(swh + 8 * e * FBN) = make_float4(__uint_as_float(rw[e][0]), __uint_as_float(rw[e][1]),
                                                                   __uint_as_float(rw[e][2]), __uint_as_float(rw[e][3]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm stores above are invisible to the compiler's counters
  };
#undef NF_A_PAIR
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nh = (K + HK - 1) / HK;
  fetch(0);
  stash(0, nh == 1 && ktail != 0);
  __syncthreads();
  const float2* as_lo = reinterpret_cast<const float2*>(smem + (wm * FBK + h) * kFullARow) + i;
  const float2* as_hi = reinterpret_cast<const float2*>(smem + (wm * FBK + h) * kFullARow) + ((i + 2) & 31);
  const float2* ws = reinterpret_cast<const float2*>(smem + kAs + h * FBN) + wn * 32 + i;
#define NF_HALF_STEP(as_, ws_)                                                              \
  _Pragma("unroll") for (int s = 0; s < HK / 2; ++s) {                                       \
    const float2 a = (as_)[s * kFullARow];                                                   \
    const float2 b = (ws_)[s * FBN];                                                         \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);          \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);          \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);          \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);          \
  }
  const float2* as_hi8 = as_hi + (HK / 2) * kFullARow;
  const float2* ws8 = ws + (HK / 2) * FBN;
  // two halves per trip so that their LDS parts are compile-time; an odd count ends with one more half
  int ht = 0;
  for (; ht + 2 <= nh; ht += 2) {
    fetch(ht + 1);
    NF_HALF_STEP(as_lo, ws)
    stash(1, ht + 2 == nh && ktail != 0);
    __syncthreads();
    if (ht + 2 < nh) fetch(ht + 2);
    NF_HALF_STEP(as_hi8, ws8)
    if (ht + 2 < nh) stash(0, ht + 3 == nh && ktail != 0);
    __syncthreads();
  }
  if (ht < nh) {
    NF_HALF_STEP(as_lo, ws)
    __syncthreads();
  }
#undef NF_HALF_STEP
}

// K, N as gemm_full_mainloop needs them?  (alignment of the bases and of lda is the fast path's own condition)
static inline bool gemm_full_ok(int N, int K) { return K >= FBK && (K % 4) == 0 && (N % FBN) == 0 && !switches().gemm_no_full; }

template <bool FULL>
__global__ __launch_bounds__(256, 4) void gemm_fast_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Cm, int ldc, int M, int N, int K, int act, int gx, int gy, const float* __restrict__ aux) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int mb, nb;
  block_to_tile(blockIdx.x, gx, gy, mb, nb);
  const int m0 = mb * FBM, n0 = nb * FBN;
  f32x16 acc[2][2];
  if (FULL) gemm_full_mainloop(A, lda, W, M, N, K, m0, n0, smem, acc);
  else gemm_fast_mainloop(A, lda, W, M, N, K, m0, n0, smem, acc);

  if (FULL) {
    // Epilogue without per-element predicates or 64-bit addresses (the guarded form below costs ~30 VALU
    // instructions per output: 1,900 per wave and tile against 1,024 MFMAs).  One buffer descriptor over the rows
    // [m0, M) of C: rows past M fall outside it and the hardware drops their stores; columns need no guard
    // (N is a multiple of 128).  The lane's part of the address is one 32-bit offset, the (t, u, r) part is uniform.
    const long long left = (long long)(M - m0) * ldc * 4;
    const unsigned bytes = left > 0x7fffffffLL ? 0x7fffffffu : (unsigned)left;
    const __amdgpu_buffer_rsrc_t crsrc =
        __builtin_amdgcn_make_buffer_rsrc(Cm + (size_t)m0 * ldc, 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(act >= 3 ? aux + (size_t)m0 * ldc : Cm + (size_t)m0 * ldc), 0, (int)bytes, 0x00020000);
    const unsigned voff = ((unsigned)(wm * 64 + 4 * h) * (unsigned)ldc + (unsigned)(n0 + wn * 64 + i)) * 4u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float bv = bias[n0 + wn * 64 + u * 32 + i];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int soff = ((t * 32 + 8 * (r >> 2) + (r & 3)) * ldc + u * 32) * 4;
          float v = acc[t][u][r] + bv;
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) v = fmaxf(v, 0.01f * v);
          else if (act >= 3) {
            const float a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, soff, 0));
            v = a > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, voff, soff, 0);
        }
    }
    return;
  }
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
        else if (act >= 3) v = aux[(size_t)row * ldc + col] > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
        Cm[(size_t)row * ldc + col] = v;
      }
    }
}

// Small batches (the reference trains at batch 128): a [128 x N] result has N / 128 tiles -- four workgroups
// for N = 512 -- and each walks the whole K alone (307 k tiles for the MNIST shape's d theta W^T: 0.6 ms on 4 of 256
// CUs).  Split K: workgroup (tile, s) multiplies the k range [s Kc, (s+1) Kc) and writes its raw tile to
// partial[s][M][N]; gemm_splitk_reduce_kernel adds the S slabs in order (run-to-run identical) and applies bias
// and activation.
__global__ __launch_bounds__(256, 4) void gemm_splitk_kernel(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ W, float* __restrict__ partial,
                                                             int M, int N, int K, int Kc, int gx, int gy) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x % (gx * gy), s = blockIdx.x / (gx * gy);
  const int m0 = (tile / gx) * FBM, n0 = (tile % gx) * FBN;
  const int k0 = s * Kc;
  f32x16 acc[2][2];
  gemm_fast_mainloop(A + k0, lda, W + (size_t)k0 * N, M, N, min(Kc, K - k0), m0, n0, smem, acc);
  float* out = partial + (size_t)s * M * N;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = n0 + wn * 64 + u * 32 + i;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (row < M) out[(size_t)row * N + col] = acc[t][u][r];
      }
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int S,
                                                                 const float* __restrict__ bias, float* __restrict__ Cm,
                                                                 int ldc, int M, int N, int act,
                                                                 const float* __restrict__ aux) {
  const size_t total = (size_t)M * N;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const int row = (int)(q / N), col = (int)(q - (size_t)row * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += partial[(size_t)s * total + q];
    v += bias[col];
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
    else if (act >= 3) v = aux[(size_t)row * ldc + col] > 0.f ? v : (act == 4 ? 0.01f * v : 0.f);
    Cm[(size_t)row * ldc + col] = v;
  }
}

// Many slabs over a small result (the backward walk's data gradient through a wide last Linear: 64 slabs of [128, 512]):
// one thread per element walks S dependent-latency rounds (11.5 us for 16.8 MB).  Here sixteen waves share an element's
// slabs -- wave w adds slabs [w c, (w + 1) c) of 64 float4 elements, all c loads in flight -- and the sixteen sums are
// added in wave order (fixed: run-to-run identical; not the order of the kernel above, which keeps the small-S cases
// whose order the spline kernel's slab sum mirrors).
__global__ __launch_bounds__(1024) void gemm_splitk_reduce_wide_kernel(const float* __restrict__ partial, int S,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ Cm, int ldc, int M, int N,
                                                                       int act, const float* __restrict__ aux) {
  __shared__ float4 part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t total4 = (size_t)M * N / 4;
  const size_t q = (size_t)blockIdx.x * 64 + lane;
  const int c = (S + 15) / 16;
  const int s0 = wave * c, s1 = min(S, s0 + c);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < total4) {
    const float4* p4 = reinterpret_cast<const float4*>(partial) + q;
    int sidx = s0;
    for (; sidx + 4 <= s1; sidx += 4) {
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = p4[(size_t)(sidx + u) * total4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
    }
    for (; sidx < s1; ++sidx) {
      const float4 t = p4[(size_t)sidx * total4];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  part[wave][lane] = v;
  __syncthreads();
  if (wave == 0 && q < total4) {
    float4 r = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) { const float4 t = part[w][lane]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    const size_t e = q * 4;
    const int row = (int)(e / N), col = (int)(e - (size_t)row * N);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
    float o[4] = {r.x + b4.x, r.y + b4.y, r.z + b4.z, r.w + b4.w};
    if (act == 1) { for (int u = 0; u < 4; ++u) o[u] = fmaxf(o[u], 0.f); }
    else if (act == 2) { for (int u = 0; u < 4; ++u) o[u] = o[u] >= 0.f ? o[u] : 0.01f * o[u]; }
    else if (act >= 3) {
      const float4 a4 = *reinterpret_cast<const float4*>(aux + (size_t)row * ldc + col);
      const float m[4] = {a4.x, a4.y, a4.z, a4.w};
      for (int u = 0; u < 4; ++u) o[u] = m[u] > 0.f ? o[u] : (act == 4 ? 0.01f * o[u] : 0.f);
    }
    *reinterpret_cast<float4*>(Cm + (size_t)row * ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// grow-only scratch for the split-K slabs, one per (host thread, stream): handles are driven from one thread each,
// and two handles of one thread run on different streams
// (use 0: the slabs; use 1: the log-det partials of nf_small_m_rqs.hip, which live across a walk's layers)
static float* splitk_scratch(hipStream_t stream, size_t floats, int use = 0) {
  struct Slot { hipStream_t stream; int use; float* ptr; size_t floats; };
  static thread_local std::vector<Slot> slots;
  for (Slot& sl : slots)
    if (sl.stream == stream && sl.use == use) {
      if (sl.floats >= floats) return sl.ptr;
      hipStreamSynchronize(stream);
      hipFree(sl.ptr);
      sl.ptr = nullptr;
      sl.floats = 0;
      if (hipMalloc(&sl.ptr, floats * sizeof(float)) != hipSuccess) return nullptr;
      sl.floats = floats;
      return sl.ptr;
    }
  Slot sl{stream, use, nullptr, 0};
  if (hipMalloc(&sl.ptr, floats * sizeof(float)) != hipSuccess) return nullptr;
  sl.floats = floats;
  slots.push_back(sl);
  return sl.ptr;
}

// act: 0 none, 1 relu, 2 leaky_relu; 3 / 4: the backward forms -- the result is multiplied by
// relu' / leaky_relu' of `aux` (an activation OUTPUT of the same shape and leading dimension as C)
int launch_gemm(const float* A, int lda, const float* W, const float* bias, float* Cm, int ldc,
                       int M, int N, int K, int act, hipStream_t stream, const float* aux, float* a_copy) {
  // few rows (the reference's batch of 128): the whole product in one launch of sixteen-wave workgroups
  if (gemm_small_m_applicable(A, lda, M, N, K) && !switches().gemm_simple)
    return launch_gemm_small_m(A, lda, W, bias, Cm, ldc, M, N, K, act, stream, aux, a_copy);
  if (a_copy != nullptr || act == 5) { set_error("this Linear's shape has no copying / accumulating form"); return NF_ERR_UNSUPPORTED; }
  const bool aligned = (lda & 3) == 0 && (N & 3) == 0 && (K & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
  if (aligned && N >= 128 && K >= 32 && !switches().gemm_simple) {
    NF_DYN_LDS(80 * 1024, &gemm_fast_kernel<false>);
    NF_DYN_LDS(80 * 1024, &gemm_fast_kernel<true>);
    const int gx = (N + FBN - 1) / FBN, gy = (M + FBM - 1) / FBM;
    // few tiles and a long K: split K over enough workgroups to fill the chip (>= 4 k tiles each)
    const int tiles = gx * gy, ktiles = (K + FBK - 1) / FBK;
    if (tiles <= 128 && ktiles >= 8 && !switches().no_splitk) {
      NF_DYN_LDS(80 * 1024, &gemm_splitk_kernel);
      int S = std::min(std::max(1, 256 / tiles), ktiles / 4);   // one resident round of workgroups (77 tiles: 3 x 77 = 231 measured 549 us per MNIST-shape call against 575 with 4 x 77)
      if (switches().splitk_s) S = std::max(1, std::min(switches().splitk_s, ktiles));
      const int Kc = ((ktiles + S - 1) / S) * FBK;
      S = (K + Kc - 1) / Kc;
      float* partial = splitk_scratch(stream, (size_t)S * M * N);
      if (partial == nullptr) { set_error("out of device memory for the split-K slabs"); return NF_ERR_HIP; }
      hipLaunchKernelGGL(gemm_splitk_kernel, dim3(tiles * S), dim3(256), kFastLdsFloats * sizeof(float), stream, A, lda,
                         W, partial, M, N, K, Kc, gx, gy);
      const size_t total = (size_t)M * N;
      const bool wide = S >= 16 && (N & 3) == 0 && (ldc & 3) == 0 && total <= ((size_t)1 << 22) && act != 5 &&
                        ((reinterpret_cast<uintptr_t>(Cm) | reinterpret_cast<uintptr_t>(bias) |
                          reinterpret_cast<uintptr_t>(aux)) & 15) == 0;
      if (wide)
        hipLaunchKernelGGL(gemm_splitk_reduce_wide_kernel, dim3((unsigned)((total / 4 + 63) / 64)), dim3(1024), 0, stream,
                           partial, S, bias, Cm, ldc, M, N, act, aux);
      else
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256),
                         0, stream, partial, S, bias, Cm, ldc, M, N, act, aux);
      NF_HIP(hipGetLastError());
      return NF_OK;
    }
    if (gemm_full_ok(N, K))
      hipLaunchKernelGGL(gemm_fast_kernel<true>, dim3(gx * gy), dim3(256), kFastLdsFloats * sizeof(float), stream, A, lda,
                         W, bias, Cm, ldc, M, N, K, act, gx, gy, aux);
    else
      hipLaunchKernelGGL(gemm_fast_kernel<false>, dim3(gx * gy), dim3(256), kFastLdsFloats * sizeof(float), stream, A,
                         lda, W, bias, Cm, ldc, M, N, K, act, gx, gy, aux);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  if (N > 64) {
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    hipLaunchKernelGGL((gemm_bias_act_kernel<128, 128, 2, 2, 16>), grid, dim3(256), 0, stream, A, lda, W,
                       bias, Cm, ldc, M, N, K, act, aux);
  } else {
    dim3 grid((N + 63) / 64, (M + 127) / 128);
    if (K <= 64 && M <= 8192)  // latency-bound sizes: one stage; at large M the smaller stage keeps more blocks per CU
      hipLaunchKernelGGL((gemm_bias_act_kernel<128, 64, 4, 1, 64>), grid, dim3(256), 0, stream, A, lda, W,
                         bias, Cm, ldc, M, N, K, act, aux);
    else
      hipLaunchKernelGGL((gemm_bias_act_kernel<128, 64, 4, 1, 16>), grid, dim3(256), 0, stream, A, lda, W,
                         bias, Cm, ldc, M, N, K, act, aux);
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------
// masked affine: params[rows][2] = (shift, log_scale), broadcast over the event;
// dims with (d % 2 == pass_parity) pass through.
__global__ __launch_bounds__(256) void affine_masked_kernel(float* __restrict__ y, float* __restrict__ ld,
                                                           const float* __restrict__ params, long long rows,
                                                           int D, int pass_parity, int inverse) {
  const long long total = rows * (long long)D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / D;
    const int d = (int)(e % D);
    const float shift = params[2 * row], ls = params[2 * row + 1];
    if ((d & 1) != pass_parity) {
      const float v = y[e];
      y[e] = inverse ? (v - shift) * expf(-ls) : fmaf(v, expf(ls), shift);
    }
    if (d == 0) {
      const int n_t = pass_parity == 0 ? D / 2 : (D + 1) / 2;  // transformed dims: d % 2 != pass_parity
      const float add = (float)n_t * ls;
      ld[row] += inverse ? -add : add;
    }
  }
}

// stand-alone distrax.ScalarAffine on explicit parameters: params[n][2] = (shift, log_scale) or (shift, scale)
__global__ __launch_bounds__(256) void scalar_affine_kernel(const float* __restrict__ x, const float2* __restrict__ params,
                                                           float* __restrict__ y, float* __restrict__ ld, long long n,
                                                           int inverse, int params_are_scale) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float2 p = params[e];
    const float v = x[e];
    float out, l;
    if (params_are_scale) {           // y = scale x + shift, log|det| = log|scale| (distrax takes log(abs(scale)))
      out = inverse ? (v - p.x) / p.y : fmaf(v, p.y, p.x);
      l = logf(fabsf(p.y));
    } else {                          // the coupling kernels' form: exp(log_scale)
      out = inverse ? (v - p.x) * expf(-p.y) : fmaf(v, expf(p.y), p.x);
      l = p.y;
    }
    if (y) y[e] = out;
    if (ld) ld[e] = inverse ? -l : l;
  }
}

int launch_scalar_affine(const float* x, const float* params, float* y, float* ld, long long n, int inverse,
                         int params_are_scale, hipStream_t stream) {
  const int grid = (int)std::max<long long>(1, std::min<long long>(8192, (n + 255) / 256));
  hipLaunchKernelGGL(scalar_affine_kernel, dim3(grid), dim3(256), 0, stream, x, reinterpret_cast<const float2*>(params), y,
                     ld, n, inverse, params_are_scale);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// split halves (variant A, any even D): params[rows][D] = (shift[d] | log_scale[d]);
// flip = 0 transforms the second half conditioned on the first.
__global__ __launch_bounds__(256) void affine_split_kernel(float* __restrict__ y, float* __restrict__ ld,
                                                          const float* __restrict__ params, long long rows,
                                                          int D, int flip, int inverse) {
  const int d = D / 2;
  for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long long)gridDim.x * 256) {
    float* v = y + row * D + (flip ? 0 : d);
    const float* p = params + row * D;
    float acc = 0.f;
    for (int c = 0; c < d; ++c) {
      const float shift = p[c], ls = p[d + c];
      v[c] = inverse ? (v[c] - shift) * expf(-ls) : fmaf(v[c], expf(ls), shift);
      acc += ls;
    }
    ld[row] += inverse ? -acc : acc;
  }
}


// One (row, transformed dim), K known only at run time: spline forward or inverse, returns the new value and
// log|det|.  p points at the 3K+1 raw parameters [widths K | heights K | slopes K+1].  Same arithmetic as the
// register form (nf_rqs_math.h: float64 bin weights, normalisation and running sums); the bins are walked twice
// (sums, then knots) instead of being kept.
__device__ __forceinline__ void rqs_eval(const float* __restrict__ p, const RqsConst& c, float v, int inverse,
                                         float* out, float* logdet) {
  const int K = c.K;
  float mw = -INFINITY, mh = -INFINITY;
  for (int k = 0; k < K; ++k) { mw = fmaxf(mw, p[k]); mh = fmaxf(mh, p[K + k]); }
  double sw = 0.0, sh = 0.0;
  for (int k = 0; k < K; ++k) {
    sw += rqsmath::bin_exp<NF_RQS_XEXP>(p[k], mw);
    sh += rqsmath::bin_exp<NF_RQS_XEXP>(p[K + k], mh);
  }
  const double iw = rqsmath::bin_scale(sw, K, c), ih = rqsmath::bin_scale(sh, K, c);
  const double mb = (double)c.min_bin;
  // walk the bins once, keeping the last one whose left knot on the search axis is <= v (bin 0 when none is);
  // a bin's size is its own float64 width rounded once, not a difference of rounded knots (nf_rqs_math.h)
  double ax = (double)c.range_min, ay = (double)c.range_min;
  float xl = c.range_min, yl = c.range_min;
  float bxl = 0.f, bw = 0.f, byl = 0.f, bh = 0.f, raw_l = 0.f, raw_r = 0.f;
  for (int k = 0; k < K; ++k) {
    const double wx = k == K - 1 ? (double)c.range_max - ax : __builtin_fma(rqsmath::bin_exp<NF_RQS_XEXP>(p[k], mw), iw, mb);
    const double wy = k == K - 1 ? (double)c.range_max - ay : __builtin_fma(rqsmath::bin_exp<NF_RQS_XEXP>(p[K + k], mh), ih, mb);
    if (k == 0 || v >= (inverse ? yl : xl)) {
      bxl = xl; bw = (float)wx; byl = yl; bh = (float)wy; raw_l = p[2 * K + k]; raw_r = p[2 * K + k + 1];
    }
    ax += wx; ay += wy;
    xl = k == K - 1 ? c.range_max : (float)ax;
    yl = k == K - 1 ? c.range_max : (float)ay;
  }
  if (inverse) rqsmath::rqs_map<1, false>(bxl, bw, byl, bh, raw_l, raw_r, p[2 * K], p[3 * K], v, c, out, logdet);
  else rqsmath::rqs_map<0, false>(bxl, bw, byl, bh, raw_l, raw_r, p[2 * K], p[3 * K], v, c, out, logdet);
}

// G lanes (power of two <= 64) share one row: each takes transformed dims dt = sub, sub+G, ...
// params[rows][Dt][3K+1] holds ONLY the transformed dims' parameters.
__global__ __launch_bounds__(256) void rqs_masked_kernel(float* __restrict__ y, float* __restrict__ ld,
                                                        const float* __restrict__ params, long long rows,
                                                        int D, int pass_parity, int inverse, int G, RqsConst c) {
  const int P = 3 * c.K + 1;
  const int Dt = pass_parity == 0 ? D / 2 : (D + 1) / 2;
  const int first_t = pass_parity == 0 ? 1 : 0;  // first transformed dim
  const long long lanes = rows * (long long)G;
  const long long span = ((lanes + 255) / 256) * 256;  // keep whole groups in lock step for the shuffles
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < span; e += (long long)gridDim.x * 256) {
    const long long row = e / G;
    const int sub = (int)(e % G);
    float acc = 0.f;
    if (row < rows) {
      for (int dt = sub; dt < Dt; dt += G) {
        const int d = first_t + 2 * dt;
        float res, l;
        rqs_eval(params + ((size_t)row * Dt + dt) * P, c, y[row * D + d], inverse, &res, &l);
        y[row * D + d] = res;
        acc += l;
      }
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (row < rows && sub == 0) ld[row] += acc;
  }
}


// Templated form (K bins known at compile time): one workgroup owns R whole rows, stages their
// parameters through LDS with coalesced 16-byte loads (a lane's 3K+1 values are then a stride-(3K+1)
// LDS read: odd stride, conflict-free), evaluates one (row, transformed dim) pair per thread and
// iteration with everything in registers, and reduces each row's log-dets in a fixed order.
// slabs.S > 0: `params` has not been formed -- it is the sum of S split-K slabs [S][rows][Dt * P] of the last Linear
// (gemm_splitk_kernel) plus its bias: added here, in slab order, while the rows are staged (small batches: the
// reduction launch, the parameter tensor's round trip through HBM and one dependent launch per layer disappear).
struct SlabSum {
  const float* slabs = nullptr;
  const float* bias = nullptr;
  int S = 0;
  size_t stride = 0;   // floats between slabs
  float* theta_out = nullptr;   // training: the summed parameters [rows, N] also go here (the backward walk reads them)
};
template <int K, int INV>
__global__ __launch_bounds__(256) void rqs_rows_kernel(float* __restrict__ y, float* __restrict__ ld,
                                                      const float* __restrict__ params, long long rows, int D,
                                                      int pass_parity, int R, RqsConst c, SlabSum slabs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int P = 3 * K + 1;
  const int Dt = pass_parity == 0 ? D / 2 : (D + 1) / 2;
  const int first_t = pass_parity == 0 ? 1 : 0;
  const int tid = threadIdx.x;
  const int tpr = 256 / R;  // threads per row
  float* red = sm + (size_t)R * Dt * P;  // [256] partial log-dets
  const long long nblk = (rows + R - 1) / R;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long row0 = blk * R;
    const int nrow = (int)((rows - row0) < R ? (rows - row0) : R);
    const int nfl = nrow * Dt * P;
    const float* src = params + (size_t)row0 * Dt * P;
    if (slabs.S > 0) {
      // N % 4 == 0 (the caller's test): 16-byte loads, two per thread and pass with up to four slabs each in flight
      // (one dependent 4-byte load at a time made this loop 50 us for a 9800-wide row: 38 memory latencies in a row)
      const int N = Dt * P;
      const float4* s0 = reinterpret_cast<const float4*>(slabs.slabs + (size_t)row0 * N);
      const float4* b4 = reinterpret_cast<const float4*>(slabs.bias);
      const size_t st4 = slabs.stride >> 2;
      const int n4 = nfl >> 2, N4 = N >> 2;
      for (int q0 = tid; q0 < n4; q0 += 512) {
        const int q1 = q0 + 256;
        const bool two = q1 < n4;
        const float4 c0 = b4[q0 % N4], c1 = two ? b4[q1 % N4] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        for (int sb = 0; sb < slabs.S; sb += 4) {
          float4 t0[4], t1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool on = sb + u < slabs.S;
            const size_t off = (size_t)(on ? sb + u : sb) * st4;
            t0[u] = s0[off + q0];
            t1[u] = s0[off + (two ? q1 : q0)];
            if (!on) t0[u] = t1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {   // slab order: the reduction kernel's order of additions
            v0.x += t0[u].x; v0.y += t0[u].y; v0.z += t0[u].z; v0.w += t0[u].w;
            v1.x += t1[u].x; v1.y += t1[u].y; v1.z += t1[u].z; v1.w += t1[u].w;
          }
        }
        const float4 w0 = make_float4(v0.x + c0.x, v0.y + c0.y, v0.z + c0.z, v0.w + c0.w);
        const float4 w1 = make_float4(v1.x + c1.x, v1.y + c1.y, v1.z + c1.z, v1.w + c1.w);
        reinterpret_cast<float4*>(sm)[q0] = w0;
        if (two) reinterpret_cast<float4*>(sm)[q1] = w1;
        if (slabs.theta_out != nullptr) {
          float4* t4 = reinterpret_cast<float4*>(slabs.theta_out + (size_t)row0 * N);
          t4[q0] = w0;
          if (two) t4[q1] = w1;
        }
      }
    } else if (((size_t)row0 * Dt * P & 3) == 0) {
      for (int q = tid * 4; q < nfl; q += 1024) {
        if (q + 3 < nfl) *reinterpret_cast<float4*>(sm + q) = *reinterpret_cast<const float4*>(src + q);
        else for (int e = q; e < nfl; ++e) sm[e] = src[e];
      }
    } else {
      for (int q = tid; q < nfl; q += 256) sm[q] = src[q];
    }
    __syncthreads();
    const int r = tid / tpr, sub = tid % tpr;
    float acc = 0.f;
    if (r < nrow) {
      float* yr = y + (row0 + r) * D;
      for (int dt = sub; dt < Dt; dt += tpr) {
        const int d = first_t + 2 * dt;
        float res, l;
        rqs_eval_reg<K, INV>(sm + ((size_t)r * Dt + dt) * P, c, yr[d], &res, &l);
        yr[d] = res;
        acc += l;
      }
    }
    red[tid] = acc;
    __syncthreads();
    for (int w = tpr >> 1; w > 0; w >>= 1) {  // fixed-order tree inside each row's thread group
      if (sub < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (sub == 0 && r < nrow) ld[row0 + r] += red[tid];
    __syncthreads();
  }
}

template <int K>
static bool launch_rqs_rows(float* y, float* ld, const float* params, long long rows, int D, int par, int inverse,
                            const RqsConst& rc, hipStream_t stream, const SlabSum& slabs = SlabSum()) {
  const int P = 3 * K + 1;
  const int Dt = par == 0 ? D / 2 : (D + 1) / 2;
  const size_t row_bytes = (size_t)Dt * P * sizeof(float);
  if (row_bytes > 60 * 1024) return false;
  int R = 1;
  while (R < 256 && (size_t)(2 * R) * row_bytes <= 48 * 1024 && 256 / (2 * R) >= 1 && (2 * R) <= rows) R <<= 1;
  const size_t lds = (size_t)R * row_bytes + 256 * sizeof(float);
  const long long nblk = (rows + R - 1) / R;
  const int grid = (int)std::min<long long>(nblk, 65536);
  if (inverse) {
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&rqs_rows_kernel<K, 1>), 64 * 1024) != NF_OK) return false;
    hipLaunchKernelGGL((rqs_rows_kernel<K, 1>), dim3(grid), dim3(256), lds, stream, y, ld, params, rows, D, par, R, rc, slabs);
  } else {
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&rqs_rows_kernel<K, 0>), 64 * 1024) != NF_OK) return false;
    hipLaunchKernelGGL((rqs_rows_kernel<K, 0>), dim3(grid), dim3(256), lds, stream, y, ld, params, rows, D, par, R, rc, slabs);
  }
  return true;
}

// ------------------------------------------------------------------------
// SURVEY's K5: the last Linear of a spline layer with the spline in its epilogue -- the
// [rows, Dt * (3K+1)] parameter tensor (2.6 GB per layer at the MNIST shape) never reaches HBM.
// Columns are laid out so that a 128-column block holds PPB = 128 / (3K+1) whole transformed dims
// (Plan::build_k5); the block's 128 x 128 result goes through LDS in two 64-row halves, every thread
// takes (row, dim) items and applies rqs_eval_reg to y in place; a row's log-dets of the block's dims
// are added in dim order and written to ldpart[column block][row]; ld_reduce_kernel adds the
// column blocks in order.
// ------------------------------------------------------------------------
constexpr int kRqsLdsFloats = (kFastLdsFloats > 64 * kCtStride ? kFastLdsFloats : 64 * kCtStride) + 64 * 8;

template <int K, int INV, bool FULL>
__global__ __launch_bounds__(256, 4) void gemm_rqs_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ y, float* __restrict__ ldpart, int M,
                                                          int Npad, int Kdim, int D, int pass_parity, int gx, int gy,
                                                          RqsConst c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int mb, nb;
  block_to_tile(blockIdx.x, gx, gy, mb, nb);
  const int m0 = mb * FBM, n0 = nb * FBN;
  f32x16 acc[2][2];
  if (FULL) gemm_full_mainloop(A, lda, W, M, Npad, Kdim, m0, n0, smem, acc);
  else gemm_fast_mainloop(A, lda, W, M, Npad, Kdim, m0, n0, smem, acc);

  rqs_tile_epilogue<K, INV>(acc, bias, n0, m0, nb, M, D, pass_parity, y, ldpart, smem, smem + kRqsLdsFloats - 64 * 8, c);
}

__global__ __launch_bounds__(256) void ld_reduce_kernel(const float* __restrict__ ldpart, int nblocks, long long rows,
                                                        float* __restrict__ ld) {
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    // blocks are added in order; their loads are issued sixteen at a time (one dependent load per block made the
    // kernel 32 us for 20 MB at the MNIST shape: 79 memory latencies in a row, four waves per CU)
    float s = 0.f;
    int b = 0;
    for (; b + 16 <= nblocks; b += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = ldpart[(size_t)(b + j) * rows + r];
#pragma unroll
      for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; b < nblocks; ++b) s += ldpart[(size_t)b * rows + r];
    ld[r] += s;
  }
}

// W[k][ (t * P + q) ] (pruned, dense) -> Wk5[k][ (t / PPB) * 128 + (t % PPB) * P + q ], zero elsewhere; row
// `Kdim` of the destination is the bias
__global__ __launch_bounds__(256) void k5_layout_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                        float* __restrict__ Wk5, float* __restrict__ bk5, int Kdim,
                                                        int N, int Npad, int P, int PPB) {
  const long long total = (long long)(Kdim + 1) * Npad;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int k = (int)(q / Npad), col = (int)(q % Npad);
    const int blk = col >> 7, in = col & 127, px = in / P, pq = in % P;
    const int src = (blk * PPB + px) * P + pq;
    const bool live = px < PPB && src < N;
    if (k < Kdim) Wk5[(size_t)k * Npad + col] = live ? W[(size_t)k * N + src] : 0.f;
    else bk5[col] = live ? b[src] : 0.f;
  }
}

// Few rows, many blocks (the small-batch spline layers leave layers x column groups of them: 448 at the MNIST shape):
// one thread per row walks 28 dependent rounds of loads (10 us at 128 rows).  Here sixteen waves share a row's blocks:
// wave w adds blocks [w c, (w + 1) c) of 64 rows (lane = row: coalesced), the sixteen sums are added in wave order.
__global__ __launch_bounds__(1024) void ld_reduce_few_rows_kernel(const float* __restrict__ ldpart, int nblocks, int rows,
                                                                  float* __restrict__ ld) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  const int chunk = (nblocks + 15) / 16;
  const int b0 = wave * chunk, b1 = min(nblocks, b0 + chunk);
  float s = 0.f;
  if (r < rows) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ldpart[(size_t)(b + j) * rows + r];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < b1; ++b) s += ldpart[(size_t)b * rows + r];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && r < rows) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][lane];
    ld[r] += t;
  }
}

static int launch_ld_reduce(const float* ldpart, int nblocks, int M, float* ld, hipStream_t stream) {
  if (M <= 1024 && nblocks >= 64) {
    hipLaunchKernelGGL(ld_reduce_few_rows_kernel, dim3((M + 63) / 64), dim3(1024), 0, stream, ldpart, nblocks, M, ld);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  hipLaunchKernelGGL(ld_reduce_kernel, dim3((unsigned)std::min<long long>(4096, ((long long)M + 255) / 256)), dim3(256), 0,
                     stream, ldpart, nblocks, (long long)M, ld);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

template <int K>
static int launch_gemm_rqs(const float* A, int lda, const float* Wk5, const float* bk5, float* y, float* ldpart,
                           float* ld, int M, int Npad, int Kdim, int D, int par, int inverse, const RqsConst& rc,
                           hipStream_t stream) {
  const int gx = Npad / FBN, gy = (M + FBM - 1) / FBM;
  const size_t lds = kRqsLdsFloats * sizeof(float);
  typedef void (*Kernel)(const float*, int, const float*, const float*, float*, float*, int, int, int, int, int, int, int,
                         RqsConst);
  static const Kernel kernels[2][2] = {{&gemm_rqs_kernel<K, 0, false>, &gemm_rqs_kernel<K, 0, true>},
                                       {&gemm_rqs_kernel<K, 1, false>, &gemm_rqs_kernel<K, 1, true>}};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      NF_DYN_LDS(80 * 1024, kernels[a][b]);
  hipLaunchKernelGGL(kernels[inverse ? 1 : 0][gemm_full_ok(Npad, Kdim) ? 1 : 0], dim3(gx * gy), dim3(256), lds, stream, A,
                     lda, Wk5, bk5, y, ldpart, M, Npad, Kdim, D, par, gx, gy, rc);
  return launch_ld_reduce(ldpart, gx, M, ld, stream);
}

// log_prob = base log-density(z) + logdet; optional float64 block sums.
__global__ __launch_bounds__(256) void base_log_prob_kernel(const float* __restrict__ z, const float* __restrict__ ld,
                                                           float* __restrict__ lp, double* __restrict__ block_sums,
                                                           long long rows, int D, int base) {
  double sum = 0.0;
  for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long long)gridDim.x * 256) {
    const float* v = z + row * D;
    float out;
    if (base == NF_BASE_STD_NORMAL) {
      float ss = 0.f;
      for (int d = 0; d < D; ++d) ss = fmaf(v[d], v[d], ss);
      out = -0.5f * ss - (float)D * 0.9189385332046727f;
    } else {
      bool inside = true;
      for (int d = 0; d < D; ++d) inside = inside && v[d] >= 0.f && v[d] <= 1.f;
      out = inside ? 0.f : -INFINITY;
    }
    if (ld) out += ld[row];
    if (lp) lp[row] = out;
    sum += (double)out;
  }
  if (block_sums) {
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Wide events (D >= 64, the image flows): one wave per row -- coalesced reads, lane sums in a fixed order -- instead of
// one thread walking its row alone (lanes 3 KB apart at D = 784: 122 us for 128 rows).
__global__ __launch_bounds__(256) void base_log_prob_wide_kernel(const float* __restrict__ z, const float* __restrict__ ld,
                                                                float* __restrict__ lp, double* __restrict__ block_sums,
                                                                long long rows, int D, int base) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum = 0.0;  // lane 0 of each wave carries its rows
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const float* v = z + row * D;
    float out;
    if (base == NF_BASE_STD_NORMAL) {
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss = fmaf(v[d], v[d], ss);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
      out = -0.5f * ss - (float)D * 0.9189385332046727f;
    } else {
      int bad = 0;
      for (int d = lane; d < D; d += 64) bad |= !(v[d] >= 0.f && v[d] <= 1.f);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) bad |= __shfl_down(bad, off);
      out = bad ? -INFINITY : 0.f;
    }
    if (lane == 0) {
      if (ld) out += ld[row];
      if (lp) lp[row] = out;
      sum += (double)out;
    }
  }
  if (block_sums) {
    __shared__ double red[4];
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

int launch_base_log_prob(const float* z, const float* ld, float* lp, double* block_sums, long long rows,
                         int D, int base, hipStream_t stream, int* out_grid) {
  if (D >= 64) {
    long long want = (rows + 3) / 4;
    const int grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    if (out_grid) *out_grid = grid;
    hipLaunchKernelGGL(base_log_prob_wide_kernel, dim3(grid), dim3(256), 0, stream, z, ld, lp, block_sums, rows, D, base);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  long long want = (rows + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  if (out_grid) *out_grid = grid;
  hipLaunchKernelGGL(base_log_prob_kernel, dim3(grid), dim3(256), 0, stream, z, ld, lp, block_sums, rows, D, base);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ------------------------------------------------------------------------
// Base sampler: Philox4x32-10, counter = index of a group of 4
// consecutive elements of the row-major [rows, D] matrix, key = the 64-bit seed.  Normals by
// Box-Muller on uniform pairs (u0,u1) -> (n0,n1), (u2,u3) -> (n2,n3); uniforms are 24-bit.
// The reference draws with JAX threefry (src/real_nvp_toy_densities.py:136,
// src/real_nvp.py:52-54); bit parity with it is out of scope, the oracle restates THIS
// generator (oracle/flows_np.py philox_*).
// ------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void base_sample_kernel(float* __restrict__ out, long long total, int base,
                                                         unsigned long long seed) {
  const long long groups = (total + 3) / 4;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)g, (unsigned)(g >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
    float v[4];
    if (base == NF_BASE_UNIFORM01) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)(r[e] >> 8) * 5.9604644775390625e-08f;  // [0, 1)
    } else {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float u0 = ((float)(r[e] >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0, 1)
        const float u1 = ((float)(r[e + 1] >> 8) + 0.5f) * 5.9604644775390625e-08f;
        const float rad = sqrtf(-2.f * logf(u0));
        float sn, cs;
        sincospif(2.f * u1, &sn, &cs);
        v[e] = rad * cs;
        v[e + 1] = rad * sn;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * g + e < total) out[4 * g + e] = v[e];
  }
}

int launch_base_sample(float* out, long long rows, int D, int base, unsigned long long seed, hipStream_t stream) {
  const long long total = rows * D;
  const long long groups = (total + 3) / 4;
  const int grid = (int)std::min<long long>(4096, (groups + 255) / 256);
  hipLaunchKernelGGL(base_sample_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, stream, out, total, base, seed);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ------------------------------------------------------------------------
// Toy-density generators: the reference's host-side tfp samplers
// (src/densities/banana.py:31-41, energies.py:47-127, gaussian_blobs.py:34-59) as one kernel.
// One Philox4x32-10 call per row (counter = row, third counter word 0x44454E53): words 0,1 ->
// a Box-Muller pair (n0, n1), word 2 -> a uniform, word 3 -> component bits.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void density_sample_kernel(float2* __restrict__ out, long long rows, int density,
                                                            unsigned long long seed) {
  const float kPi = 3.14159265358979f;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    unsigned w[4];
    philox4x32_10((unsigned)r, (unsigned)(r >> 32), 0x44454E53u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    const float u0 = ((float)(w[0] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float u1 = ((float)(w[1] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float u2 = (float)(w[2] >> 8) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.f * logf(u0));
    float sn, cs;
    sincospif(2.f * u1, &sn, &cs);
    const float n0 = rad * cs, n1 = rad * sn;
    const bool pick = (w[3] & 1u) != 0u;
    float a, b;
    if (density == NF_DENSITY_BANANA) {               // x2 ~ N(0, sqrt 3); x1 ~ N(x2^2/3 - 2, sqrt(3)/2)
      b = 1.7320508075688772f * n0;
      a = b * b / 3.f - 2.f + 0.8660254037844386f * n1;
    } else if (density == NF_DENSITY_ENERGY1) {       // z ~ N(0,2); v ~ N(+-(2 - z^2/32), 0.3); row (v, z)
      b = 2.f * n0;
      const float mu = 2.f - b * b / 32.f;
      a = (pick ? mu : -mu) + 0.3f * n1;
    } else if (density == NF_DENSITY_BLOBS) {         // 8 blobs on the circle of radius 2, sigma 0.3
      float sk, ck;
      sincospif((float)(w[3] & 7u) * 0.25f, &sk, &ck);
      a = 2.f * ck + 0.3f * n0;
      b = 2.f * sk + 0.3f * n1;
    } else {                                          // energy2-4: x1 ~ U(-1.5 pi, 1.5 pi), x2 ~ N(m(x1), 0.5)
      a = (u2 * 3.f - 1.5f) * kPi;
      float m = sinpif(0.5f * a);                     // w1(z) = sin(2 pi z / 4)
      if (density == NF_DENSITY_ENERGY3 && pick) { const float t = (a - 1.f) / 0.6f; m -= 3.f * expf(-0.5f * t * t); }
      if (density == NF_DENSITY_ENERGY4 && pick) m += 3.f / (1.f + expf(-(a - 1.f) / 0.3f));
      b = m + 0.5f * n0;
    }
    out[r] = make_float2(a, b);
  }
}

int launch_density_sample(float* out, long long rows, int density, unsigned long long seed, hipStream_t stream) {
  const int grid = (int)std::min<long long>(4096, (rows + 255) / 256);
  hipLaunchKernelGGL(density_sample_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, stream,
                     reinterpret_cast<float2*>(out), rows, density, seed);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// Streaming read of a [rows, 2] float32 matrix with the hot path's access shape (one float2
// per lane); exists to calibrate rocprofv3's FETCH_SIZE on a known byte count
// (MI355X_MICROARCH.md, HBM section: widths other than 16 B/lane are uncalibrated).
__global__ __launch_bounds__(256) void debug_read_f2_kernel(const float2* __restrict__ x, long long rows,
                                                           float* __restrict__ out) {
  float acc = 0.f;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float2 v = x[r];
    acc += v.x + v.y;
  }
  if (acc == 123.456f) out[0] = acc;  // keeps the loads alive, practically never stores
}

int launch_debug_read_f2(const float* x, long long rows, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(debug_read_f2_kernel, dim3(2048), dim3(256), 0, stream,
                     reinterpret_cast<const float2*>(x), rows, out);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ------------------------------------------------------------------------
// host side: weights and the layer walk
// ------------------------------------------------------------------------
int Plan::build(const nf_flow_desc& d, const float* packed) {
  release();
  desc = d;
  std::vector<int> in, out;
  const int nlin = d.num_hidden + 1;
  int first_in, last_out;
  const int P = 3 * d.rqs_bins + 1;
  if (d.kind == NF_KIND_AFFINE_SPLIT) { first_in = d.event_dim / 2; last_out = d.event_dim; }
  else if (d.kind == NF_KIND_AFFINE_MASKED) { first_in = d.event_dim; last_out = 2; }
  else { first_in = d.event_dim; last_out = d.event_dim * P; }
  int cur = first_in;
  for (int j = 0; j < d.num_hidden; ++j) { in.push_back(cur); out.push_back(d.hidden[j]); cur = d.hidden[j]; }
  in.push_back(cur); out.push_back(last_out);
  size_t per_layer = 0;
  for (int j = 0; j < nlin; ++j) per_layer += (size_t)in[j] * out[j] + out[j];

  max_width = 0;
  act_width = 0;
  std::vector<float> host;
  canon_index.clear();
  linears.assign((size_t)d.num_layers * nlin, Linear());
  for (int l = 0; l < d.num_layers; ++l) {
    const float* src = packed + per_layer * l;
    const int pass_parity = (d.first_mask_parity + l) & 1;  // masked kinds
    for (int j = 0; j < nlin; ++j) {
      const float* W = src;
      const float* b = src + (size_t)in[j] * out[j];
      src = b + out[j];
      Linear& L = linears[(size_t)l * nlin + j];
      L.K = in[j];
      L.N = out[j];
      std::vector<float> w(W, W + (size_t)in[j] * out[j]);
      std::vector<float> bb(b, b + out[j]);
      // where each kept value sits in `packed` (-1: a structural zero): weights, then biases
      std::vector<int> wi(w.size()), bi(bb.size());
      for (size_t q = 0; q < w.size(); ++q) wi[q] = (int)(W - packed + q);
      for (size_t q = 0; q < bb.size(); ++q) bi[q] = (int)(b - packed + q);
      if (d.kind != NF_KIND_AFFINE_SPLIT && j == 0) {
        // where(mask, x, 0) @ W  ==  x @ W with the transformed dims' rows zeroed
        for (int r = 0; r < in[j]; ++r)
          if ((r & 1) != pass_parity)
            for (int c = 0; c < out[j]; ++c) { w[(size_t)r * out[j] + c] = 0.f; wi[(size_t)r * out[j] + c] = -1; }
      }
      if (d.kind == NF_KIND_RQS_MASKED && j == nlin - 1) {
        // keep the parameter columns of the transformed dims only
        const int D = d.event_dim;
        std::vector<int> tdims;
        for (int e = 0; e < D; ++e) if ((e & 1) != pass_parity) tdims.push_back(e);
        const int Nn = (int)tdims.size() * P;
        std::vector<float> w2((size_t)in[j] * Nn), b2(Nn);
        std::vector<int> wi2(w2.size()), bi2(Nn);
        for (size_t t = 0; t < tdims.size(); ++t)
          for (int q = 0; q < P; ++q) {
            const int cs = tdims[t] * P + q, cd = (int)t * P + q;
            b2[cd] = bb[cs];
            bi2[cd] = bi[cs];
            for (int r = 0; r < in[j]; ++r) {
              w2[(size_t)r * Nn + cd] = w[(size_t)r * out[j] + cs];
              wi2[(size_t)r * Nn + cd] = wi[(size_t)r * out[j] + cs];
            }
          }
        w.swap(w2);
        bb.swap(b2);
        wi.swap(wi2);
        bi.swap(bi2);
        L.N = Nn;
      }
      max_width = std::max(max_width, L.N);
      if (j < nlin - 1) act_width = std::max(act_width, L.N);
      else param_width = std::max(param_width, L.N);
      L.w_off = host.size();
      host.insert(host.end(), w.begin(), w.end());
      canon_index.insert(canon_index.end(), wi.begin(), wi.end());
      while (host.size() % 4) { host.push_back(0.f); canon_index.push_back(-1); }
      L.b_off = host.size();
      host.insert(host.end(), bb.begin(), bb.end());
      canon_index.insert(canon_index.end(), bi.begin(), bi.end());
      while (host.size() % 4) { host.push_back(0.f); canon_index.push_back(-1); }
    }
  }
  NF_HIP(hipMalloc(&d_weights, host.size() * sizeof(float)));
  NF_HIP(hipMemcpy(d_weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  host_weights.swap(host);
  if (d_wsmr) { hipFree(d_wsmr); d_wsmr = nullptr; }   // the small-batch copy of the last Linears: laid again on first use
  wsmr_stale = true;
  // compact first Linear (masked kinds, wide events; see nf_internal.h)
  w1c = false;
  if (d_w1c) { hipFree(d_w1c); d_w1c = nullptr; }
  if (d.kind != NF_KIND_AFFINE_SPLIT && d.event_dim >= 128 && !switches().no_compact_l1) {
    w1c_k[0] = (((d.event_dim + 1) / 2) + 3) & ~3;
    w1c_k[1] = ((d.event_dim / 2) + 3) & ~3;
    w1c_off.assign(d.num_layers, 0);
    size_t total = 0;
    for (int l = 0; l < d.num_layers; ++l) {
      w1c_off[l] = total;
      total += (size_t)w1c_k[(d.first_mask_parity + l) & 1] * linears[(size_t)l * nlin].N;
    }
    NF_HIP(hipMalloc(&d_w1c, total * sizeof(float)));
    w1c = true;
    NF_TRY(refresh_w1c(nullptr));
    NF_HIP(hipStreamSynchronize(nullptr));
  }
  // K5: the last Linear of a spline layer through gemm_rqs_kernel
  k5 = false;
  if (d_wk5) { hipFree(d_wk5); d_wk5 = nullptr; }
  if (d.kind == NF_KIND_RQS_MASKED && !switches().no_k5 &&
      (d.rqs_bins == 6 || d.rqs_bins == 8 || d.rqs_bins == 10 || d.rqs_bins == 12)) {
    const Linear& last = linears[nlin - 1];
    const int ppb = 128 / P;
    int dt_max = (d.event_dim + 1) / 2;
    if ((last.K & 3) == 0 && last.K >= 32 && !(last.K <= 64 && last.N <= 64)) {
      k5_npad = ((dt_max + ppb - 1) / ppb) * 128;
      k5_layer_floats = (size_t)(last.K + 1) * k5_npad;
      NF_HIP(hipMalloc(&d_wk5, k5_layer_floats * d.num_layers * sizeof(float)));
      k5 = true;
      k5_stale = false;
      NF_TRY(refresh_k5(nullptr));
      NF_HIP(hipDeviceSynchronize());
    }
  }
  if (x3) return set_x3(true);  // new weights under an option that was already on
  return NF_OK;
}

int Plan::refresh_k5(hipStream_t stream) {
  if (!k5) return NF_OK;
  const int nlin = desc.num_hidden + 1, P = 3 * desc.rqs_bins + 1, ppb = 128 / P;
  for (int l = 0; l < desc.num_layers; ++l) {
    const Linear& L = linears[(size_t)l * nlin + nlin - 1];
    float* dst = d_wk5 + (size_t)l * k5_layer_floats;
    const long long total = (long long)(L.K + 1) * k5_npad;
    hipLaunchKernelGGL(k5_layout_kernel, dim3((unsigned)std::min<long long>(2048, (total + 255) / 256)), dim3(256), 0,
                       stream, d_weights + L.w_off, d_weights + L.b_off, dst, dst + (size_t)L.K * k5_npad, L.K, L.N,
                       k5_npad, P, ppb);
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// Build (or drop) the split-bf16 weight images of every Linear wide enough for gemm_x3_kernel.
int Plan::set_x3(bool on) {
  if (d_wx3) { hipFree(d_wx3); d_wx3 = nullptr; }
  x3 = on;
  if (!on || linears.empty()) return NF_OK;
  size_t total = 0;
  for (Linear& L : linears) {
    L.x3_off = total;
    if (L.N >= 64 && L.K >= 32) total += x3_image_halfwords(L.K, L.N);
  }
  // K5 under the mode: the last Linear of every layer once more, in the K5 column layout
  const int nlin = desc.num_hidden + 1, P = 3 * desc.rqs_bins + 1, ppb = k5 ? 128 / P : 1;
  k5_x3_off.assign(desc.num_layers, 0);
  if (k5)
    for (int l = 0; l < desc.num_layers; ++l) {
      k5_x3_off[l] = total;
      total += x3_image_halfwords(linears[(size_t)l * nlin + nlin - 1].K, k5_npad);
    }
  if (total == 0) return NF_OK;
  std::vector<unsigned short> img(total);
  for (const Linear& L : linears)
    if (L.N >= 64 && L.K >= 32) pack_w_x3(host_weights.data() + L.w_off, L.K, L.N, img.data() + L.x3_off);
  if (k5) {
    std::vector<float> wk;
    for (int l = 0; l < desc.num_layers; ++l) {
      const Linear& L = linears[(size_t)l * nlin + nlin - 1];
      wk.assign((size_t)L.K * k5_npad, 0.f);
      const float* W = host_weights.data() + L.w_off;
      for (int col = 0; col < k5_npad; ++col) {  // same map as k5_layout_kernel
        const int blk = col >> 7, in = col & 127, px = in / P, pq = in % P;
        const int src = (blk * ppb + px) * P + pq;
        if (px >= ppb || src >= L.N) continue;
        for (int k = 0; k < L.K; ++k) wk[(size_t)k * k5_npad + col] = W[(size_t)k * L.N + src];
      }
      pack_w_x3(wk.data(), L.K, k5_npad, img.data() + k5_x3_off[l]);
    }
  }
  NF_HIP(hipMalloc(&d_wx3, total * sizeof(unsigned short)));
  NF_HIP(hipMemcpy(d_wx3, img.data(), total * sizeof(unsigned short), hipMemcpyHostToDevice));
  return NF_OK;
}

void Plan::release() {
  if (d_weights) { hipFree(d_weights); d_weights = nullptr; }
  if (d_wk5) { hipFree(d_wk5); d_wk5 = nullptr; }
  k5 = false;
  if (d_w1c) { hipFree(d_w1c); d_w1c = nullptr; }
  w1c = false;
  if (d_wsmr) { hipFree(d_wsmr); d_wsmr = nullptr; }
  hipFree(bwd.d_grad); hipFree(bwd.d_wT); hipFree(bwd.d_canon_index); hipFree(bwd.d_zero); hipFree(bwd.d_partials);
  hipFree(bwd.d_small_partials); hipFree(bwd.d_live_bits);
  hipFree(bwd.d_scratch); hipFree(bwd.d_tape); hipFree(bwd.d_acts); hipFree(bwd.d_jobs);
  if (bwd.side) { hipStreamSynchronize(bwd.side); hipStreamDestroy(bwd.side); }
  for (hipEvent_t e : bwd.ev) hipEventDestroy(e);
  hipFree(bwd.d_layer_grads);
  bwd = Backward();
  if (d_wx3) { hipFree(d_wx3); d_wx3 = nullptr; }
  host_weights.clear();
  if (d_scratch) { hipFree(d_scratch); d_scratch = nullptr; }
  scratch_rows = 0;
  linears.clear();
}

int Plan::ensure_scratch(long long rows) {
  // per row: two activation ping-pong buffers + the conditioner's output; up to 4 GiB of the
  // 288 GB so that even the 784-D flow walks 2^16 rows in one chunk (big M keeps the GEMMs of the
  // narrow hidden layers at full occupancy)
  const size_t per_row = ((size_t)2 * act_width + param_width + (k5 ? k5_npad / 128 : 0) +
                          (w1c ? std::max(w1c_k[0], w1c_k[1]) : 0)) * sizeof(float);
  long long cap = (long long)(((size_t)4 << 30) / per_row);
  cap = std::max<long long>(128, cap / 128 * 128);
  const long long want = std::min(rows, cap);
  if (want <= scratch_rows) return NF_OK;
  if (d_scratch) { hipFree(d_scratch); d_scratch = nullptr; scratch_rows = 0; }
  NF_HIP(hipMalloc(&d_scratch, per_row * (size_t)want + 16));
  scratch_rows = want;
  xc_off = ((((size_t)2 * act_width + param_width + (k5 ? k5_npad / 128 : 0)) * (size_t)want) + 3) & ~(size_t)3;
  return NF_OK;
}

RqsConst make_rqs_const(const nf_flow_desc& d) {
  RqsConst rc;
  rc.K = d.rqs_bins;
  rc.range_min = d.rqs_range_min;
  rc.range_max = d.rqs_range_max;
  rc.min_bin = d.rqs_min_bin_size;
  rc.min_slope = d.rqs_min_knot_slope;
  rc.slope_offset = logf(expf(1.f - d.rqs_min_knot_slope) - 1.f);
  return rc;
}

// One coupling transform on rows [0, n): y and ld updated in place from the conditioner output.
int launch_transform(const nf_flow_desc& d, float* yc, float* ldc, const float* params, long long n, int par,
                     int inverse_op, const RqsConst& rc, hipStream_t stream) {
  const int D = d.event_dim;
  if (d.kind == NF_KIND_AFFINE_MASKED) {
    const long long total = n * D;
    const int grid = (int)std::min<long long>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(affine_masked_kernel, dim3(grid), dim3(256), 0, stream, yc, ldc, params, n, D, par, inverse_op);
  } else if (d.kind == NF_KIND_AFFINE_SPLIT) {
    const int grid = (int)std::min<long long>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(affine_split_kernel, dim3(grid), dim3(256), 0, stream, yc, ldc, params, n, D, par, inverse_op);
  } else {
    bool done = false;
    if (!switches().rqs_simple) {
      switch (rc.K) {  // the bin counts the register form is built for; others take the run-time-K kernel below
        case 4: done = launch_rqs_rows<4>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        case 6: done = launch_rqs_rows<6>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        case 8: done = launch_rqs_rows<8>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        case 10: done = launch_rqs_rows<10>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        case 12: done = launch_rqs_rows<12>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        case 16: done = launch_rqs_rows<16>(yc, ldc, params, n, D, par, inverse_op, rc, stream); break;
        default: break;
      }
    }
    if (!done) {
      const int Dt = par == 0 ? D / 2 : (D + 1) / 2;
      int G = 1;
      while (G < 64 && G < Dt) G <<= 1;
      const long long lanes = n * G;
      const int grid = (int)std::min<long long>(8192, (lanes + 255) / 256);
      hipLaunchKernelGGL(rqs_masked_kernel, dim3(grid), dim3(256), 0, stream, yc, ldc, params, n, D, par, inverse_op,
                         G, rc);
    }
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// ---- compact first Linear of wide masked flows (Plan::w1c) --------------------------------------------------------
// xc[row][j] = y[row][2 j + parity] for j < kc (the dims the layer passes through and shows its conditioner), 0 up to kpad
__global__ __launch_bounds__(256) void gather_pass_dims_kernel(const float* __restrict__ y, int D, float* __restrict__ xc,
                                                               int kc, int kpad, long long rows, int parity) {
  const long long total = rows * (long long)kpad;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / kpad;
    const int j = (int)(e - row * kpad);
    xc[e] = j < kc ? y[row * D + 2 * j + parity] : 0.f;
  }
}
// the same for D % 4 == 0 (kc = D / 2, no padding): one 16-byte load and one 8-byte store per thread
__global__ __launch_bounds__(256) void gather_pass_dims4_kernel(const float4* __restrict__ y4, float2* __restrict__ xc2,
                                                                long long quads, int parity) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < quads; e += (long long)gridDim.x * 256) {
    const float4 v = y4[e];
    xc2[e] = parity ? make_float2(v.y, v.w) : make_float2(v.x, v.z);
  }
}
// wc[j][n] = W[2 j + parity][n] (the rows d_weights keeps non-zero), zero rows up to kpad
__global__ __launch_bounds__(256) void gather_live_rows_kernel(const float* __restrict__ W, float* __restrict__ wc, int N,
                                                               int kc, int kpad, int parity) {
  const long long total = (long long)kpad * N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e / N), n = (int)(e - (long long)j * N);
    wc[e] = j < kc ? W[(size_t)(2 * j + parity) * N + n] : 0.f;
  }
}

int Plan::refresh_w1c(hipStream_t stream) {
  const int nlin = desc.num_hidden + 1;
  for (int l = 0; l < desc.num_layers; ++l) {
    const Linear& L = linears[(size_t)l * nlin];
    const int par = (desc.first_mask_parity + l) & 1;
    const int kc = par == 0 ? (desc.event_dim + 1) / 2 : desc.event_dim / 2;
    const long long total = (long long)w1c_k[par] * L.N;
    hipLaunchKernelGGL(gather_live_rows_kernel, dim3((unsigned)std::min<long long>(2048, (total + 255) / 256)), dim3(256), 0,
                       stream, d_weights + L.w_off, d_w1c + w1c_off[l], L.N, kc, w1c_k[par], par);
  }
  NF_HIP(hipGetLastError());
  w1c_stale = false;
  return NF_OK;
}

int Plan::ensure_wsmr(hipStream_t stream) {
  const int nlin = desc.num_hidden + 1;
  if (d_wsmr == nullptr) {
    wsmr_off.assign(desc.num_layers, 0);
    size_t total = 0;
    for (int l = 0; l < desc.num_layers; ++l) {
      const Linear& L = linears[(size_t)l * nlin + nlin - 1];
      wsmr_off[l] = total;
      total += small_m_rqs_packed_floats(L.N, L.K, desc.rqs_bins);
    }
    NF_HIP(hipMalloc(&d_wsmr, std::max<size_t>(total, 4) * sizeof(float)));
    wsmr_stale = true;
  }
  if (wsmr_stale) {
    for (int l = 0; l < desc.num_layers; ++l) {
      const Linear& L = linears[(size_t)l * nlin + nlin - 1];
      NF_TRY(launch_small_m_rqs_pack(d_weights + L.w_off, d_wsmr + wsmr_off[l], L.N, L.K, desc.rqs_bins, stream));
    }
    wsmr_stale = false;
  }
  return NF_OK;
}

// The last Linear of a spline layer at a small batch: split-K slabs (the test is launch_gemm's), added up inside the
// spline kernel instead of by a reduction launch.  *done = false: not this shape -- the caller takes the plain route.
static int launch_gemm_splitk_rqs(const nf_flow_desc& d, const float* A, int lda, const float* W, const float* bias,
                                  int M, int N, int K, float* yc, float* ldc, int par, int inverse_op, const RqsConst& rc,
                                  hipStream_t stream, bool* done, float* theta_out = nullptr) {
  *done = false;
  const int D = d.event_dim;
  const int Dt = par == 0 ? D / 2 : (D + 1) / 2;
  const bool aligned = (lda & 3) == 0 && (N & 3) == 0 && (K & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
  if (!aligned || N < 128 || K < 32 || switches().gemm_simple || switches().no_splitk || switches().rqs_simple ||
      gemm_small_m_applicable(A, lda, M, N, K) || N != Dt * (3 * rc.K + 1))
    return NF_OK;
  const int gx = (N + FBN - 1) / FBN, gy = (M + FBM - 1) / FBM;
  const int tiles = gx * gy, ktiles = (K + FBK - 1) / FBK;
  if (tiles > 128 || ktiles < 8) return NF_OK;
  if (rc.K != 4 && rc.K != 6 && rc.K != 8 && rc.K != 10 && rc.K != 12 && rc.K != 16) return NF_OK;
  if ((size_t)N * sizeof(float) > 60 * 1024) return NF_OK;   // a row's parameters must fit the spline kernel's LDS
  NF_DYN_LDS(80 * 1024, &gemm_splitk_kernel);
  int S = std::min(std::max(1, 256 / tiles), ktiles / 4);   // one resident round of workgroups (77 tiles: 3 x 77 = 231 measured 549 us per MNIST-shape call against 575 with 4 x 77)
  if (switches().splitk_s) S = std::max(1, std::min(switches().splitk_s, ktiles));
  const int Kc = ((ktiles + S - 1) / S) * FBK;
  S = (K + Kc - 1) / Kc;
  float* partial = splitk_scratch(stream, (size_t)S * M * N);
  if (partial == nullptr) { set_error("out of device memory for the split-K slabs"); return NF_ERR_HIP; }
  hipLaunchKernelGGL(gemm_splitk_kernel, dim3(tiles * S), dim3(256), kFastLdsFloats * sizeof(float), stream, A, lda, W,
                     partial, M, N, K, Kc, gx, gy);
  NF_HIP(hipGetLastError());
  SlabSum ss;
  ss.slabs = partial;
  ss.bias = bias;
  ss.S = S;
  ss.stride = (size_t)M * N;
  ss.theta_out = theta_out;
  bool ok = false;
  switch (rc.K) {
    case 4: ok = launch_rqs_rows<4>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
    case 6: ok = launch_rqs_rows<6>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
    case 8: ok = launch_rqs_rows<8>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
    case 10: ok = launch_rqs_rows<10>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
    case 12: ok = launch_rqs_rows<12>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
    case 16: ok = launch_rqs_rows<16>(yc, ldc, nullptr, M, D, par, inverse_op, rc, stream, ss); break;
  }
  if (!ok) { set_error("spline kernel refused a shape it was checked for"); return NF_ERR_UNSUPPORTED; }
  NF_HIP(hipGetLastError());
  *done = true;
  return NF_OK;
}

// y (in place) and ld (in place) already hold the chain's input state.
int Plan::walk(float* y, float* ld, const ChainIO& io, hipStream_t stream) {
  const nf_flow_desc& d = desc;
  const int D = d.event_dim;
  const int nlin = d.num_hidden + 1;
  const int nlay = io.layer_end - io.layer_begin;
  NF_TRY(ensure_scratch(io.rows));
  if (k5 && k5_stale && !record_tape) {
    NF_TRY(refresh_k5(stream));
    k5_stale = false;
  }
  if (w1c && w1c_stale && !record_tape) NF_TRY(refresh_w1c(stream));
  const int act = d.activation == NF_ACT_RELU ? 1 : 2;
  const RqsConst rc = make_rqs_const(d);
  // 257..512 rows of a wide masked spline flow: in two blocks through the small-batch kernels (0.76 / 0.95 ms at the
  // MNIST shape and 384 / 512 rows against 0.89 / 1.10 ms through the split-K route; three blocks -- 640 rows,
  // 1.23 ms -- lose to the large-batch kernels)
  long long step_rows = scratch_rows;
  if (d.kind == NF_KIND_RQS_MASKED && !record_tape && !(x3 && d_wx3) && io.rows > 256 && io.rows <= 512 && nlin > 1 &&
      !switches().no_splitk) {
    const Linear& L = linears[(size_t)io.layer_begin * nlin + nlin - 1];
    const int par = (d.first_mask_parity + io.layer_begin) & 1;
    if (L.K >= 8 * 32 && (L.K & 3) == 0 &&
        small_m_rqs_groups(d_scratch, L.K, 256, L.N, L.K, par == 0 ? D / 2 : (D + 1) / 2, rc.K) > 0)
      step_rows = std::min<long long>(scratch_rows, 256);
  }
  for (long long r0 = 0; r0 < io.rows; r0 += step_rows) {
    const long long n = std::min(step_rows, io.rows - r0);
    float* yc = y + r0 * D;
    float* ldc = ld + r0;
    float* buf[3] = {d_scratch, d_scratch + (size_t)scratch_rows * act_width,
                     d_scratch + (size_t)2 * scratch_rows * act_width};
    int ld_parts = 0;                // [column group][row] log-det partials waiting to be added into ldc
    float* ld_parts_at = nullptr;
    for (int li = 0; li < nlay; ++li) {
      const int l = io.descending ? io.layer_end - 1 - li : io.layer_begin + li;
      const int par = (d.first_mask_parity + l) & 1;  // masked: pass-through parity; split: flip
      const float* a = yc;
      int lda = D;
      if (d.kind == NF_KIND_AFFINE_SPLIT && par) a = yc + D / 2;  // flipped: second half conditions
      const Linear* Ls = &linears[(size_t)l * nlin];
      bool fused_transform = false;
      const bool narrow = !(x3 && d_wx3) && mlp_small_applicable(Ls, nlin);
      // training: keep every layer's input (indexed by LAYER, not by position); the narrow conditioner's
      // kernel writes it while it loads its input tile
      float* tape = record_tape ? bwd.d_tape + ((size_t)l * bwd.tape_rows + r0) * D : nullptr;
      const bool tape_in_kernel = narrow && a == yc && Ls[0].K == D;
      // small batches: the first Linear's one-launch kernel leaves its input in the tape as it reads it
      const bool tape_by_l1 = record_tape && !narrow && nlin > 1 && a == yc && Ls[0].K == D && !(x3 && d_wx3) &&
                              !switches().gemm_simple && gemm_small_m16_applicable(yc, D, (int)n, Ls[0].N, Ls[0].K);
      if (record_tape && !tape_in_kernel && !tape_by_l1)
        NF_HIP(hipMemcpyAsync(tape, yc, sizeof(float) * n * D, hipMemcpyDeviceToDevice, stream));
      if (narrow) {
        // narrow conditioner: one launch, activations stay in LDS (recorded only for the training walk)
        float* outs[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < nlin; ++j) {
          if (record_tape) outs[j] = bwd.d_acts + bwd.act_off[(size_t)l * nlin + j] + (size_t)r0 * Ls[j].N;
          else if (j == nlin - 1) outs[j] = buf[2];
        }
        NF_TRY(launch_mlp_small(a, lda, Ls, nlin, d_weights, act, outs, n, stream, tape_in_kernel ? tape : nullptr));
        a = outs[nlin - 1];
      } else
      for (int j = 0; j < nlin; ++j) {
        const Linear& L = linears[(size_t)l * nlin + j];
        // (a batch of one or two row blocks leaves the fused kernel ~80 workgroups that each walk the whole K; the
        // split-K GEMM + the transform kernel are quicker there: 45 -> 32 us per layer at the MNIST shape, batch 128)
        // (<= 256 rows of a wide flow: nf_small_m_rqs.hip's one-launch last Linear + spline -- 580 -> 480 us per
        // 8-layer call at the MNIST shape and 192 / 256 rows, where this kernel's 158 workgroups took 50 + 5 us per layer)
        const bool small_m_rqs_first =
            j == nlin - 1 && d.kind == NF_KIND_RQS_MASKED && !record_tape && !(x3 && d_wx3) && n <= 256 && L.K >= 8 * 32 &&
            !switches().no_splitk &&   // (NF_NO_SPLITK: the kernels of large batches on any batch)
            small_m_rqs_groups(a, lda, (int)n, L.N, L.K, par == 0 ? D / 2 : (D + 1) / 2, rc.K) > 0;
        const bool k5_small = small_m_rqs_first ||
                              ((long long)(k5_npad / 128) * ((n + 127) / 128) <= 128 && L.K >= 8 * 32 &&
                               !switches().no_splitk);
        if (j == nlin - 1 && k5 && !k5_small && !record_tape && (lda & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(a) & 15) == 0) {
          // K5: GEMM + spline in one kernel; theta stays in LDS
          const float* wk = d_wk5 + (size_t)l * k5_layer_floats;
          float* ldpart = d_scratch + ((size_t)2 * act_width + param_width) * scratch_rows;
          int st = NF_ERR_UNSUPPORTED;
          if (x3 && d_wx3) {
            st = launch_gemm_x3_rqs(a, lda, d_wx3 + k5_x3_off[l], wk + (size_t)L.K * k5_npad, yc, ldpart, (int)n,
                                    k5_npad, L.K, D, par, io.inverse_op, rc, stream);
            if (st == NF_OK) st = launch_ld_reduce(ldpart, k5_npad / 128, (int)n, ldc, stream);
          } else
          switch (rc.K) {
            case 6: st = launch_gemm_rqs<6>(a, lda, wk, wk + (size_t)L.K * k5_npad, yc, ldpart, ldc, (int)n, k5_npad, L.K, D, par, io.inverse_op, rc, stream); break;
            case 8: st = launch_gemm_rqs<8>(a, lda, wk, wk + (size_t)L.K * k5_npad, yc, ldpart, ldc, (int)n, k5_npad, L.K, D, par, io.inverse_op, rc, stream); break;
            case 10: st = launch_gemm_rqs<10>(a, lda, wk, wk + (size_t)L.K * k5_npad, yc, ldpart, ldc, (int)n, k5_npad, L.K, D, par, io.inverse_op, rc, stream); break;
            case 12: st = launch_gemm_rqs<12>(a, lda, wk, wk + (size_t)L.K * k5_npad, yc, ldpart, ldc, (int)n, k5_npad, L.K, D, par, io.inverse_op, rc, stream); break;
          }
          NF_TRY(st);
          fused_transform = true;
          break;
        }
        if (j == nlin - 1 && d.kind == NF_KIND_RQS_MASKED && !record_tape && !(x3 && d_wx3) && n <= 256) {
          // small batch, float32: the Linear and the spline in one launch on every CU (nf_small_m_rqs.hip); the
          // log-det partials of all layers are added after the last one
          const int Dt = par == 0 ? D / 2 : (D + 1) / 2;
          const int cgs = small_m_rqs_groups(a, lda, (int)n, L.N, L.K, Dt, rc.K);
          if (cgs > 0) {
            const int cg_max = (D + 1) / 2;   // >= any layer's column groups
            float* ldp = splitk_scratch(stream, (size_t)nlay * cg_max * n, 1);
            if (ldp == nullptr) { set_error("out of device memory for the log-det partials"); return NF_ERR_HIP; }
            NF_TRY(ensure_wsmr(stream));
            NF_TRY(launch_small_m_rqs(a, lda, d_wsmr + wsmr_off[l], d_weights + L.b_off, yc, ldp + (size_t)ld_parts * n,
                                      nullptr, (int)n, L.N, L.K, D, par, io.inverse_op, rc, stream));
            ld_parts += cgs;
            ld_parts_at = ldp;
            fused_transform = true;
            break;
          }
        }
        if (j == nlin - 1 && d.kind == NF_KIND_RQS_MASKED && !(x3 && d_wx3)) {
          // small batch, float32: the slabs of the split-K product go straight into the spline kernel (training: which
          // also leaves the summed parameters where the backward walk looks for them)
          bool done = false;
          float* theta = record_tape ? bwd.d_acts + bwd.act_off[(size_t)l * nlin + j] + (size_t)r0 * L.N : nullptr;
          NF_TRY(launch_gemm_splitk_rqs(d, a, lda, d_weights + L.w_off, d_weights + L.b_off, (int)n, L.N, L.K, yc, ldc, par,
                                        io.inverse_op, rc, stream, &done, theta));
          if (done) { fused_transform = true; break; }
        }
        float* dst = buf[j == nlin - 1 ? 2 : (j & 1)];
        if (record_tape)  // training: every Linear's output stays (hidden activations and theta)
          dst = bwd.d_acts + bwd.act_off[(size_t)l * nlin + j] + (size_t)r0 * L.N;
        // bf16x3 mode: grids of a few tiles (small batches) take the float32 GEMM, which splits K over the chip
        // (the same test as launch_gemm's)
        const bool few_tiles = (long long)((L.N + 127) / 128) * ((n + 127) / 128) <= 128 && L.K >= 8 * 32 &&
                               !switches().no_splitk;
        if (x3 && d_wx3 && !few_tiles && gemm_x3_applicable(a, lda, L.N, L.K))
          NF_TRY(launch_gemm_x3(a, lda, d_wx3 + L.x3_off, d_weights + L.b_off, dst, L.N, (int)n, L.N, L.K,
                                j == nlin - 1 ? 0 : act, stream));
        else if (j == 0 && w1c && !record_tape && n >= 1024 && a == yc && nlin > 1) {
          // wide masked flow, large batch: the pass-through dims gathered into a dense matrix, the live rows of W1
          const int kpad = w1c_k[par], kc = par == 0 ? (D + 1) / 2 : D / 2;
          float* xc = d_scratch + xc_off;
          if ((D & 3) == 0 && kpad == D / 2) {
            const long long quads = n * (long long)(D / 4);
            hipLaunchKernelGGL(gather_pass_dims4_kernel, dim3((unsigned)std::min<long long>(16384, (quads + 255) / 256)),
                               dim3(256), 0, stream, reinterpret_cast<const float4*>(yc), reinterpret_cast<float2*>(xc),
                               quads, par);
          } else {
            const long long total = n * (long long)kpad;
            hipLaunchKernelGGL(gather_pass_dims_kernel, dim3((unsigned)std::min<long long>(16384, (total + 255) / 256)),
                               dim3(256), 0, stream, yc, D, xc, kc, kpad, n, par);
          }
          NF_TRY(launch_gemm(xc, kpad, d_w1c + w1c_off[l], d_weights + L.b_off, dst, L.N, (int)n, L.N, kpad, act, stream));
        } else
          NF_TRY(launch_gemm(a, lda, d_weights + L.w_off, d_weights + L.b_off, dst, L.N, (int)n, L.N, L.K,
                             j == nlin - 1 ? 0 : act, stream, nullptr, (j == 0 && tape_by_l1) ? tape : nullptr));
        a = dst;
        lda = L.N;
      }
      if (!fused_transform)
        NF_TRY(launch_transform(d, yc, ldc, a /* the last Linear's output */, n, par, io.inverse_op, rc, stream));
      NF_HIP(hipGetLastError());
    }
    if (ld_parts > 0) NF_TRY(launch_ld_reduce(ld_parts_at, ld_parts, (int)n, ldc, stream));
  }
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
