// General path, training, narrow conditioners: the MLP backward of one coupling layer in ONE launch when
// every Linear has at most 64 inputs and 64 outputs (the counterpart of nf_mlp_small.hip; the update of
// src/maf_toy_densities.py:167-176 / src/real_nvp_toy_densities.py:164-173 for 2-D and low-dimensional
// flows).  Plan::backward's generic form runs, per Linear, a weight-gradient product, its reduction and a
// data-gradient GEMM (9 launches per layer, every gradient matrix through HBM); here a wave keeps a
// 32-row tile's gradient in LDS from d(theta) down to dL/dx and the weight gradients of ALL the
// layer's Linears in registers across all its tiles.
//
// Per 32-row tile and Linear j (last to first), with g_j the gradient of Linear j's output (g_last =
// d theta from the transform's backward kernel) and a_{j-1} its recorded input:
//   dW_j += a_{j-1}^T g_j      f32 MFMA with the SAMPLE index on the k dimension: both operands are plain
//                              row reads of the two LDS tiles; accumulators stay in registers
//   db_j += column sums of g_j lane-local, from the B operands of that product
//   g_{j-1} = (g_j W_j^T) * act'(a_{j-1})   W_j^T staged once per workgroup; written over a_{j-1}'s tile
//   j = 0: dL/dx += g_0 W_1^T  (rows of W_1 that belong to transformed dims are structural zeros)
// One wave per SIMD (up to 256 accumulator registers + operands), one workgroup per CU; a wave's tiles
// are fixed by the grid, partial sums are joined in a fixed order (waves through LDS, workgroups by
// small_bwd_reduce_kernel): run-to-run identical gradients, as in the generic form.
#include "nf_internal.h"

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpSmallBwdArgs {
  const float* x;       // the layer's input [rows][ldx]; the first K[0] columns feed Linear 0
  int ldx;
  const float* W[4];    // [K][N] row-major
  const float* WT[4];   // the same Linear transposed, [N][K] row-major (the plan keeps these for the data-gradient GEMMs)
  int K[4], N[4];
  const float* a[4];    // a[j]: recorded output of Linear j (after its activation), [rows][N[j]], j < nlin - 1
  const float* dth;     // [rows][N[nlin-1]]
  int act;              // 1 relu, 2 leaky_relu(0.01)
  float* G;             // [rows][D]: += g_0 W_1^T
  int D;
  float* partial;       // [gridDim.x][span]: this workgroup's sums, in the plan's weight layout of the layer
  float* grad_direct;   // one workgroup only: the layer's span of the gradient itself, no slab and no reduction launch
  int accumulate;       // grad_direct: add to what is there (a later row chunk) instead of storing (the first chunk:
                        // the gradient was zeroed before the walk, and a read-modify-write per accumulator tile behind
                        // a barrier each is a dozen exposed memory round trips -- most of a batch-128 layer's time)
  int span;
  int w_rel[4], b_rel[4];  // where dW_j / db_j sit inside a slab
  long long rows;
};

constexpr int kBwdTileStride = 65;
constexpr int kBwdTileFloats = 32 * kBwdTileStride;

// A 32-row tile travels in two steps so that its memory latency can sit behind MFMAs (one wave per SIMD:
// nothing else hides it).  tile_fetch: rows [row0, row0+32) x column `lane` of src [rows][ld] into registers,
// one row per instruction, all 32 in flight, no branch (addresses clamped; a predicated load would cost one
// memory latency per row).  tile_store: into the LDS tile, zero beyond n (up to 64 columns) and beyond `rows`.
__device__ __forceinline__ void tile_fetch(float (&v)[32], const float* __restrict__ src, int ld, int n,
                                           long long row0, long long rows, int lane) {
  const int col = lane < n ? lane : n - 1;
  const long long last = rows - 1;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    long long row = row0 + r;
    row = row < last ? row : last;
    v[r] = src[row * ld + col];
  }
}
__device__ __forceinline__ void tile_store(float* T, const float (&v)[32], int n, long long row0, long long rows,
                                           int lane) {
#pragma unroll
  for (int r = 0; r < 32; ++r) T[r * kBwdTileStride + lane] = (lane < n && row0 + r < rows) ? v[r] : 0.f;
}

// dW (+)= TA^T TG over the tile's 32 samples (16 MFMA k steps of 2), K2 / N2: a second 32-wide tile of rows /
// columns exists.  Branch-free, four steps per iteration: the LDS reads of a group run ahead of its MFMAs.
template <bool K2, bool N2>
__device__ __forceinline__ void dw_tile(f32x16 (&acc)[2][2], float (&db)[2], const float* TA, const float* TG, int i,
                                        int h) {
#pragma unroll 1
  for (int s4 = 0; s4 < 16; s4 += 4)
#pragma unroll
  for (int s = s4; s < s4 + 4; ++s) {
    const float* ta = TA + (2 * s + h) * kBwdTileStride + i;
    const float* tg = TG + (2 * s + h) * kBwdTileStride + i;
    const float a0 = ta[0], g0 = tg[0];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, g0, acc[0][0], 0, 0, 0);
    db[0] += g0;
    if (N2) {
      const float g1 = tg[32];
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, g1, acc[0][1], 0, 0, 0);
      db[1] += g1;
      if (K2) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[32], g1, acc[1][1], 0, 0, 0);
    }
    if (K2) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[32], g0, acc[1][0], 0, 0, 0);
  }
}
// o [32][K] = TG [32][N] x W^T [N][K] in chunks of 8 k steps (the tile's and the image's padding is zero)
template <bool K2>
__device__ __forceinline__ void dgrad_tile(f32x16 (&o)[2], const float* TG, const float* WT, int N, int i, int h) {
  const int chunks = (N + 15) >> 4;
  for (int c = 0; c < chunks; ++c) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int s = 8 * c + s8;
      const float g = TG[i * kBwdTileStride + 2 * s + h];
      const float* w = WT + (2 * s + h) * 64 + i;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, w[0], o[0], 0, 0, 0);
      if (K2) o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, w[32], o[1], 0, 0, 0);
    }
  }
}

template <int NLIN>
__global__ __launch_bounds__(256, 1) void mlp_small_bwd_kernel(MlpSmallBwdArgs a) {
  // Two LDS objects: the W^T images (read-only while tiles are walked) and the waves' tiles.  In one array the
  // compiler must assume that a store to a tile may change an image and cannot fetch the images' MFMA operands ahead
  // of the tile stores (nf_affine2d_bwd.hip, round 3).
  extern __shared__ __attribute__((aligned(16))) float smem[];             // [NLIN][64 n][64 k] W^T
  __shared__ __attribute__((aligned(16))) float tiles_lds[4 * 2 * kBwdTileFloats];  // [4 waves][2 tiles]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // W^T [64 n][64 k], zero beyond N and K, from the plan's transposed copies: consecutive threads read consecutive k of
  // a row and write consecutive LDS words.  (Staged from W [K][N] itself, as the first build did, consecutive threads
  // wrote LDS words 64 apart -- every store of a wave on one bank -- after a separate zero pass: at the reference's
  // batch of 128 a layer's launch is mostly such fixed costs.)
#pragma unroll
  for (int j = 0; j < NLIN; ++j) {
    const int K = a.K[j], N = a.N[j];
    float* WT = smem + j * 4096;
    const float* __restrict__ src = a.WT[j];
    for (int q = tid; q < 4096; q += 256) {
      const int n = q >> 6, k = q & 63;
      WT[q] = (n < N && k < K) ? src[n * K + k] : 0.f;
    }
  }
  __syncthreads();
  float* tiles_base = tiles_lds;
  float* T0 = tiles_base + wave * 2 * kBwdTileFloats;

  f32x16 acc[NLIN][2][2];
  float db[NLIN][2];
#pragma unroll
  for (int j = 0; j < NLIN; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      db[j][p] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][p][q][r] = 0.f;
    }

  const long long tiles = (a.rows + 31) / 32;
  const long long tstep = (long long)gridDim.x * 4;
  long long tile = (long long)blockIdx.x * 4 + wave;
  // the tile's d(theta) and the last hidden activation, fetched one tile ahead (during Linear 0's products)
  const float* top_src = NLIN >= 2 ? a.a[NLIN - 2] : a.x;
  const int top_ld = NLIN >= 2 ? a.K[NLIN - 1] : a.ldx, top_n = a.K[NLIN - 1];
  // (four Linears: 256 accumulator registers leave no room to hold a tile ahead -- fetched at the top instead)
  constexpr bool kAhead = NLIN <= 3;
  float vG[32], vA[32];
  if (kAhead && tile < tiles) {
    tile_fetch(vG, a.dth, a.N[NLIN - 1], a.N[NLIN - 1], tile * 32, a.rows, lane);
    tile_fetch(vA, top_src, top_ld, top_n, tile * 32, a.rows, lane);
  }
  for (; tile < tiles; tile += tstep) {
    const long long row0 = tile * 32;
    float* TG = T0;
    float* TA = T0 + kBwdTileFloats;
    if (!kAhead) {
      tile_fetch(vG, a.dth, a.N[NLIN - 1], a.N[NLIN - 1], row0, a.rows, lane);
      tile_fetch(vA, top_src, top_ld, top_n, row0, a.rows, lane);
    }
    tile_store(TG, vG, a.N[NLIN - 1], row0, a.rows, lane);
    tile_store(TA, vA, top_n, row0, a.rows, lane);
#pragma unroll
    for (int jj = 0; jj < NLIN; ++jj) {
      const int j = NLIN - 1 - jj;
      const int K = a.K[j], N = a.N[j];
      const bool k2 = K > 32, n2 = N > 32;
      // memory traffic of the NEXT stage, issued before this stage's products
      float vN[32];   // j > 0: Linear j-1's input tile
      float gG[2][16];  // j == 0: the rows of G this lane adds to
      if (j > 1) tile_fetch(vN, a.a[j - 2], a.K[j - 1], a.K[j - 1], row0, a.rows, lane);
      else if (j == 1) tile_fetch(vN, a.x, a.ldx, a.K[0], row0, a.rows, lane);
      else {
        if (kAhead) {
          const long long nrow0 = (tile + tstep < tiles ? tile + tstep : tile) * 32;  // past the end: re-fetch, unused
          tile_fetch(vG, a.dth, a.N[NLIN - 1], a.N[NLIN - 1], nrow0, a.rows, lane);
          tile_fetch(vA, top_src, top_ld, top_n, nrow0, a.rows, lane);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = 32 * u + i < a.D ? 32 * u + i : a.D - 1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            row = row < a.rows ? row : a.rows - 1;
            gG[u][r] = (u == 0 || k2) ? a.G[row * a.D + col] : 0.f;
          }
        }
      }
      // weight gradient: samples on the MFMA k dimension
      if (k2) {
        if (n2) dw_tile<true, true>(acc[j], db[j], TA, TG, i, h);
        else dw_tile<true, false>(acc[j], db[j], TA, TG, i, h);
      } else {
        if (n2) dw_tile<false, true>(acc[j], db[j], TA, TG, i, h);
        else dw_tile<false, false>(acc[j], db[j], TA, TG, i, h);
      }
      // data gradient: out [32][K] = TG [32][N] x W^T [N][K]
      const float* WT = smem + j * 4096;
      f32x16 o[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
      if (k2) dgrad_tile<true>(o, TG, WT, N, i, h);
      else dgrad_tile<false>(o, TG, WT, N, i, h);
      if (j > 0) {
        // times act'(a_{j-1}), over a_{j-1}'s own tile (columns >= K hold zeros and stay zero)
        const float neg = a.act == 2 ? 0.01f : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !k2) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* p = TA + ((r & 3) + 8 * (r >> 2) + 4 * h) * kBwdTileStride + 32 * u + i;
            *p = *p > 0.f ? o[u][r] : neg * o[u][r];
          }
        }
        float* t = TG; TG = TA; TA = t;
        tile_store(TA, vN, a.K[j - 1], row0, a.rows, lane);  // over the gradient tile this stage has consumed
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = 32 * u + i;
          if (col >= a.D || col >= K) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.rows) a.G[row * a.D + col] = gG[u][r] + o[u][r];
          }
        }
      }
    }
  }

  // join the four waves' sums in a fixed order; the tile area is free once every wave is done
  __syncthreads();
  float* red = tiles_base;  // [4 waves][1024]
  const bool direct = a.grad_direct != nullptr;  // small batches: no slab, no reduction launch
  float* slab = direct ? a.grad_direct : a.partial + (size_t)blockIdx.x * a.span;
#pragma unroll
  for (int j = 0; j < NLIN; ++j) {
    const int K = a.K[j], N = a.N[j];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (32 * p >= K || 32 * q >= N) continue;  // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[j][p][q][r];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
          const int r = e >> 6, l = e & 63;
          const int kk = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), nn = 32 * q + (l & 31);
          if (kk < K && nn < N) {
            const float t = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
            float* dst = slab + a.w_rel[j] + kk * N + nn;
            *dst = (direct && a.accumulate) ? *dst + t : t;
          }
        }
        __syncthreads();
      }
    // biases: lane (i, h) holds the sum over the samples 2s + h of column 32 q + i
    red[wave * 128 + lane] = db[j][0];
    red[wave * 128 + 64 + lane] = db[j][1];
    __syncthreads();
    if (tid < 64 && tid < N) {
      const int q = tid >> 5, c = tid & 31;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[w * 128 + q * 64 + c] + red[w * 128 + q * 64 + 32 + c];
      slab[a.b_rel[j] + tid] = (direct && a.accumulate) ? slab[a.b_rel[j] + tid] + s : s;
    }
    __syncthreads();
  }
}

// grad[q] += sum over workgroups of partial[b][q], q < span: 64 outputs per workgroup, its four waves each add
// every fourth slab in order, LDS joins the four sums in a fixed order
__global__ __launch_bounds__(256) void small_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int span,
                                                               float* __restrict__ grad) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (q < span)
    for (int b = grp; b < nblocks; b += 4) s += partial[(size_t)b * span + q];
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && q < span) grad[q] += ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

bool mlp_small_bwd_applicable(const Linear* Ls, int nlin, int D) {
  if (nlin < 2 || nlin > 4 || D > 64 || switches().no_mlp_small_bwd) return false;
  for (int j = 0; j < nlin; ++j)
    if (Ls[j].K > 64 || Ls[j].N > 64) return false;
  return true;
}

int mlp_small_bwd_max_blocks() { return 256; }

// The layer's Linears occupy [Ls[0].w_off, Ls[nlin-1].b_off + N) of the plan's weight vector: `grad` points at
// the gradient vector's start, `partial` holds mlp_small_bwd_max_blocks() slabs of that span.
int launch_mlp_small_bwd(const float* x, int ldx, const Linear* Ls, int nlin, const float* weights,
                         const float* const* wT, const float* const* acts, const float* dth, int act, float* G, int D, float* partial,
                         float* grad, long long rows, hipStream_t stream, bool first_chunk) {
  MlpSmallBwdArgs a;
  a.accumulate = first_chunk ? 0 : 1;
  const size_t lo = Ls[0].w_off, hi = Ls[nlin - 1].b_off + Ls[nlin - 1].N;
  a.x = x;
  a.ldx = ldx;
  a.dth = dth;
  a.act = act;
  a.G = G;
  a.D = D;
  a.partial = partial;
  a.span = (int)(hi - lo);
  a.rows = rows;
  for (int j = 0; j < 4; ++j) {
    const bool live = j < nlin;
    a.W[j] = live ? weights + Ls[j].w_off : nullptr;
    a.WT[j] = live ? wT[j] : nullptr;
    a.K[j] = live ? Ls[j].K : 0;
    a.N[j] = live ? Ls[j].N : 0;
    a.a[j] = live && j < nlin - 1 ? acts[j] : nullptr;
    a.w_rel[j] = live ? (int)(Ls[j].w_off - lo) : 0;
    a.b_rel[j] = live ? (int)(Ls[j].b_off - lo) : 0;
  }
  const size_t lds_bytes = (size_t)nlin * 4096 * sizeof(float);   // + the kernel's static tiles (66.6 KB)
  const long long tiles = (rows + 31) / 32;
  const int blocks = (int)std::max<long long>(1, std::min<long long>(mlp_small_bwd_max_blocks(), (tiles + 3) / 4));
  a.grad_direct = blocks == 1 ? grad + lo : nullptr;
  if (blocks > 1)
    NF_HIP(hipMemsetAsync(partial, 0, (size_t)blocks * a.span * sizeof(float), stream));  // alignment gaps of the layout
  if (nlin == 2) hipLaunchKernelGGL(mlp_small_bwd_kernel<2>, dim3(blocks), dim3(256), lds_bytes, stream, a);
  else if (nlin == 3) hipLaunchKernelGGL(mlp_small_bwd_kernel<3>, dim3(blocks), dim3(256), lds_bytes, stream, a);
  else hipLaunchKernelGGL(mlp_small_bwd_kernel<4>, dim3(blocks), dim3(256), lds_bytes, stream, a);
  if (blocks > 1)
    hipLaunchKernelGGL(small_bwd_reduce_kernel, dim3((a.span + 63) / 64), dim3(256), 0, stream, partial, blocks, a.span,
                       grad + lo);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
