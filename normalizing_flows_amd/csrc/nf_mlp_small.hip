// General path, narrow conditioners: the whole MLP of one coupling layer in ONE launch when every
// Linear has at most 64 inputs and 64 outputs (2-D and low-dimensional flows with hidden widths other
// than the fused kernels' 64/64 -- e.g. the 50 the reference's command line defaults to).  The chain of
// GEMM launches of nf_generic.hip is HBM-bound there (every activation matrix is written and read
// back: 0.5 GB per Linear at 2^20 rows); here the activations of a 32-row tile stay in the wave's own
// LDS tile between Linears and only the conditioner's output (and, when the training walk records
// them, the hidden activations) goes to HBM.
//
// 256 threads = 4 waves; the workgroup stages EVERY Linear's W (K x N floats, zero padded to 64 x 64) and
// bias in LDS once, then its waves walk 32-row tiles independently (grid-stride, no barrier after the
// staging): each wave multiplies its tile on the f32 MFMA (rows = samples, columns = output units,
// both column tiles before anything is overwritten), applies bias and activation and writes the
// result back over its own tile for the next Linear.
#include "nf_internal.h"

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpSmallArgs {
  const float* x;      // [rows][ldx], first K[0] columns used
  int ldx;
  int nlin;
  const float* W[4];   // [K][N] row-major
  const float* b[4];
  int K[4], N[4];
  int act[4];          // 0 none, 1 relu, 2 leaky_relu
  float* out[4];       // [rows][N] dense, or null (hidden activations nobody needs)
  float* tape;         // training walk: the input rows are copied here ([rows][K[0]] dense, ldx == K[0]), or null
  long long rows;
};

constexpr int kTileStride = 65;

constexpr int kWFloats = 64 * 64 + 64;  // one Linear's staged weights + bias

// acc (+)= T [32][K] x Ws [K][N] in groups of four k steps (no branch inside: the LDS reads of a group run ahead
// of its MFMAs).  Reads up to 7 columns / rows past K: the caller guarantees zeros there on both sides.
template <bool N2>
__device__ __forceinline__ void mlp_tile_product(f32x16 (&acc)[2], const float* T, const float* Ws, int K, int i,
                                                 int h) {
  const int groups = (K + 7) >> 3;
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int s = 4 * g + s4;
      const float av = T[i * kTileStride + 2 * s + h];
      const float* w = Ws + (2 * s + h) * 64 + i;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w[0], acc[0], 0, 0, 0);
      if (N2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w[32], acc[1], 0, 0, 0);
    }
  }
}

template <int kWaves>
__global__ __launch_bounds__(kWaves * 64, 1) void mlp_small_kernel(MlpSmallArgs a) {
  // two LDS objects (staged weights: read-only while tiles are walked; the waves' tiles), so that the compiler may
  // fetch a product's weight operands ahead of the previous Linear's tile stores (nf_affine2d_bwd.hip, round 3)
  extern __shared__ __attribute__((aligned(16))) float smem[];                        // [nlin][kWFloats]
  __shared__ __attribute__((aligned(16))) float tiles_lds[kWaves * 32 * kTileStride];  // [waves][32 * kTileStride]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  for (int j = 0; j < a.nlin; ++j) {
    const int K = a.K[j], N = a.N[j];
    float* Wj = smem + j * kWFloats;
    for (int q = tid; q < 64 * 64; q += kWaves * 64) {
      const int k = q >> 6, n = q & 63;
      Wj[q] = (k < K && n < N) ? a.W[j][(size_t)k * N + n] : 0.f;
    }
    if (tid < 64) Wj[4096 + tid] = tid < N ? a.b[j][tid] : 0.f;
  }
  __syncthreads();
  float* T = tiles_lds + wave * (32 * kTileStride);

  const long long tiles = (a.rows + 31) / 32;
  for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += (long long)gridDim.x * kWaves) {
    const long long row0 = tile * 32;
    // the wave's 32 input rows, zero beyond K[0] (all 64 columns) and beyond `rows`.  No predicated load (each
    // would cost a memory latency of its own): addresses are clamped, zeros come from a select.  Up to 8 input
    // columns with a dense matrix: the tile is one contiguous run, four loads per lane cover it; wider inputs
    // take one row per instruction, all 32 in flight.
    {
      const int K0 = a.K[0];
      const long long last = a.rows - 1;
      if (K0 <= 8 && a.ldx == K0) {
        const long long e0 = row0 * K0, elast = a.rows * K0 - 1;
        const unsigned magic = ((1u << 20) + K0 - 1) / K0;  // e / K0 == (e * magic) >> 20 for e < 2048, K0 <= 64
        float v[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const long long e = e0 + lane + 64 * it;
          v[it] = a.x[e < elast ? e : elast];
        }
        for (int q = lane; q < 32 * 16; q += 64) T[(q >> 4) * kTileStride + (q & 15)] = 0.f;  // what the product reads
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const unsigned e = lane + 64 * it;
          const unsigned r = (e * magic) >> 20, k = e - r * K0;
          if (e < 32u * K0 && row0 + r <= last) {
            T[r * kTileStride + k] = v[it];
            if (a.tape != nullptr) a.tape[e0 + e] = v[it];
          }
        }
      } else {
        float v[32];
        const int col = lane < K0 ? lane : K0 - 1;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          long long row = row0 + r;
          row = row < last ? row : last;
          v[r] = a.x[row * a.ldx + col];
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) T[r * kTileStride + lane] = (lane < K0 && row0 + r <= last) ? v[r] : 0.f;
        if (a.tape != nullptr && lane < K0)
#pragma unroll
          for (int r = 0; r < 32; ++r)
            if (row0 + r <= last) a.tape[(row0 + r) * K0 + lane] = v[r];
      }
    }
    for (int j = 0; j < a.nlin; ++j) {
      const int K = a.K[j], N = a.N[j];
      const float* Ws = smem + j * kWFloats;
      const float* bs = Ws + 4096;
      const int ntiles = (N + 31) >> 5;
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      // every column of T is defined (zeros beyond this Linear's K) and Ws is zero padded to 64 x 64
      if (ntiles > 1) mlp_tile_product<true>(acc, T, Ws, K, i, h);
      else mlp_tile_product<false>(acc, T, Ws, K, i, h);
      // bias + activation, back over the wave's own tile (all its reads are done), and out to HBM if wanted
      float* out = a.out[j];
      const int act = a.act[j];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u >= ntiles) continue;
        const int n = 32 * u + i;
        const float bias = bs[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = acc[u][r] + bias;
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
          T[rr * kTileStride + n] = n < N ? v : 0.f;
          const long long row = row0 + rr;
          if (out != nullptr && n < N && row < a.rows) out[row * N + n] = v;
        }
      }
      // columns beyond this Linear's tiles must read as zero for the next one (its K = this N <= 64)
      if (ntiles < 2)
        for (int q = lane; q < 32 * 32; q += 64) T[(q >> 5) * kTileStride + 32 + (q & 31)] = 0.f;
    }
  }
}

bool mlp_small_applicable(const Linear* Ls, int nlin) {
  if (nlin < 1 || nlin > 4 || switches().no_mlp_small) return false;
  for (int j = 0; j < nlin; ++j)
    if (Ls[j].K > 64 || Ls[j].N > 64) return false;
  return true;
}

// One coupling layer's conditioner: x [rows][ldx] -> out[nlin-1] (= theta), hidden outputs optional.
int launch_mlp_small(const float* x, int ldx, const Linear* Ls, int nlin, const float* weights, int act,
                     float* const* outs, long long rows, hipStream_t stream, float* tape) {
  MlpSmallArgs a;
  a.tape = tape;
  a.x = x;
  a.ldx = ldx;
  a.nlin = nlin;
  a.rows = rows;
  for (int j = 0; j < 4; ++j) {
    const bool live = j < nlin;
    a.W[j] = live ? weights + Ls[j].w_off : nullptr;
    a.b[j] = live ? weights + Ls[j].b_off : nullptr;
    a.K[j] = live ? Ls[j].K : 0;
    a.N[j] = live ? Ls[j].N : 0;
    a.act[j] = live && j < nlin - 1 ? act : 0;
    a.out[j] = live ? outs[j] : nullptr;
  }
  // dynamic part: up to four staged Linears (66.5 KB) beside the static tiles (8 x 8.3 KB / 12 x 8.3 KB)
  NF_DYN_LDS(4 * kWFloats * (int)sizeof(float), &mlp_small_kernel<8>);
  NF_DYN_LDS(3 * kWFloats * (int)sizeof(float), &mlp_small_kernel<12>);
  // one workgroup per CU; as many waves as the LDS left beside the staged weights holds tiles for
  // (three Linears: 50 KB of weights + 12 x 8.3 KB; four: 66.5 KB + 8 x 8.3 KB)
  const int waves = nlin <= 3 ? 12 : 8;
  const size_t lds_bytes = (size_t)nlin * kWFloats * sizeof(float);   // + the kernel's static tiles
  const long long tiles = (rows + 31) / 32;
  const long long blocks = std::max<long long>(1, std::min<long long>(256, (tiles + waves - 1) / waves));
  if (waves == 12) hipLaunchKernelGGL(mlp_small_kernel<12>, dim3((unsigned)blocks), dim3(12 * 64), lds_bytes, stream, a);
  else hipLaunchKernelGGL(mlp_small_kernel<8>, dim3((unsigned)blocks), dim3(8 * 64), lds_bytes, stream, a);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
