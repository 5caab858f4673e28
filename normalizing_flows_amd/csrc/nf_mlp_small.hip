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
  long long rows;
};

constexpr int kTileStride = 65;

constexpr int kWFloats = 64 * 64 + 64;  // one Linear's staged weights + bias

template <int kWaves>
__global__ __launch_bounds__(kWaves * 64, 1) void mlp_small_kernel(MlpSmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [nlin][kWFloats] then [4 waves][32 * kTileStride]
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
  float* T = smem + a.nlin * kWFloats + wave * (32 * kTileStride);

  const long long tiles = (a.rows + 31) / 32;
  for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += (long long)gridDim.x * kWaves) {
    const long long row0 = tile * 32;
    // the wave's 32 input rows: K[0] columns and one zero column behind them (an odd K reads it; what
    // lies further right is the previous tile's and is never read by this Linear)
    const int k0w = a.K[0] + 1;
    for (int q = lane; q < 32 * k0w; q += 64) {
      const int r = q / k0w, k = q - r * k0w;
      const long long row = row0 + r;
      T[r * kTileStride + k] = (row < a.rows && k < a.K[0]) ? a.x[row * a.ldx + k] : 0.f;
    }
    for (int j = 0; j < a.nlin; ++j) {
      const int K = a.K[j], N = a.N[j];
      const float* Ws = smem + j * kWFloats;
      const float* bs = Ws + 4096;
      const int ntiles = (N + 31) >> 5, steps = (K + 1) >> 1;
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      if (ntiles > 1) {
        for (int s = 0; s < steps; ++s) {
          const float av = T[i * kTileStride + 2 * s + h];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Ws[(2 * s + h) * 64 + i], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Ws[(2 * s + h) * 64 + 32 + i], acc[1], 0, 0, 0);
        }
      } else {
        for (int s = 0; s < steps; ++s)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(T[i * kTileStride + 2 * s + h], Ws[(2 * s + h) * 64 + i],
                                                        acc[0], 0, 0, 0);
      }
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
  if (nlin < 1 || nlin > 4 || getenv("NF_NO_MLP_SMALL") != nullptr) return false;
  for (int j = 0; j < nlin; ++j)
    if (Ls[j].K > 64 || Ls[j].N > 64) return false;
  return true;
}

// One coupling layer's conditioner: x [rows][ldx] -> out[nlin-1] (= theta), hidden outputs optional.
int launch_mlp_small(const float* x, int ldx, const Linear* Ls, int nlin, const float* weights, int act,
                     float* const* outs, long long rows, hipStream_t stream) {
  MlpSmallArgs a;
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
  static bool attr_set = false;
  if (!attr_set) {
    NF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_small_kernel<8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    NF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_small_kernel<12>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  // one workgroup per CU; as many waves as the LDS left beside the staged weights holds tiles for
  // (three Linears: 50 KB of weights + 12 x 8.3 KB; four: 66.5 KB + 8 x 8.3 KB)
  const int waves = nlin <= 3 ? 12 : 8;
  const size_t lds_bytes = ((size_t)nlin * kWFloats + (size_t)waves * 32 * kTileStride) * sizeof(float);
  const long long tiles = (rows + 31) / 32;
  const long long blocks = std::max<long long>(1, std::min<long long>(256, (tiles + waves - 1) / waves));
  if (waves == 12) hipLaunchKernelGGL(mlp_small_kernel<12>, dim3((unsigned)blocks), dim3(12 * 64), lds_bytes, stream, a);
  else hipLaunchKernelGGL(mlp_small_kernel<8>, dim3((unsigned)blocks), dim3(8 * 64), lds_bytes, stream, a);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace generic
}  // namespace nf
