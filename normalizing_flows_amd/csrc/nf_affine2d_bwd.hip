// Backward pass of the masked affine coupling chain (variant B, D = 2, conditioner
// 2 -> 64 -> 64 -> 2): d(NLL)/d(weights), i.e. the jax.grad(loss_fn) inside the reference's
// `update` (src/real_nvp_toy_densities.py:139-173), SURVEY.md section 8f rank 2.
//
// No activation is stored: the flow is invertible, so the backward walk (layer 0 first, the
// reverse of the order log_prob applies them) rebuilds each layer's input from its output while
// it recomputes the conditioner.  One launch for the chain (a wave walks the same tiles in every layer, the state
// goes through HBM between layers: 16 B per sample and layer, and each layer's image through LDS); per 32-sample
// tile and layer a wave runs three 64x64 products on the f32 MFMA, all operands staying in registers except the
// transposes through LDS:
//   1. recompute  a2 = W2^T h1 + b2          (as the forward kernel; h1 generated from x_c)
//   2. lane-local: (shift, log_scale), the inverse affine map, the output gradients, and
//      g2 = (W3 g_out) * [a2 > 0]  on the accumulator
//   3. g1 = W2 g2                            (accumulator-fed MFMA pass, k in accumulator order)
//   4. dW2 += h1 (x) g2 summed over the tile's samples (MFMA whose K dimension is the sample
//      index: h1 regenerated with samples on K, g2 read back transposed from LDS)
//   5. the small tensors -- dW3, db2, db1, dW1: per unit a sum over the SAMPLES of (a per-sample factor) x (a value
//      a lane holds for its sample) -- on v_mfma_f32_4x4x1 (sample_sums below): relu(a2), g2 and the first Linear's
//      masked gradient pass through the wave's transposed LDS tile one after the other, lane u of the result is
//      unit u.  The first build accumulated them per (unit, lane) in 160 registers and summed the lanes at the end:
//      404 registers, ONE wave per SIMD, every LDS and HBM latency exposed (MFMA pipe 52 % busy, a third of the
//      cycles idle: profiles/r03/train_b_2p20_before_*).  Now 8 registers, 229 in all, two waves per SIMD (eight
//      per workgroup): 306 -> 259 us per layer at 2^20 rows, MFMA pipe 68 % busy.
// Every wave keeps its partial gradient in registers across all its tiles and writes it once; a second kernel adds
// the workgroups' partials in a fixed order, scales by 1/rows_global and scatters into the canonical
// (nf_flow_set_weights) layout.
#include "nf_bf16x3.h"
#include "nf_internal.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace nf {
namespace affine2d_bwd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
using bf16x3::bf16x8;
constexpr int kReluShift = 64;

// Image of the bf16x3 form of this kernel (hidden_matmul = bf16x3 on a training handle): the two
// 64x64 operands as three bf16 parts each in 16-deep k steps, then the small arrays of the float32
// image unchanged.  Built on the device from the float32 image (build_x3_image_kernel).
constexpr int xA2 = 0;                       // [3 part][4 k16][2 mtile][64 lane][8] bf16  W2 * 2^64
constexpr int xA2T = 6144;                   // same shape: W2 rows, k in accumulator order
constexpr int xSmall = 12288;                // the float32 image from oW1B1 on
constexpr int kImageFloatsX3 = xSmall + (kImageFloats - oW1B1);
template <bool X3>
__host__ __device__ constexpr int small_off(int o) { return X3 ? xSmall + (o - oW1B1) : o; }

static inline int unit_of(int m, int r, int h) { return 32 * m + 8 * (r >> 2) + 4 * h + (r & 3); }

void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, float* img) {
  memset(img, 0, sizeof(float) * kImageFloats);
  for (int s = 0; s < 32; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int m = 0; m < 2; ++m)
        img[oA2 + (s * 64 + lane) * 2 + m] = ldexpf(W2[(2 * s + (lane >> 5)) * 64 + 32 * m + (lane & 31)], kReluShift);
  for (int m = 0; m < 2; ++m)
    for (int r = 0; r < 16; ++r)
      for (int lane = 0; lane < 64; ++lane)
        for (int mo = 0; mo < 2; ++mo)
          img[oA2T + ((m * 16 + r) * 64 + lane) * 2 + mo] = W2[(32 * mo + (lane & 31)) * 64 + unit_of(m, r, lane >> 5)];
  for (int k = 0; k < 64; ++k) {
    img[oW1B1 + 2 * k + 0] = ldexpf(W1[cond_dim * 64 + k], -kReluShift);
    img[oW1B1 + 2 * k + 1] = ldexpf(b1[k], -kReluShift);
  }
  for (int h = 0; h < 2; ++h)
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < 16; ++r) {
        const int u = unit_of(m, r, h), q = (h * 2 + m) * 16 + r;
        img[oW1acc + q] = W1[cond_dim * 64 + u];
        img[oB1acc + q] = b1[u];
        img[oB2 + q] = b2[u];
        img[oW3 + 2 * q + 0] = W3[u * 2 + 0];
        img[oW3 + 2 * q + 1] = W3[u * 2 + 1];
      }
  img[oB3 + 0] = b3[0];
  img[oB3 + 1] = b3[1];
}

// canonical (per layer: W1[2][64], b1[64], W2[64][64], b2[64], W3[64][2], b3[2]) -> index into a
// wave's partial block, and the factor that undoes the 2^-64 carried by h1 in dW2
void build_gather_map(int cond_dim, int* map, float* scale) {
  for (int e = 0; e < kCanonFloats; ++e) { map[e] = -1; scale[e] = 1.f; }
  const int cW1 = 0, cb1 = 128, cW2 = 192, cb2 = 192 + 4096, cW3 = cb2 + 64, cb3 = cW3 + 128;
  for (int mk = 0; mk < 2; ++mk)
    for (int mj = 0; mj < 2; ++mj)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int k = unit_of(mk, r, lane >> 5), j = 32 * mj + (lane & 31);
          map[cW2 + k * 64 + j] = pW2 + ((mk * 2 + mj) * 16 + r) * 64 + lane;
          scale[cW2 + k * 64 + j] = ldexpf(1.f, kReluShift);
        }
  for (int u = 0; u < 64; ++u) {   // the small tensors leave the kernel indexed by unit (lane u of sample_sums)
    map[cb2 + u] = pB2 + u;
    map[cW3 + 2 * u + 0] = pW3 + 2 * u + 0;
    map[cW3 + 2 * u + 1] = pW3 + 2 * u + 1;
    map[cb1 + u] = pB1 + u;
    map[cW1 + cond_dim * 64 + u] = pW1 + u;  // the masked input row has zero gradient
  }
  map[cb3 + 0] = pB3 + 0;
  map[cb3 + 1] = pB3 + 1;
}

struct Params {
  const float* img;       // images of the layers, layer 0 first (img_stride floats apart)
  long long img_stride;
  float2* v;              // [rows] state: in = z (the chain's output), out = the chain's input
  float2* G;              // [rows] d(loss_i)/d(state); layer 0 starts from G := v (the base density's gradient)
  float* partials;        // [num_layers][gridDim.x][kPartialFloats]
  long long rows;
  int num_layers;
  int parity0;            // first_mask_parity: layer l conditions on dim (parity0 + l) & 1
};

// A layer reads the state the previous layer of the SAME launch stored (same wave, other half-wave's lanes too): read
// past the CU's vector cache, which may still hold the line from the previous layer's own read.
__device__ __forceinline__ float2 load_state(const float2* q) {
  const unsigned long long u =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}

__device__ __forceinline__ float add_halves(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
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
// sum over the 32 lanes of this lane's half-wave
__device__ __forceinline__ float half_sum(float x) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

// Development: -DNF_BWD_SKIP=<mask> builds a TIMING-ONLY kernel without some of its phases (wrong results, same
// remaining instruction stream): 1 pass 1 (recompute a2), 2 the lane-local block that forms g2, 4 pass 3 (g1 = W2 g2),
// 8 the first Linear's gradients, 16 pass 4 (dW2).  profiles/scratch/bwd_phase_timing.sh.
#ifndef NF_BWD_SKIP
#define NF_BWD_SKIP 0
#endif


constexpr int kWaves = 8;               // one workgroup per CU, two waves per SIMD
constexpr int kGT = 32 * 65;            // one transposed tile [sample][unit], padded rows
constexpr int kAopRow = 36;             // (rows 16 bytes apart in bank terms: the four rows a 16-byte factor read touches
                                        // then sit on different banks -- with 32 it was a 4-way conflict, 96 cycles per tile)
constexpr int kAop = 6 * kAopRow;       // per-sample factors as rows [g_s | g_ls | 0 | 0 | 1 | c] (A operands below)
constexpr int kWaveScratch = kGT + kAop;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Sum over the 32 SAMPLES of a tile of (a per-sample factor) x (a per-(unit, sample) value), for all 64 units and
// four factors at once, on v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, K = 1; layout and cost measured in
// profiles/microbench/mfma_f32_4x4x1_layout.hip: D[reg r][lane 4b+j] += A[lane 4b+r] * B[lane 4b+j], 8 cycles):
// one instruction per sample with B = lane u's value X[u][sample] read from the transposed tile and A = factor
// row (lane & 3) of that sample, so D[r][lane u] accumulates sum_i factor_r[i] X[u][i] -- over every tile the wave
// walks, in sample order (a fixed order).  This replaces 160 lane-held accumulators (dW3, db2, db1, dW1 per unit and
// lane, summed over lanes at the end) by 8 registers: 404 -> under 256 registers, two waves per SIMD instead of one.
__device__ __forceinline__ void sample_sums(f32x4& D, const float* __restrict__ frow, const float* __restrict__ xt,
                                            int lane) {
  const float4* f4 = reinterpret_cast<const float4*>(frow);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 a = f4[q];
    D = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, xt[(4 * q + 0) * 65 + lane], D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, xt[(4 * q + 1) * 65 + lane], D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, xt[(4 * q + 2) * 65 + lane], D, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, xt[(4 * q + 3) * 65 + lane], D, 0, 0, 0);
  }
}

// X3: passes 1 and 3 (recompute a2; g1 = W2 g2) as six bf16 MFMAs on three-way operand splits
// (nf_bf16x3.h); pass 4 (the weight-gradient product) and everything lane-local stay float32.
template <bool X3>
__global__ __launch_bounds__(kWaves * 64) void chain_kernel(Params p) {
  // Two LDS objects: the weight image (read-only while tiles are walked) and the waves' scratch.  In ONE array the
  // compiler must assume that a store to the scratch may change the image: it then sinks the image reads of a
  // select's operands into branches behind the stores (a read and a full wait per accumulator register, seen in
  // the first build of this form) instead of fetching them ahead.
  extern __shared__ __attribute__((aligned(16))) float lds[];                     // the image
  __shared__ __attribute__((aligned(16))) float scratch[kWaves * kWaveScratch];  // per wave: tile + factor rows
  const int tid = threadIdx.x;
  constexpr int kImg = X3 ? kImageFloatsX3 : kImageFloats;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  float* gT = scratch + wave * kWaveScratch;      // the wave's transposed tile: relu(a2), then g2, then the
                                                  // first Linear's masked gradient, one after the other
  float* aop = gT + kGT;
  float* cbuf = aop + 5 * kAopRow;                // row 5: the tile's conditioning values
  const float* L = lds;
  const float2* a2 = reinterpret_cast<const float2*>(L + oA2) + lane;     // float32 form only
  const float2* a2t = reinterpret_cast<const float2*>(L + oA2T) + lane;
  const bf16x8* xa2 = reinterpret_cast<const bf16x8*>(L + xA2) + lane;     // bf16x3 form: + ((part*4+s)*2+m)*64
  const bf16x8* xa2t = reinterpret_cast<const bf16x8*>(L + xA2T) + lane;
  const float2* w1b1 = reinterpret_cast<const float2*>(L + small_off<X3>(oW1B1));
  constexpr int OB2 = small_off<X3>(oB2), OW3 = small_off<X3>(oW3), OB3 = small_off<X3>(oB3);
  constexpr int OW1A = small_off<X3>(oW1acc), OB1A = small_off<X3>(oB1acc);
  const int hq = h * 32;  // offset of this half's [m][r] block in the accumulator-order images
  const float* f_hv = aop + (lane & 3) * kAopRow;        // factor rows (g_s, g_ls, 0, 0)
  const float* f_gz = aop + ((lane & 3) + 2) * kAopRow;  // factor rows (0, 0, 1, c)
  float* gTw = gT + i * 65 + 4 * h;                 // this lane's 32 units of sample i: + 32 m + 8 (r >> 2) + (r & 3)

  const long long rows = p.rows;
  const long long tiles = (rows + 31) / 32;
  const long long tstride = (long long)gridDim.x * kWaves;

  // ONE launch walks the chain, layer 0 (the one log_prob applied last) first: a sample's backward depends on no other
  // sample, a wave takes the same tiles in every layer, so a layer reads only state its own wave stored -- no grid-wide
  // step between layers, just the workgroup's barrier around re-staging the image.  (One launch PER layer before:
  // at the reference's batch of 128 the step was eight launch gaps long.)
  for (int layer = 0; layer < p.num_layers; ++layer) {
  const int cond_dim = (p.parity0 + layer) & 1;
  const bool first = layer == 0;
  __syncthreads();  // every wave is done with the previous layer's image and with the scratch
  {
    const float4* src = reinterpret_cast<const float4*>(p.img + (size_t)layer * p.img_stride);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int q = tid; q < kImg / 4; q += kWaves * 64) dst[q] = src[q];
  }
  if (lane < 32) { aop[2 * kAopRow + lane] = 0.f; aop[3 * kAopRow + lane] = 0.f; aop[4 * kAopRow + lane] = 1.f; }
  __syncthreads();

  f32x16 dW2[2][2];
  f32x4 Dx = {0.f, 0.f, 0.f, 0.f};   // lane u: (dW3[u][shift], dW3[u][log_scale], db1[u], dW1[u])
  f32x4 Dy = {0.f, 0.f, 0.f, 0.f};   // lane u: (., ., db2[u], .)
  float db3s = 0.f, db3l = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW2[a][0][r] = 0.f; dW2[a][1][r] = 0.f; }

  // The next tile's state is requested before the current tile is walked (the first build loaded it at the top of
  // its own iteration -- v, then behind a branch on `first` G: two exposed round trips per tile); `first`: G := v,
  // read from the same line instead of branching.
  const float2* Gsrc = first ? p.v : p.G;
  long long tile = (long long)blockIdx.x * kWaves + wave;
  float2 v = make_float2(0.f, 0.f), G = make_float2(0.f, 0.f);
  if (tile < tiles && tile * 32 + i < rows) { v = load_state(p.v + tile * 32 + i); G = load_state(Gsrc + tile * 32 + i); }
  // the loop is entered with no load in flight (the waits the compiler places follow the register that is read;
  // a load still pending at the loop head would make it wait for the NEXT tile's request at the first use of v)
  asm volatile("; tile state loaded" : "+v"(v.x), "+v"(v.y), "+v"(G.x), "+v"(G.y));
  for (; tile < tiles; tile += tstride) {
    const long long idx = tile * 32 + i;
    const bool valid = idx < rows;
    float2 v_next = make_float2(0.f, 0.f), G_next = make_float2(0.f, 0.f);
    {
      const long long nidx = (tile + tstride) * 32 + i;
      if (tile + tstride < tiles && nidx < rows) { v_next = load_state(p.v + nidx); G_next = load_state(Gsrc + nidx); }
    }
    const float c = cond_dim ? v.y : v.x;
    const float tp = cond_dim ? v.x : v.y;
    float Gc = cond_dim ? G.y : G.x;
    const float Gtp = cond_dim ? G.x : G.y;

    // 1. a2 = W2^T relu(w1 c + b1) + b2
    f32x16 acc[2];
    if constexpr (X3) {
      acc[0] = lds_f32x16(L + OB2 + hq);
      acc[1] = lds_f32x16(L + OB2 + hq + 16);
      const float4* wb = reinterpret_cast<const float4*>(w1b1) + 4 * h;  // (w1, b1) of k = 16 s + 8 h + 0..7
      // the operand of step s+1 (52 VALU) is issued between the 12 MFMAs of step s
      bf16x8 B[2][3];
      auto make_B = [&](int s) {
        float hv8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 f = wb[8 * s + k];
          hv8[2 * k] = __builtin_amdgcn_fmed3f(fmaf(c, f.x, f.y), 0.f, 1.f);
          hv8[2 * k + 1] = __builtin_amdgcn_fmed3f(fmaf(c, f.z, f.w), 0.f, 1.f);
        }
        bf16x3::split3(hv8, B[s & 1][0], B[s & 1][1], B[s & 1][2]);
      };
      make_B(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) make_B(s + 1);
#pragma unroll
        for (int m = 0; m < 2; ++m)
          bf16x3::mfma6(acc[m], xa2[((0 * 4 + s) * 2 + m) * 64], xa2[((1 * 4 + s) * 2 + m) * 64],
                        xa2[((2 * 4 + s) * 2 + m) * 64], B[s & 1][0], B[s & 1][1], B[s & 1][2]);
        if (s < 3) {
#pragma unroll
          for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
      }
    } else if ((NF_BWD_SKIP & 1) != 0) {
      acc[0] = lds_f32x16(L + OB2 + hq);
      acc[1] = lds_f32x16(L + OB2 + hq + 16);
    } else {
      {
        const f32x16 bias0 = lds_f32x16(L + OB2 + hq);
        const f32x16 bias1 = lds_f32x16(L + OB2 + hq + 16);
        const float2 a = a2[0];
        const float2 wb = w1b1[h];
        const float b = __builtin_amdgcn_fmed3f(fmaf(c, wb.x, wb.y), 0.f, 1.f);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, bias0, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, bias1, 0, 0, 0);
      }
#pragma unroll 8
      for (int s = 1; s < 32; ++s) {
        const float2 a = a2[s * 64];
        const float2 wb = w1b1[2 * s + h];
        const float b = __builtin_amdgcn_fmed3f(fmaf(c, wb.x, wb.y), 0.f, 1.f);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
      }
    }

    // 2. (shift, log_scale), undo the affine map, gradients of the layer's two outputs; relu(a2) goes to the
    //    transposed tile for the last Linear's weight gradient
    const float2* w3 = reinterpret_cast<const float2*>(L + OW3) + hq;
    float ps = 0.f, pl = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 w = w3[m * 16 + r];
        const float hv = fmaxf(acc[m][r], 0.f);
        ps = fmaf(hv, w.x, ps);
        pl = fmaf(hv, w.y, pl);
        gTw[32 * m + 8 * (r >> 2) + (r & 3)] = hv;
      }
    const float shift = add_halves(ps) + L[OB3 + 0];
    const float ls = add_halves(pl) + L[OB3 + 1];
    const float e = exp_f32(ls);
    const float tpm = tp - shift;
    const float t = tpm * exp_f32(-ls);            // the layer's input on the transformed dim
    const float g_s = Gtp;                          // d loss / d shift
    const float g_ls = valid ? fmaf(Gtp, tpm, -1.f) : 0.f;  // t e^ls = t' - shift; -1 from the -log_scale term
    const float Gt = Gtp * e;
    db3s += h == 0 ? g_s : 0.f;
    db3l += h == 0 ? g_ls : 0.f;
    aop[h * kAopRow + i] = h ? g_ls : g_s;         // both halves hold both (identical bits)
    if (h == 0) cbuf[i] = c;
    // g2 = (W3 g_out) * [a2 > 0], in `acc`: the next pass's operand
    if ((NF_BWD_SKIP & 2) == 0)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 w = w3[m * 16 + r];
        const float a = acc[m][r];
        float gfull = fmaf(w.x, g_s, w.y * g_ls);
        // (opaque use: otherwise the select becomes a branch with W3's LDS read and a full wait inside it,
        // once per accumulator register)
        asm volatile("" : "+v"(gfull));
        acc[m][r] = a > 0.f ? gfull : 0.f;
      }

    // 3. g1 = W2 g2 (k-steps in accumulator order)
    f32x16 g1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { g1[0][r] = 0.f; g1[1][r] = 0.f; }
    if constexpr (X3) {
      bf16x8 B[2][3];
      auto make_B = [&](int s) {  // k step (m, g) = accumulator registers acc[m][8g .. 8g+7]
        float gv8[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) gv8[jj] = acc[s >> 1][8 * (s & 1) + jj];
        bf16x3::split3(gv8, B[s & 1][0], B[s & 1][1], B[s & 1][2]);
      };
      make_B(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) make_B(s + 1);
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
          bf16x3::mfma6(g1[mo], xa2t[((0 * 4 + s) * 2 + mo) * 64], xa2t[((1 * 4 + s) * 2 + mo) * 64],
                        xa2t[((2 * 4 + s) * 2 + mo) * 64], B[s & 1][0], B[s & 1][1], B[s & 1][2]);
        if (s < 3) {
#pragma unroll
          for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          }
        }
      }
      // dW3[u][.] += sum_i (g_s, g_ls)[i] relu(a2)[u][i]
      sample_sums(Dx, f_hv, gT, lane);
    } else if ((NF_BWD_SKIP & 4) == 0) {
      // ... with the last Linear's weight gradient written between its MFMAs, one sample per k step (the compiler
      // clusters the small MFMAs after the large ones all the same; pinned in place with sched_barrier the step
      // measured the same, 2.60 against 2.61 ms: the small MFMAs' dependent chain is not what the kernel waits for)
      //   dW3[u][.] += sum_i (g_s, g_ls)[i] relu(a2)[u][i]
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 a = a2t[(m * 16 + r) * 64];
          g1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[m][r], g1[0], 0, 0, 0);
          Dx = __builtin_amdgcn_mfma_f32_4x4x1f32(f_hv[m * 16 + r], gT[(m * 16 + r) * 65 + lane], Dx, 0, 0, 0);
          g1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[m][r], g1[1], 0, 0, 0);
        }
    } else {
      sample_sums(Dx, f_hv, gT, lane);
    }
    // g2 transposed into LDS (over relu(a2): a wave's LDS operations execute in order)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) gTw[32 * m + 8 * (r >> 2) + (r & 3)] = acc[m][r];
    // the first Linear: gz = g1 * [w1 c + b1 > 0] -> d loss / d c; gz stays in g1's registers until pass 4 has read g2
    float gcm = 0.f;
    if ((NF_BWD_SKIP & 8) == 0)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w1a = L[OW1A + hq + m * 16 + r], b1a = L[OB1A + hq + m * 16 + r];
        const float gz = fmaf(w1a, c, b1a) > 0.f ? g1[m][r] : 0.f;
        gcm = fmaf(gz, w1a, gcm);
        g1[m][r] = gz;
      }
    Gc += add_halves(gcm);

    // 4. dW2[k][j] += sum_i h1[k][i] g2[j][i]: samples on the MFMA's K dimension; between them
    //    db2[u] += sum_i g2[u][i], two samples per step
    if ((NF_BWD_SKIP & 16) == 0) {
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float ci = cbuf[2 * s + h];
        const float2 wb0 = w1b1[i], wb1 = w1b1[32 + i];
        const float a0 = __builtin_amdgcn_fmed3f(fmaf(ci, wb0.x, wb0.y), 0.f, 1.f);
        const float a1 = __builtin_amdgcn_fmed3f(fmaf(ci, wb1.x, wb1.y), 0.f, 1.f);
        const float b0 = gT[(2 * s + h) * 65 + i], b1v = gT[(2 * s + h) * 65 + 32 + i];
        dW2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, dW2[0][0], 0, 0, 0);
        Dy = __builtin_amdgcn_mfma_f32_4x4x1f32(f_gz[2 * s], gT[(2 * s) * 65 + lane], Dy, 0, 0, 0);
        dW2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1v, dW2[0][1], 0, 0, 0);
        dW2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, dW2[1][0], 0, 0, 0);
        Dy = __builtin_amdgcn_mfma_f32_4x4x1f32(f_gz[2 * s + 1], gT[(2 * s + 1) * 65 + lane], Dy, 0, 0, 0);
        dW2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1v, dW2[1][1], 0, 0, 0);
      }
    } else {
      sample_sums(Dy, f_gz, gT, lane);
    }

    // (db1, dW1)[u] += sum_i (1, c[i]) gz[u][i]  (gz over g2 in the tile)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) gTw[32 * m + 8 * (r >> 2) + (r & 3)] = g1[m][r];
    sample_sums(Dx, f_gz, gT, lane);

    const float2 v_out = cond_dim ? make_float2(t, c) : make_float2(c, t);
    const float2 G_out = cond_dim ? make_float2(Gt, Gc) : make_float2(Gc, Gt);
    // (before this tile's stores: the wait for the loads must not wait for the stores as well)
    v = v_next;
    G = G_next;
    asm volatile("; next tile's state arrived" : "+v"(v.x), "+v"(v.y), "+v"(G.x), "+v"(G.y));
    if (h == 0 && valid) {
      p.v[idx] = v_out;
      p.G[idx] = G_out;
    }
  }

  // one partial gradient per workgroup: the waves lay theirs out in the scratch (dead by now), four at a time, and
  // they are added in wave order
  static_assert(4 * kPartialFloats <= kWaves * kWaveScratch, "staging of four partials");
  constexpr int kPer = (kPartialFloats + kWaves * 64 - 1) / (kWaves * 64);
  float total[kPer];
#pragma unroll
  for (int e = 0; e < kPer; ++e) total[e] = 0.f;
  const float t3s = half_sum(db3s), t3l = half_sum(db3l);
#pragma unroll
  for (int round = 0; round < kWaves / 4; ++round) {
    __syncthreads();
    if ((wave >> 2) == round) {
      float* P = scratch + (wave & 3) * kPartialFloats;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) P[pW2 + ((a * 2 + b) * 16 + r) * 64 + lane] = dW2[a][b][r];
      P[pW3 + 2 * lane + 0] = Dx[0];
      P[pW3 + 2 * lane + 1] = Dx[1];
      P[pB1 + lane] = Dx[2];
      P[pW1 + lane] = Dx[3];
      P[pB2 + lane] = Dy[2];
      if (lane == 0) { P[pB3 + 0] = t3s; P[pB3 + 1] = t3l; P[pB3 + 2] = 0.f; P[pB3 + 3] = 0.f; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const int q = tid + e * kWaves * 64;
      if (q < kPartialFloats)
        total[e] = (((total[e] + scratch[q]) + scratch[kPartialFloats + q]) + scratch[2 * kPartialFloats + q]) +
                   scratch[3 * kPartialFloats + q];
    }
  }
  float* out = p.partials + ((size_t)layer * gridDim.x + blockIdx.x) * kPartialFloats;
#pragma unroll
  for (int e = 0; e < kPer; ++e) {
    const int q = tid + e * kWaves * 64;
    if (q < kPartialFloats) out[q] = total[e];
  }
  }  // layer
}

// grad[e] = scale[e] / rows_global * sum over workgroups of partial[b][map[e]].  Fixed order: chunk c
// adds workgroups c, c+4, c+8, ... (Kahan), then the four chunks are added in order.
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ partials_all, int nparts,
                                                     const int* __restrict__ map0, const int* __restrict__ map1,
                                                     const float* __restrict__ scale0, const float* __restrict__ scale1,
                                                     int parity0, float* __restrict__ grad_all, float inv_rows) {
  // blockIdx.y = layer: its workgroups' partials, the map of its conditioning dim, its slice of the gradient
  const int layer = blockIdx.y;
  const float* __restrict__ partials = partials_all + (size_t)layer * nparts * kPartialFloats;
  const int cond = (parity0 + layer) & 1;
  const int* __restrict__ map = cond ? map1 : map0;
  const float* __restrict__ scale = cond ? scale1 : scale0;
  float* __restrict__ grad = grad_all + (size_t)layer * kCanonFloats;
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int src = e < kCanonFloats ? map[e] : -1;
  float s = 0.f, c = 0.f;
  if (src >= 0) {
    for (int w0 = chunk; w0 < nparts; w0 += 32) {
      float y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int w = w0 + 4 * u;
        y[u] = w < nparts ? partials[(size_t)w * kPartialFloats + src] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float yy = y[u] - c;
        const float t = s + yy;
        c = (t - s) - yy;
        s = t;
      }
    }
  }
  red[chunk][el] = s;
  __syncthreads();
  if (chunk == 0 && e < kCanonFloats)
    grad[e] = (((red[0][el] + red[1][el]) + red[2][el]) + red[3][el]) * scale[e] * inv_rows;
}

int grid_blocks() { return 256; }

// float32 image -> bf16x3 image, every layer in one launch (after each optimiser step the float32
// images are refreshed by repack_kernel; this follows them)
__global__ __launch_bounds__(256) void build_x3_image_kernel(const float* __restrict__ img32, float* __restrict__ imgx3) {
  const float* src = img32 + (size_t)blockIdx.y * kImageFloats;
  float* dst = imgx3 + (size_t)blockIdx.y * kImageFloatsX3;
  unsigned short* half = reinterpret_cast<unsigned short*>(dst);
  constexpr int kPart = 4 * 2 * 64 * 8;  // bf16 per part of one operand block
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * kPart; e += gridDim.x * 256) {
    const int block = e / kPart, q = e % kPart;           // 0: A2, 1: A2T
    const int j = q & 7, lane = (q >> 3) & 63, m = (q >> 9) & 1, s = q >> 10;
    const int hk = lane >> 5, u = lane & 31;
    float w;
    if (block == 0) {
      const int k = 16 * s + 8 * hk + j;                  // float32 image: k = 2 s32 + h32
      w = src[oA2 + ((k >> 1) * 64 + (k & 1) * 32 + u) * 2 + m];
    } else {
      const int mk = s >> 1, r = 8 * (s & 1) + j;          // float32 image: k step (mk, r), same lane
      w = src[oA2T + ((mk * 16 + r) * 64 + lane) * 2 + m];
    }
    const __bf16 hi = (__bf16)w;
    const float r1 = w - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    unsigned short* base = half + (block == 0 ? xA2 : xA2T) * 2;
    base[0 * kPart + q] = __builtin_bit_cast(unsigned short, hi);
    base[1 * kPart + q] = __builtin_bit_cast(unsigned short, mid);
    base[2 * kPart + q] = __builtin_bit_cast(unsigned short, lo);
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < kImageFloats - oW1B1; e += gridDim.x * 256)
    dst[xSmall + e] = src[oW1B1 + e];
}

int x3_image_floats() { return kImageFloatsX3; }

int launch_build_x3_images(const float* img32, float* imgx3, int num_layers, hipStream_t stream) {
  hipLaunchKernelGGL(build_x3_image_kernel, dim3(16, num_layers), dim3(256), 0, stream, img32, imgx3);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int launch_chain(const float* images, int num_layers, int first_mask_parity, float* v, float* G, float* partials,
                 long long rows, hipStream_t stream, int* out_parts, bool x3) {
  const int image = x3 ? kImageFloatsX3 : kImageFloats;
  const size_t lds_bytes = (size_t)image * sizeof(float);   // + the kernel's static scratch (72.7 KB)
  Params p;
  p.img = images;
  p.img_stride = image;
  p.v = reinterpret_cast<float2*>(v);
  p.G = reinterpret_cast<float2*>(G);
  p.partials = partials;
  p.rows = rows;
  p.num_layers = num_layers;
  p.parity0 = first_mask_parity & 1;
  const long long tiles = (rows + 31) / 32;
  const int grid = (int)std::min<long long>(grid_blocks(), std::max<long long>(1, (tiles + kWaves - 1) / kWaves));
  if (out_parts) *out_parts = grid;
  if (x3) hipLaunchKernelGGL(chain_kernel<true>, dim3(grid), dim3(kWaves * 64), lds_bytes, stream, p);
  else hipLaunchKernelGGL(chain_kernel<false>, dim3(grid), dim3(kWaves * 64), lds_bytes, stream, p);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int launch_gather(const float* partials, int nparts, int num_layers, int first_mask_parity, const int* map0,
                  const int* map1, const float* scale0, const float* scale1, float* grad, long long rows_global,
                  hipStream_t stream) {
  hipLaunchKernelGGL(gather_kernel, dim3((kCanonFloats + 63) / 64, num_layers), dim3(256), 0, stream, partials, nparts,
                     map0, map1, scale0, scale1, first_mask_parity & 1, grad, 1.0f / (float)rows_global);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace affine2d_bwd

// ---------------------------------------------------------------------------
// optimiser and image refresh on the canonical weight vector
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                   float b1, float b2, float eps, float c1, float c2,
                                                   const int* __restrict__ skip, const unsigned* __restrict__ live) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n || (skip != nullptr && *skip != 0)) return;
  // plan layout (generic::Plan): entries without a canonical weight behind them (masked rows of the first Linear,
  // alignment gaps) hold zeros that the products rely on; their "gradient" is not one
  if (live != nullptr && ((live[e >> 5] >> (e & 31)) & 1u) == 0) return;
  // optax.adam (src/real_nvp_toy_densities.py:162): m, v moments, bias-corrected by c1 = 1-b1^t, c2 = 1-b2^t
  const float ge = g[e];
  const float me = b1 * m[e] + (1.f - b1) * ge;
  const float ve = b2 * v[e] + (1.f - b2) * ge * ge;
  m[e] = me;
  v[e] = ve;
  w[e] -= lr * (me / c1) / (sqrtf(ve / c2) + eps);
}

// four entries per thread (16-byte accesses; n % 4 == 0, bases 16-byte aligned: the plan layout)
__global__ __launch_bounds__(256) void adam4_kernel(float4* __restrict__ w, const float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, size_t n4, float lr,
                                                    float b1, float b2, float eps, float c1, float c2,
                                                    const int* __restrict__ skip, const unsigned* __restrict__ live) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n4 || (skip != nullptr && *skip != 0)) return;
  const unsigned bits = live != nullptr ? (live[q >> 3] >> ((q & 7) * 4)) & 0xFu : 0xFu;
  if (bits == 0) return;
  const float4 g4 = g[q];
  float4 m4 = m[q], v4 = v[q], w4 = w[q];
  const float ge[4] = {g4.x, g4.y, g4.z, g4.w};
  float me[4] = {m4.x, m4.y, m4.z, m4.w}, ve[4] = {v4.x, v4.y, v4.z, v4.w}, we[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (bits & (1u << u)) {   // the same arithmetic as adam_kernel, entry by entry
      me[u] = b1 * me[u] + (1.f - b1) * ge[u];
      ve[u] = b2 * ve[u] + (1.f - b2) * ge[u] * ge[u];
      we[u] -= lr * (me[u] / c1) / (sqrtf(ve[u] / c2) + eps);
    }
  m[q] = make_float4(me[0], me[1], me[2], me[3]);
  v[q] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  w[q] = make_float4(we[0], we[1], we[2], we[3]);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, size_t n, float lr,
                                                  const int* __restrict__ skip, const unsigned* __restrict__ live) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n && !(skip != nullptr && *skip != 0) && !(live != nullptr && ((live[e >> 5] >> (e & 31)) & 1u) == 0)) w[e] -= lr * g[e];  // src/real_nvp.py:298
}

// *flag = 1 when any gradient entry is NaN or infinite (flag zeroed by the caller): the "skip_nonfinite_updates"
// option, optax.apply_if_finite's test
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, size_t n, int* __restrict__ flag) {
  bool bad = false;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    bad |= (__float_as_uint(g[e]) & 0x7f800000u) == 0x7f800000u;
  if (bad) *flag = 1;
}

// image[i] = canon[src[i]] * 2^shift[i]  (src < 0: 0)
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ canon, const int* __restrict__ src,
                                                     const signed char* __restrict__ shift, float* __restrict__ img,
                                                     size_t n) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int s = src[e];
  img[e] = s < 0 ? 0.f : ldexpf(canon[s], (int)shift[e]);
}

int launch_nonfinite_flag(const float* g, size_t n, int* flag, hipStream_t stream) {
  NF_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream));
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)std::min<size_t>(256, (n + 255) / 256)), dim3(256), 0, stream, g, n,
                     flag);
  NF_HIP(hipGetLastError());
  return NF_OK;
}
// With a skip flag (option "skip_nonfinite_updates" = optax.apply_if_finite) the step count that enters the bias
// correction lives on the device and advances only when the update is applied, as the wrapped optimiser's own
// count does: bias[0..1] = (1 - b1^t, 1 - b2^t), bias[2] = t.
__global__ void adam_bias_kernel(float* __restrict__ bias, const int* __restrict__ skip, float b1, float b2) {
  if (*skip != 0) return;
  const float t = bias[2] + 1.f;
  bias[2] = t;
  bias[0] = 1.f - powf(b1, t);
  bias[1] = 1.f - powf(b2, t);
}
__global__ __launch_bounds__(256) void adam_dev_bias_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, size_t n,
                                                            float lr, float b1, float b2, float eps,
                                                            const float* __restrict__ bias,
                                                            const int* __restrict__ skip, const unsigned* __restrict__ live) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n || *skip != 0) return;
  if (live != nullptr && ((live[e >> 5] >> (e & 31)) & 1u) == 0) return;
  const float c1 = bias[0], c2 = bias[1];
  const float ge = g[e];
  const float me = b1 * m[e] + (1.f - b1) * ge;
  const float ve = b2 * v[e] + (1.f - b2) * ge * ge;
  m[e] = me;
  v[e] = ve;
  w[e] -= lr * (me / c1) / (sqrtf(ve / c2) + eps);
}

int launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                long long step, hipStream_t stream, const int* skip, float* dev_bias, const unsigned* live) {
  if (skip != nullptr && dev_bias != nullptr) {
    hipLaunchKernelGGL(adam_bias_kernel, dim3(1), dim3(1), 0, stream, dev_bias, skip, b1, b2);
    hipLaunchKernelGGL(adam_dev_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, g, m, v, n, lr,
                       b1, b2, eps, dev_bias, skip, live);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  const float c1 = 1.f - powf(b1, (float)step), c2 = 1.f - powf(b2, (float)step);
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
    hipLaunchKernelGGL(adam4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<float4*>(w),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n / 4,
                       lr, b1, b2, eps, c1, c2, skip, live);
    NF_HIP(hipGetLastError());
    return NF_OK;
  }
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, g, m, v, n, lr, b1, b2,
                     eps, c1, c2, skip, live);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int launch_sgd(float* w, const float* g, size_t n, float lr, hipStream_t stream, const int* skip, const unsigned* live) {
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, g, n, lr, skip, live);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// two images from one canonical vector in one launch (the forward and the backward image after an optimiser step)
__global__ __launch_bounds__(256) void repack2_kernel(const float* __restrict__ canon, const int* __restrict__ src0,
                                                      const signed char* __restrict__ shift0, float* __restrict__ img0,
                                                      size_t n0, const int* __restrict__ src1,
                                                      const signed char* __restrict__ shift1, float* __restrict__ img1,
                                                      size_t n1) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n0) {
    const int s = src0[e];
    img0[e] = s < 0 ? 0.f : ldexpf(canon[s], (int)shift0[e]);
  } else if ((e -= n0) < n1) {
    const int s = src1[e];
    img1[e] = s < 0 ? 0.f : ldexpf(canon[s], (int)shift1[e]);
  }
}

int launch_repack2(const float* canon, const int* src0, const signed char* shift0, float* img0, size_t n0,
                   const int* src1, const signed char* shift1, float* img1, size_t n1, hipStream_t stream) {
  hipLaunchKernelGGL(repack2_kernel, dim3((unsigned)((n0 + n1 + 255) / 256)), dim3(256), 0, stream, canon, src0, shift0,
                     img0, n0, src1, shift1, img1, n1);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

int launch_repack(const float* canon, const int* src, const signed char* shift, float* img, size_t n,
                  hipStream_t stream) {
  hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, canon, src, shift, img, n);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

}  // namespace nf
