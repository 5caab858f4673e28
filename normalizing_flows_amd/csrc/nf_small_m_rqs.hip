// The last Linear of a masked spline layer AND the spline itself at SMALL batches (M <= 256 rows: the reference's image
// flow trains and evaluates at batch 128, src/maf_mnist.py:159-171,228-230), in ONE launch:
//     theta[M, N] = A[M, Kdim] W[Kdim, N] + bias,   N = Dt (3K + 1);   y[:, transformed dims] <- spline(theta; y)
//
// What this replaces: gemm_splitk_kernel (128 x 128 tiles, K split three ways: 231 workgroups, 20.5 us per launch at the
// MNIST shape [128, 512] x [512, 9800], its 15 MB of slabs through HBM) + rqs_rows_kernel adding the slabs (one workgroup
// per row: 128 of 256 CUs, five dependent rounds of loads, 10.2 us): 30.7 of the 45 us a layer of that flow took at batch
// 128 once the hidden Linears were one launch each (rocprofv3 kernel trace, round 4).  This kernel: 24.4 us per launch
// in a chain over the eight layers' weights (profiles/microbench/small_m_rqs_clocks.hip); the 8-layer log_prob call at
// batch 128 went from 369 to 296 us.  The product is 1.28 GFLOP -- 8.2 us of the whole chip's f32 MFMA rate:
//
//   * one workgroup = 32 rows x DW transformed dims (DW (3K + 1) <= 176 columns = 11 MFMA column tiles; K = 8: 7 dims,
//     175 columns), so [128, 9800] is 4 x 56 = 224 workgroups -- one resident round, 87 % of the CUs, every MFMA column
//     but one per workgroup useful;
//   * its EIGHT waves split Kdim between them (Kdim = 512: four groups of 16 k each); a wave holds all 22 result tiles
//     (16 x 16, v_mfma_f32_16x16x4_f32: 88 accumulator registers) and loads its operands from global memory straight
//     into MFMA operand registers -- every element of A and W is loaded ONCE per workgroup, each W element feeds two
//     MFMAs;
//   * W is read from a copy laid out in exactly that operand order (pack_kernel; Plan::ensure_wsmr makes it on the
//     first small-batch walk and again after the weights change): a workgroup's share is one contiguous 360 KB block,
//     a wave-load 1 KB, a lane's four k of one column tile one 16-byte load.  Read in place the share is 512 pieces of
//     700 bytes, 39 KB apart: 28-31 us per launch;
//   * the waves' tiles are added through LDS in wave order (run-to-run identical), one row tile at a time (8 x 11 KB),
//     bias added last, and the 32 x DW (3K + 1) parameters stay in LDS for the spline: one thread per (row, dim),
//     rqs_eval_reg on y in place (the same arithmetic as every other spline kernel here: nf_rqs_math.h);
//   * a row's log-dets of the workgroup's dims are added in dim order into ldpart[column group][row]; the caller adds
//     the column groups of ALL layers of a walk in order with one ld_reduce launch at the end (a per-layer reduction
//     launch would cost more than it adds).
// Where the 24 us go (shader-clock stamps of every wave, same microbenchmark): loads + MFMAs 13.5 us (a group's loads
// land after 2.3 / 2.3 / 1.1 / 1.1 us, its 88 MFMAs take 1.4-2.0 us next to the SIMD's other wave), waiting for the
// workgroup's slowest wave + the cross-wave sum 4.4 us, spline 1.3 us, launch boundary 2.4 us.  The loads are bound by
// what one CU can pull (422 KB per workgroup: the four row blocks re-read W, the 56 column groups re-read A), the
// same with the weights resident in L2 (one matrix instead of eight: 24.1 us) -- not by HBM.
// Lane (i = lane & 15, kq = lane >> 4) holds, per group of 16 k, A[row i][16 g + 4 kq .. + 3] per row tile and
// W[16 g + 4 kq + j][column i] per column tile; MFMA j multiplies k = 16 g + 4 kq + j.  Result register r of lane l:
// row 4 (l >> 4) + r, column l & 15.
#include "nf_internal.h"
#include "nf_rqs_eval.h"

#include <algorithm>

namespace nf {
namespace generic {

typedef float f32x4s __attribute__((ext_vector_type(4)));

// TIMING-ONLY builds (profiles/scratch/build_smr_skip.sh): wrong results, one phase removed each.
// 1: no MFMAs (one FMA per MFMA keeps the loads alive); 2: no loads of W; 4: no spline; 8: no cross-wave sum
#ifndef NF_SMR_SKIP
#define NF_SMR_SKIP 0
#endif
#ifndef NF_SMR_LOOP
#define NF_SMR_LOOP 0
#endif
#ifndef NF_SMR_NT
#define NF_SMR_NT 0   // 1: non-temporal loads of W (26.2 us per launch against 24.4: the four row blocks share W through L2)
#endif
// -DNF_SMR_CLOCKS (profiles/microbench/small_m_rqs_clocks.hip): theta_out is a debug buffer, every wave leaves the shader
// clock at six program points and the 100 MHz wall clock at entry and exit
#ifdef NF_SMR_CLOCKS
#define SMR_STAMP(i)                                                                                                     \
  do {                                                                                                                   \
    if (lane == 0) {                                                                                                     \
      unsigned long long* dbg_ = reinterpret_cast<unsigned long long*>(theta_out) + ((size_t)blockIdx.x * kSrWaves + wave) * 16; \
      dbg_[i] = ((i) == 6 || (i) == 7) ? wall_clock64() : clock64();                                                                   \
    }                                                                                                                    \
  } while (0)
#else
#define SMR_STAMP(i) do {} while (0)
#endif

constexpr int kSrWaves = 8;
constexpr int kSrMaxTiles = 11;   // 16-column tiles per workgroup: 2 x 11 tiles x 4 registers of accumulators per lane

template <int K>
struct SrShape {
  static constexpr int P = 3 * K + 1;
  static constexpr int DW = (16 * kSrMaxTiles) / P;   // transformed dims per workgroup
  static constexpr int COLS = DW * P;
  static constexpr int CT = (COLS + 15) / 16;
  static constexpr int STRIDE = COLS | 1;             // odd row stride of the parameter tile in LDS
  static constexpr int RED_FLOATS = kSrWaves * CT * 256;
  static constexpr int lds_floats(int RT) { return RED_FLOATS + 16 * RT * STRIDE + 16 * RT * DW; }
};

#if NF_SMR_SKIP & 1
__device__ __forceinline__ f32x4s SMR_MFMA(float a, float b, f32x4s c, int, int, int) { c[0] = fmaf(a, b, c[0]); return c; }
#else
#define SMR_MFMA __builtin_amdgcn_mfma_f32_16x16x4f32
#endif

// RT: 16-row tiles per workgroup; 2 is what runs (see launch_k for the measurement of 4)
template <int K, int INV, int RT>
__global__ __launch_bounds__(kSrWaves * 64) void linear_rqs_small_m_kernel(const float* __restrict__ A, int lda,
                                                                          const float4* __restrict__ Wp,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ y, float* __restrict__ ldpart,
                                                                          float* __restrict__ theta_out, int M, int N,
                                                                          int Kdim, int D, int pass_parity, int CG,
                                                                          RqsConst c) {
  using S = SrShape<K>;
  constexpr int CT = S::CT, P = S::P, DW = S::DW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                           // [wave][column tile][register][lane]
  constexpr int ROWS = 16 * RT;
  float* prm = sm + S::RED_FLOATS;           // [ROWS][STRIDE]
  float* ldv = prm + ROWS * S::STRIDE;       // [ROWS][DW]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  // workgroup -> (column group, row block), XCD-aware: workgroups are dealt to the eight XCDs round-robin, and the row
  // blocks that share a column group's W should meet in ONE L2 (23.7 MB of HBM traffic per launch at the MNIST shape
  // against 20.2 MB of re-laid W: rocprofv3 FETCH_SIZE) -- whatever CG is
  const int RG = (M + 16 * RT - 1) / (16 * RT);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = 8 * (slot / RG) + xcd, rg = slot % RG;
  if (cg >= CG) return;   // (the grid is rounded up to whole rounds of eight column groups; no barrier reached yet)
  const int m0 = rg * (16 * RT);
  const int Dt = N / P;
  const int dt0 = cg * DW;
  const int ndim = min(DW, Dt - dt0);
  const int c0 = dt0 * P;
  const int ncol = ndim * P;

  SMR_STAMP(6);
  SMR_STAMP(0);
  const int groups = Kdim >> 4;   // Kdim % 16 == 0 (the launcher's test)
  const int per = (groups + kSrWaves - 1) / kSrWaves;
  const int g_begin = wave * per;
  const int g_end = min(groups, g_begin + per);
  // addresses as a wave-uniform base (the group's k row) + loop-invariant 32-bit byte offsets per lane: rows / columns
  // past the edge are clamped, their results never used
  unsigned a_off[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) a_off[rt] = (unsigned)(((size_t)min(m0 + 16 * rt + i, M - 1) * lda + 4 * kq) * sizeof(float));
  // W comes re-laid in operand order (pack_kernel below): the workgroup's share is one contiguous block, a wave's group
  // of 16 k is CT consecutive kilobytes, a lane's four k of one column tile one 16-byte load
  const float4* wl = Wp + (size_t)cg * groups * (CT * 64) + lane;

  f32x4s acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4s{0.f, 0.f, 0.f, 0.f};

  SMR_STAMP(1);
  float4 a[2][RT];
  float4 b[2][CT];
  // (readfirstlane on an index that is wave-uniform anyway: it keeps the compiler's loop strength reduction from
  // turning every load's address into its own 64-bit induction pointer in vector registers)
  auto fetch = [&](int buf, int g) {
    const char* ag = reinterpret_cast<const char*>(A + __builtin_amdgcn_readfirstlane(16 * g));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[buf][rt] = *reinterpret_cast<const float4*>(ag + a_off[rt]);
    const float4* wg = wl + __builtin_amdgcn_readfirstlane(g * (CT * 64));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#if NF_SMR_NT
      {
        typedef float f32x4nt __attribute__((ext_vector_type(4)));
        const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(wg + ct * 64));
        b[buf][ct] = make_float4(t[0], t[1], t[2], t[3]);
      }
#else
      b[buf][ct] = wg[ct * 64];
#endif
      if (NF_SMR_SKIP & 2) b[buf][ct] = make_float4((float)g, 1.f, 2.f, (float)ct);
    }
  };
  auto multiply = [&](int buf) {
#define SMR_K_STEP(c_)                                                                            \
  _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                              \
      acc[rt][ct] = SMR_MFMA(a[buf][rt].c_, b[buf][ct].c_, acc[rt][ct], 0, 0, 0);
    SMR_K_STEP(x) SMR_K_STEP(y) SMR_K_STEP(z) SMR_K_STEP(w)
#undef SMR_K_STEP
  };
  // The k loop: a group's loads, then its 88 MFMAs -- no overlap between them.  What bounds this phase is how many bytes a
  // CU can pull (a workgroup reads 358 KB of W and 64 KB of A; 13 us at ~40 GB/s per CU, between the L2-hit and the
  // Infinity-Cache rates of MI355X_MICROARCH.md's gather table), not latency: with the next group's loads in flight under
  // the MFMAs (NF_SMR_LOOP=2: two landing buffers, scheduling barriers so that the loads stay ahead of the MFMAs) a
  // wave's first group lands after 6.8 us instead of 2.3 and the launch takes 25.5 us against 24.3 (shader-clock stamps,
  // profiles/microbench/small_m_rqs_clocks.hip).  The buffer index is a compile-time constant in both forms (a run-time
  // index sends the operand arrays to scratch memory).
#if NF_SMR_LOOP == 2
  if (g_begin < g_end) {
    int g = g_begin;
    fetch(0, g);
    __builtin_amdgcn_sched_barrier(0);
    for (; g + 2 < g_end; g += 2) {
      fetch(1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
      __builtin_amdgcn_sched_barrier(0);
      SMR_STAMP(9);
      fetch(0, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1);
      __builtin_amdgcn_sched_barrier(0);
      SMR_STAMP(11);
    }
    if (g + 1 < g_end) {
      fetch(1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
      __builtin_amdgcn_sched_barrier(0);
      SMR_STAMP(13);
      multiply(1);
    } else {
      multiply(0);
    }
  }
#else
  {
#ifdef NF_SMR_CLOCKS
    int slot = 8;
#endif
    for (int g = g_begin; g < g_end; ++g) {
      fetch(0, g);
#ifdef NF_SMR_CLOCKS
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      if (slot < 16) SMR_STAMP(slot);
      ++slot;
      __builtin_amdgcn_sched_barrier(0);
#endif
      multiply(0);
#ifdef NF_SMR_CLOCKS
      __builtin_amdgcn_sched_barrier(0);
      if (slot < 16) SMR_STAMP(slot);
      ++slot;
#endif
    }
  }
#endif
  SMR_STAMP(2);

  // the eight waves' tiles -> one parameter tile, one row tile (16 rows) at a time
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (rt) __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * CT + ct) * 4 + r) * 64 + lane] = acc[rt][ct][r];
    __syncthreads();
    for (int idx = tid; idx < CT * 256; idx += kSrWaves * 64) {
      const int ct = idx >> 8, r = (idx >> 6) & 3, l = idx & 63;
      const int row = 16 * rt + 4 * (l >> 4) + r, cl = ct * 16 + (l & 15);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < ((NF_SMR_SKIP & 8) ? 1 : kSrWaves); ++w) v += red[((w * CT + ct) * 4 + r) * 64 + l];
      if (cl < ncol) {
        v += bias[c0 + cl];
        prm[row * S::STRIDE + cl] = v;
#ifndef NF_SMR_CLOCKS
        if (theta_out != nullptr && m0 + row < M) theta_out[(size_t)(m0 + row) * N + c0 + cl] = v;
#endif
      }
    }
  }
  __syncthreads();
  SMR_STAMP(3);

  // the spline: one thread per (row, transformed dim)
  const int first_t = pass_parity == 0 ? 1 : 0;
  for (int it = tid; it < ROWS * DW; it += kSrWaves * 64) {
    const int rl = it / DW, dl = it % DW;
    float l = 0.f;
    if (dl < ndim && m0 + rl < M) {
      float* yp = y + (size_t)(m0 + rl) * D + first_t + 2 * (dt0 + dl);
      float res;
      if (NF_SMR_SKIP & 4) { res = prm[rl * S::STRIDE + dl * P] + *yp; l = res; }
      else rqs_eval_reg<K, INV>(prm + rl * S::STRIDE + dl * P, c, *yp, &res, &l);
      *yp = res;
    }
    ldv[it] = l;
  }
  __syncthreads();
  SMR_STAMP(4);
  if (tid < ROWS && m0 + tid < M) {
    float s = 0.f;
    for (int dl = 0; dl < ndim; ++dl) s += ldv[tid * DW + dl];
    ldpart[(size_t)cg * M + m0 + tid] = s;
  }
  SMR_STAMP(5);
  SMR_STAMP(7);
}

template <int K>
static int column_groups(int Dt) { return (Dt + SrShape<K>::DW - 1) / SrShape<K>::DW; }

// W[Kdim, N] -> the operand order of the kernel above: float4 (column group, group of 16 k, column tile, lane) =
// W[16 g + 4 (lane >> 4) + 0..3][first column of the group + 16 tile + (lane & 15)], zero past the group's columns.
// Why a second copy of the weights: read in place, a workgroup's share of W is 512 pieces of 700 bytes, 39 KB apart, in
// 64-byte requests -- 224 workgroups of them at once ran HBM at 2 TB/s and a wave waited 2-3 us (up to 10) for each
// of its four groups of loads (shader-clock stamps, profiles/microbench/small_m_rqs_clocks.hip); re-laid, the share is
// one contiguous 360 KB block read in 1 KB wave loads.
template <int K>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ W, float4* __restrict__ Wp, int N, int Kdim,
                                                   int CG) {
  using S = SrShape<K>;
  const int groups = Kdim >> 4, Dt = N / S::P;
  const long long total = (long long)CG * groups * S::CT * 64;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int lane = (int)(q & 63);
    long long t = q >> 6;
    const int ct = (int)(t % S::CT);
    t /= S::CT;
    const int g = (int)(t % groups), cg = (int)(t / groups);
    const int ncol = min(S::DW, Dt - cg * S::DW) * S::P;
    const int cl = 16 * ct + (lane & 15);
    const size_t k = (size_t)16 * g + 4 * (lane >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cl < ncol) {
      const float* src = W + k * N + (size_t)cg * S::COLS + cl;
      v = make_float4(src[0], src[(size_t)N], src[(size_t)2 * N], src[(size_t)3 * N]);
    }
    Wp[q] = v;
  }
}

template <int K>
static size_t packed_floats_k(int N, int Kdim) {
  return (size_t)column_groups<K>(N / SrShape<K>::P) * (Kdim >> 4) * SrShape<K>::CT * 256;
}

// floats of the re-laid copy of one layer's last Linear (0: not a bin count this kernel has)
size_t small_m_rqs_packed_floats(int N, int Kdim, int bins) {
  switch (bins) {
    case 4: return packed_floats_k<4>(N, Kdim);
    case 6: return packed_floats_k<6>(N, Kdim);
    case 8: return packed_floats_k<8>(N, Kdim);
    case 10: return packed_floats_k<10>(N, Kdim);
    case 12: return packed_floats_k<12>(N, Kdim);
    case 16: return packed_floats_k<16>(N, Kdim);
  }
  return 0;
}

template <int K>
static void pack_k(const float* W, float* Wp, int N, int Kdim, hipStream_t stream) {
  const int CG = column_groups<K>(N / SrShape<K>::P);
  const long long total = (long long)CG * (Kdim >> 4) * SrShape<K>::CT * 64;
  hipLaunchKernelGGL(pack_kernel<K>, dim3((unsigned)std::min<long long>(4096, (total + 255) / 256)), dim3(256), 0, stream, W,
                     reinterpret_cast<float4*>(Wp), N, Kdim, CG);
}

int launch_small_m_rqs_pack(const float* W, float* Wp, int N, int Kdim, int bins, hipStream_t stream) {
  switch (bins) {
    case 4: pack_k<4>(W, Wp, N, Kdim, stream); break;
    case 6: pack_k<6>(W, Wp, N, Kdim, stream); break;
    case 8: pack_k<8>(W, Wp, N, Kdim, stream); break;
    case 10: pack_k<10>(W, Wp, N, Kdim, stream); break;
    case 12: pack_k<12>(W, Wp, N, Kdim, stream); break;
    case 16: pack_k<16>(W, Wp, N, Kdim, stream); break;
    default: set_error("spline bin count %d has no small-batch kernel", bins); return NF_ERR_UNSUPPORTED;
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// Column groups a layer of this shape is cut into; 0: not a shape for this kernel (the caller keeps the split-K route).
int small_m_rqs_groups(const float* A, int lda, int M, int N, int Kdim, int Dt, int bins) {
  if (switches().no_small_m_rqs || M < 1 || M > 256 || Kdim < 64 || Kdim > 2048 || (Kdim & 15) || (lda & 3) ||
      (size_t)Kdim * N * sizeof(float) >= (1ull << 31) || (size_t)M * lda * sizeof(float) >= (1ull << 31) ||
      (reinterpret_cast<uintptr_t>(A) & 15) != 0 || N != Dt * (3 * bins + 1))
    return 0;
  int cg = 0;
  switch (bins) {
    case 4: cg = column_groups<4>(Dt); break;
    case 6: cg = column_groups<6>(Dt); break;
    case 8: cg = column_groups<8>(Dt); break;
    case 10: cg = column_groups<10>(Dt); break;
    case 12: cg = column_groups<12>(Dt); break;
    case 16: cg = column_groups<16>(Dt); break;
    default: return 0;
  }
  // a handful of workgroups is no way to run a chip: narrow flows keep the split-K route
  return cg * ((M + 31) / 32) >= 32 ? cg : 0;   // (counted in 32-row blocks, whichever block form runs)
}

template <int K, int RT>
static int launch_krt(const float* A, int lda, const float* Wp, const float* bias, float* y, float* ldpart, float* theta_out,
                      int M, int N, int Kdim, int D, int par, int inverse_op, const RqsConst& rc, hipStream_t stream) {
  using S = SrShape<K>;
  const int Dt = N / S::P;
  const int CG = column_groups<K>(Dt);
  const dim3 grid(8 * ((CG + 7) / 8) * ((M + 16 * RT - 1) / (16 * RT)));
  const size_t lds = sizeof(float) * S::lds_floats(RT);
  if (inverse_op) {
    NF_DYN_LDS((int)lds, &linear_rqs_small_m_kernel<K, 1, RT>);
    hipLaunchKernelGGL((linear_rqs_small_m_kernel<K, 1, RT>), grid, dim3(kSrWaves * 64), lds, stream, A, lda,
                       reinterpret_cast<const float4*>(Wp), bias, y, ldpart, theta_out, M, N, Kdim, D, par, CG, rc);
  } else {
    NF_DYN_LDS((int)lds, &linear_rqs_small_m_kernel<K, 0, RT>);
    hipLaunchKernelGGL((linear_rqs_small_m_kernel<K, 0, RT>), grid, dim3(kSrWaves * 64), lds, stream, A, lda,
                       reinterpret_cast<const float4*>(Wp), bias, y, ldpart, theta_out, M, N, Kdim, D, par, CG, rc);
  }
  NF_HIP(hipGetLastError());
  return NF_OK;
}

// 32-row blocks whatever M: 64-row blocks (RT = 4: W read once per 64 rows, one round of workgroups at 129..256 rows,
// 250 registers) took 577-593 us per 8-layer call at 192 / 256 rows against 478-481 us for two rounds of 32-row blocks.
template <int K>
static int launch_k(const float* A, int lda, const float* Wp, const float* bias, float* y, float* ldpart, float* theta_out,
                    int M, int N, int Kdim, int D, int par, int inverse_op, const RqsConst& rc, hipStream_t stream) {
  return launch_krt<K, 2>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
}

// Wp: launch_small_m_rqs_pack()'s copy of the layer's W; ldpart: [small_m_rqs_groups()][M] floats of this layer;
// theta_out (may be null): the parameters [M, N], for training.
int launch_small_m_rqs(const float* A, int lda, const float* Wp, const float* bias, float* y, float* ldpart,
                       float* theta_out, int M, int N, int Kdim, int D, int par, int inverse_op, const RqsConst& rc,
                       hipStream_t stream) {
  switch (rc.K) {
    case 4: return launch_k<4>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
    case 6: return launch_k<6>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
    case 8: return launch_k<8>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
    case 10: return launch_k<10>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
    case 12: return launch_k<12>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
    case 16: return launch_k<16>(A, lda, Wp, bias, y, ldpart, theta_out, M, N, Kdim, D, par, inverse_op, rc, stream);
  }
  set_error("spline bin count %d has no small-batch kernel", rc.K);
  return NF_ERR_UNSUPPORTED;
}

}  // namespace generic
}  // namespace nf
