// Where does VALU work hide behind v_mfma_f32_32x32x16_bf16 on gfx950?
//   mode 0: every wave runs MFMA + VALU interleaved in program order (1 MFMA : NV VALU)
//   mode 1: wave roles -- waves 0..3 of a workgroup only MFMA, waves 4..7 only VALU (one of each per SIMD)
//   mode 2: MFMA waves alone (VALU waves exit), mode 3: VALU waves alone
// Same instruction totals per SIMD in modes 0 and 1.  build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&v)[8], float a, float b) {
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], a, b);
}

template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b + j); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const int wave = threadIdx.x >> 6;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m], 0, 0, 0);
        valu_block<NV>(v, a, b);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
    }
  } else {
    const bool mfma_role = wave < 4;
    if (mfma_role && MODE != 3) {
      for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m], 0, 0, 0);
      }
    }
    if (!mfma_role && MODE != 2) {
      for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) valu_block<NV>(v, a, b);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mode 0 with the exact-float32 MFMA (64 cycles): how many VALU fit behind one?
template <int NV>
__global__ __launch_bounds__(512) void kf32(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[m], acc[m], 0, 0, 0);
      valu_block<NV>(v, a, b);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run_f32() {
  const int threads = 512, blocks = 256, iters = 20000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kf32<NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kf32<NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("f32 MFMA, interleaved NV=%2d: wall %.3f ms -> %.2f ns per (1 MFMA + %d VALU) per SIMD\n", NV, ms,
         ms * 1e6 / (2.0 * iters * 4), NV);
  hipFree(out);
}

template <int MODE, int NV>
void run() {
  const int threads = 512, blocks = 256, iters = 20000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  // per SIMD: 2*iters*4 MFMAs and 2*iters*4*NV VALU in every mode that runs both
  const double per_mfma_ns = ms * 1e6 / (2.0 * iters * 4);
  printf("mode %d NV=%2d: wall %.3f ms  -> %.2f ns per (1 MFMA + %d VALU) per SIMD\n", MODE, NV, ms, per_mfma_ns, NV);
  hipFree(out);
}

int main() {
  run<2, 4>();
  run<3, 4>(); run<3, 6>(); run<3, 8>();
  run<0, 4>(); run<0, 6>(); run<0, 8>();
  run<1, 4>(); run<1, 6>(); run<1, 8>();
  run_f32<0>(); run_f32<2>(); run_f32<4>(); run_f32<8>(); run_f32<12>(); run_f32<14>(); run_f32<16>(); run_f32<20>();
  return 0;
}
