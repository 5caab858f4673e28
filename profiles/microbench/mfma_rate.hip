// Issue rate of the float32 MFMA shapes on gfx950: N independent accumulators per wave, W waves per SIMD, a long loop.
// Prints nanoseconds and cycles (at 2.4 GHz) per MFMA per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, float a, float b) {
  float r = 0.f;
  if (SHAPE == 0) {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3];
  } else if (SHAPE == 1) {
    f16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][15];
  } else {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int SHAPE>
static void run(const char* name, int threads, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<NACC, SHAPE>), dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<NACC, SHAPE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * NACC * (threads / 256);   // MFMAs issued on one SIMD
  const double ns = ms * 1e6 / per_simd;
  printf("%-18s %d accumulators, %d waves per SIMD: %6.2f ns = %5.1f cycles at 2.4 GHz per MFMA\n", name, NACC, threads / 256, ns,
         ns * 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<8, 0>("16x16x4 f32", 256, out);
  run<8, 0>("16x16x4 f32", 512, out);
  run<22, 0>("16x16x4 f32", 512, out);
  run<2, 0>("16x16x4 f32", 256, out);
  run<4, 1>("32x32x2 f32", 256, out);
  run<4, 1>("32x32x2 f32", 512, out);
  run<6, 1>("32x32x2 f32", 512, out);
  run<1, 1>("32x32x2 f32", 256, out);
  run<8, 2>("4x4x1 f32", 256, out);
  run<8, 2>("4x4x1 f32", 512, out);
  return 0;
}
