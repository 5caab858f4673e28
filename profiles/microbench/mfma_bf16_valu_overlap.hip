// Does v_mfma_f32_32x32x16_bf16 overlap with f32 VALU work on one SIMD of gfx950?
// (companion of mfma_f32_valu_overlap.hip).  Loop body = NM independent bf16 MFMAs + NV v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV>
__global__ void k(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b + j); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], a, b);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV>
void run(int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 20000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)blocks * (threads / 64) * iters * NM * 32768.0;
  const double valu = (double)blocks * (threads / 64) * iters * NV;
  printf("NM=%d NV=%2d waves/SIMD=%d  wall %.3f ms  MFMA %.0f TF/s  VALU %.2f G wave-instr/s  ns per iteration per SIMD %.2f\n", NM, NV,
         waves_per_simd, ms, flop / ms / 1e9, valu / ms / 1e6, ms * 1e6 / iters / waves_per_simd);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<4, 0>(w); run<4, 8>(w); run<4, 16>(w); run<4, 24>(w); run<4, 32>(w); run<0, 32>(w);
  }
  return 0;
}
