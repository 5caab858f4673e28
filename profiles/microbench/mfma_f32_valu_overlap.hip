// Microbenchmark: does v_mfma_f32_32x32x2_f32 overlap with f32 VALU work on one SIMD of gfx950?
// Loop body = NM independent MFMAs + NV independent v_fma_f32, measured with s_memtime for
// 1, 2 and 4 waves per SIMD.  Prints cycles per loop iteration per wave and per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_f32_valu_overlap.hip -o mfma_overlap && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV>
__global__ void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NM, int NV>
void run(int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 20000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += (double)c; mean /= h.size();
  // s_memtime ticks at 100 MHz on gfx9?  report raw ticks per iteration per wave; and per-SIMD cost
  const double flop = (double)blocks * (threads / 64) * iters * NM * 4096.0;
  printf("NM=%d NV=%2d waves/SIMD=%d  ticks/iter/wave=%8.2f  per SIMD per (iter)=%8.2f  wall %.3f ms  MFMA %.1f TF/s  implied clock %.2f GHz\n", NM, NV,
         waves_per_simd, mean / iters, mean / iters / waves_per_simd, ms, flop / ms / 1e9, mean / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 3, 4}) {
    run<4, 0>(w); run<4, 4>(w); run<4, 8>(w); run<4, 16>(w); run<8, 8>(w);
  }
  return 0;
}
