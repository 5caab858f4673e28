// v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) which lane / register holds A[b][i], B[b][j], D[b][i][j] (16 blocks of
// 4x4, K = 1) -- the fused affine backward uses it to sum lane-held per-sample values over the SAMPLES of a tile into
// per-unit accumulators (rows = four per-sample factors, columns = 64 units, one instruction per sample);
// (2) its issue cost alone and between v_mfma_f32_32x32x2_f32.
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_4x4x1_layout.hip -o mfma_f32_4x4x1_layout
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  // A lane (b, i) = (1 + i) + 10 b, B lane (b, j) = 10^j: D = A * B shows which operand lane feeds which result
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  const float pw[4] = {1.f, 10.f, 100.f, 1000.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + (l & 3) + 10 * (l >> 2)), pw[l & 3], d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[4 * l + r] = d[r];
}

template <int MODE>
__global__ void cost(float* out, int iters) {
  f32x4 d[4];
  f32x16 big[2];
  for (int i = 0; i < 4; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0 || MODE == 2) {  // 8 x (4 small)
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[i], 0, 0, 0);
      }
      if (MODE == 1 || MODE == 2) {  // 8 x (2 big)
        big[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[0], 0, 0, 0);
        big[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[1], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][3];
  for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * (1 << 20));
  layout<<<1, 64>>>(out);
  float h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("expected if D[reg i][lane 4b+j] = A[lane 4b+i] * B[lane 4b+j]: (1 + i + 10 b) * 10^j\n");
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const float pw[4] = {1.f, 10.f, 100.f, 1000.f};
      const float want = (float)(1 + r + 10 * (l >> 2)) * pw[l & 3];
      if (h[4 * l + r] != want) { if (bad < 8) printf("  lane %d reg %d: %g, expected %g\n", l, r, h[4 * l + r], want); ++bad; }
    }
  printf("layout %s (%d mismatches)\n", bad ? "DIFFERS" : "confirmed", bad);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves = 1; waves <= 2; ++waves) {
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) cost<0><<<256, 256 * waves>>>(out, iters);
        if (mode == 1) cost<1><<<256, 256 * waves>>>(out, iters);
        if (mode == 2) cost<2><<<256, 256 * waves>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
      }
    }
    // per loop iteration and wave: mode 0: 32 small; mode 1: 16 big; mode 2: both
    const double ns0 = ms[0] * 1e6 / iters / waves, ns1 = ms[1] * 1e6 / iters / waves, ns2 = ms[2] * 1e6 / iters / waves;
    printf("%d wave(s)/SIMD: 32 x 4x4x1 %.1f ns (%.2f ns each); 16 x 32x32x2 %.1f ns (%.2f each); both %.1f ns -> a 4x4x1 beside the "
           "big ones costs %.2f ns (%.1f cycles at 2.4 GHz)\n", waves, ns0, ns0 / 32, ns1, ns1 / 16, ns2, (ns2 - ns1) / 32,
           (ns2 - ns1) / 32 * 2.4);
  }
  return 0;
}
