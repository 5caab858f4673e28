// What does ONE vector instruction cost next to v_mfma_f32_32x32x2_f32, by operand FORM?  valu_f32_operands.hip found
// (no MFMA, 8 waves per SIMD) VGPR-only v_fma/mul/add at 2.3-3.3 cycles and SGPR-operand forms, v_max, float64 and
// packed forms at 4.1-4.4; mfma_f32_valu_overlap.hip had measured "4-6 cycles of MFMA time per VALU" with an FMA that
// takes SGPR operands.  Loop body: 4 MFMAs (256 pipe cycles) + NV instructions of one form, 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_valu_forms.hip -o mfma_f32_valu_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP, int NV>
__global__ void k(float* out, int iters, float sa, float sb) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float d[8], a[8], b[8];
  double dd[8], da[8];
  for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 1e-3f + i; a[i] = 0.999f + 1e-4f * i; b[i] = 1e-3f * i; dd[i] = d[i]; da[i] = a[i]; }
  const float ma = sa + threadIdx.x * 1e-6f, mb = sb;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc[m], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int i = q & 7;
      if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
      else if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "s"(sa));
      else if (OP == 2) asm volatile("v_mul_f32 %0, 0x3c23d70a, %0" : "+v"(d[i]));
      else if (OP == 3) asm volatile("v_max_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
      else if (OP == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
      else if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(d[i]));
      else if (OP == 6) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dd[i]) : "v"(da[i]));
      else if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
      else if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      else if (OP == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[i]) : "v"(a[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += d[i] + (float)dd[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int NV>
double run() {
  const int threads = 1024, blocks = 256, iters = 20000;  // 4 waves per SIMD
  float* out;
  (void)hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e0, 0);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  (void)hipFree(out);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 4);  // cycles per loop iteration per wave-slot of the SIMD (4 waves share it)
}

template <int OP>
void form(const char* name, double base) {
  const double c4 = run<OP, 4>(), c8 = run<OP, 8>(), c16 = run<OP, 16>();
  printf("%-36s  +%5.2f  +%5.2f  +%5.2f cycles of SIMD time per instruction (4 / 8 / 16 per 4 MFMAs)\n", name,
         (c4 - base) / 4, (c8 - base) / 8, (c16 - base) / 16);
}

int main() {
  const double base = run<0, 0>();
  printf("4 MFMAs alone: %.1f cycles per iteration (256 = the pipe's time)\n", base);
  form<0>("v_fma_f32 d, a, b, d (VGPRs)", base);
  form<1>("v_fma_f32 d, a, s, d (SGPR)", base);
  form<2>("v_mul_f32 d, literal, d", base);
  form<4>("v_add_f32 d, a, d", base);
  form<7>("v_fma_f32 d, a, b, d clamp", base);
  form<8>("v_mov_b32 d, a", base);
  form<3>("v_max_f32 d, a, d", base);
  form<5>("v_exp_f32 d, d", base);
  form<6>("v_add_f64 d, a, d", base);
  form<9>("v_cvt_f64_f32 d, a", base);
  return 0;
}
