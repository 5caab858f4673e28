// Issue cost of f32 VALU instructions on gfx950 as a function of their OPERANDS (8 waves per SIMD, independent
// chains, no MFMA): pk_f32_opsel.hip found v_fma_f32 v0,v0,v0,v0 at 2.3 cycles per wave64 instruction but the same
// opcode with an SGPR source, v_max_f32 and every packed form at 4.2 -- which forms does real code get?
//   hipcc -O3 --offload-arch=gfx950 valu_f32_operands.hip -o valu_f32_operands
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define REP 16

template <int OP>
__global__ void cost(float* out, int iters, float sc) {
  float d[CHAINS], a[CHAINS], b[CHAINS];
  v2f pd[CHAINS], pa[CHAINS], pb[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    d[i] = threadIdx.x * 1e-3f + i; a[i] = 0.999f + i * 1e-4f; b[i] = 1e-3f * i;
    pd[i] = v2f{d[i], d[i]}; pa[i] = v2f{a[i], a[i]}; pb[i] = v2f{b[i], b[i]};
  }
  double dd[CHAINS], da[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { dd[i] = d[i]; da[i] = a[i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if (OP == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 3) asm volatile("v_max_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pd[i]) : "v"(pa[i]), "v"(pb[i]));
        else if (OP == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "s"(sc));
        else if (OP == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if (OP == 7) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 8) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(d[i]) : "s"(sc));
        else if (OP == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d[i]) : "s"(sc), "v"(b[i]));
        else if (OP == 10) asm volatile("v_min_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 11) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 12) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if (OP == 13) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if (OP == 14) asm volatile("v_mul_f32 %0, %1, %0 clamp" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 15) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 16) asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(d[i]) : "v"(a[i]));
        else if (OP == 17) asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if (OP == 18) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dd[i]) : "v"(da[i]));
        else if (OP == 19) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(dd[i]) : "v"(da[i]));
        else if (OP == 20) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[i]) : "v"(d[i]));
        else if (OP == 21) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d[i]) : "v"(dd[i]));
        else if (OP == 22) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dd[i]) : "v"(da[i]));
        else if (OP == 23) asm volatile("v_log_f32 %0, %0" : "+v"(d[i]));
        else if (OP == 24) asm volatile("v_rcp_f32 %0, %0" : "+v"(d[i]));
        else if (OP == 25) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(d[i]) : "v"(a[i]) : "vcc");
        else if (OP == 26) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(d[i]) : "v"(a[i]) : "vcc");
      }
  }
  float s = 0;
  for (int i = 0; i < CHAINS; ++i) s += d[i] + pd[i].x + pd[i].y + (float)dd[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern) {
  const int nb = 512, threads = 1024, iters = 2000;  // 8 waves per SIMD
  float* out;
  (void)hipMalloc(&out, sizeof(float) * nb * threads);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, out, iters, 1.0001f);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * REP * CHAINS * 8;
  printf("%-44s %6.2f cycles per wave64 instruction (2.4 GHz, 8 waves/SIMD)\n", name, ms * 1e6 / n * 2.4);
  (void)hipFree(out);
}

int main() {
  run("v_fma_f32 d, a, b, d (3 VGPRs)", cost<0>);
  run("v_mul_f32 d, a, d", cost<1>);
  run("v_fma_f32 d, d, a, a", cost<2>);
  run("v_max_f32 d, a, d", cost<3>);
  run("v_min_f32 d, a, d", cost<10>);
  run("v_pk_fma_f32 d, a, b, d", cost<4>);
  run("v_fma_f32 d, a, s, d (SGPR)", cost<5>);
  run("v_fmac_f32 d, a, b", cost<6>);
  run("v_fmac_f32 d, s, b (SGPR)", cost<9>);
  run("v_add_f32 d, a, d", cost<7>);
  run("v_sub_f32 d, a, d", cost<15>);
  run("v_mul_f32 d, s, d (SGPR)", cost<8>);
  run("v_med3_f32 d, a, b, d", cost<12>);
  run("v_fma_f32 d, a, b, d clamp", cost<13>);
  run("v_mul_f32 d, a, d clamp", cost<14>);
  run("v_mul_legacy_f32 d, a, d", cost<16>);
  run("v_fma_f32 d, |a|, b, d", cost<17>);
  run("v_add_f64 d, a, d", cost<18>);
  run("v_fma_f64 d, a, a, d", cost<19>);
  run("v_mul_f64 d, a, d", cost<22>);
  run("v_cvt_f64_f32", cost<20>);
  run("v_cvt_f32_f64", cost<21>);
  run("v_log_f32", cost<23>);
  run("v_rcp_f32", cost<24>);
  run("v_cmp_ge_f32 + v_cndmask_b32 (pair)", cost<25>);
  run("v_cmp_ge_f32 + v_addc_co_u32 (pair)", cost<26>);
  return 0;
}
