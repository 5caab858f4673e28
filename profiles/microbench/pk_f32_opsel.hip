// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950 WITHOUT MFMAs around (the split-halves flow of
// src/real_nvp.py has no matrix work): (1) op_sel / op_sel_hi / neg_lo / neg_hi semantics checked against scalar
// arithmetic, (2) issue cost per wave64 instruction at 1, 2, 4, 8 waves per SIMD, next to v_fma_f32.
//   hipcc -O3 --offload-arch=gfx950 pk_f32_opsel.hip -o pk_f32_opsel
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void semantics(const float* in, float* out) {
  const int t = threadIdx.x;
  v2f a = {in[4 * t + 0], in[4 * t + 1]};
  v2f w = {in[4 * t + 2], in[4 * t + 3]};
  v2f c = {in[4 * t + 1] * 3.f, in[4 * t + 2] * 5.f};
  v2f d0, d1, d2, d3, d4, d5;
  asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d0) : "v"(a), "v"(w));   // a*w.x + w.y
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d1) : "v"(a), "v"(w), "v"(c));            // a*w.x + c
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d2) : "v"(a), "v"(w), "v"(c));  // a*w.y + c
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d3) : "v"(a), "v"(w), "v"(c));  // a*w.y + c.y
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d4) : "v"(a), "v"(c));                // a - c
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d5) : "v"(a), "v"(w), "v"(c));  // a*w.x - c
  float* o = out + 12 * t;
  o[0] = d0.x; o[1] = d0.y; o[2] = d1.x; o[3] = d1.y; o[4] = d2.x; o[5] = d2.y;
  o[6] = d3.x; o[7] = d3.y; o[8] = d4.x; o[9] = d4.y; o[10] = d5.x; o[11] = d5.y;
}

#define CHAINS 8
#define REP 16
template <int PK>
__global__ void cost(float* out, int iters) {
  v2f v[CHAINS];
  for (int i = 0; i < CHAINS; ++i) v[i] = v2f{threadIdx.x * 1e-3f + i, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (PK == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        else if (PK == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(v[i]));
        else if (PK == 3) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v[i]));
        else if (PK == 4) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i].x));
        else if (PK == 5) asm volatile("v_max_f32 %0, %0, %0" : "+v"(v[i].x));
        else if (PK == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i].x) : "v"(v[i].y));
        else if (PK == 7) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i].x));
        else if (PK == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
        else if (PK == 9) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v[i]));
        else if (PK == 10) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i].x) : "v"(v[i].y));
        else if (PK == 11) asm volatile("v_mul_f32 %0, 0x3c23d70a, %0" : "+v"(v[i].x));
        else if (PK == 12) asm volatile("v_fma_f32 %0, %0, s2, %0" : "+v"(v[i].x));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i].x));
      }
  }
  float s = 0;
  for (int i = 0; i < CHAINS; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, int waves_per_simd) {
  const int blocks = 256, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int nb = blocks * (256 * waves_per_simd / threads);
  const int iters = 2000;
  float* out;
  hipMalloc(&out, sizeof(float) * nb * threads);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * REP * CHAINS * waves_per_simd;  // wave instructions per SIMD
  printf("%-34s %d waves/SIMD: %6.2f cycles per wave64 instruction (2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / n * 2.4);
  hipFree(out);
}

int main() {
  const int T = 64;
  float h_in[4 * T], h_out[12 * T];
  for (int i = 0; i < 4 * T; ++i) h_in[i] = sinf(i * 0.37f) * 3.f + 0.1f * i;
  float *d_in, *d_out;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(semantics, dim3(1), dim3(T), 0, 0, d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < T; ++t) {
    const float ax = h_in[4 * t], ay = h_in[4 * t + 1], wx = h_in[4 * t + 2], wy = h_in[4 * t + 3];
    const float cx = ay * 3.f, cy = wx * 5.f;
    const float ref[12] = {fmaf(ax, wx, wy), fmaf(ay, wx, wy), fmaf(ax, wx, cx), fmaf(ay, wx, cy), fmaf(ax, wy, cx), fmaf(ay, wy, cy),
                           fmaf(ax, wy, cy), fmaf(ay, wy, cy), ax - cx, ay - cy, fmaf(ax, wx, -cx), fmaf(ay, wx, -cy)};
    for (int j = 0; j < 12; ++j)
      if (ref[j] != h_out[12 * t + j]) { if (bad < 8) printf("mismatch t=%d j=%d got %g want %g\n", t, j, h_out[12 * t + j], ref[j]); ++bad; }
  }
  printf("op_sel / neg semantics: %s (%d mismatches)\n", bad ? "WRONG" : "as assumed, bit-exact against fmaf", bad);
  for (int w : {1, 2, 4, 8}) {
    run("v_fma_f32", cost<0>, w);
    run("v_pk_fma_f32", cost<1>, w);
    run("v_pk_fma_f32 op_sel broadcast", cost<2>, w);
    run("v_pk_mul_f32", cost<3>, w);
  }
  run("v_mul_f32", cost<4>, 8);
  run("v_max_f32", cost<5>, 8);
  run("v_mov_b32", cost<6>, 8);
  run("v_add_f32", cost<7>, 8);
  run("v_exp_f32", cost<8>, 8);
  run("v_pk_add_f32", cost<9>, 8);
  run("v_fmac_f32 (VOP2)", cost<10>, 8);
  run("v_mul_f32 with a 32-bit literal", cost<11>, 8);
  run("v_fma_f32 with an SGPR operand", cost<12>, 8);
  return 0;
}
