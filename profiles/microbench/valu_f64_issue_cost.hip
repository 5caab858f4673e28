// Issue cost (cycles per wave64 instruction, one wave per SIMD, independent chains) of the float64 and
// transcendental instructions the accurate spline knots could be built from (nf_rqs_math.h): which of
// v_fma_f64 / v_add_f64 / v_mul_f64 / v_cvt_* / v_rndne_f64 / v_ldexp_f64 / v_rcp_f64 run at the float32 rate?
//   hipcc -O3 --offload-arch=gfx950 valu_f64_issue_cost.hip -o valu_f64_issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAINS 8
#define REP 16

// OPS: asm statement applied to 8 independent registers per iteration (x REP)
#define KERNEL(NAME, DECL, INIT, ASM, CONSTR, SINK)                                            \
  __global__ void NAME(float* out, unsigned long long* cyc, int iters) {                       \
    DECL v[CHAINS];                                                                            \
    for (int i = 0; i < CHAINS; ++i) v[i] = INIT;                                              \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
    for (int it = 0; it < iters; ++it) {                                                       \
      _Pragma("unroll") for (int r = 0; r < REP; ++r) {                                        \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : CONSTR(v[i]));   \
      }                                                                                        \
    }                                                                                          \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
    float s = 0;                                                                               \
    for (int i = 0; i < CHAINS; ++i) s += SINK(v[i]);                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
  }

#define CD(x) "+v"(x)
#define FSINK(x) (x)
#define DSINK(x) ((float)(x))

KERNEL(k_fma_f32, float, threadIdx.x * 1e-3f + i, "v_fma_f32 %0, %0, %0, %0", CD, FSINK)
KERNEL(k_exp_f32, float, threadIdx.x * 1e-3f + i, "v_exp_f32 %0, %0", CD, FSINK)
KERNEL(k_log_f32, float, threadIdx.x * 1e-3f + i + 1, "v_log_f32 %0, %0", CD, FSINK)
KERNEL(k_rcp_f32, float, threadIdx.x * 1e-3f + i + 1, "v_rcp_f32 %0, %0", CD, FSINK)
KERNEL(k_rndne_f32, float, threadIdx.x * 1e-3f + i, "v_rndne_f32 %0, %0", CD, FSINK)
KERNEL(k_cvt_i32_f32, float, threadIdx.x * 1e-3f + i, "v_cvt_i32_f32 %0, %0", CD, FSINK)
KERNEL(k_ldexp_f32, float, threadIdx.x * 1e-3f + i, "v_ldexp_f32 %0, %0, 1", CD, FSINK)
KERNEL(k_add_u32, float, threadIdx.x * 1e-3f + i, "v_add_u32 %0, %0, %0", CD, FSINK)
KERNEL(k_fma_f64, double, threadIdx.x * 1e-3 + i, "v_fma_f64 %0, %0, %0, %0", CD, DSINK)
KERNEL(k_add_f64, double, threadIdx.x * 1e-3 + i, "v_add_f64 %0, %0, %0", CD, DSINK)
KERNEL(k_mul_f64, double, threadIdx.x * 1e-3 + i, "v_mul_f64 %0, %0, %0", CD, DSINK)
KERNEL(k_rndne_f64, double, threadIdx.x * 1e-3 + i, "v_rndne_f64 %0, %0", CD, DSINK)
KERNEL(k_ldexp_f64, double, threadIdx.x * 1e-3 + i, "v_ldexp_f64 %0, %0, 1", CD, DSINK)
KERNEL(k_rcp_f64, double, threadIdx.x * 1e-3 + i + 1, "v_rcp_f64 %0, %0", CD, DSINK)
// conversions change the register width: a float and a double register per chain, converted back and forth
#define KERNEL2(NAME, TA, TB, ASM_AB, ASM_BA)                                                  \
  __global__ void NAME(float* out, unsigned long long* cyc, int iters) {                       \
    TA a[CHAINS];                                                                              \
    TB b[CHAINS];                                                                              \
    for (int i = 0; i < CHAINS; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 0; }              \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
    for (int it = 0; it < iters; ++it) {                                                       \
      _Pragma("unroll") for (int r = 0; r < REP; ++r) {                                        \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) {                                   \
          asm volatile(ASM_AB : "=v"(b[i]) : "v"(a[i]));                                       \
          asm volatile(ASM_BA : "=v"(a[i]) : "v"(b[i]));                                       \
        }                                                                                      \
      }                                                                                        \
    }                                                                                          \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
    float s = 0;                                                                               \
    for (int i = 0; i < CHAINS; ++i) s += (float)a[i] + (float)b[i];                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
  }
KERNEL2(k_cvt_pair, double, float, "v_cvt_f32_f64 %0, %1", "v_cvt_f64_f32 %0, %1")
KERNEL2(k_cvt_i32_f64_pair, double, int, "v_cvt_i32_f64 %0, %1", "v_cvt_f64_i32 %0, %1")

template <class K>
void run(const char* name, K kern, int per_iter) {
  const int blocks = 256, threads = 256, iters = 2000;  // one wave per SIMD
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * REP * CHAINS * per_iter;
  printf("%-28s %6.2f ns per instruction per SIMD  (x 2.4 GHz = %5.1f cycles)\n", name, ms * 1e6 / n, ms * 1e6 / n * 2.4);
  hipFree(out); hipFree(cyc);
}

int main() {
  run("v_fma_f32", k_fma_f32, 1);
  run("v_exp_f32", k_exp_f32, 1);
  run("v_log_f32", k_log_f32, 1);
  run("v_rcp_f32", k_rcp_f32, 1);
  run("v_rndne_f32", k_rndne_f32, 1);
  run("v_cvt_i32_f32", k_cvt_i32_f32, 1);
  run("v_ldexp_f32", k_ldexp_f32, 1);
  run("v_add_u32", k_add_u32, 1);
  run("v_fma_f64", k_fma_f64, 1);
  run("v_add_f64", k_add_f64, 1);
  run("v_mul_f64", k_mul_f64, 1);
  run("v_rndne_f64", k_rndne_f64, 1);
  run("v_ldexp_f64", k_ldexp_f64, 1);
  run("v_rcp_f64", k_rcp_f64, 1);
  run("v_cvt_f32_f64 + v_cvt_f64_f32", k_cvt_pair, 2);
  run("v_cvt_i32_f64 + v_cvt_f64_i32", k_cvt_i32_f64_pair, 2);
  return 0;
}
