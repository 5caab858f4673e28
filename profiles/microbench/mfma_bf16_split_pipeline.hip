// Which ingredient of the bf16x3 inner step keeps VALU work from hiding behind the bf16 MFMAs?
// One loop iteration = the kernel q-step: 12 x v_mfma_f32_32x32x16_bf16 + the 52-VALU operand
// split for the next step, interleaved 1 MFMA : 4-5 VALU with sched_group_barrier.  Variants:
//   V0 kernel-like: two accumulators, six dependent MFMAs each; B operands are the split's output
//   V1 as V0 but MFMAs rotate over four accumulators, consecutive ones on the same m are 2 apart
//   V2 as V0 but the MFMAs read constant B registers (split results only folded into a checksum)
//   V3 as V0 but the split is replaced by 96 independent v_fma_f32
//   V4 MFMAs only, V5 split only
//   V7 as V0 but the data never changes (xc made opaque instead of incremented): same instruction
//      stream, MFMA inputs identical every iteration -- separates data-dependent power from issue
//   V8 as V2 plus one v_mov per step that rewrites a word of the constant B (VALU write -> MFMA read)
//   V9 as V6 but the MFMAs consume the operands produced TWO steps earlier
//   V10 as V4 (MFMAs only) but A and B hold dense pseudo-random mantissas instead of small round numbers
//   V11 as V2 with the same dense constants
//   V6 as V0 with three rotating B buffers, each kept live (empty asm) until the MFMAs of the
//      following step have issued, so that no VALU result lands in a register an in-flight MFMA reads
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = v[2 * q], b = v[2 * q + 1];
    const unsigned h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    ph[q] = h; pm[q] = m; pl[q] = cvt_pk_bf16(sa, sb);
  }
  hi = __builtin_bit_cast(bf16x8, ph); mid = __builtin_bit_cast(bf16x8, pm); lo = __builtin_bit_cast(bf16x8, pl);
}

template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 A[6], Bc[3], B[3][3];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) A[i][j] = (__bf16)(a * (i + 1) + j);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) { Bc[i][j] = (__bf16)(b * (i + 1) + j); B[0][i][j] = Bc[i][j]; B[1][i][j] = Bc[i][j]; B[2][i][j] = Bc[i][j]; }
  if (V == 10 || V == 11) {
    unsigned hsh = threadIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { hsh = hsh * 1664525u + 1013904223u; A[i][j] = (__bf16)(0.5f + (hsh >> 8) * (1.0f / 16777216.0f)); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) { hsh = hsh * 1664525u + 1013904223u; Bc[i][j] = (__bf16)((0.5f + (hsh >> 8) * (1.0f / 16777216.0f)) * (i == 0 ? 1.f : (i == 1 ? 0.0039f : 1.5e-5f))); }
  }
  float w1[8], bb[8], f[8];
  for (int j = 0; j < 8; ++j) { w1[j] = a + j * 1e-3f; bb[j] = b * 1e-3f * j; f[j] = threadIdx.x * 1e-3f + j; }
  float xc = threadIdx.x * 1e-4f;
  unsigned chk = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < ((V == 6 || V == 9) ? 3 : 2); ++q) {
      constexpr int NB = (V == 6 || V == 9) ? 3 : 2;
      if (V != 4 && V != 10) {
        if (V == 3) {
#pragma unroll
          for (int n = 0; n < 96; ++n) f[n & 7] = fmaf(f[n & 7], a, b);
        } else {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(fmaf(xc, w1[j], bb[j]), 0.f, 1.f);
          split3(v, B[(q + 1) % NB][0], B[(q + 1) % NB][1], B[(q + 1) % NB][2]);
          if (V == 7) asm volatile("" : "+v"(xc)); else xc += 1e-6f;
          if (V == 2 || V == 5 || V == 8 || V == 11) {
#pragma unroll
            for (int part = 0; part < 3; ++part) {
              u32x4 t = __builtin_bit_cast(u32x4, B[(q + 1) % NB][part]);
              chk += t[0] + t[1] + t[2] + t[3];
            }
          }
        }
      }
      if (V != 5) {
        if (V == 8) {
          u32x4 t = __builtin_bit_cast(u32x4, Bc[2]);
          asm volatile("v_mov_b32 %0, %0" : "+v"(t[3]));
          Bc[2] = __builtin_bit_cast(bf16x8, t);
        }
        const int RD = V == 9 ? (q + NB - 1) % NB : q % NB;  // V9: written during step q-2
        const bf16x8 Bh = (V == 2 || V == 3 || V == 8 || V == 10 || V == 11) ? Bc[0] : B[RD][0];
        const bf16x8 Bm = (V == 2 || V == 3 || V == 8 || V == 10 || V == 11) ? Bc[1] : B[RD][1];
        const bf16x8 Bl = (V == 2 || V == 3 || V == 8 || V == 10 || V == 11) ? Bc[2] : B[RD][2];
        if (V == 1) {
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int m = 0; m < 4; ++m)
              acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(g * 4 + m) % 6], (g + m) % 3 == 0 ? Bh : ((g + m) % 3 == 1 ? Bm : Bl), acc[m], 0, 0, 0);
        } else {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2 + m * 3], Bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0 + m * 3], Bl, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1 + m * 3], Bm, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1 + m * 3], Bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0 + m * 3], Bm, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0 + m * 3], Bh, acc[m], 0, 0, 0);
          }
        }
      }
      if (V == 6) {  // the operands of the PREVIOUS step stay live up to here
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          u32x4 keep = __builtin_bit_cast(u32x4, B[(q + NB - 1) % NB][part]);
          asm volatile("" ::"v"(keep));
        }
      }
      if (V == 3) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
      } else if (V != 4 && V != 5 && V != 10) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
    }
  }
  float s = __uint_as_float(chk & 0x3fffffffu) + xc;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char* what) {
  const int threads = 512, blocks = 256, iters = 10000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  // per SIMD: 2 waves x iters x 2 q-steps
  printf("V%d %-44s wall %.3f ms  -> %.1f ns per q-step (12 MFMA + 52 VALU (V3: 96)) per SIMD\n", V, what, ms,
         ms * 1e6 / (2.0 * iters * ((V == 6 || V == 9) ? 3 : 2)));
  hipFree(out);
}

int main() {
  run<4>("MFMAs only");
  run<5>("split only");
  run<0>("kernel-like");
  run<1>("four accumulators");
  run<2>("MFMAs read constant B");
  run<3>("independent fma instead of the split");
  run<6>("three B buffers kept live one step longer");
  run<7>("kernel-like, identical data every step");
  run<8>("constant B, one v_mov into it per step");
  run<10>("MFMAs only, dense mantissas");
  run<11>("constant dense B + split alongside");
  run<9>("MFMAs consume operands two steps old");
  return 0;
}
