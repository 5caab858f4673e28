// The "bf16x3" arithmetic shared by nf_affine2d_x3.hip, nf_rqs2d_x3.hip and nf_gemm_x3.hip:
// x = hi + mid + lo with three bf16 numbers, round-to-nearest-even at each level (8 + 8 + 8 = 24
// significant bits: the split of a float32 is exact), and the six-term product
//   A B ~= A_l B_h + A_h B_l + A_m B_m + A_m B_h + A_h B_m + A_h B_h        (float32 accumulate,
// smallest terms first; the dropped A_m B_l, A_l B_m, A_l B_l are <= 2^-25 of the product).
// Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <string.h>

namespace nf {
namespace bf16x3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---- host: weights are split once, when they are packed ---------------------------------------
inline unsigned short host_bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float host_bf16_to_float(unsigned short b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// part[0..2] = (hi, mid, lo) of w
inline void host_split3(float w, unsigned short part[3]) {
  part[0] = host_bf16_rne(w);
  const float r1 = w - host_bf16_to_float(part[0]);
  part[1] = host_bf16_rne(r1);
  part[2] = host_bf16_rne(r1 - host_bf16_to_float(part[1]));
}

// ---- device: activations are split where they are produced -------------------------------------
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32: a low, b high
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two values -> their packed (hi, mid, lo) parts: 11 VALU
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(sa, sb);
}
// eight values (one lane's share of a 16-deep MFMA k step) -> three operands
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h, m, l;
    split_pair(v[2 * q], v[2 * q + 1], h, m, l);
    ph[q] = h;
    pm[q] = m;
    pl[q] = l;
  }
  hi = __builtin_bit_cast(bf16x8, ph);
  mid = __builtin_bit_cast(bf16x8, pm);
  lo = __builtin_bit_cast(bf16x8, pl);
}
// acc += A x B, both given as (hi, mid, lo)
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8& Ah, const bf16x8& Am, const bf16x8& Al,
                                      const bf16x8& Bh, const bf16x8& Bm, const bf16x8& Bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc, 0, 0, 0);
}

}  // namespace bf16x3
}  // namespace nf
