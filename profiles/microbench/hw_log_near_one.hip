// How accurate is v_log_f32 for arguments just above 1?  softplus(v) = log(1 + exp(v)) needs log(u), u = fl(1 + e), with a
// RELATIVE error of ~1e-7 down to e = 2^-23 (nf_rqs_math.h log1p_exp): if the hardware log2 is accurate relative to its
// (tiny) result there, the series branch of log1p_exp is unnecessary.  Prints the worst relative error of
// v_log_f32(u) * ln2 against log(u) in float64 per binade of u - 1, and of v_exp_f32 against exp2 for reference.
//   hipcc -O3 --offload-arch=gfx950 hw_log_near_one.hip -o hw_log_near_one
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* u, float* lg, float* ex, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    lg[i] = __builtin_amdgcn_logf(u[i]);
    ex[i] = __builtin_amdgcn_exp2f(-(u[i] - 1.f) * 20.f);
  }
}

int main() {
  const int per = 4096, bins = 24;
  std::vector<float> u(per * bins);
  for (int b = 0; b < bins; ++b)
    for (int j = 0; j < per; ++j) u[b * per + j] = 1.f + ldexpf(1.f + (float)j / per, -(b + 1));  // u - 1 in [2^-(b+1), 2^-b)
  float *du, *dl, *de;
  hipMalloc(&du, u.size() * 4); hipMalloc(&dl, u.size() * 4); hipMalloc(&de, u.size() * 4);
  hipMemcpy(du, u.data(), u.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((u.size() + 255) / 256), dim3(256), 0, 0, du, dl, de, (int)u.size());
  std::vector<float> l(u.size()), e(u.size());
  hipMemcpy(l.data(), dl, u.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, u.size() * 4, hipMemcpyDeviceToHost);
  printf("u - 1 in [2^-(b+1), 2^-b): worst relative error of v_log_f32(u) vs log2(u); of v_exp_f32(x) vs exp2(x), x = -20 (u - 1)\n");
  for (int b = 0; b < bins; ++b) {
    double wl = 0, we = 0;
    for (int j = 0; j < per; ++j) {
      const int i = b * per + j;
      const double ref = log2((double)u[i]);
      wl = fmax(wl, fabs(l[i] - ref) / ref);
      const double x = -(double)(float)((u[i] - 1.f) * 20.f);
      we = fmax(we, fabs(e[i] - exp2(x)) / exp2(x));
    }
    printf("b = %2d   log %.3e   exp %.3e\n", b + 1, wl, we);
  }
  return 0;
}
