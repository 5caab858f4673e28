// Where does linear_rqs_small_m_kernel (nf_small_m_rqs.hip) spend its time at the MNIST shape ([128, 512] x [512, 9800],
// K = 8)?  The kernel source is compiled INTO this program with -DNF_SMR_CLOCKS: every wave leaves the shader clock at
// six program points and the 100 MHz wall clock at entry and exit.  Launched in a chain over eight weight matrices (the
// eight layers' 20 MB each), like the flow does.  Prints microseconds per launch on the stream, the shader clock the
// two counters imply, and the mean / max over waves of each phase.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNF_SMR_CLOCKS [-DNF_SMR_LOOP=2] -I../../include -I../../normalizing_flows_amd/csrc \
//     small_m_rqs_clocks.hip -L../../normalizing_flows_amd -lnf_hip -Wl,-rpath,'$ORIGIN/../../normalizing_flows_amd' -o small_m_rqs_clocks
// (own names: the library holds the same kernel, and one host stub cannot stand for two device functions)
#define linear_rqs_small_m_kernel linear_rqs_small_m_kernel_clocks
#define launch_small_m_rqs launch_small_m_rqs_clocks
#define small_m_rqs_groups small_m_rqs_groups_clocks
#define launch_small_m_rqs_pack launch_small_m_rqs_pack_clocks
#define small_m_rqs_packed_floats small_m_rqs_packed_floats_clocks
#include "../../normalizing_flows_amd/csrc/nf_small_m_rqs.hip"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char** argv) {
  using namespace nf::generic;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int M = 128, Kdim = 512, D = 784, Dt = 392, N = Dt * 25, LMAX = 8;
  const int L = argc > 1 ? atoi(argv[1]) : LMAX;   // weight matrices cycled through (8: the flow; 1: cache-resident)
  float *A, *W, *bias, *y, *ldp;
  unsigned long long* dbg;
  hipMalloc(&A, sizeof(float) * M * Kdim);
  hipMalloc(&W, sizeof(float) * (size_t)LMAX * Kdim * N);
  hipMalloc(&bias, sizeof(float) * N);
  hipMalloc(&y, sizeof(float) * M * D);
  hipMalloc(&ldp, sizeof(float) * 64 * M);
  const int waves = 224 * 8;
  hipMalloc(&dbg, sizeof(unsigned long long) * waves * 16);
  std::vector<float> h((size_t)Kdim * N);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u >> 20) % 17) - 0.08f;
  for (int l = 0; l < L; ++l) hipMemcpy(W + (size_t)l * Kdim * N, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  hipMemcpy(A, h.data(), sizeof(float) * M * Kdim, hipMemcpyHostToDevice);
  const size_t pf = small_m_rqs_packed_floats(N, Kdim, 8);
  float* Wp;
  hipMalloc(&Wp, sizeof(float) * pf * L);
  for (int l = 0; l < L; ++l) launch_small_m_rqs_pack(W + (size_t)l * Kdim * N, Wp + (size_t)l * pf, N, Kdim, 8, st);
  hipStreamSynchronize(st);
  hipMemset(bias, 0, sizeof(float) * N);
  std::vector<float> hy((size_t)M * D, 0.4f);
  hipMemcpy(y, hy.data(), sizeof(float) * hy.size(), hipMemcpyHostToDevice);
  RqsConst rc{8, 0.f, 1.f, 1e-3f, 1e-3f, 0.f};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto go = [&](int i) {
    return launch_small_m_rqs(A, Kdim, Wp + (size_t)(i % L) * pf, bias, y, ldp, reinterpret_cast<float*>(dbg), M, N, Kdim, D,
                              i & 1, 0, rc, st);
  };
  for (int i = 0; i < 64; ++i) if (go(i) != NF_OK) { printf("launch failed: %s\n", nf_last_error()); return 1; }
  hipStreamSynchronize(st);
  const int n = 800;
  hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) go(i);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("loop form %d, %d weight matrices: %.2f us per launch on the stream\n", NF_SMR_LOOP, L, ms * 1e3 / n);
  std::vector<unsigned long long> t((size_t)waves * 16);
  hipMemcpy(t.data(), dbg, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost);
  const char* names[5] = {"set-up (addresses, zeroing)", "loads + MFMAs", "cross-wave sum (2 row tiles)", "spline", "log-det partials"};
  double ghz = 0.0;
  unsigned long long wall_lo = ~0ull, wall_hi = 0;
  for (int w = 0; w < waves; ++w) {
    const unsigned long long* q = &t[(size_t)w * 16];
    ghz += (double)(q[5] - q[0]) / ((double)(q[7] - q[6]) * 10.0);
    wall_lo = q[6] < wall_lo ? q[6] : wall_lo;
    wall_hi = q[7] > wall_hi ? q[7] : wall_hi;
  }
  ghz /= waves;
  printf("shader clock %.2f GHz; first wave in to last wave out: %.2f us\n", ghz, (double)(wall_hi - wall_lo) / 100.0);
  for (int p = 0; p < 5; ++p) {
    double mean = 0.0, mx = 0.0;
    for (int w = 0; w < waves; ++w) {
      const double c = (double)(t[(size_t)w * 16 + p + 1] - t[(size_t)w * 16 + p]);
      mean += c;
      mx = c > mx ? c : mx;
    }
    mean /= waves;
    printf("  %-30s mean %8.0f cycles = %6.2f us   max %8.0f cycles = %6.2f us\n", names[p], mean, mean / (ghz * 1e3), mx, mx / (ghz * 1e3));
  }
#if NF_SMR_LOOP == 2
  {
    const int at[4] = {1, 9, 11, 13};
    const char* what[3] = {"group 0 landed + its MFMAs", "group 1 (loads overlapped)", "group 2 (loads overlapped)"};
    for (int p = 0; p < 3; ++p) {
      double mean = 0.0, mx = 0.0;
      for (int w = 0; w < waves; ++w) {
        const double c = (double)(t[(size_t)w * 16 + at[p + 1]] - t[(size_t)w * 16 + at[p]]);
        mean += c;
        mx = c > mx ? c : mx;
      }
      mean /= waves;
      printf("    %-28s mean %8.0f cycles = %6.2f us   max %8.0f cycles = %6.2f us\n", what[p], mean, mean / (ghz * 1e3), mx, mx / (ghz * 1e3));
    }
  }
#endif
#if NF_SMR_LOOP == 0
  const char* inner[8] = {"group 0: loads landed", "group 0: MFMAs issued", "group 1: loads landed", "group 1: MFMAs issued",
                          "group 2: loads landed", "group 2: MFMAs issued", "group 3: loads landed", "group 3: MFMAs issued"};
  for (int p = 0; p < 8; ++p) {
    double mean = 0.0, mx = 0.0;
    for (int w = 0; w < waves; ++w) {
      const double c = (double)(t[(size_t)w * 16 + 8 + p] - t[(size_t)w * 16 + (p == 0 ? 1 : 7 + p)]);
      mean += c;
      mx = c > mx ? c : mx;
    }
    mean /= waves;
    printf("    %-26s mean %8.0f cycles = %6.2f us   max %8.0f cycles = %6.2f us\n", inner[p], mean, mean / (ghz * 1e3), mx, mx / (ghz * 1e3));
  }
#endif
  return 0;
}
