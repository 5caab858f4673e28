// How long does ONE gemm_small_m_kernel launch take inside a chain of dependent launches, outside any profiler?
// (rocprofv3 --kernel-trace shows >= 4.2 us for EVERY kernel of the batch-128 MNIST-shape call, empty ones included --
// launch_boundary.hip shows 2.7-3.0 us per dependent launch, host-bound, without it.)  Calls the library's own launcher
// (libnf_hip.so, C++ symbol) in a loop on one stream: [M, K] x [K, N] for the hidden Linear (128 x 512 x 512), the
// first Linear (K = 784) and a chain alternating two buffers (true data dependence).
//   hipcc -O3 --offload-arch=gfx950 -I../../include -I../../normalizing_flows_amd/csrc small_m_chain.hip \
//         -L../../normalizing_flows_amd -lnf_hip -Wl,-rpath,'$ORIGIN/../../normalizing_flows_amd' -o small_m_chain
#include "nf_internal.h"
#include <stdio.h>
#include <chrono>
#include <vector>

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int M = 128;
  float *A, *B, *W1, *W2, *bias;
  hipMalloc(&A, sizeof(float) * M * 784);
  hipMalloc(&B, sizeof(float) * M * 784);
  hipMalloc(&W1, sizeof(float) * 784 * 512);
  hipMalloc(&W2, sizeof(float) * 512 * 512);
  hipMalloc(&bias, sizeof(float) * 512);
  hipMemset(A, 0, sizeof(float) * M * 784);
  hipMemset(B, 0, sizeof(float) * M * 784);
  hipMemset(W1, 0, sizeof(float) * 784 * 512);
  hipMemset(W2, 0, sizeof(float) * 512 * 512);
  hipMemset(bias, 0, sizeof(float) * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, int K, const float* W) {
    const int N = 2000;
    for (int i = 0; i < 100; ++i) nf::generic::launch_gemm_small_m(i & 1 ? B : A, K, W, bias, i & 1 ? A : B, 512, M, 512, K, 1, st, nullptr);
    hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) nf::generic::launch_gemm_small_m(i & 1 ? B : A, K, W, bias, i & 1 ? A : B, 512, M, 512, K, 1, st, nullptr);
    hipEventRecord(e1, st);
    const auto t1 = std::chrono::steady_clock::now();
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %6.2f us per launch on the stream, %5.2f us of host time to enqueue one\n", name, ms * 1e3 / N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  };
  run("hidden Linear 128 x 512 x 512 (chain)", 512, W2);
  run("first Linear 128 x 784 x 512 (chain)", 784, W1);
  return 0;
}
