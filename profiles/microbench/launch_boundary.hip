// What does ONE dependent kernel boundary cost on this box, on a hipStreamNonBlocking stream like a handle's?  The
// small-batch analysis of DESIGN.md 3.8 rests on it: a chain of N kernels that each depend on the previous one
// (same stream), timed with events around the whole chain, for
//   (a) an empty kernel, 1 workgroup of 64 threads          (b) an empty kernel, 256 workgroups of 256 threads
//   (c) 64 workgroups of 1024 threads with 64 KB of LDS (gemm_small_m_kernel's geometry), each storing one float
//   (d) like (c) but every workgroup also reads 128 KB from a 32 MB buffer (the hidden Linear's operand volume)
// plus the host's time to enqueue the chain.
//   hipcc -O3 --offload-arch=gfx950 launch_boundary.hip -o launch_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void empty_kernel(float* out) { if (out == nullptr) __builtin_trap(); }

__global__ __launch_bounds__(1024) void lds_kernel(float* out) {
  __shared__ float s[16384];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[1023];
}

__global__ __launch_bounds__(1024) void lds_read_kernel(float* out, const float4* __restrict__ src) {
  __shared__ float s[16384];
  // 128 KB per workgroup: 1024 threads x 8 float4
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = src + (size_t)blockIdx.x * 8192 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const float4 v = p[q * 1024]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  s[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[1023];
}

template <typename F>
static void run(const char* name, hipStream_t st, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int N = 2000;
  for (int i = 0; i < 200; ++i) launch();
  hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  hipEventRecord(e0, st);
  for (int i = 0; i < N; ++i) launch();
  hipEventRecord(e1, st);
  const auto t1 = std::chrono::steady_clock::now();
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
  printf("%-58s %6.2f us per kernel on the stream, %5.2f us of host time to enqueue one\n", name, ms * 1e3 / N, host_us);
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float* out;
  float4* src;
  hipMalloc(&out, 1 << 20);
  hipMalloc(&src, 32 << 20);
  hipMemset(src, 0, 32 << 20);
  run("(a) empty, 1 x 64", st, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, out); });
  run("(b) empty, 256 x 256", st, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, out); });
  run("(c) 64 x 1024 threads, 64 KB LDS, one store each", st, [&] { hipLaunchKernelGGL(lds_kernel, dim3(64), dim3(1024), 0, st, out); });
  run("(d) the same + 128 KB read per workgroup", st, [&] { hipLaunchKernelGGL(lds_read_kernel, dim3(64), dim3(1024), 0, st, out, src); });
  run("(e) (d) on 256 workgroups (32 MB read)", st, [&] { hipLaunchKernelGGL(lds_read_kernel, dim3(256), dim3(1024), 0, st, out, src); });
  return 0;
}
