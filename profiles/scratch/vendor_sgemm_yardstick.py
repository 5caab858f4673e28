#!/usr/bin/env python3
"""Yardstick only (not part of the product, which never imports torch): what the vendor's float32 GEMM (rocBLAS /
hipBLASLt behind torch.mm, full float32) reaches at the MNIST-shape Linears, next to this library's gemm_fast_kernel /
gemm_rqs_kernel (0.83-0.86 of the 157.3 TFLOP/s f32 MFMA peak).  Prints TFLOP/s and the fraction."""
import torch

assert torch.cuda.is_available()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
PEAK = 157.3
for (M, K, N) in [(65536, 512, 512), (65536, 784, 512), (65536, 512, 9800), (65536, 512, 10112), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float32)
    b = torch.randn(K, N, device=dev, dtype=torch.float32)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    tf = 2.0 * M * K * N / best / 1e9
    print(f"torch.mm f32 {M} x {K} x {N}: {best:.3f} ms, {tf:.1f} TFLOP/s, {tf / PEAK:.3f} of the f32 MFMA peak", flush=True)
    del a, b, c
