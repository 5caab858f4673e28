#!/usr/bin/env python3
"""Yardstick only (the product never imports torch): the headline workload -- 8 masked affine coupling layers, MLP
2-64-64-2, 2^20 rows, log_prob + mean -- written the obvious way in PyTorch eager float32 on the same MI355X, i.e.
what a framework user gets without a fused kernel (every activation matrix goes through HBM)."""
import math

import torch

assert torch.cuda.is_available()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(534)
L, rows = 8, 1 << 20
layers = []
for l in range(L):
    W1 = (torch.randn(2, 64, generator=g) / math.sqrt(2)).to(dev); b1 = torch.zeros(64, device=dev)
    W2 = (torch.randn(64, 64, generator=g) / 8).to(dev); b2 = torch.zeros(64, device=dev)
    W3 = (torch.randn(64, 4, generator=g) * 0.02).to(dev); b3 = torch.zeros(4, device=dev)
    mask = torch.tensor([float((l + d) % 2) for d in range(2)], device=dev)
    layers.append((W1, b1, W2, b2, W3, b3, mask))
x = torch.randn(rows, 2, device=dev)


def nll(x):
    ld = torch.zeros(x.shape[0], device=dev)
    y = x
    for W1, b1, W2, b2, W3, b3, mask in reversed(layers):      # log_prob walks the chain backwards (distrax.Inverse)
        h = torch.relu((y * mask) @ W1 + b1)
        h = torch.relu(h @ W2 + b2)
        p = (h @ W3 + b3).reshape(-1, 2, 2)
        shift, ls = p[..., 0], p[..., 1]
        y = torch.where(mask.bool(), y, (y - shift) * torch.exp(-ls))
        ld = ld - (ls * (1 - mask)).sum(-1)
    lp = -0.5 * (y * y).sum(-1) - math.log(2 * math.pi) + ld
    return -lp.mean()


for _ in range(3):
    v = nll(x)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v = nll(x)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"PyTorch eager float32, 8 affine layers, 2^20 rows: {best:.2f} ms per log_prob + NLL = {rows / best * 1e3:.3e} evals/s (nll {float(v):.4f})")
