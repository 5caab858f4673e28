"""Per-call latency of the reference's call sites through haiku_like (batch 128, the reference's size)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import normalizing_flows_amd as distrax
import normalizing_flows_amd.haiku_like as hk
from test_abi_and_host import _reference_style_log_prob
log_prob = _reference_style_log_prob(distrax, hk, num_layers=8)
params = log_prob.init(42, np.zeros((1, 2)))
batch = np.random.default_rng(0).standard_normal((128, 2)).astype(np.float32)
for _ in range(20): log_prob.apply(params, batch)
t0 = time.perf_counter()
N = 300
for _ in range(N): out = log_prob.apply(params, batch)
dt = (time.perf_counter() - t0) / N
print(f"log_prob.apply(params, batch[128]): {dt*1e6:.0f} us per call")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): log_prob.apply(params, batch)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
