"""Throughput of the drop-in call with HOST buffers (NumPy in, NumPy out: NF_MEM_HOST), i.e. what
model.log_prob(batch) costs a script that keeps its data in host memory.  python profiles/host_path.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synthetic as O  # product generators; the oracle is not used here
from normalizing_flows_amd import _lib, flow as nfa

spec = O.make_affine_spec(8, final_std=0.02)
cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), first_mask_parity=1)
f = nfa.CouplingFlow(cfg, spec.layers)
for lg in (12, 16, 20, 23):
    rows = 1 << lg
    x = O.make_inputs("banana", rows, seed=1)
    for _ in range(3):
        f.log_prob(x)
    n = 20 if lg <= 20 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        lp = f.log_prob(x)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        v = f.nll(x)
    dn = (time.perf_counter() - t0) / n
    print(f"rows 2^{lg}: log_prob(host) {dt*1e3:8.3f} ms  {rows/dt/1e6:9.1f} M evals/s   "
          f"PCIe bytes {rows*12/dt/1e9:6.2f} GB/s | nll(host) {dn*1e3:8.3f} ms {rows/dn/1e6:9.1f} M evals/s", flush=True)
