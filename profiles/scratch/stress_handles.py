import os, sys, gc
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import _synthetic as O  # product generators
from normalizing_flows_amd import _lib, flow as nfa, training
import ctypes as C
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
def free_mem():
    a, b = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(a), C.byref(b))
    return a.value
specs = [("affine", O.make_affine_spec(4, final_std=0.02)), ("rqs", O.make_rqs_spec(3, final_std=0.05)),
         ("gen", O.make_affine_spec(2, event_size=8, hidden=(96, 64), final_std=0.03)),
         ("pad", O.make_affine_spec(4, hidden=(50, 50), final_std=0.02)),          # zero-padded onto the fused kernel
         ("narrow", O.make_affine_spec(2, event_size=6, hidden=(32, 48), final_std=0.03)),  # one-launch backward
         ("wide", O.make_affine_spec(2, event_size=8, hidden=(512, 256), final_std=0.03))]  # split-K GEMMs
x2 = O.make_inputs("banana", 4096)
x8 = np.random.default_rng(0).standard_normal((4096, 8)).astype(np.float32)
x6 = np.random.default_rng(1).standard_normal((4096, 6)).astype(np.float32)
base = None
for it in range(60):
    for kind, spec in specs:
        if kind == "rqs":
            cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=3, hidden=(64, 64), rqs_bins=12, rqs_range_min=-8.0, rqs_range_max=8.0)
        else:
            cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=spec.event_size, num_layers=len(spec.layers), hidden=tuple(w.shape[1] for w, _ in spec.layers[0][:-1]))
        f = nfa.CouplingFlow(cfg, spec.layers)
        x = x8 if kind in ("gen", "wide") else (x6 if kind == "narrow" else x2)
        f.log_prob(x); f.nll(x)
        f.set_option("hidden_matmul", "bf16x3"); f.log_prob(x)
        f.set_option("skip_nonfinite_updates", "1")
        f.train_step(x, training.adam(1e-4))
        f.log_prob(x)
        f.close()
    gc.collect()
    if it == 4: base = free_mem()
    if it % 20 == 19: print("iter", it, "free MB", free_mem() / 2**20, "delta vs iter 4 (MB)", (base - free_mem()) / 2**20, flush=True)
print("ok")
