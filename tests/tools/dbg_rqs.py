import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import flows_np as O
from normalizing_flows_amd import _lib, flow as nfa
K, L, parity, rows, lo, hi = 12, 1, 0, 33, -8.0, 8.0
spec = O.make_rqs_spec(L, num_bins=K, range_min=lo, range_max=hi, final_std=0.05, first_parity=parity)
cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=L, hidden=(64, 64), first_mask_parity=parity,
                     rqs_bins=K, rqs_range_min=lo, rqs_range_max=hi)
f = nfa.CouplingFlow(cfg, spec.layers)
x = O.make_inputs("banana", rows, seed=K * 100 + L)
zr, ldr = O.chain_forward(spec, x)
for mode in ("f32", "bf16x3"):
    f.set_option("hidden_matmul", mode)
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    xr, ldr2 = O.chain_inverse(spec, z)
    x32, l32 = O.chain_inverse(spec, z, np.float32)
    print(mode, "fwd err", np.abs(z - zr).max(), "inv err", np.abs(xb - xr).max(), "np32 inv err", np.abs(x32 - xr).max(),
          "roundtrip", np.abs(xb - x).max())
    bad = np.argmax(np.abs(xb - xr).max(axis=1))
    print("  worst row", bad, "x", x[bad], "z", z[bad], "xb", xb[bad], "xr", xr[bad], "x32", x32[bad], "ld", ldb[bad], ldr2[bad])
