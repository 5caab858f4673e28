"""Debug probe: the fused spline epilogue against the oracle and the unfused path on one small case."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flows_np as O
from normalizing_flows_amd import _lib, flow as nfa
from test_parity_gpu import rqs_flow

K, D, hidden, L, rows, parity = 8, 64, (256,), 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 1
spec = O.make_rqs_spec(L, event_size=D, hidden=hidden, num_bins=K, range_min=-5.0, range_max=5.0,
                       final_std=float(sys.argv[2]) if len(sys.argv) > 2 else 0.08, first_parity=parity)
x = (np.random.default_rng(K * 100 + D).standard_normal((rows, D)) * 2.5).astype(np.float32)
x[0, 0] = 7.5
x[-1, D - 1] = -5.0
for no_k5 in (0, 1):
    if no_k5:
        os.environ["NF_NO_K5"] = "1"
    f = rqs_flow(nfa, spec, parity)
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    zr, ldr = O.chain_forward(spec, x)
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    xr, ldr2 = O.chain_inverse(spec, z)
    x32, _ = O.chain_inverse(spec, z, np.float32)
    e = np.abs(xb - xr)
    j = np.unravel_index(np.argmax(e), e.shape)
    print("no_k5", no_k5, "fwd err", np.abs(z - zr).max(), "inv err", e.max(), "at", j, "x", x[j], "z", z[j], "gpu", xb[j],
          "ref", xr[j], "np32", x32[j], "ld", ldb[j[0]], ldr2[j[0]])
