"""Error of the two hidden-matmul modes against the float64 oracle (development aid).
usage: python tests/tools/x3_error_stats.py   (GPU box).  Lives under tests/: it uses the oracle as the checker."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import flows_np as O
from normalizing_flows_amd import _lib, flow as nfa

def build(spec):
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=len(spec.layers), hidden=(64, 64),
                         first_mask_parity=1)
    return nfa.CouplingFlow(cfg, spec.layers)

for density in ["banana", "energy4", "gaussian_blobs"]:
    for std in (0.02, 0.05):
        spec = O.make_affine_spec(8, final_std=std)
        x = O.make_inputs(density, 1 << 16, seed=3)
        ref = O.log_prob(spec, x)
        e32 = O.log_prob(spec, x, np.float32).astype(np.float64) - ref
        f = build(spec)
        a = f.log_prob(x).astype(np.float64) - ref
        f.set_option("hidden_matmul", "bf16x3")
        b = f.log_prob(x).astype(np.float64) - ref
        scale = np.maximum(1.0, np.abs(ref))
        def st(e):
            return "max %.2e  max/scale %.2e  rms %.2e  mean %+.2e" % (np.abs(e).max(), (np.abs(e) / scale).max(),
                                                                   np.sqrt((e * e).mean()), e.mean())
        print(f"{density:15s} std {std}: |ref|max {np.abs(ref).max():.1f}")
        print("   numpy f32 :", st(e32))
        print("   hip f32   :", st(a))
        print("   hip bf16x3:", st(b))
