import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import flows_np as O
from normalizing_flows_amd import _lib, flow as nfa
np.set_printoptions(linewidth=220, precision=5, suppress=True)
def rqs_flow(spec, parity):
    cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=spec.event_size, num_layers=len(spec.layers),
                         hidden=tuple(w.shape[1] for w, _ in spec.layers[0][:-1]), first_mask_parity=parity,
                         rqs_bins=spec.num_bins, rqs_range_min=spec.range_min, rqs_range_max=spec.range_max)
    return nfa.CouplingFlow(cfg, spec.layers)
D, hidden, K, L, parity, lo, hi = 5, (24,), 8, 2, 1, -4.0, 4.0
spec = O.make_rqs_spec(L, event_size=D, hidden=hidden, num_bins=K, range_min=lo, range_max=hi, final_std=0.2, first_parity=parity)
f = rqs_flow(spec, parity)
x = np.random.default_rng(D + K).standard_normal((1500, D)).astype(np.float32) * 2.0
for rows in (1500, 200, 50):
    xs = x[:rows]
    nll, g = f.nll_grad(xs)
    rn, rg = O.flow_nll_and_grad(spec, xs)
    print("rows", rows, "nll", nll, rn)
    for l in range(L):
        for j in range(len(spec.layers[l])):
            for which in (0, 1):
                e = np.abs(g[l][j][which] - rg[l][j][which]); s = np.abs(rg[l][j][which]).max()
                print(f"  l{l} lin{j} {'Wb'[which]} err {e.max():.3e} scale {s:.3e} rel {e.max()/max(s,1e-9):.2e} argmax {np.unravel_index(e.argmax(), e.shape)}")
# per-sample: which single row produces the discrepancy?  gradient of one-row batches
bad = []
for r in range(60):
    xs = x[r:r+1]
    nll, g = f.nll_grad(xs)
    rn, rg = O.flow_nll_and_grad(spec, xs)
    e = max(np.abs(g[l][j][w] - rg[l][j][w]).max() / max(np.abs(rg[l][j][w]).max(), 1e-9) for l in range(L) for j in range(2) for w in (0, 1))
    if e > 1e-3: bad.append((r, e, xs))
print("single-row batches with rel err > 1e-3:", [(r, float(e)) for r, e, _ in bad])
for r, e, xs in bad[:3]:
    print(r, xs, "z", O.chain_forward(spec, xs)[0])
