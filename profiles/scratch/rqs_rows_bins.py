"""General-path spline transform, register form against the run-time-K kernel (NF_RQS_SIMPLE=1) for several bin counts."""
import os, sys
_h = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.join(_h, "..", "..")); sys.path.insert(0, os.path.join(_h, ".."))
import _synthetic as O
from normalizing_flows_amd import _lib, flow as nfa
rows = 1 << 20
for K in (4, 6, 16):
    spec = O.make_rqs_spec(6, event_size=6, hidden=(48, 48), num_bins=K, range_min=-4.0, range_max=4.0, final_std=0.05)
    cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=6, num_layers=6, hidden=(48, 48), rqs_bins=K, rqs_range_min=-4.0, rqs_range_max=4.0)
    f = nfa.CouplingFlow(cfg, spec.layers)
    import numpy as np
    x = nfa.DeviceArray.from_numpy((np.random.default_rng(1).standard_normal((rows, 6)) * 1.5).astype(np.float32))
    out = nfa.DeviceArray((rows,))
    for _ in range(3): f.log_prob(x, out)
    f.synchronize(); f.timer_start()
    for _ in range(10): f.log_prob(x, out)
    print(os.environ.get("NF_RQS_SIMPLE"), "K", K, "ms per 2^20 rows x 6 layers (D=6)", f.timer_stop() / 10, flush=True)
