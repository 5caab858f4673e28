"""Training step of a 2-D flow with THREE hidden layers of 64 (four Linears: the narrow backward keeps 256 accumulator
registers) in both backward forms: python3 profiles/scratch/train_nlin4.py; NF_NO_MLP_SMALL_BWD=1 python3 ..."""
import os, sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_here, "..", ".."))
sys.path.insert(0, os.path.join(_here, ".."))
import _synthetic as O
from normalizing_flows_amd import _lib, flow, training
spec = O.make_affine_spec(8, hidden=(64, 64, 64), final_std=0.02)
cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64, 64))
f = flow.CouplingFlow(cfg, spec.layers)
opt = training.adam(1e-4)
for lg in (7, 12, 17, 20):
    rows = 1 << lg
    xd = flow.DeviceArray.from_numpy(O.make_inputs("banana", rows))
    for _ in range(3):
        f.train_step(xd, opt, want_nll=False)
    f.synchronize()
    f.timer_start()
    for _ in range(10):
        f.train_step(xd, opt, want_nll=False)
    print(os.environ.get("NF_NO_MLP_SMALL_BWD"), "rows 2^%d" % lg, "ms/step", f.timer_stop() / 10, flush=True)
