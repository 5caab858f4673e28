"""Training-step latency of the MNIST-shaped flow (784-D, MLP 512x3, K = 8, 8 layers, batch 128: the reference's own
training configuration, src/maf_mnist.py:34-89,159-171,228-230), device-resident rows.  python profiles/mnist_train_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import _synthetic as O
from normalizing_flows_amd import _lib, flow as nfa, training

spec = O.make_rqs_spec(8, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                       base="uniform01", final_std=0.01)
cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=8, hidden=(512, 512, 512), rqs_bins=8,
                     rqs_range_min=0.0, rqs_range_max=1.0, base=_lib.BASE_UNIFORM01)
f = nfa.CouplingFlow(cfg, spec.layers)
opt = training.adam(1e-4)
for rows in [int(v) for v in os.environ.get("NF_MNIST_ROWS", "128").split(",")]:
    x = nfa.DeviceArray.from_numpy((np.random.default_rng(1).integers(0, 256, (rows, 784)) / 256.0).astype(np.float32))
    for _ in range(3):
        f.train_step(x, opt, want_nll=False)
    f.synchronize()
    f.timer_start()
    for _ in range(20):
        f.train_step(x, opt, want_nll=False)
    ms = f.timer_stop() / 20
    print(f"rows {rows:6d}: {ms * 1e3:8.1f} us per training step", flush=True)
