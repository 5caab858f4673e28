#!/usr/bin/env python3
"""NLL step (chain kernel + block-sum kernel + scalar to the host ring) at BASELINE configs[0]: 4 layers, 4096 rows.
Median of 20 blocks of 200 steps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from normalizing_flows_amd import _lib, flow  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synthetic as O  # noqa: E402

spec = O.make_affine_spec(4, final_std=0.02)
cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
f = flow.CouplingFlow(cfg, spec.layers)
rows = 4096
xd = flow.DeviceArray.from_numpy(O.make_inputs("banana", rows))
lpd = flow.DeviceArray((rows,))
for i in range(50):
    f.nll_enqueue(xd, rows, i % _lib.NF_NLL_SLOTS, lpd)
f.synchronize()
walls = []
for b in range(20):
    t0 = time.perf_counter()
    for i in range(200):
        f.nll_enqueue(xd, rows, i % _lib.NF_NLL_SLOTS, lpd)
    f.synchronize()
    walls.append((time.perf_counter() - t0) / 200 * 1e6)
print(f"configs[0] NLL step: {np.median(walls):.2f} us per step (min {min(walls):.2f}), nll {f.nll_result(199 % _lib.NF_NLL_SLOTS):.9f}")
