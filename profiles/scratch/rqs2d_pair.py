#!/usr/bin/env python3
"""A/B of the fused spline kernel's geometries (NF_RQS2D_TUNE = "T,WAVES"): bit-equality of log_prob / forward /
inverse against the default geometry and time per launch at 2^20 rows (configs[2])."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from normalizing_flows_amd import _lib, flow  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synthetic as O  # noqa: E402

spec = O.make_rqs_spec(8, final_std=0.05)
cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), rqs_bins=12,
                      rqs_range_min=-8.0, rqs_range_max=8.0)
f = flow.CouplingFlow(cfg, spec.layers)
rows = int(os.environ.get("ROWS", 1 << 20))
ref = {}
for density in ("gaussian_blobs", "banana"):
    x = O.make_inputs(density, rows)
    xd = flow.DeviceArray.from_numpy(x)
    out = flow.DeviceArray((rows,))
    for tune in ("1,16", "2,8", "2,12"):
        os.environ["NF_RQS2D_TUNE"] = tune
        for _ in range(3):
            f.log_prob(xd, out)
        f.synchronize()
        best = []
        for rep in range(5):
            f.timer_start()
            for _ in range(20):
                f.log_prob(xd, out)
            best.append(f.timer_stop() / 20)
        lp = out.to_numpy()
        y, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x[: (1 << 16) - 37])
        xi, ldi = f.chain_apply(_lib.DIR_CHAIN_INVERSE, y)
        key = density
        if tune == "1,16":
            ref[key] = (lp, y, ld, xi, ldi)
        r = ref[key]
        same = [bool(np.array_equal(a, b)) for a, b in zip(r, (lp, y, ld, xi, ldi))]
        print(json.dumps({"density": density, "tune": tune, "ms_median": float(np.median(best)), "ms_min": float(min(best)),
                          "frac": rows * 105472 / float(np.median(best)) / 1e9 / 157.3,
                          "bit_equal_lp_y_ld_xi_ldi": same}), flush=True)
