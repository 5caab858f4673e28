#!/usr/bin/env python3
"""Times the other BASELINE.json configs (device-resident inputs, HIP events on the handle's
stream).  One JSON line per config; these are measurement notes, bench.py holds the contract line.
    python3 profiles/bench_configs.py [a] [c] [d] [b4k] [train] [traina] [trainc]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from normalizing_flows_amd import _lib, flow  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synthetic as O  # product generators; the oracle is not used here

which = sys.argv[1:] or ["a", "b4k", "c", "d", "train"]


def time_log_prob(f, xd, rows, steps=10, warm=2):
    out = flow.DeviceArray((rows,))
    for _ in range(warm):
        f.log_prob(xd, out)
    f.synchronize()
    f.timer_start()
    for _ in range(steps):
        f.log_prob(xd, out)
    ms = f.timer_stop() / steps
    return ms, out


if "a" in which:   # src/real_nvp.py variant, 8 layers, batch 2^20 (and 2^24 to leave the caches)
    params, flips = O.make_a_params(8)
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,),
                          activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
    f = flow.CouplingFlow(cfg, [[(W, b) for W, b in net] for net in params])
    for lg in [int(v) for v in os.environ.get("NF_BENCH_A_LG", "20,24").split(",")]:
        rows = 1 << lg
        x = O.make_inputs("energy2", rows)
        xd = flow.DeviceArray.from_numpy(x)
        ms, _ = time_log_prob(f, xd, rows, steps=20)
        print(json.dumps({"config": f"variant A (MLP 1-4-2) L=8 energy2 batch=2^{lg}", "kernel": f.kernel_name,
                          "ms": ms, "evals_per_s": rows / ms * 1e3, "hbm_GBps_algorithmic": rows * 12 / ms / 1e6,
                          "frac_hbm_6.29TBps": rows * 12 / ms / 1e6 / 6290}))
        del xd

if "b4k" in which:  # configs[0]: 4 layers, banana, batch 4096 (latency-bound on a GPU)
    spec = O.make_affine_spec(4, final_std=0.02)
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
    f = flow.CouplingFlow(cfg, spec.layers)
    x = O.make_inputs("banana", 4096)
    xd = flow.DeviceArray.from_numpy(x)
    for mode in ("f32", "bf16x3"):
        f.set_option("hidden_matmul", mode)
        ms, _ = time_log_prob(f, xd, 4096, steps=200, warm=20)
        print(json.dumps({"config": "variant B L=4 banana batch=4096 (configs[0])", "hidden_matmul": mode,
                          "kernel": f.kernel_name, "ms": ms, "evals_per_s": 4096 / ms * 1e3}))

if "c" in which:   # configs[2]: spline coupling, 8 layers, K=12, [-8,8], gaussian_blobs, batch 2^20
    spec = O.make_rqs_spec(8, final_std=0.05)
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), rqs_bins=12,
                          rqs_range_min=-8.0, rqs_range_max=8.0)
    f = flow.CouplingFlow(cfg, spec.layers)
    rows = 1 << 20
    x = O.make_inputs("gaussian_blobs", rows)
    xd = flow.DeviceArray.from_numpy(x)
    for mode in ("f32", "bf16x3"):
        f.set_option("hidden_matmul", mode)
        ms, _ = time_log_prob(f, xd, rows, steps=5)
        print(json.dumps({"config": "variant C (RQS K=12) L=8 gaussian_blobs batch=2^20 (configs[2])",
                          "hidden_matmul": mode, "tune": os.environ.get("NF_RQS2D_X3_TUNE", "default"),
                          "kernel": f.kernel_name, "ms": ms, "evals_per_s": rows / ms * 1e3,
                          "useful_TFLOPs": rows * 105472 / ms / 1e9,
                          "frac_f32_mfma_peak": rows * 105472 / ms / 1e9 / 157.3}), flush=True)

if "d" in which:   # configs[4]: MNIST shape, 6 layers, batch 2^16
    t0 = time.time()
    spec = O.make_rqs_spec(6, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                           base="uniform01", final_std=0.05)
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=6, hidden=(512, 512, 512), rqs_bins=8,
                          rqs_range_min=0.0, rqs_range_max=1.0, base=_lib.BASE_UNIFORM01)
    f = flow.CouplingFlow(cfg, spec.layers)
    rows = 1 << 16
    x = O.make_inputs("mnist_like", rows)
    xd = flow.DeviceArray.from_numpy(x)
    lp32 = None
    for mode in ("f32", "bf16x3"):
        f.set_option("hidden_matmul", mode)
        ms, out = time_log_prob(f, xd, rows, steps=int(os.environ.get("NF_BENCH_STEPS", 2)), warm=1)
        lp = out.to_numpy()
        if lp32 is None:
            lp32 = lp
        print(json.dumps({"config": "variant D (MNIST 784-D, MLP 512x3, RQS K=8) L=6 batch=2^16 (configs[4])",
                          "hidden_matmul": mode, "kernel": f.kernel_name, "ms": ms, "evals_per_s": rows / ms * 1e3,
                          "useful_TFLOPs": rows * 71.3e6 / ms / 1e9,
                          "frac_f32_mfma_peak": rows * 71.3e6 / ms / 1e9 / 157.3,
                          "finite": bool(np.isfinite(lp).all()),
                          "max_rel_diff_to_f32_mode": float(np.max(np.abs(lp - lp32) / np.maximum(1, np.abs(lp32)))),
                          "setup_s": time.time() - t0}), flush=True)

if "train" in which:   # the reference's `update` step (src/real_nvp_toy_densities.py:164-173), 8 layers
    from normalizing_flows_amd import training
    spec = O.make_affine_spec(8, final_std=0.02)
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64))
    f = flow.CouplingFlow(cfg, spec.layers)
    opt = training.adam(1e-4)
    cases = [(7, "f32"), (12, "f32"), (20, "f32"), (20, "bf16x3")]
    if os.environ.get("NF_TRAIN_LG"):
        cases = [(int(v), "f32") for v in os.environ["NF_TRAIN_LG"].split(",")]
    for lg, mode in cases:
        rows = 1 << lg
        f.set_option("hidden_matmul", mode)
        xd = flow.DeviceArray.from_numpy(O.make_inputs("banana", rows))
        for _ in range(3):
            f.train_step(xd, opt, want_nll=False)
        f.synchronize()
        steps = 20
        t0 = time.perf_counter()
        f.timer_start()
        for _ in range(steps):
            f.train_step(xd, opt, want_nll=False)
        ms = f.timer_stop() / steps
        f.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        # forward 1x + backward 3x the 64x64 products (+ the small layers): 4 * 69,632 FLOP per sample
        print(json.dumps({"config": f"train step (fwd + bwd + Adam) variant B L=8 banana batch=2^{lg}", "hidden_matmul": mode,
                          "ms_stream": ms,
                          "ms_wall": wall, "samples_per_s": rows / ms * 1e3,
                          "TFLOPs_at_4x_forward": rows * 4 * 69632 / ms / 1e9}))

if "traina" in which:   # the SGD step of src/real_nvp.py:295-306 (variant A: MLP 1-4-2, 8 layers)
    from normalizing_flows_amd import training
    params, flips = O.make_a_params(8)
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,),
                          activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
    f = flow.CouplingFlow(cfg, [[(W, b) for W, b in net] for net in params])
    opt = training.sgd(1e-3)
    for lg in [int(v) for v in os.environ.get("NF_TRAINA_LG", "7,17,20,24").split(",")]:
        rows = 1 << lg
        xd = flow.DeviceArray.from_numpy(O.make_inputs("energy2", rows))
        for _ in range(3):
            f.train_step(xd, opt, want_nll=False)
        f.synchronize()
        steps = 20
        f.timer_start()
        for _ in range(steps):
            f.train_step(xd, opt, want_nll=False)
        ms = f.timer_stop() / steps
        f.synchronize()
        print(json.dumps({"config": f"train step (fwd + bwd + SGD) variant A (MLP 1-4-2) L=8 batch=2^{lg}",
                          "ms_stream": ms, "samples_per_s": rows / ms * 1e3}), flush=True)

if "trainc" in which:   # `update` of src/maf_toy_densities.py:167-176 (spline flow, general backward)
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(8, final_std=0.05)
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), rqs_bins=12,
                          rqs_range_min=-8.0, rqs_range_max=8.0)
    f = flow.CouplingFlow(cfg, spec.layers)
    opt = training.adam(1e-4)
    for lg in [int(v) for v in os.environ.get("NF_TRAINC_LG", "7,12,17,20").split(",")]:
        rows = 1 << lg
        xd = flow.DeviceArray.from_numpy(O.make_inputs("gaussian_blobs", rows))
        for _ in range(3):
            f.train_step(xd, opt, want_nll=False)
        f.synchronize()
        steps = 20 if lg < 20 else 5
        t0 = time.perf_counter()
        f.timer_start()
        for _ in range(steps):
            f.train_step(xd, opt, want_nll=False)
        ms = f.timer_stop() / steps
        f.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        print(json.dumps({"config": f"train step (fwd + bwd + Adam) variant C (RQS K=12) L=8 batch=2^{lg}",
                          "ms_stream": ms, "ms_wall": wall, "samples_per_s": rows / ms * 1e3}), flush=True)

