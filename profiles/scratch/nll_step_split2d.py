import os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
from normalizing_flows_amd import _lib, flow, densities
STEPS = int(os.environ.get("STEPS", "50"))
lib=_lib.load(); _lib.require_device()
rng = np.random.default_rng(534)
layers = [[((rng.standard_normal((1, 4)) * 0.1).astype(np.float32), np.zeros(4, np.float32)),
           ((rng.standard_normal((4, 2)) * 0.1).astype(np.float32), np.zeros(2, np.float32))] for _ in range(8)]
cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,), activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
for lg in (20, 22, 24):
    rows = 1 << lg
    xd = next(densities.get_density("energy2")(seed=2123, batch_size=rows, num_batches=1, device=True))
    f = flow.CouplingFlow(cfg, layers); lpd = flow.DeviceArray((rows,))
    for i in range(10): f.nll_enqueue(xd, rows, i % _lib.NF_NLL_SLOTS, lpd)
    f.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); f.timer_start()
        for i in range(STEPS): f.nll_enqueue(xd, rows, i % _lib.NF_NLL_SLOTS, lpd)
        ev = f.timer_stop(); f.synchronize(); w = time.perf_counter() - t0; ev = ev * 50 / STEPS
        best = min(best, w / STEPS)
    print(f"2^{lg}: NLL step {best*1e6:.1f} us wall, {ev/50*1e3:.1f} us stream; nll {f.nll_result((STEPS - 1) % _lib.NF_NLL_SLOTS):.6f}")
    f.close(); xd.free(); lpd.free()
