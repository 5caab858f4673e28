"""Kernel time of the bf16x3 chain kernel per (T, WAVES) instantiation at the headline batch.
usage: python profiles/x3_tune.py   (GPU box)"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one():
    import _synthetic as O  # product generators; the oracle is not used here
    from normalizing_flows_amd import _lib, flow as nfa
    spec = O.make_affine_spec(8, final_std=0.02)
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), first_mask_parity=1)
    f = nfa.CouplingFlow(cfg, spec.layers)
    if os.environ.get("MODE", "bf16x3") != "f32":
        f.set_option("hidden_matmul", "bf16x3")
    rows = int(os.environ.get("ROWS", 1 << 20))
    x = nfa.DeviceArray.from_numpy(O.make_inputs("banana", rows, seed=1))
    out = nfa.DeviceArray((rows,))
    for _ in range(5):
        f.log_prob(x, out)
    f.synchronize()
    best = 1e9
    for rep in range(5):
        f.timer_start()
        for _ in range(20):
            f.log_prob(x, out)
        ms = f.timer_stop() / 20
        best = min(best, ms)
    print(f"{os.environ.get('MODE','bf16x3'):7s} tune={os.environ.get('NF_AFFINE2D_X3_TUNE','default'):6s} rows={rows}: {best*1e3:8.1f} us  "
          f"{rows/best/1e6:8.3f} Gevals/s", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for mode, tune in [("f32", ""), ("bf16x3", "1,8"), ("bf16x3", "2,8"), ("bf16x3", "4,8"), ("bf16x3", "1,16"), ("bf16x3", "1,4"), ("bf16x3", "2,4"), ("bf16x3", "4,4")]:
            env = dict(os.environ, MODE=mode)
            if tune:
                env["NF_AFFINE2D_X3_TUNE"] = tune
            subprocess.run([sys.executable, __file__, "one"], env=env, check=True)
