"""log_prob rate of 2-D flows whose conditioner is NOT the fused kernels' 64/64: as shipped (two hidden layers up to
64 wide are zero-padded onto the fused kernels by nf_flow_create), on the general path with one launch per
conditioner (NF_NO_PAD=1: nf_mlp_small.hip) and with the GEMM chain (+ NF_NO_MLP_SMALL=1).
python profiles/narrow_flows.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

def one():
    import _synthetic as O
    from normalizing_flows_amd import _lib, flow as nfa
    rows = 1 << 20
    x = nfa.DeviceArray.from_numpy(O.make_inputs("banana", rows, seed=1))
    out = nfa.DeviceArray((rows,))
    for name, hidden, kind in [("affine 50/50", (50, 50), "affine"), ("affine 32/32", (32, 32), "affine"),
                               ("spline K=4 50/50", (50, 50), "rqs"), ("affine 64/64/64", (64, 64, 64), "affine")]:
        if kind == "affine":
            spec = O.make_affine_spec(8, hidden=hidden, final_std=0.02)
            cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=hidden)
        else:
            spec = O.make_rqs_spec(8, hidden=hidden, num_bins=4, final_std=0.05)
            cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=hidden, rqs_bins=4,
                                 rqs_range_min=-8.0, rqs_range_max=8.0)
        f = nfa.CouplingFlow(cfg, spec.layers)
        for _ in range(3):
            f.log_prob(x, out)
        f.synchronize()
        f.timer_start()
        for _ in range(10):
            f.log_prob(x, out)
        ms = f.timer_stop() / 10
        label = "GEMM chain" if os.environ.get("NF_NO_MLP_SMALL") else ("one launch" if os.environ.get("NF_NO_PAD") else "as shipped")
        print(f"{label:11s} {name:18s} {f.kernel_name}: "
              f"{ms:7.3f} ms per 2^20 rows x 8 layers = {rows / ms / 1e3:7.1f} M evals/s", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for env in ({}, {"NF_NO_PAD": "1"}, {"NF_NO_PAD": "1", "NF_NO_MLP_SMALL": "1"}):
            subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, **env), check=True)
