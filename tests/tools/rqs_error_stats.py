"""Scaled error of the spline kernels against the float64 oracle, no floor (development aid + the numbers
BASELINE.md quotes).  usage: python tests/tools/rqs_error_stats.py [rows]   (GPU box).
Lives under tests/: it uses the oracle as the checker.

For BASELINE.json configs[2] (8 spline layers, K = 12, [-8, 8]) on three densities, and for every path a spline
flow can take (fused 2-D kernel; general path with the spline in the last GEMM's epilogue; general path with the
separate transform kernel), prints max |hip - f64| / max(1, |f64|) for log_prob, z, logdet, the inverse walk and
its logdet, beside the NumPy float32 restatement's own figure, and how many samples sit outside
1e-5 * max(1, |f64|)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import flows_np as O
from normalizing_flows_amd import _lib, flow as nfa


def build(spec, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=spec.event_size, num_layers=len(spec.layers),
                             hidden=tuple(w.shape[1] for w, _ in spec.layers[0][:-1]), first_mask_parity=1,
                             base=_lib.BASE_STD_NORMAL if spec.base == "normal" else _lib.BASE_UNIFORM01,
                             rqs_bins=spec.num_bins, rqs_range_min=spec.range_min, rqs_range_max=spec.range_max)
        return nfa.CouplingFlow(cfg, spec.layers)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def scaled(got, ref):
    got = np.asarray(got, np.float64)
    e = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    if e.ndim > 1:
        e = e.max(axis=-1)
    return e


def line(name, e):
    return f"   {name:22s} max {e.max():.2e}  p99.9 {np.quantile(e, 0.999):.2e}  rms {np.sqrt((e * e).mean()):.2e}  " \
           f"outside 1e-5: {int((e > 1e-5).sum())}/{e.size}"


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    paths = [("fused rqs2d", {}), ("general + K5 epilogue", {"NF_FORCE_GENERIC": "1"}),
             ("general + rows kernel", {"NF_FORCE_GENERIC": "1", "NF_NO_K5": "1"})]
    worst = {}
    for density in ["gaussian_blobs", "banana", "energy2"]:
        spec = O.make_rqs_spec(8, final_std=0.05)
        x = O.make_inputs(density, rows)
        ref = O.log_prob(spec, x)
        zr, ldr = O.chain_forward(spec, x)
        lp32 = O.log_prob(spec, x, np.float32)
        z32, ld32 = O.chain_forward(spec, x, np.float32)
        print(f"{density}: rows {rows}, |log_prob| max {np.abs(ref).max():.1f}")
        print(line("numpy f32 log_prob", scaled(lp32, ref)))
        print(line("numpy f32 z", scaled(z32, zr)))
        print(line("numpy f32 logdet", scaled(ld32, ldr)))
        zf = None
        for name, env in paths:
            f = build(spec, env)
            lp = f.log_prob(x)
            z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
            if zf is None:
                zf = z
            xr, ldir = O.chain_inverse(spec, zf)
            xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, zf)
            print(f"  {name} ({f.kernel_name})")
            for what, e in (("log_prob", scaled(lp, ref)), ("z", scaled(z, zr)), ("logdet", scaled(ld, ldr)),
                            ("inverse x", scaled(xb, xr)), ("inverse logdet", scaled(ldb, ldir))):
                print(line(what, e))
                key = (name, what)
                worst[key] = max(worst.get(key, 0.0), float(e.max()))
            # per-sample bound the tests use: max(1e-5 * scale, 2 * |f32_i - f64_i|)
            def outside(got, r64, r32):
                e = np.abs(np.asarray(got, np.float64) - r64)
                bound = np.maximum(1e-5 * np.maximum(1.0, np.abs(r64)), 2 * np.abs(np.asarray(r32, np.float64) - r64))
                return int((e > bound).sum())
            x32, ldi32 = O.chain_inverse(spec, zf, np.float32)
            zb, ldf = O.chain_forward(spec, xb)
            bwd = (np.abs(zb - zf) <= 1e-5 * np.maximum(1, np.abs(zf))).all(-1) & \
                  (np.abs(ldb + ldf) <= 1e-5 * np.maximum(1, np.abs(ldf)))
            fwd = (np.abs(xb - xr) <= np.maximum(1e-5 * np.maximum(1, np.abs(xr)), 2 * np.abs(x32 - xr))).all(-1) & \
                  (np.abs(ldb - ldir) <= np.maximum(1e-5 * np.maximum(1, np.abs(ldir)), 2 * np.abs(ldi32 - ldir)))
            print(f"   outside the per-sample bound: log_prob {outside(lp, ref, lp32)}, z {outside(z, zr, z32)}, "
                  f"logdet {outside(ld, ldr, ld32)}; inverse rows failing forward AND backward bound: "
                  f"{int((~(fwd | bwd)).sum())} (forward alone {int((~fwd).sum())})")
    print("worst over the three densities:")
    for (name, what), v in worst.items():
        print(f"   {name:24s} {what:16s} {v:.2e}")


if __name__ == "__main__":
    main()
