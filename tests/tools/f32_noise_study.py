"""How reproducible is a float32 evaluation of a spline flow, sample by sample?  (CPU only; the evidence behind the
per-sample bound of the spline parity tests, tests/test_parity_gpu.py.  Uses the oracle: lives under tests/.)

The NumPy float32 restatement is evaluated on the same inputs with the conditioner's hidden units permuted -- the same
function, another order of the float32 sums.  For an ensemble of n evaluations the bound
max(1e-5 * max(1, |f64|), F * max_n |f32_k - f64|) is formed per sample, and FURTHER float32 evaluations are tested
against it: how often does float32 violate a bound derived from float32?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import flows_np as O  # noqa: E402
import test_parity_gpu as P  # noqa: E402


def study(name, spec, x, draws):
    zr, ldr = O.chain_forward(spec, x)
    ez, el = [], []
    for c in P.permuted_specs(spec, draws):
        z, ld = O.chain_forward(c, x, np.float32)
        ez.append(np.abs(z - zr).max(-1))
        el.append(np.abs(ld - ldr))
    ez, el = np.array(ez), np.array(el)
    tz, tl = 1e-5 * np.maximum(1, np.abs(zr)).max(-1), 1e-5 * np.maximum(1, np.abs(ldr))
    print(f"{name}: {x.shape[0]} rows, {draws} float32 evaluations; outside 1e-5 scaled per evaluation: "
          f"z {(ez > tz).mean() * 100:.2f} %, logdet {(el > tl).mean() * 100:.2f} %")
    for n in (1, 4, 8):
        for fac in (2, 4, 8):
            bz = np.maximum(tz, fac * ez[:n].max(0))
            bl = np.maximum(tl, fac * el[:n].max(0))
            print(f"   bound from {n} evaluation(s) x {fac}: violated by the other {draws - n}: z {(ez[n:] > bz).sum()}/{ez[n:].size}"
                  f"  logdet {(el[n:] > bl).sum()}/{el[n:].size}")


if __name__ == "__main__":
    spec = O.make_rqs_spec(8, final_std=0.05)
    study("configs[2] flow, energy2", spec, O.make_inputs("energy2", 4096), 24)
    study("configs[2] flow, gaussian_blobs", spec, O.make_inputs("gaussian_blobs", 4096), 24)
    spec = O.make_rqs_spec(2, event_size=10, hidden=(128,), num_bins=6, range_min=-5.0, range_max=5.0, final_std=0.08)
    x = (np.random.default_rng(610).standard_normal((777, 10)) * 2.5).astype(np.float32)
    study("general path case (D = 10, 128 hidden, K = 6)", spec, x, 40)
