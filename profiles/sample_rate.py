"""Transformed.sample(1e6) -- the reference's only large-batch use of the chain
(src/real_nvp_toy_densities.py:197-203, src/maf_toy_densities.py) -- through the distrax surface.
python profiles/sample_rate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as distrax

def model(kind, rng):
    mask = (np.arange(2) % 2).astype(bool)
    layers = []
    for _ in range(8):
        if kind == "affine":
            fn = lambda p: distrax.ScalarAffine(shift=np.split(p, 2, axis=-1)[0], log_scale=np.split(p, 2, axis=-1)[1])
            cond = distrax.make_conditioner((2,), [64, 64], 2, rng=rng, final_std=0.02)
        else:
            fn = lambda p: distrax.RationalQuadraticSpline(p, range_min=-8.0, range_max=8.0)
            cond = distrax.make_conditioner((2,), [64, 64], 37, per_dim=True, rng=rng, final_std=0.05)
        layers.append(distrax.MaskedCoupling(mask=mask, bijector=fn, conditioner=cond))
        mask = np.logical_not(mask)
    base = distrax.Independent(distrax.Normal(loc=np.zeros(2), scale=np.ones(2)), reinterpreted_batch_ndims=1)
    return distrax.Transformed(base, distrax.Inverse(distrax.Chain(layers)))

for kind in ("affine", "rqs"):
    m = model(kind, np.random.default_rng(1))
    for _ in range(3):
        s = m.sample(seed=3, sample_shape=(1000000,))
    t0 = time.perf_counter()
    for i in range(10):
        s = m.sample(seed=i, sample_shape=(1000000,))
    dt = (time.perf_counter() - t0) / 10
    print(f"{kind}: sample(1e6) {dt*1e3:.2f} ms  ({1e6/dt/1e6:.0f} M samples/s, host array out)  mean {s.mean(0)}")
