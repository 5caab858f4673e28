"""End-to-end latency of the drop-in surface at the reference's batch size (128 rows, NumPy in / out):
what one iteration of the scripts' loops costs through Python + ctypes + the library.
python profiles/api_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as distrax
from normalizing_flows_amd import densities, training

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

def timeit(fn, n=300):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6

for kind in ("affine", "rqs"):
    m = model(kind, np.random.default_rng(1))
    batch = next(densities.make_dataset_banana(seed=1, batch_size=128, num_batches=1))
    opt = training.adam(1e-4)
    print(f"{kind:6s} batch 128:  log_prob {timeit(lambda: m.log_prob(batch)):7.1f} us   nll {timeit(lambda: m.nll(batch)):7.1f} us   "
          f"update {timeit(lambda: training.update(m, opt, batch, want_loss=False)):7.1f} us   "
          f"update+loss {timeit(lambda: training.update(m, opt, batch)):7.1f} us   "
          f"sample(128) {timeit(lambda: m.sample(seed=3, sample_shape=(128,))):7.1f} us", flush=True)
