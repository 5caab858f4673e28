"""Debug probe: non-finite samples of the affine example after a short training run, padded vs general path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "examples"))
import real_nvp_toy_densities as ex
from normalizing_flows_amd import training

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
make_dataset = ex.densities.get_density("banana")
opt = training.adam(1e-4)
model = ex.make_flow_model((2,), ex.FLOW_NUM_LAYERS, [ex.HIDDEN_SIZE] * ex.MLP_NUM_LAYERS, np.random.default_rng(42))
ds = make_dataset(seed=2123, batch_size=128, num_batches=2 * steps)
for step in range(steps):
    training.update(model, opt, next(ds), want_loss=False)
training.sync_params(model)
s = model.sample(seed=7, sample_shape=(1000000,))
bad = ~np.isfinite(s).all(axis=1)
print("NF_NO_PAD", os.environ.get("NF_NO_PAD"), "steps", steps, "non-finite samples", int(bad.sum()), flush=True)
x = next(ds)
lp = model.log_prob(x)
print("log_prob of data: non-finite", int((~np.isfinite(lp)).sum()), "mean", np.nanmean(lp), flush=True)
w = model._lowered()[0].get_weights()
if w is not None:
    flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in w for W, b in layer])
    print("weights: non-finite", int((~np.isfinite(flat)).sum()), "max |w|", np.nanmax(np.abs(flat)), flush=True)
print("first samples", s[:3].tolist())
