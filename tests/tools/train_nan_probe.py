"""Debug probe: first training step of the affine example whose batch, loss or weights are not finite."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "examples"))
import real_nvp_toy_densities as ex
from normalizing_flows_amd import training

steps = 1500
make_dataset = ex.densities.get_density("banana")
opt = training.adam(1e-4)
model = ex.make_flow_model((2,), ex.FLOW_NUM_LAYERS, [ex.HIDDEN_SIZE] * ex.MLP_NUM_LAYERS, np.random.default_rng(42))
ds = make_dataset(seed=2123, batch_size=128, num_batches=2 * steps)
flow = model._lowered()[0]
for step in range(steps):
    batch = next(ds)
    b = np.asarray(batch.to_numpy() if hasattr(batch, "to_numpy") else batch)
    loss = training.update(model, opt, batch, want_loss=True)
    if not np.isfinite(b).all() or not np.isfinite(loss) or step % 250 == 0:
        print("step", step, "batch finite", bool(np.isfinite(b).all()), "max |x|", float(np.abs(b).max()), "loss", loss, flush=True)
    if not np.isfinite(loss):
        lp = model.log_prob(b)
        bad = np.where(~np.isfinite(lp))[0]
        print("rows with non-finite log_prob", bad.tolist()[:10], b[bad][:5].tolist())
        break
