#!/usr/bin/env python3
"""The reference's src/maf_toy_densities.py (spline coupling flow: RationalQuadraticSpline with 12
bins on [-8, 8], conditioner 2-64-64-2*37, zero-initialised last Linear) on an MI355X: same
functions, same defaults (the module constants the script's main() uses, :32-35), `distrax` /
`haiku` / `jax` / `optax` swapped for normalizing_flows_amd.  Plotting is
left out.

    python examples/maf_toy_densities.py --density gaussian_blobs --training-steps 2000

log_prob / sampling run on the fused spline kernel (nf_rqs2d_h64.hip), the training step on the
general backward (nf_generic_bwd.hip).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as distrax  # noqa: E402
from normalizing_flows_amd import densities, training  # noqa: E402

FLOW_NUM_LAYERS = 8
HIDDEN_SIZE = 64
MLP_NUM_LAYERS = 2
FLOW_NUM_PARAMS = 12  # num_bins


def make_flow_model(event_shape, num_layers, hidden_sizes, flow_num_params, rng):
    """src/maf_toy_densities.py:74-111."""
    mask = np.arange(0, np.prod(event_shape)) % 2
    mask = np.reshape(mask, event_shape).astype(bool)

    def bijector_fn(params):
        return distrax.RationalQuadraticSpline(params, range_min=-8.0, range_max=8.0)

    num_bijector_params = 3 * flow_num_params + 1
    layers = []
    for _ in range(num_layers):
        layers.append(distrax.MaskedCoupling(
            mask=mask, bijector=bijector_fn,
            conditioner=distrax.make_conditioner(event_shape, hidden_sizes, num_bijector_params, per_dim=True,
                                                 rng=rng)))
        mask = np.logical_not(mask)
    flow = distrax.Inverse(distrax.Chain(layers))
    base = distrax.Independent(distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape)),
                               reinterpreted_batch_ndims=len(event_shape))
    return distrax.Transformed(base, flow)


def main(batch_size, learning_rate, training_steps, eval_frequency, density):
    make_dataset = densities.get_density(density)
    optimizer = training.adam(learning_rate)
    model = make_flow_model((2,), FLOW_NUM_LAYERS, [HIDDEN_SIZE] * MLP_NUM_LAYERS, FLOW_NUM_PARAMS,
                            np.random.default_rng(42))
    train_ds = make_dataset(seed=2123, batch_size=batch_size, num_batches=2 * training_steps)
    valid_ds = make_dataset(seed=2235, batch_size=batch_size, num_batches=training_steps)
    # not in the reference script: optax.apply_if_finite's guard.  This data stream holds a 4.4-sigma banana sample at
    # step 1147 whose log_prob overflows float32 under the weights of that moment; without the guard every weight is
    # NaN from there on (as it would be in the reference's float32 arithmetic on the same batch and weights).
    training.apply_if_finite(model)
    for step in range(training_steps):
        training.update(model, optimizer, next(train_ds), want_loss=False)
        if step % eval_frequency == 0:
            train_loss = model.nll(next(train_ds))
            val_loss = model.nll(next(valid_ds))
            print(f"{density} STEP: {step}; training loss: {train_loss}, validation loss: {val_loss}", flush=True)
    training.sync_params(model)
    samples = model.sample(seed=7, sample_shape=(1000000,))
    print("samples", samples.shape, "mean", samples.mean(0), "std", samples.std(0))
    return model


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--batch-size", default=128, type=int)
    parser.add_argument("--learning-rate", default=1e-4, type=float)
    parser.add_argument("--training-steps", default=5000, type=int)
    parser.add_argument("--eval-frequency", default=100, type=int)
    parser.add_argument("--density", default="banana", type=str)
    args = parser.parse_args()
    main(args.batch_size, args.learning_rate, args.training_steps, args.eval_frequency, args.density)
