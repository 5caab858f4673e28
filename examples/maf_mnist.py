#!/usr/bin/env python3
"""The reference's src/maf_mnist.py (spline coupling flow on 28x28x1 images: RationalQuadraticSpline
with 8 bins on [0, 1], conditioner 784-512-512-512-19600, Uniform(0, 1) base, 8 layers) on an MI355X:
same functions and module constants (:25-31), `distrax` / `haiku` / `jax` / `optax` swapped for
normalizing_flows_amd.

The reference loads MNIST through tensorflow_datasets (a download); this container has neither the
package nor a network, so `load_dataset` below yields SYNTHETIC uint8 images of the same shape and
dtype (blurred strokes on black) unless --npz points at a file with an `image` array
[N, 28, 28, 1] uint8.  Plotting of the samples is left out.

    python examples/maf_mnist.py --training-steps 200 --eval-frequency 50
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as distrax  # noqa: E402
from normalizing_flows_amd import prepare_data, training  # noqa: E402

MNIST_IMAGE_SHAPE = (28, 28, 1)
FLOW_NUM_LAYERS = 8
HIDDEN_SIZE = 512
MLP_NUM_LAYERS = 3
FLOW_NUM_PARAMS = 8


def make_flow_model(event_shape, num_layers, hidden_sizes, flow_num_params, rng):
    """src/maf_mnist.py:52-89."""
    mask = np.arange(0, np.prod(event_shape)) % 2
    mask = np.reshape(mask, event_shape).astype(bool)

    def bijector_fn(params):
        return distrax.RationalQuadraticSpline(params, range_min=0.0, range_max=1.0)

    num_bijector_params = 3 * flow_num_params + 1
    layers = []
    for _ in range(num_layers):
        layers.append(distrax.MaskedCoupling(
            mask=mask, bijector=bijector_fn,
            conditioner=distrax.make_conditioner(event_shape, hidden_sizes, num_bijector_params, per_dim=True,
                                                 rng=rng)))
        mask = np.logical_not(mask)
    flow = distrax.Inverse(distrax.Chain(layers))
    base = distrax.Independent(distrax.Uniform(low=np.zeros(event_shape), high=np.ones(event_shape)),
                               reinterpreted_batch_ndims=len(event_shape))
    return distrax.Transformed(base, flow)


def load_dataset(seed, batch_size, npz=None):
    """Stand-in for src/maf_mnist.py:92-98 (tfds.load("mnist")): an endless iterator of {"image": uint8}."""
    rng = np.random.default_rng(seed)
    if npz:
        images = np.load(npz)["image"].astype(np.uint8).reshape(-1, *MNIST_IMAGE_SHAPE)
        while True:
            yield {"image": images[rng.integers(0, len(images), batch_size)]}
    yy, xx = np.mgrid[0:28, 0:28]
    while True:
        img = np.zeros((batch_size, 28, 28), np.float32)
        for b in range(batch_size):
            for _ in range(3):   # a few soft strokes
                cx, cy, r = rng.uniform(6, 22), rng.uniform(6, 22), rng.uniform(2.5, 6)
                img[b] += np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))
        img = np.clip(img / max(img.max(), 1e-6) * 255.0, 0, 255).astype(np.uint8)
        yield {"image": img[..., None]}


def main(batch_size, learning_rate, training_steps, eval_frequency, npz=None):
    optimizer = training.adam(learning_rate)
    model = make_flow_model(MNIST_IMAGE_SHAPE, FLOW_NUM_LAYERS, [HIDDEN_SIZE] * MLP_NUM_LAYERS, FLOW_NUM_PARAMS,
                            np.random.default_rng(42))
    train_ds = load_dataset(1, batch_size, npz)
    valid_ds = load_dataset(2, batch_size, npz)
    for step in range(training_steps):
        # loss_fn dequantises with a fresh key per step (src/maf_mnist.py:133-138), eval_fn does not (:141-146)
        training.update(model, optimizer, prepare_data(next(train_ds), prng_key=step + 1), want_loss=False)
        if step % eval_frequency == 0:
            val_loss = model.nll(prepare_data(next(valid_ds)))
            print(f"STEP: {step}; validation loss: {val_loss}", flush=True)
    training.sync_params(model)
    samples = model.sample(seed=214, sample_shape=(16,))
    print("samples", samples.shape, "min", float(samples.min()), "max", float(samples.max()))
    return model


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--batch-size", default=128, type=int)
    parser.add_argument("--learning-rate", default=1e-4, type=float)
    parser.add_argument("--training-steps", default=5000, type=int)
    parser.add_argument("--eval-frequency", default=100, type=int)
    parser.add_argument("--npz", default=None, help="optional .npz with an `image` array [N, 28, 28, 1] uint8")
    args = parser.parse_args()
    main(args.batch_size, args.learning_rate, args.training_steps, args.eval_frequency, args.npz)
