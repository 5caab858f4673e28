"""Synthetic weights and inputs for the measurement scripts in this directory, made with the PRODUCT's
own generators (normalizing_flows_amd.make_conditioner, densities.sample_density) -- the oracle is test
infrastructure and is not imported by anything here.  The objects mimic what the scripts need of a spec:
`.layers` (per layer [(W, b), ...])."""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from normalizing_flows_amd import densities  # noqa: E402
from normalizing_flows_amd.conditioner import make_conditioner  # noqa: E402


def make_affine_spec(num_layers=8, event_size=2, hidden=(64, 64), final_std=0.05, first_parity=1, seed=534):
    layers = [make_conditioner((event_size,), list(hidden), 2, rng=np.random.default_rng(seed + l),
                               final_std=final_std).linears for l in range(num_layers)]
    return SimpleNamespace(layers=layers, event_size=event_size)


def make_rqs_spec(num_layers=8, event_size=2, hidden=(64, 64), num_bins=12, range_min=-8.0, range_max=8.0,
                  base="normal", final_std=0.05, first_parity=1, seed=534):
    layers = [make_conditioner((event_size,), list(hidden), 3 * num_bins + 1, per_dim=True,
                               rng=np.random.default_rng(seed + l), final_std=final_std).linears
              for l in range(num_layers)]
    return SimpleNamespace(layers=layers, event_size=event_size, num_bins=num_bins, range_min=range_min,
                           range_max=range_max, base=base)


def make_a_params(num_layers=8, sizes=(1, 4, 2), seed=534):
    """src/real_nvp.py:187-194: W ~ 0.1 N(0, 1), b = 0; returns (params, flips)."""
    params, flips = [], []
    for l in range(num_layers):
        rng = np.random.default_rng(seed + l)
        params.append([[(rng.standard_normal((m, n)) * 0.1).astype(np.float32), np.zeros(n, np.float32)]
                       for m, n in zip(sizes[:-1], sizes[1:])])
        flips.append(bool(l & 1))
    return params, flips


def make_inputs(density, rows, seed=2123):
    if density == "mnist_like":   # k / 256 pixels, as prepare_data's evaluation path yields
        return (np.random.default_rng(seed).integers(0, 256, (rows, 784)) / 256.0).astype(np.float32)
    return np.asarray(densities.sample_density(density, seed, rows, False), np.float32)
