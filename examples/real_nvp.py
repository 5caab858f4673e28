#!/usr/bin/env python3
"""The training loop of the reference's hand-rolled src/real_nvp.py (:286-315) on an MI355X."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from normalizing_flows_amd import densities  # noqa: E402
from normalizing_flows_amd import real_nvp as R  # noqa: E402

DENSITY = "energy2"

if __name__ == "__main__":
    training_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5001
    learning_rate = 1e-2
    params, fns_config = R.init_nvp_chain(n=8)
    train_ds = densities.get_density(DENSITY)(seed=231, batch_size=128, num_batches=training_steps)
    for step in range(training_steps):
        nll, params = R.sgd_update(params, fns_config, next(train_ds), learning_rate)
        if step % 1000 == 0:
            print(f"Step: {step}; NLL: {nll}")
    y = R.sample_nvp_chain(params, fns_config, R.sample_N01, 1000000, 7809)
    print("samples", y.shape, "mean", y.mean(0), "std", y.std(0))
