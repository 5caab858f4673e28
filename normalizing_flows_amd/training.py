"""The training step of the reference's scripts on the GPU.

`update` in src/real_nvp_toy_densities.py:164-173 is
`grads = jax.grad(loss_fn)(params, batch); updates = optax.adam(lr).update(grads, state);
params = optax.apply_updates(params, updates)`.  Here the whole step -- forward, backward walk,
gradient all-reduce when sharded, Adam, refresh of the kernels' weight images -- is one call into
libnf_hip (`nf_train_step`); parameters stay on the device between steps and are copied back into
the model's conditioners on request.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import _lib
from .bijectors import lower_for_log_prob
from .conditioner import layers_to_haiku_params
from .distributions import Transformed


def adam(learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> _lib.Optimizer:
    """optax.adam(learning_rate) with optax's defaults (src/real_nvp_toy_densities.py:162)."""
    return _lib.Optimizer(kind=_lib.OPT_ADAM, learning_rate=learning_rate, beta1=b1, beta2=b2, eps=eps)


def sgd(learning_rate: float) -> _lib.Optimizer:
    """p - learning_rate * g (src/real_nvp.py:298)."""
    return _lib.Optimizer(kind=_lib.OPT_SGD, learning_rate=learning_rate, beta1=0.0, beta2=0.0, eps=0.0)


def apply_if_finite(model: Transformed, enable: bool = True) -> None:
    """optax.apply_if_finite around the optimiser: a step whose gradient holds a NaN or an infinity (one far outlier
    in a batch is enough for an unconstrained exp(log_scale) flow in float32) leaves weights and moments untouched
    instead of turning every weight into NaN.  The reference's scripts do not wrap their optimiser; off by default."""
    model.set_option("skip_nonfinite_updates", "1" if enable else "0")


def _flow_of(model: Transformed):
    flow, direction = model._lowered()
    if direction != _lib.DIR_CHAIN_FORWARD:
        raise NotImplementedError("training is wired for Transformed(base, Inverse(Chain(...)))")
    return flow


def value_and_grad(model: Transformed, batch) -> Tuple[float, Dict[str, Dict[str, np.ndarray]]]:
    """(loss, grads) = jax.value_and_grad(loss_fn)(params, batch); grads keyed like the haiku
    parameter dict ('mlp/~/linear_0', ..., 'linear', 'mlp_1/~/linear_0', ...)."""
    flow = _flow_of(model)
    rows = np.asarray(batch).reshape(-1, flow.config.event_dim)
    nll, grads = flow.nll_grad(rows)
    return nll, layers_to_haiku_params(_unpermute(model, grads))


def update(model: Transformed, optimizer: _lib.Optimizer, batch, want_loss: bool = True):
    """One `update` step; returns the loss before the step (or None)."""
    flow = _flow_of(model)
    rows = batch if hasattr(batch, "ptr") else np.asarray(batch).reshape(-1, flow.config.event_dim)
    return flow.train_step(rows, optimizer, want_nll=want_loss)


def sync_params(model: Transformed) -> None:
    """Copy the device-resident parameters back into the model's conditioner objects."""
    flow = _flow_of(model)
    layers = _unpermute(model, flow.get_weights())
    chain, _ = lower_for_log_prob(model.bijector)
    for coupling, lin in zip(chain.bijectors, layers):
        cond = coupling.conditioner
        cond.linears = [(np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32)) for W, b in lin]
        # keep the cached handle: its weights are already these
    chain._flows[flow.config.base] = (tuple((id(b.conditioner), b.conditioner.version) for b in chain.bijectors), flow)


def _unpermute(model: Transformed, layers: List[List[Tuple[np.ndarray, np.ndarray]]]):
    """Undo the (shift, log_scale) column swap a bijector_fn may have asked for."""
    chain, _ = lower_for_log_prob(model.bijector)
    out = []
    for coupling, lin in zip(chain.bijectors, layers):
        lin = list(lin)
        if coupling._perm == (1, 0):
            W, b = lin[-1]
            lin[-1] = (np.ascontiguousarray(W[:, ::-1]), np.ascontiguousarray(b[::-1]))
        out.append(lin)
    return out
