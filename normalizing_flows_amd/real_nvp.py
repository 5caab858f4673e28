"""Free-function RealNVP API of the reference's hand-rolled script (src/real_nvp.py).

Same names, same argument meaning, same parameter structure
(`params[layer][k] = [W, b]`, `fns_config[layer] = (fn, flip)`), with every
evaluation lowered to the fused split-coupling kernel of libnf_hip
(`NF_KIND_AFFINE_SPLIT`).  NumPy arrays in, NumPy arrays out; no CPU fallback.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import _lib
from .distributions import MultivariateNormalDiag
from .flow import CouplingFlow, FlowConfig

NetParams = List[List[np.ndarray]]  # [[W, b], ..., [W_last, b_last]]

_N01 = MultivariateNormalDiag(loc=np.zeros(2, np.float32), scale_diag=np.ones(2, np.float32))


def sample_N01(N: int, seed) -> np.ndarray:
    """src/real_nvp.py:42-54 (Philox4x32-10 on the GPU instead of JAX threefry)."""
    return _N01.sample(seed=seed, sample_shape=(N,))


def log_prob_N01(x) -> np.ndarray:
    """src/real_nvp.py:57-68, evaluated on the GPU."""
    return _N01.log_prob(x)


def shift_and_log_scale_fn(net_params: NetParams, x) -> Tuple[np.ndarray, np.ndarray]:
    """src/real_nvp.py:161-171: MLP with leaky_relu, output split into (shift, log_scale).

    Evaluated on the GPU through the coupling kernel: with the transformed half set
    to zero, forward_nvp returns y2 = 0 * exp(log_scale) + shift = shift and the
    layer's log-det is log_scale (event_dim 2 only).
    """
    x = np.ascontiguousarray(x, np.float32)
    if x.ndim != 2 or x.shape[1] != 1:
        raise NotImplementedError("stand-alone conditioner evaluation is wired for the 2-D flow ([rows, 1] input)")
    flow = _chain_for([net_params], [False])
    full = np.concatenate([x, np.zeros_like(x)], axis=1)
    y, ld = flow.chain_apply(_lib.DIR_FIRST_TO_LAST_FORWARD, full)
    return y[:, 1:2], ld[:, None]


# Handles are cached by parameter CONTENT (shapes + a digest of the bytes; the reference's conditioner is 18
# floats per layer, so hashing costs less than the ctypes call).  JAX arrays are immutable and a reference
# script can only ever pass new arrays; NumPy callers can also update in place (`W -= lr * g`), which an
# identity key would miss -- the digest sees it and the flow is rebuilt with the new weights.
_CACHE: "OrderedDict[tuple, CouplingFlow]" = OrderedDict()
_CACHE_MAX = 16


def _signature(params: Sequence[NetParams]):
    h = hashlib.blake2b(digest_size=16)
    shapes = []
    for net in params:
        for lin in net:
            for a in lin:
                a = np.ascontiguousarray(a, np.float32)
                shapes.append(a.shape)
                h.update(a.tobytes())
    return (tuple(shapes), h.digest())


def _cache_put(key, flow: CouplingFlow) -> None:
    _CACHE[key] = flow
    _CACHE.move_to_end(key)
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)


def _chain_for(params: Sequence[NetParams], flips: Sequence[bool]) -> CouplingFlow:
    flips = [bool(f) for f in flips]
    for a, b in zip(flips[:-1], flips[1:]):
        if a == b:
            raise NotImplementedError("flips must alternate from layer to layer (init_nvp_chain does)")
    key = (_signature(params), tuple(flips))
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit
    first = params[0]
    d_in = np.asarray(first[0][0]).shape[0]
    d_out = np.asarray(first[-1][0]).shape[1]
    if d_out != 2 * d_in:
        raise ValueError(f"conditioner maps {d_in} inputs to {d_out} outputs; expected {2 * d_in} (shift | log_scale)")
    cfg = FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2 * d_in, num_layers=len(params),
                     hidden=tuple(np.asarray(W).shape[1] for W, _ in first[:-1]),
                     activation=_lib.ACT_LEAKY_RELU, first_mask_parity=int(flips[0]))
    flow = CouplingFlow(cfg, [[(np.asarray(W), np.asarray(b)) for W, b in net] for net in params])
    _cache_put(key, flow)
    return flow


def _require_fn(fn: Callable) -> None:
    if fn is not shift_and_log_scale_fn:
        raise NotImplementedError("only shift_and_log_scale_fn (the reference's conditioner) is lowered to the GPU")


def forward_nvp(net_params: NetParams, shift_and_log_scale_fn: Callable, x, flip: bool = False) -> np.ndarray:
    """src/real_nvp.py:71-101."""
    _require_fn(shift_and_log_scale_fn)
    y, _ = _chain_for([net_params], [flip]).chain_apply(_lib.DIR_FIRST_TO_LAST_FORWARD, x, want_logdet=False)
    return y


def inverse_nvp(net_params: NetParams, shift_and_log_scale_fn: Callable, y, flip: bool = False):
    """src/real_nvp.py:104-134: returns (x, log_scale)."""
    _require_fn(shift_and_log_scale_fn)
    flow = _chain_for([net_params], [flip])
    x, ld = flow.chain_apply(_lib.DIR_LAST_TO_FIRST_INVERSE, y)
    if flow.config.event_dim != 2:
        raise NotImplementedError("per-dimension log_scale is returned for the 2-D flow only; "
                                  "use log_prob_nvp for the summed log-det")
    return x, -ld[:, None]


def log_prob_nvp(net_params: NetParams, shift_and_log_scale_fn: Callable, base_log_prob_fn: Callable, y,
                 flip: bool = False) -> np.ndarray:
    """src/real_nvp.py:137-158."""
    return log_prob_nvp_chains([net_params], [(shift_and_log_scale_fn, flip)], base_log_prob_fn, y)


def init_mlp(hidden_units: Sequence[int] = (4,), rng=None) -> NetParams:
    """src/real_nvp.py:174-195: sizes are forced to [1, 4, 2]; W ~ 0.1 N(0,1), b = 0."""
    sizes = [1, 4, 2]
    rng = rng if rng is not None else np.random.Generator(np.random.Philox(534))
    params = []
    for m, n in zip(sizes[:-1], sizes[1:]):
        params.append([(rng.standard_normal((m, n)) * 0.1).astype(np.float32), np.zeros(n, np.float32)])
    return params


def init_nvp_chain(n: int = 2):
    """src/real_nvp.py:198-220.  As in the reference every layer is initialised from the
    same key, so all layers start identical."""
    flip = False
    params, fns_config = [], []
    for _ in range(n):
        params.append(init_mlp())
        fns_config.append((shift_and_log_scale_fn, flip))
        flip = not flip
    return params, fns_config


def sample_nvp_chain(params, fns_config, base_sample_fn: Callable, N: int, key) -> np.ndarray:
    """src/real_nvp.py:228-248: z = base_sample_fn(N, key); forward_nvp through layers 0..n-1."""
    for fn, _ in fns_config:
        _require_fn(fn)
    z = base_sample_fn(N, key)
    flow = _chain_for(params, [flip for _, flip in fns_config])
    y, _ = flow.chain_apply(_lib.DIR_FIRST_TO_LAST_FORWARD, z, want_logdet=False)
    return y


def make_log_prob_fn(net_params, log_prob_fn: Callable, fn_config) -> Callable:
    """src/real_nvp.py:251-265."""
    fn, flip = fn_config
    return lambda x: log_prob_nvp(net_params, fn, log_prob_fn, x, flip=flip)


def log_prob_nvp_chains(params, fns_config, base_log_prob_fn: Callable, y) -> np.ndarray:
    """src/real_nvp.py:268-283: last layer's inverse first, base density last -- one launch."""
    for fn, _ in fns_config:
        _require_fn(fn)
    if base_log_prob_fn is not log_prob_N01:
        raise NotImplementedError("only log_prob_N01 (the reference's base density) is fused into the kernel")
    return _chain_for(params, [flip for _, flip in fns_config]).log_prob(y)


def loss(params, fns_config, batch) -> float:
    """src/real_nvp.py:292-293: -mean(log_prob), reduced in float64 on the GPU."""
    for fn, _ in fns_config:
        _require_fn(fn)
    return _chain_for(params, [flip for _, flip in fns_config]).nll(batch)


def value_and_grad_loss(params, fns_config, batch):
    """jax.value_and_grad(loss)(params, batch) (src/real_nvp.py:297): (nll, grads shaped like params)."""
    for fn, _ in fns_config:
        _require_fn(fn)
    flow = _chain_for(params, [flip for _, flip in fns_config])
    nll, grads = flow.nll_grad(np.ascontiguousarray(batch, np.float32))
    return nll, [[[W, b] for W, b in layer] for layer in grads]


def sgd_update(params, fns_config, batch, learning_rate: float = 1e-2):
    """src/real_nvp.py:295-306: `nll, grads = value_and_grad(loss)(params, batch);
    params = tree_multimap(lambda p, g: p - learning_rate * g, params, grads)`; returns (nll, params).
    The step runs on the device (nf_train_step); the returned params are new arrays."""
    from . import training
    for fn, _ in fns_config:
        _require_fn(fn)
    flips = [flip for _, flip in fns_config]
    old_key = (_signature(params), tuple(bool(f) for f in flips))
    flow = _chain_for(params, flips)
    nll = flow.train_step(np.ascontiguousarray(batch, np.float32), training.sgd(learning_rate))
    new_params = [[[W, b] for W, b in layer] for layer in flow.get_weights()]
    # the handle now holds the NEW weights: move it to their key, so that the old params no longer reach it
    # (a later loss(old_params, ...) builds a flow of its own) and the 50,001-step loop keeps one entry
    _CACHE.pop(old_key, None)
    _cache_put((_signature(new_params), tuple(bool(f) for f in flips)), flow)
    return nll, new_params
