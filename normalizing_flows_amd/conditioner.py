"""Conditioner networks of the coupling layers.

The reference builds them with haiku (`make_conditioner`,
src/real_nvp_toy_densities.py:56-71, src/maf_toy_densities.py:55-70,
src/maf_mnist.py:34-49):

    hk.Sequential([Flatten, hk.nets.MLP(hidden, activate_final=True),
                   hk.Linear(n_out, w_init=zeros, b_init=zeros), Reshape])

Here a conditioner is a plain object that OWNS its float32 parameters; the HIP
library evaluates it fused with the coupling transform (there is no stand-alone
host evaluation and no CPU fallback).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

LinearParams = Tuple[np.ndarray, np.ndarray]


def _trunc_normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


class MLPConditioner:
    """MLP (hidden layers with activation, final Linear without).

    linears     [(W [in,out], b [out]), ...]; the last entry is the final Linear.
    activation  'relu' (haiku default) or 'leaky_relu' (slope 0.01, src/real_nvp.py:167)
    event_shape shape of one event; the input is flattened to prod(event_shape)
    params_per_dim  None when the output is used as is ([..., n_out]); an int P when
                    it is reshaped to event_shape + (P,) as in src/maf_toy_densities.py:69
    """

    def __init__(self, linears: Sequence[LinearParams], activation: str = "relu",
                 event_shape: Sequence[int] = (2,), params_per_dim: Optional[int] = None):
        if activation not in ("relu", "leaky_relu"):
            raise ValueError(f"unknown activation {activation!r}")
        if len(linears) < 2:
            raise ValueError("a conditioner needs at least one hidden layer and the final Linear")
        self.activation = activation
        self.event_shape = tuple(int(s) for s in event_shape)
        self.params_per_dim = params_per_dim
        self.hk_names: Optional[Tuple[str, ...]] = None   # haiku module names (set under haiku_like.transform)
        self.version = 0
        self.set_params(linears)

    def set_params(self, linears: Sequence[LinearParams]) -> None:
        lin = []
        prev = None
        for j, (W, b) in enumerate(linears):
            W = np.ascontiguousarray(W, np.float32)
            b = np.ascontiguousarray(b, np.float32)
            if W.ndim != 2 or b.shape != (W.shape[1],):
                raise ValueError(f"linear {j}: W must be [in,out] and b [out], got {W.shape}, {b.shape}")
            if prev is not None and W.shape[0] != prev:
                raise ValueError(f"linear {j}: input width {W.shape[0]} does not follow output width {prev}")
            prev = W.shape[1]
            lin.append((W, b))
        self.linears: List[LinearParams] = lin
        self.version += 1

    @property
    def in_size(self) -> int:
        return self.linears[0][0].shape[0]

    @property
    def out_size(self) -> int:
        return self.linears[-1][0].shape[1]

    @property
    def hidden_sizes(self) -> Tuple[int, ...]:
        return tuple(W.shape[1] for W, _ in self.linears[:-1])

    def __call__(self, x):
        raise NotImplementedError(
            "MLPConditioner has no stand-alone host evaluation: it runs fused inside "
            "MaskedCoupling / the real_nvp chain on the GPU (no CPU fallback)")


def make_conditioner(event_shape: Sequence[int], hidden_sizes: Sequence[int], num_bijector_params: int,
                     per_dim: bool = False, rng: Optional[np.random.Generator] = None,
                     final_std: float = 0.0) -> MLPConditioner:
    """Same arguments as the reference's `make_conditioner`.

    per_dim=False: final Linear has `num_bijector_params` outputs
                   (src/real_nvp_toy_densities.py:64-69).
    per_dim=True:  prod(event_shape) * num_bijector_params outputs reshaped to
                   event_shape + (num_bijector_params,) (src/maf_toy_densities.py:63-69).
    Initialisation follows haiku: hidden w ~ TruncNormal(+-2 sigma), sigma = 1/sqrt(fan_in),
    b = 0; final Linear zeros (final_std > 0 draws N(0, final_std^2) instead, for tests
    and benchmarks that need a non-identity flow).
    """
    d = int(np.prod(event_shape))
    sizes = [d, *[int(h) for h in hidden_sizes]]
    n_out = d * num_bijector_params if per_dim else num_bijector_params
    from . import haiku_like
    frame = haiku_like.current_frame()
    if frame is not None:
        # inside haiku_like.transform: parameters are created (init) or taken from the params dict (apply) under
        # haiku's module names, in creation order -- what the reference's hk.Sequential does
        def hidden(r, fi, fo):
            return _trunc_normal(r, (fi, fo), 1.0 / math.sqrt(fi)), np.zeros(fo, np.float32)

        def final(r, fi, fo):
            if final_std > 0:
                return ((r.standard_normal((fi, fo)) * final_std).astype(np.float32),
                        (r.standard_normal(fo) * final_std).astype(np.float32))
            return np.zeros((fi, fo), np.float32), np.zeros(fo, np.float32)
        shapes = list(zip(sizes, sizes[1:] + [n_out]))
        lin = haiku_like.conditioner_params(frame, shapes, hidden, final)
        cond = MLPConditioner(lin, "relu", event_shape, num_bijector_params if per_dim else None)
        cond.hk_names = frame.last_names      # which modules of `params` this conditioner reads
        return cond
    rng = rng if rng is not None else np.random.default_rng(42)
    lin: List[LinearParams] = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        lin.append((_trunc_normal(rng, (fi, fo), 1.0 / math.sqrt(fi)), np.zeros(fo, np.float32)))
    if final_std > 0:
        W = (rng.standard_normal((sizes[-1], n_out)) * final_std).astype(np.float32)
        b = (rng.standard_normal(n_out) * final_std).astype(np.float32)
    else:
        W = np.zeros((sizes[-1], n_out), np.float32)
        b = np.zeros(n_out, np.float32)
    lin.append((W, b))
    return MLPConditioner(lin, "relu", event_shape, num_bijector_params if per_dim else None)


_HK_MLP = re.compile(r"^mlp(?:_(\d+))?/~/linear_(\d+)$")
_HK_LIN = re.compile(r"^linear(?:_(\d+))?$")


def layers_from_haiku_params(params: Mapping[str, Mapping[str, np.ndarray]]) -> List[List[LinearParams]]:
    """Map a haiku parameter dict of the reference's flows to per-layer Linear lists.

    Names follow haiku's creation order: coupling layer k owns
    'mlp/~/linear_i' ('mlp_k/~/linear_i' for k > 0) and the final 'linear'
    ('linear_k'); each module has 'w' [in,out] and 'b' [out].
    """
    hidden: Dict[int, Dict[int, LinearParams]] = {}
    final: Dict[int, LinearParams] = {}
    for name, mod in params.items():
        m = _HK_MLP.match(name)
        if m:
            k = int(m.group(1) or 0)
            hidden.setdefault(k, {})[int(m.group(2))] = (np.asarray(mod["w"]), np.asarray(mod["b"]))
            continue
        m = _HK_LIN.match(name)
        if m:
            final[int(m.group(1) or 0)] = (np.asarray(mod["w"]), np.asarray(mod["b"]))
            continue
        raise ValueError(f"unexpected haiku module name {name!r}")
    if sorted(hidden) != list(range(len(hidden))) or sorted(final) != sorted(hidden):
        raise ValueError("haiku params do not form consecutive coupling layers")
    out = []
    for k in range(len(hidden)):
        idx = sorted(hidden[k])
        if idx != list(range(len(idx))):
            raise ValueError(f"coupling layer {k}: MLP linears are not consecutive")
        out.append([hidden[k][i] for i in idx] + [final[k]])
    return out


def layers_to_haiku_params(layers: Sequence[Sequence[LinearParams]]) -> Dict[str, Dict[str, np.ndarray]]:
    """Inverse of `layers_from_haiku_params`: per-layer Linear lists -> haiku-named dict."""
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for k, layer in enumerate(layers):
        sfx = "" if k == 0 else f"_{k}"
        for i, (W, b) in enumerate(layer[:-1]):
            out[f"mlp{sfx}/~/linear_{i}"] = {"w": np.asarray(W, np.float32), "b": np.asarray(b, np.float32)}
        W, b = layer[-1]
        out[f"linear{sfx}"] = {"w": np.asarray(W, np.float32), "b": np.asarray(b, np.float32)}
    return out


def save_params(path: str, params: Mapping[str, Mapping[str, np.ndarray]]) -> None:
    """Weight file (the reference keeps parameters in process memory only):
    one .npz whose keys are '<haiku module name>/w' and '<haiku module name>/b', float32."""
    flat = {}
    for name, mod in params.items():
        flat[f"{name}/w"] = np.asarray(mod["w"], np.float32)
        flat[f"{name}/b"] = np.asarray(mod["b"], np.float32)
    np.savez(path, **flat)


def load_params(path: str) -> Dict[str, Dict[str, np.ndarray]]:
    out: Dict[str, Dict[str, np.ndarray]] = {}
    with np.load(path) as f:
        for key in f.files:
            name, leaf = key.rsplit("/", 1)
            if leaf not in ("w", "b"):
                raise ValueError(f"unexpected entry {key!r} in weight file")
            out.setdefault(name, {})[leaf] = f[key]
    for name, mod in out.items():
        if set(mod) != {"w", "b"}:
            raise ValueError(f"module {name!r} lacks w or b")
    return out
