"""The haiku calls the reference's flow scripts make AROUND the hot path, so that their call sites stay as written:

    @hk.without_apply_rng
    @hk.transform
    def log_prob(data):                                        # src/real_nvp_toy_densities.py:113-123
        model = make_flow_model(event_shape=data.shape[-1:], num_layers=..., hidden_sizes=[...])
        return model.log_prob(data)

    params = log_prob.init(next(prng_seq), np.zeros((1, *event_shape)))      # :175-177
    loss = -jnp.mean(log_prob.apply(params, batch))                          # :143,150

`import normalizing_flows_amd.haiku_like as hk` provides `transform`, `without_apply_rng` and `PRNGSequence` with
those semantics for functions whose only parameterised modules are this package's conditioners
(`make_conditioner`, which stands for the reference's hk.Sequential([Flatten, hk.nets.MLP, hk.Linear, Reshape])):

  * `init(rng, *args)` runs the function once; every `make_conditioner` call creates its parameters with haiku's
    initialisers (TruncatedNormal(1/sqrt(fan_in)) weights, zero biases, zero final Linear) and files them under the
    names haiku gives in creation order -- 'mlp/~/linear_0', ..., 'linear', 'mlp_1/~/linear_0', ..., 'linear_1', ...
    ('w' [in, out], 'b' [out]) -- and returns that dict.  The function's result is discarded
    (haiku evaluates it; here nothing is computed: with the zero final Linear the flow is the identity anyway).
  * `apply(params, [rng,] *args)` runs the function with every `make_conditioner` call taking its parameters from
    `params` by the same names (ValueError on a missing module, a wrong shape or unused modules, as haiku raises:
    a tree with new module names is first taken through a dry run of the function that evaluates nothing).
    The device handles the function's models lower to are kept by the transformed function and reused from call to
    call while `params` holds the same values, so a loop of `apply` calls costs one kernel launch each, not one
    handle construction and weight upload each (the reference gets the same from jax.jit's trace cache).

Nothing here computes: evaluation stays in libnf_hip.so.  jax.grad / optax have no counterpart in this module -- the
training step is `training.update` (nf_train_step), see INTEGRATION.md.
"""
from __future__ import annotations

import hashlib
import weakref
from typing import Callable, Dict, List, Mapping, Optional, Tuple

import numpy as np

_FRAMES: List["_Frame"] = []


class _Frame:
    def __init__(self, mode: str, params: Optional[Mapping[str, Mapping[str, np.ndarray]]], rng, owner: "Transformed"):
        self.mode = mode                  # 'init' | 'plan' (apply's dry run: names and shapes only) | 'apply'
        self.params = {} if mode == "init" else params
        self.rng = rng
        self.owner = owner
        self.layer = 0                    # conditioners created so far
        self._mod_keys = {}               # module name -> content key (apply)

    def check_all_used(self) -> None:
        """apply: the params dict holds exactly the coupling layers the function built.  Called ONCE, when the
        function has returned -- a function may build several models (and evaluate one before it builds the next),
        so a count taken in the middle says nothing."""
        have = sum(1 for name in self.params if name.startswith("linear"))
        if self.mode != "init" and have != self.layer:
            raise ValueError(f"params hold {have} coupling layers, the function built {self.layer}")

    def modules_key(self, names) -> tuple:
        """Content key of the named modules only (memoised per apply call): what one chain's device weights
        depend on."""
        out = []
        for name in names:
            k = self._mod_keys.get(name)
            if k is None:
                mod = self.params[name]
                k = self._mod_keys[name] = (name, _array_key(mod["w"]), _array_key(mod["b"]))
            out.append(k)
        return tuple(out)

    def next_names(self, n_hidden: int) -> Tuple[List[str], str]:
        k = self.layer
        self.layer += 1
        sfx = "" if k == 0 else f"_{k}"
        return [f"mlp{sfx}/~/linear_{i}" for i in range(n_hidden)], f"linear{sfx}"


def current_frame() -> Optional[_Frame]:
    return _FRAMES[-1] if _FRAMES else None


try:                                      # xxh3: ~10 GB/s, so the 350 MB of an MNIST-shape flow cost ~30 ms per apply
    from xxhash import xxh3_128_digest as _digest
except ImportError:                       # pragma: no cover - blake2b: ~1 GB/s, same guarantee
    def _digest(buf) -> bytes:
        return hashlib.blake2b(buf, digest_size=16).digest()


_FROZEN_DIGESTS: dict = {}   # id(array) -> (weakref to it, key): arrays nothing can write to are hashed once


def _immutable(a) -> bool:
    """True when no NumPy handle can write to a's bytes: a itself and every ndarray it is a view of are read-only
    (np.asarray of a jax.Array is such an array; a read-only VIEW of a writable array is not)."""
    b = a
    while isinstance(b, np.ndarray):
        if b.flags.writeable:
            return False
        b = b.base
    return True


def _array_key(a: np.ndarray) -> tuple:
    """Shape, dtype and a digest of EVERY byte: an optimiser step, a reassigned array and an in-place edit of a
    single entry all change it (a sparse sample of large arrays, as an earlier version took, lets such an edit
    evaluate stale device weights silently).  The digest of an array that CANNOT change (read-only all the way down:
    what np.asarray gives for the reference's jax parameters) is computed once per array object; writable arrays are
    hashed on every call (30 ms per apply for the 351 MB of the MNIST-shape flow with xxh3: INTEGRATION.md)."""
    if isinstance(a, np.ndarray) and _immutable(a):
        hit = _FROZEN_DIGESTS.get(id(a))
        if hit is not None and hit[0]() is a:
            return hit[1]
        c = np.ascontiguousarray(a)
        key = (c.shape, c.dtype.num, _digest(memoryview(c).cast("B")))
        try:
            ident = id(a)
            _FROZEN_DIGESTS[ident] = (weakref.ref(a, lambda _r, ident=ident: _FROZEN_DIGESTS.pop(ident, None)), key)
        except TypeError:      # (an ndarray subclass without weak references: hash it every time)
            pass
        return key
    a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.num, _digest(memoryview(a).cast("B")))   # (str(dtype) alone cost 3 us per array)


def params_key(params: Mapping[str, Mapping[str, np.ndarray]]) -> tuple:
    return tuple((name, _array_key(mod["w"]), _array_key(mod["b"])) for name, mod in sorted(params.items()))


class Transformed:
    """What hk.transform returns: `.init(rng, *args)` and `.apply(params, rng, *args)`."""

    def __init__(self, fn: Callable, apply_takes_rng: bool = True):
        self._fn = fn
        self._apply_takes_rng = apply_takes_rng
        # (FlowConfig, column orders, haiku module names of the chain) -> (content key of THOSE modules,
        # CouplingFlow): see Chain.lowered.  Two models of one architecture built by one function have different
        # module names, hence different slots.
        self.flow_cache: Dict[tuple, tuple] = {}
        self._validated: Optional[frozenset] = None   # (module name, w.shape, b.shape) of the last params tree the dry
                                                      # run accepted

    def init(self, rng, *args, **kwargs) -> Dict[str, Dict[str, np.ndarray]]:
        seed = rng if isinstance(rng, (int, np.integer)) else int(np.asarray(rng).reshape(-1)[-1])
        frame = _Frame("init", None, np.random.default_rng(seed), self)
        _FRAMES.append(frame)
        try:
            self._fn(*args, **kwargs)
        finally:
            _FRAMES.pop()
        return frame.params

    def apply(self, params, *args, **kwargs):
        if self._apply_takes_rng:
            if not args:
                raise TypeError("apply(params, rng, *args): rng is missing (wrap with without_apply_rng to drop it)")
            args = args[1:]
        if not isinstance(params, Mapping):
            raise ValueError("params must be the dict init() returned (haiku module name -> {'w', 'b'})")
        # (names AND shapes: a tree with the same module names but a wrong shape in a later module must not skip the
        # dry run and fail after earlier models have already run on the GPU)
        names = frozenset((name, tuple(np.shape(mod.get("w"))), tuple(np.shape(mod.get("b")))) if isinstance(mod, Mapping)
                          else (name, None, None) for name, mod in params.items())
        if names != self._validated:
            # dry run: the function builds its models against `params` (missing module, wrong shape and a wrong
            # number of coupling layers raise here), evaluates nothing -- so a bad tree fails before anything
            # reaches the GPU even when the function builds several models.  (Every evaluation entry point of the
            # surface is a stub in this mode: Transformed.log_prob / sample / sample_and_log_prob / nll and the
            # bijectors' forward / inverse walks return zeros of the right shape.  The function itself runs twice on
            # such a call -- side effects of user code in it, e.g. drawing from a host rng, happen twice.)
            self._run(_Frame("plan", params, None, self), args, kwargs)
            self._validated = names
        return self._run(_Frame("apply", params, None, self), args, kwargs)

    def _run(self, frame: _Frame, args, kwargs):
        _FRAMES.append(frame)
        try:
            out = self._fn(*args, **kwargs)
        finally:
            _FRAMES.pop()
        frame.check_all_used()
        return out


def transform(fn: Callable) -> Transformed:
    """hk.transform."""
    return Transformed(fn, apply_takes_rng=True)


def without_apply_rng(t: Transformed) -> Transformed:
    """hk.without_apply_rng: `apply(params, *args)`."""
    out = Transformed(t._fn, apply_takes_rng=False)
    out.flow_cache = t.flow_cache
    out._validated = t._validated
    return out


class PRNGSequence:
    """hk.PRNGSequence(seed): an iterator of keys (here: integers drawn from a seeded generator)."""

    def __init__(self, seed):
        self._rng = np.random.default_rng(int(seed))

    def __iter__(self):
        return self

    def __next__(self) -> int:
        return int(self._rng.integers(0, 2 ** 63 - 1))


def conditioner_params(frame: _Frame, shapes: List[Tuple[int, int]], make_hidden, make_final):
    """Called by make_conditioner under an active frame: the [(W, b), ...] of the next conditioner, created
    (init) or looked up (apply) under haiku's names."""
    hidden_names, final_name = frame.next_names(len(shapes) - 1)
    names = hidden_names + [final_name]
    frame.last_names = tuple(names)
    out = []
    for j, (name, (fi, fo)) in enumerate(zip(names, shapes)):
        if frame.mode == "init":
            W, b = (make_final if j == len(shapes) - 1 else make_hidden)(frame.rng, fi, fo)
            frame.params[name] = {"w": W, "b": b}
        else:
            mod = frame.params.get(name)
            if mod is None:
                raise ValueError(f"params lack module {name!r} (modules are named in creation order: "
                                 f"'mlp/~/linear_0', ..., 'linear', 'mlp_1/~/linear_0', ...)")
            W, b = np.asarray(mod["w"]), np.asarray(mod["b"])
            if W.shape != (fi, fo) or b.shape != (fo,):
                raise ValueError(f"module {name!r}: expected w {(fi, fo)} and b {(fo,)}, got {W.shape} and {b.shape}")
        out.append((W, b))
    return out
