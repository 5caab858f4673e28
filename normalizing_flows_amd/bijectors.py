"""distrax-shaped bijectors over the HIP coupling-chain kernels.

The reference's flows are written against `distrax.Bijector`
(src/real_nvp_toy_densities.py:84-104, src/maf_toy_densities.py:86-105,
src/maf_mnist.py:65-83; the protocol itself is shown in README.md:32-57 and
src/rotation_flow.py:13-37).  The classes here keep that surface --
`forward`, `inverse`, `forward_and_log_det`, `inverse_and_log_det`,
`forward_log_det_jacobian`, `inverse_log_det_jacobian` -- and lower every
composition of `MaskedCoupling` layers (alone, in a `Chain`, under `Inverse`) to
ONE fused launch of libnf_hip through the C ABI.

There is no host arithmetic here and no CPU fallback: a structure the library
has no kernel for raises `NotImplementedError`; a missing library or GPU raises
`NFError`.
"""
from __future__ import annotations

import types
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .conditioner import MLPConditioner
from .flow import CouplingFlow, DeviceArray, DeviceView, FlowConfig

Array = np.ndarray


class Bijector:
    """The distrax.Bijector protocol (README.md:32-57 of the reference)."""

    def __init__(self, event_ndims_in: int, event_ndims_out: Optional[int] = None,
                 is_constant_jacobian: bool = False, is_constant_log_det: Optional[bool] = None):
        if event_ndims_out is None:
            event_ndims_out = event_ndims_in
        if event_ndims_in < 0 or event_ndims_out < 0:
            raise ValueError("event_ndims must be non-negative")
        if is_constant_log_det is None:
            is_constant_log_det = is_constant_jacobian
        if is_constant_jacobian and not is_constant_log_det:
            raise ValueError("a constant Jacobian has a constant log-determinant")
        self._event_ndims_in = event_ndims_in
        self._event_ndims_out = event_ndims_out
        self._is_constant_jacobian = is_constant_jacobian
        self._is_constant_log_det = is_constant_log_det

    event_ndims_in = property(lambda self: self._event_ndims_in)
    event_ndims_out = property(lambda self: self._event_ndims_out)
    is_constant_jacobian = property(lambda self: self._is_constant_jacobian)
    is_constant_log_det = property(lambda self: self._is_constant_log_det)

    def forward(self, x):
        return self.forward_and_log_det(x)[0]

    def inverse(self, y):
        return self.inverse_and_log_det(y)[0]

    def forward_log_det_jacobian(self, x):
        return self.forward_and_log_det(x)[1]

    def inverse_log_det_jacobian(self, y):
        return self.inverse_and_log_det(y)[1]

    def forward_and_log_det(self, x):
        raise NotImplementedError(f"{type(self).__name__} does not implement forward_and_log_det")

    def inverse_and_log_det(self, y):
        raise NotImplementedError(f"{type(self).__name__} does not implement inverse_and_log_det")

    @property
    def name(self) -> str:
        return type(self).__name__


# --------------------------------------------------------------------------
# inner (element-wise) bijectors: descriptions the coupling layer is built from
# --------------------------------------------------------------------------
def _elementwise(desc: "_lib.ElementwiseDesc", inverse: bool, x, params: np.ndarray):
    """x (any shape) and params (x.shape + (P,)) through nf_elementwise_apply; returns (y, logdet), each of x's
    shape.  Host arrays in, host arrays out (the parameters of a stand-alone inner bijector are host arrays)."""
    import ctypes as C
    x = np.asarray(x, np.float32)
    P = params.shape[-1]
    shape = np.broadcast_shapes(x.shape, params.shape[:-1])
    xb = np.ascontiguousarray(np.broadcast_to(x, shape))
    pb = np.ascontiguousarray(np.broadcast_to(params, shape + (P,)), np.float32)
    y = np.empty(shape, np.float32)
    ld = np.empty(shape, np.float32)
    _lib.check(_lib.load().nf_elementwise_apply(C.byref(desc), 1 if inverse else 0, xb.ctypes.data_as(C.c_void_p),
                                                pb.ctypes.data_as(C.c_void_p), xb.size, y.ctypes.data_as(C.c_void_p),
                                                ld.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
    return y, ld


class ScalarAffine(Bijector):
    """y = scale * x + shift with scale = exp(log_scale) (distrax.ScalarAffine; used at
    src/real_nvp_toy_densities.py:86).  Inside `MaskedCoupling` its parameters come from the conditioner and the
    map is evaluated by the fused kernel; on its own (explicit `shift` and `log_scale` or `scale` arrays, broadcast
    against the input as distrax does) it runs through nf_elementwise_apply."""

    def __init__(self, shift, scale=None, log_scale=None):
        super().__init__(event_ndims_in=0, is_constant_jacobian=True)
        if scale is not None and log_scale is not None:
            raise ValueError("only one of scale and log_scale may be given")
        self.shift = shift
        if scale is not None:
            self.scale = scale
            self.log_scale = None       # distrax derives log|scale| lazily; the kernel does (NF_ELEM_AFFINE_SCALE)
        else:
            self.scale = None
            self.log_scale = log_scale if log_scale is not None else np.zeros_like(np.asarray(shift))

    def _apply(self, v, inverse: bool):
        second = self.log_scale if self.scale is None else self.scale
        shift, second = np.broadcast_arrays(np.asarray(self.shift, np.float32), np.asarray(second, np.float32))
        desc = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION,
                                    kind=_lib.ELEM_AFFINE_LOG_SCALE if self.scale is None else _lib.ELEM_AFFINE_SCALE)
        return _elementwise(desc, inverse, v, np.stack([shift, second], axis=-1))

    def forward_and_log_det(self, x):
        return self._apply(x, False)

    def inverse_and_log_det(self, y):
        return self._apply(y, True)


class RationalQuadraticSpline(Bijector):
    """distrax.RationalQuadraticSpline (used at src/maf_toy_densities.py:86-87 and
    src/maf_mnist.py:65-66): monotone rational-quadratic spline on
    [range_min, range_max] with linear tails; `params[..., 3K+1]`.  Inside `MaskedCoupling` the fused kernels
    evaluate it; on its own (explicit `params`, batch dimensions broadcast against the input) it runs through
    nf_elementwise_apply -- the same spline arithmetic (csrc/nf_rqs_math.h)."""

    def __init__(self, params, range_min: float, range_max: float,
                 boundary_slopes: str = "unconstrained", min_bin_size: float = 1e-4,
                 min_knot_slope: float = 1e-4):
        super().__init__(event_ndims_in=0)
        p = np.asarray(params)
        if p.shape[-1] % 3 != 1 or p.shape[-1] < 4:
            raise ValueError(f"the last dimension of params must be 3K+1 with K >= 1, got {p.shape[-1]}")
        if range_min >= range_max:
            raise ValueError(f"range_min {range_min} must be below range_max {range_max}")
        if boundary_slopes != "unconstrained":
            raise NotImplementedError("only boundary_slopes='unconstrained' (the reference's) has a kernel")
        if min_knot_slope >= 1.0 or min_knot_slope <= 0.0:
            raise ValueError("min_knot_slope must be in (0, 1)")
        self.params = params
        self.num_bins = (p.shape[-1] - 1) // 3
        if self.num_bins * min_bin_size > range_max - range_min:
            raise ValueError("num_bins * min_bin_size exceeds the range")
        self.range_min = float(range_min)
        self.range_max = float(range_max)
        self.min_bin_size = float(min_bin_size)
        self.min_knot_slope = float(min_knot_slope)

    def _apply(self, v, inverse: bool):
        desc = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION, kind=_lib.ELEM_RQS, rqs_bins=self.num_bins,
                                    rqs_range_min=self.range_min, rqs_range_max=self.range_max,
                                    rqs_min_bin_size=self.min_bin_size, rqs_min_knot_slope=self.min_knot_slope)
        return _elementwise(desc, inverse, v, np.asarray(self.params, np.float32))

    def forward_and_log_det(self, x):
        return self._apply(x, False)

    def inverse_and_log_det(self, y):
        return self._apply(y, True)


# --------------------------------------------------------------------------
# lowering helpers
# --------------------------------------------------------------------------
def _flatten_rows(x, event_shape: Tuple[int, ...]):
    """[..., *event_shape] -> ([rows, D], batch_shape).  Device arrays must already be [rows, D]."""
    D = int(np.prod(event_shape))
    if isinstance(x, (DeviceArray, DeviceView)):
        if tuple(x.shape[1:]) != (D,):
            raise ValueError(f"device input must be [rows, {D}], got {tuple(x.shape)}")
        return x, (x.shape[0],)
    a = np.asarray(x)
    n = len(event_shape)
    if a.ndim < n or tuple(a.shape[a.ndim - n:]) != tuple(event_shape):
        raise ValueError(f"input shape {a.shape} does not end with the event shape {tuple(event_shape)}")
    batch = a.shape[:a.ndim - n]
    return np.ascontiguousarray(a.reshape(-1, D), np.float32), batch


def _unflatten(y, ld, batch, event_shape):
    if isinstance(y, (DeviceArray, DeviceView)) or isinstance(ld, (DeviceArray, DeviceView)):
        return y, ld
    if y is not None:
        y = y.reshape(tuple(batch) + tuple(event_shape))
    if ld is not None:
        ld = ld.reshape(tuple(batch))
    return y, ld


class _LayerSignature:
    __slots__ = ("kind", "event_shape", "parity", "hidden", "activation", "rqs")

    def key(self):
        return (self.kind, self.event_shape, self.hidden, self.activation, self.rqs)


_SIGNATURES: dict = {}   # MaskedCoupling._signature_cache_key()[0] -> (_LayerSignature, column order, kept objects)

_BY_IDENTITY = (types.ModuleType, type, types.FunctionType, types.BuiltinFunctionType, types.MethodType)


def _key_part(v, keep: list, depth: int = 0):
    """One captured value as a hashable key part, or None if it has no safe one (see _signature_cache_key)."""
    if isinstance(v, (bool, int, float, str, bytes, type(None))):
        return ("v", type(v).__name__, v)
    if isinstance(v, np.generic):
        return ("g", type(v).__name__, v.item())
    if isinstance(v, tuple) and depth < 4:
        parts = tuple(_key_part(e, keep, depth + 1) for e in v)
        return None if any(p is None for p in parts) else ("t", parts)
    if isinstance(v, _BY_IDENTITY):
        keep.append(v)
        return ("o", id(v))
    return None


class MaskedCoupling(Bijector):
    """distrax.MaskedCoupling (src/real_nvp_toy_densities.py:90-98).

    mask         bool array of the event shape; True = passed through unchanged and
                 shown to the conditioner.  The kernels implement the reference's
                 alternating masks, `arange(D) % 2 == p` (any p in {0, 1}).
    bijector     callable params -> inner Bijector.  Recognised forms: the reference's
                 `split(params, 2, -1) -> ScalarAffine(shift, log_scale)` (either
                 column order) and `RationalQuadraticSpline(params, range_min, range_max)`.
    conditioner  an `MLPConditioner`.
    """

    def __init__(self, mask, bijector: Callable, conditioner: MLPConditioner,
                 event_ndims: Optional[int] = None, inner_event_ndims: int = 0):
        mask = np.asarray(mask)
        if mask.dtype != np.bool_:
            raise ValueError("mask must be a boolean array")
        if event_ndims is not None and event_ndims < inner_event_ndims:
            raise ValueError("event_ndims must be at least inner_event_ndims")
        if inner_event_ndims != 0:
            raise NotImplementedError("only scalar inner bijectors (inner_event_ndims=0) have kernels")
        event_ndims = mask.ndim if event_ndims is None else event_ndims
        if event_ndims != mask.ndim:
            raise NotImplementedError("the mask must span exactly the event dimensions")
        super().__init__(event_ndims_in=event_ndims)
        if not isinstance(conditioner, MLPConditioner):
            raise NotImplementedError(
                "conditioner must be an MLPConditioner: arbitrary Python callables cannot be "
                "lowered to the GPU and there is no CPU fallback")
        if not callable(bijector):
            raise ValueError("bijector must be a callable params -> Bijector")
        self._mask = mask
        self._bijector_fn = bijector
        self._conditioner = conditioner
        self._sig: Optional[_LayerSignature] = None
        self._perm: Optional[Tuple[int, int]] = None
        self._solo: Optional["Chain"] = None

    mask = property(lambda self: self._mask)
    conditioner = property(lambda self: self._conditioner)

    # -- what does bijector_fn build? -------------------------------------------
    def _signature_cache_key(self):
        """haiku_like's apply re-runs the model-building function on every call, so every call meets new
        MaskedCoupling objects around a NEW closure of the same `bijector_fn` source.  What the probe below finds
        depends on that function's code, on what it closes over, on the module globals it reads and on the layer's
        shapes only: cached per (code object, closure contents, globals read, defaults, mask, conditioner shape) -- 8
        probes of 2 calls each were 300 us of a 445 us `log_prob.apply(params, batch)` at the reference's batch of 128.
        Returns (key, keep) or None (not cacheable: the probe runs every time).  Values enter the key BY VALUE
        (Python scalars, NumPy scalars -- a fresh `np.float32(...)` per call is the usual case -- strings, tuples of
        those); modules, classes and functions by identity, with a strong reference in `keep` that the cache entry
        holds, so that their id cannot be recycled while the entry lives; anything else (arrays, lists, config
        objects, 0-d device scalars ...) makes the layer uncacheable rather than keyed by an address."""
        fn = self._bijector_fn
        code = getattr(fn, "__code__", None)
        if code is None:
            return None
        keep: list = []
        parts = [code]
        try:
            for cell in (fn.__closure__ or ()):
                part = _key_part(cell.cell_contents, keep)
                if part is None:
                    return None
                parts.append(part)
            glob = getattr(fn, "__globals__", {})
            for name in code.co_names:       # globals (and attribute names: those are not in __globals__, or are
                if name in glob:             # keyed harmlessly) the function may read
                    part = _key_part(glob[name], keep)
                    if part is None:
                        return None
                    parts.append((name, part))
            part = _key_part((fn.__defaults__, tuple(sorted((fn.__kwdefaults__ or {}).items()))), keep)
            if part is None:
                return None
            parts.append(part)
            c = self._conditioner
            parts.append((self._mask.shape, self._mask.tobytes(), c.in_size, c.out_size, c.params_per_dim,
                          c.hidden_sizes, c.activation))
            key = tuple(parts)
            hash(key)
            return key, keep
        except (TypeError, ValueError):
            return None

    def signature(self) -> _LayerSignature:
        if self._sig is not None:
            return self._sig
        keyed = self._signature_cache_key()
        if keyed is not None:
            hit = _SIGNATURES.get(keyed[0])
            if hit is not None:
                self._sig, self._perm = hit[0], hit[1]
                return self._sig
        sig = self._probe_signature()
        if keyed is not None:
            if len(_SIGNATURES) > 256:
                _SIGNATURES.clear()
            _SIGNATURES[keyed[0]] = (sig, self._perm, keyed[1])   # keyed[1]: the identity-keyed objects, kept alive
        return sig

    def _probe_signature(self) -> _LayerSignature:
        cond = self._conditioner
        flat = self._mask.reshape(-1)
        D = flat.size
        if D < 2:
            raise NotImplementedError("coupling needs at least 2 event elements")
        parity = None
        for p in (0, 1):
            if np.array_equal(flat, (np.arange(D) % 2) == p):
                parity = p
        if parity is None:
            raise NotImplementedError("only the alternating masks arange(D) % 2 == p have kernels")
        if cond.in_size != D:
            raise ValueError(f"conditioner expects {cond.in_size} inputs, the event has {D} elements")
        n_out = cond.out_size
        shape = (1, n_out) if cond.params_per_dim is None else (1,) + self._mask.shape + (cond.params_per_dim,)
        probe_a = (np.arange(n_out, dtype=np.float32) * 0.5 - 1.25).reshape(shape)
        probe_b = (np.arange(n_out, dtype=np.float32)[::-1] * 0.25 + 3.0).reshape(shape)
        inner_a = self._bijector_fn(probe_a)
        inner_b = self._bijector_fn(probe_b)
        sig = _LayerSignature()
        sig.event_shape = tuple(self._mask.shape)
        sig.parity = parity
        sig.hidden = cond.hidden_sizes
        sig.activation = _lib.ACT_RELU if cond.activation == "relu" else _lib.ACT_LEAKY_RELU
        sig.rqs = None
        if isinstance(inner_a, ScalarAffine) and isinstance(inner_b, ScalarAffine):
            if n_out != 2 or cond.params_per_dim is not None:
                raise NotImplementedError("the affine kernel takes exactly 2 conditioner outputs "
                                          "(shift, log_scale) broadcast over the event")
            if inner_a.log_scale is None or inner_b.log_scale is None:
                raise NotImplementedError("inside MaskedCoupling the affine kernels take ScalarAffine(shift, log_scale); "
                                          "the scale= form is evaluated stand-alone only")
            perm = None
            for cand in ((0, 1), (1, 0)):
                ok = True
                for probe, inner in ((probe_a, inner_a), (probe_b, inner_b)):
                    s = np.asarray(inner.shift, np.float32).reshape(-1)
                    ls = np.asarray(inner.log_scale, np.float32).reshape(-1)
                    ok &= s.size == 1 and ls.size == 1 and s[0] == probe.reshape(-1)[cand[0]] \
                        and ls[0] == probe.reshape(-1)[cand[1]]
                if ok:
                    perm = cand
            if perm is None:
                raise NotImplementedError("bijector_fn must map the two conditioner outputs to "
                                          "ScalarAffine(shift, log_scale) without further arithmetic")
            sig.kind = _lib.KIND_AFFINE_MASKED
            self._perm = perm
        elif isinstance(inner_a, RationalQuadraticSpline) and isinstance(inner_b, RationalQuadraticSpline):
            if cond.params_per_dim is None:
                raise NotImplementedError("the spline kernel needs conditioner outputs shaped event_shape + (3K+1,)")
            if not (np.array_equal(np.asarray(inner_a.params, np.float32), probe_a)
                    and np.array_equal(np.asarray(inner_b.params, np.float32), probe_b)):
                raise NotImplementedError("bijector_fn must pass the conditioner output to "
                                          "RationalQuadraticSpline unchanged")
            sig.kind = _lib.KIND_RQS_MASKED
            sig.rqs = (inner_a.num_bins, inner_a.range_min, inner_a.range_max,
                       inner_a.min_bin_size, inner_a.min_knot_slope)
        else:
            raise NotImplementedError(
                f"inner bijector {type(inner_a).__name__} has no fused kernel (ScalarAffine and "
                "RationalQuadraticSpline do); there is no CPU fallback")
        self._sig = sig
        return sig

    def packed_linears(self):
        """Conditioner parameters in the column order the kernel expects."""
        self.signature()
        lin = list(self._conditioner.linears)
        if self._perm == (1, 0):
            W, b = lin[-1]
            lin[-1] = (np.ascontiguousarray(W[:, ::-1]), np.ascontiguousarray(b[::-1]))
        return lin

    def _as_chain(self) -> "Chain":
        if self._solo is None:
            self._solo = Chain([self])
        return self._solo

    def forward_and_log_det(self, x):
        return self._as_chain().forward_and_log_det(x)

    def inverse_and_log_det(self, y):
        return self._as_chain().inverse_and_log_det(y)


class Chain(Bijector):
    """distrax.Chain: `Chain([b0, ..., bn]).forward = b0 o b1 o ... o bn` (bn is applied
    first); `.inverse` applies b0^-1 first (src/real_nvp_toy_densities.py:104)."""

    def __init__(self, bijectors: Sequence[Bijector]):
        if not bijectors:
            raise ValueError("the sequence of bijectors cannot be empty")
        for a, b in zip(bijectors[:-1], bijectors[1:]):
            if a.event_ndims_in != b.event_ndims_out:
                raise ValueError("bijectors in a Chain must have matching event_ndims")
        super().__init__(event_ndims_in=bijectors[-1].event_ndims_in,
                         event_ndims_out=bijectors[0].event_ndims_out,
                         is_constant_jacobian=all(b.is_constant_jacobian for b in bijectors),
                         is_constant_log_det=all(b.is_constant_log_det for b in bijectors))
        self._bijectors = list(bijectors)
        self._flows = {}  # base code -> (parameter key, CouplingFlow)

    bijectors = property(lambda self: self._bijectors)

    # -- lowering ----------------------------------------------------------------
    def lowered(self, base: int = _lib.BASE_STD_NORMAL) -> Tuple[CouplingFlow, Tuple[int, ...]]:
        """The device chain for this Chain (built once, rebuilt when parameters change)."""
        layers = self._bijectors
        if not all(isinstance(b, MaskedCoupling) for b in layers):
            other = next(type(b).__name__ for b in layers if not isinstance(b, MaskedCoupling))
            raise NotImplementedError(
                f"Chain contains {other}: only chains of MaskedCoupling layers have a fused kernel "
                "and there is no CPU fallback")
        sigs = [b.signature() for b in layers]
        first = sigs[0]
        for l, s in enumerate(sigs):
            if s.key() != first.key():
                raise NotImplementedError("all coupling layers of a Chain must share one architecture")
            if s.parity != (first.parity + l) % 2:
                raise NotImplementedError("masks must alternate from layer to layer "
                                          "(mask = logical_not(mask), as in the reference)")
        key = tuple((id(b.conditioner), b.conditioner.version) for b in layers)
        cached = self._flows.get(base)
        if cached is None or cached[0] != key:
            rqs = first.rqs or (0, 0.0, 1.0, 1e-4, 1e-4)
            cfg = FlowConfig(kind=first.kind, event_dim=int(np.prod(first.event_shape)),
                             num_layers=len(layers), hidden=first.hidden, activation=first.activation,
                             first_mask_parity=first.parity, base=base, rqs_bins=rqs[0],
                             rqs_range_min=rqs[1], rqs_range_max=rqs[2], rqs_min_bin_size=rqs[3],
                             rqs_min_knot_slope=rqs[4])
            from . import haiku_like
            frame = haiku_like.current_frame()
            names = tuple(b.conditioner.hk_names for b in layers)
            if frame is not None and frame.mode == "apply" and all(n is not None for n in names):
                # built inside haiku_like's apply(params, ...): this Chain object lives for one call, the device
                # handle is kept by the transformed function for as long as THIS chain's modules keep their values.
                # The slot carries the chain's module names: a second model of the same architecture built by the
                # same function reads other modules and gets its own handle.
                slot = (cfg, tuple(b._perm for b in layers), names)
                mods_key = frame.modules_key([n for layer in names for n in layer])
                held = frame.owner.flow_cache.get(slot)
                if held is None:
                    held = (mods_key, CouplingFlow(cfg, [b.packed_linears() for b in layers]))
                elif held[0] != mods_key:
                    held[1].set_weights([b.packed_linears() for b in layers])
                    held = (mods_key, held[1])
                frame.owner.flow_cache[slot] = held
                self._flows[base] = (key, held[1])
                self._borrowed = True
                return held[1], first.event_shape
            if cached is not None and not getattr(self, "_borrowed", False):
                cached[1].close()
            cached = (key, CouplingFlow(cfg, [b.packed_linears() for b in layers]))
            self._flows[base] = cached
        return cached[1], first.event_shape

    def _walk(self, direction: int, v):
        from . import haiku_like
        frame = haiku_like.current_frame()
        if frame is not None and frame.mode in ("init", "plan"):
            # hk.transform(...).init and apply's dry run build the model and evaluate nothing (haiku_like.py)
            if isinstance(v, (DeviceArray, DeviceView)):
                return v, DeviceArray((v.shape[0],))
            a = np.asarray(v, np.float32)
            return np.zeros_like(a), np.zeros(a.shape[:a.ndim - self.event_ndims_in], np.float32)
        # the base density plays no part in a chain walk: reuse whichever handle exists
        base = next(iter(self._flows), _lib.BASE_STD_NORMAL)
        flow, event_shape = self.lowered(base)
        rows, batch = _flatten_rows(v, event_shape)
        y, ld = flow.chain_apply(direction, rows)
        return _unflatten(y, ld, batch, event_shape)

    def forward_and_log_det(self, x):
        return self._walk(_lib.DIR_CHAIN_FORWARD, x)

    def inverse_and_log_det(self, y):
        return self._walk(_lib.DIR_CHAIN_INVERSE, y)


class Inverse(Bijector):
    """distrax.Inverse: swaps forward and inverse (src/real_nvp_toy_densities.py:104)."""

    def __init__(self, bijector: Bijector):
        super().__init__(event_ndims_in=bijector.event_ndims_out,
                         event_ndims_out=bijector.event_ndims_in,
                         is_constant_jacobian=bijector.is_constant_jacobian,
                         is_constant_log_det=bijector.is_constant_log_det)
        self._bijector = bijector

    bijector = property(lambda self: self._bijector)

    def forward_and_log_det(self, x):
        return self._bijector.inverse_and_log_det(x)

    def inverse_and_log_det(self, y):
        return self._bijector.forward_and_log_det(y)


def lower_for_log_prob(bijector: Bijector):
    """(Chain, direction) such that `bijector.inverse_and_log_det` is that chain walk."""
    inverted = False
    b = bijector
    while isinstance(b, Inverse):
        inverted = not inverted
        b = b.bijector
    if isinstance(b, MaskedCoupling):
        b = b._as_chain()
    if not isinstance(b, Chain):
        raise NotImplementedError(
            f"{type(b).__name__} cannot be lowered to the fused log_prob kernel (no CPU fallback)")
    # bijector.inverse == chain.forward when the chain sits under an odd number of Inverse
    return b, (_lib.DIR_CHAIN_FORWARD if inverted else _lib.DIR_CHAIN_INVERSE)
