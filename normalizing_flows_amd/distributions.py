"""distrax-shaped distributions used by the reference's flows.

`Transformed(Independent(Normal(0,1)), Inverse(Chain(layers)))`
(src/real_nvp_toy_densities.py:104-110) and the Uniform(0,1) base of the MNIST
flow (src/maf_mnist.py:84-89).  `Transformed.log_prob` is ONE fused launch of
libnf_hip (chain walk + base log-density); `sample` draws the base on the GPU
(Philox4x32-10) and walks the chain in the same call.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import check
from .bijectors import Bijector, Chain, Inverse, MaskedCoupling, _flatten_rows, lower_for_log_prob
from .flow import DeviceArray, DeviceView

IntLike = Union[int, Sequence[int]]


def _as_shape(sample_shape: IntLike) -> Tuple[int, ...]:
    if isinstance(sample_shape, (int, np.integer)):
        return (int(sample_shape),)
    return tuple(int(s) for s in sample_shape)


def _seed_u64(seed) -> int:
    """An int, or a key array as the reference passes (`next(prng_seq)`): its words folded to 64 bits."""
    words = np.asarray(seed).astype(np.uint64).ravel()
    out = 0
    for w in words:
        out = (out * 0x9E3779B97F4A7C15 + int(w)) & 0xFFFFFFFFFFFFFFFF
    return out


def _device_base_sample(base: int, event_shape: Tuple[int, ...], seed, sample_shape: IntLike) -> np.ndarray:
    """Philox4x32-10 draw on the GPU (nf_base_sample): [*sample_shape, *event_shape] float32."""
    shape = _as_shape(sample_shape)
    rows = int(np.prod(shape)) if shape else 1
    D = int(np.prod(event_shape)) if event_shape else 1
    lib = _lib.load()
    _lib.require_device()
    out = np.empty((rows, D), np.float32)
    check(lib.nf_base_sample(base, D, C.c_uint64(_seed_u64(seed)), rows, out.ctypes.data_as(C.c_void_p),
                             _lib.MEM_HOST))
    return out.reshape(shape + tuple(event_shape))


def _device_base_log_prob(base: int, event_shape: Tuple[int, ...], value):
    D = int(np.prod(event_shape)) if event_shape else 1
    rows, batch = _flatten_rows(value, event_shape if event_shape else ())
    lib = _lib.load()
    _lib.require_device()
    if isinstance(rows, (DeviceArray, DeviceView)):
        out = DeviceArray((rows.shape[0],))
        check(lib.nf_base_log_prob(base, D, rows.ptr, rows.shape[0], out.ptr, _lib.MEM_DEVICE))
        return out
    out = np.empty(rows.shape[0], np.float32)
    check(lib.nf_base_log_prob(base, D, rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                               out.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
    return out.reshape(batch)


class Distribution:
    event_shape: Tuple[int, ...] = ()
    batch_shape: Tuple[int, ...] = ()

    def _base_code(self) -> Optional[int]:
        """_lib.BASE_* when this is one of the base densities the kernels implement."""
        return None

    def log_prob(self, value):
        raise NotImplementedError

    def sample(self, *, seed, sample_shape: IntLike = ()):
        raise NotImplementedError


class Normal(Distribution):
    """distrax.Normal(loc, scale) (src/real_nvp_toy_densities.py:106)."""

    def __init__(self, loc, scale):
        self.loc = np.asarray(loc, np.float32)
        self.scale = np.asarray(scale, np.float32)
        self.batch_shape = np.broadcast(self.loc, self.scale).shape
        self.event_shape = ()

    def is_standard(self) -> bool:
        return bool(np.all(self.loc == 0) and np.all(self.scale == 1))

    def sample(self, *, seed, sample_shape: IntLike = ()):
        if not self.is_standard():
            raise NotImplementedError("only Normal(0, 1) draws have a kernel (no CPU fallback)")
        return _device_base_sample(_lib.BASE_STD_NORMAL, tuple(self.batch_shape), seed, sample_shape)

    def log_prob(self, value):
        raise NotImplementedError("element-wise Normal.log_prob is off the accelerated path; wrap it in "
                                  "Independent(...) to get the per-event base density on the GPU")


class Uniform(Distribution):
    """distrax.Uniform(low, high) (src/maf_mnist.py:85)."""

    def __init__(self, low=0.0, high=1.0):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.batch_shape = np.broadcast(self.low, self.high).shape
        self.event_shape = ()

    def is_standard(self) -> bool:
        return bool(np.all(self.low == 0) and np.all(self.high == 1))

    def sample(self, *, seed, sample_shape: IntLike = ()):
        if not self.is_standard():
            raise NotImplementedError("only Uniform(0, 1) draws have a kernel (no CPU fallback)")
        return _device_base_sample(_lib.BASE_UNIFORM01, tuple(self.batch_shape), seed, sample_shape)

    def log_prob(self, value):
        raise NotImplementedError("element-wise Uniform.log_prob is off the accelerated path; wrap it in "
                                  "Independent(...)")


class Independent(Distribution):
    """distrax.Independent: the last `reinterpreted_batch_ndims` batch dims become the event."""

    def __init__(self, distribution: Distribution, reinterpreted_batch_ndims: Optional[int] = None):
        nb = len(distribution.batch_shape)
        n = nb if reinterpreted_batch_ndims is None else int(reinterpreted_batch_ndims)
        if n < 0 or n > nb:
            raise ValueError(f"reinterpreted_batch_ndims {n} outside 0..{nb}")
        self.distribution = distribution
        self.reinterpreted_batch_ndims = n
        self.batch_shape = tuple(distribution.batch_shape[:nb - n])
        self.event_shape = tuple(distribution.batch_shape[nb - n:]) + tuple(distribution.event_shape)

    def _base_code(self) -> Optional[int]:
        d = self.distribution
        if self.batch_shape:
            return None
        if isinstance(d, Normal) and d.is_standard():
            return _lib.BASE_STD_NORMAL
        if isinstance(d, Uniform) and d.is_standard():
            return _lib.BASE_UNIFORM01
        return None

    def sample(self, *, seed, sample_shape: IntLike = ()):
        return self.distribution.sample(seed=seed, sample_shape=sample_shape)

    def log_prob(self, value):
        code = self._base_code()
        if code is None:
            raise NotImplementedError("only Independent(Normal(0,1)) and Independent(Uniform(0,1)) have kernels")
        return _device_base_log_prob(code, self.event_shape, value)


class MultivariateNormalDiag(Distribution):
    """tfd / distrax MultivariateNormalDiag (src/real_nvp.py:52-54,66-68)."""

    def __init__(self, loc=None, scale_diag=None):
        if loc is None and scale_diag is None:
            raise ValueError("at least one of loc and scale_diag must be given")
        loc = np.zeros_like(np.asarray(scale_diag, np.float32)) if loc is None else np.asarray(loc, np.float32)
        scale_diag = np.ones_like(loc) if scale_diag is None else np.asarray(scale_diag, np.float32)
        if loc.ndim != 1 or scale_diag.shape != loc.shape:
            raise ValueError("loc and scale_diag must be vectors of one length")
        self.loc, self.scale_diag = loc, scale_diag
        self.event_shape = loc.shape
        self.batch_shape = ()

    def _base_code(self) -> Optional[int]:
        if np.all(self.loc == 0) and np.all(self.scale_diag == 1):
            return _lib.BASE_STD_NORMAL
        return None

    def sample(self, *, seed, sample_shape: IntLike = ()):
        if self._base_code() is None:
            raise NotImplementedError("only MultivariateNormalDiag(0, I) draws have a kernel")
        return _device_base_sample(_lib.BASE_STD_NORMAL, tuple(self.event_shape), seed, sample_shape)

    def log_prob(self, value):
        code = self._base_code()
        if code is None:
            raise NotImplementedError("only MultivariateNormalDiag(0, I) has a kernel")
        return _device_base_log_prob(code, self.event_shape, value)


class Transformed(Distribution):
    """distrax.Transformed(distribution, bijector) (src/real_nvp_toy_densities.py:110)."""

    def __init__(self, distribution: Distribution, bijector: Bijector):
        if len(distribution.event_shape) != bijector.event_ndims_in:
            raise ValueError(f"base event has {len(distribution.event_shape)} dims, the bijector expects "
                             f"{bijector.event_ndims_in}")
        self.distribution = distribution
        self.bijector = bijector
        self.event_shape = tuple(distribution.event_shape)
        self.batch_shape = tuple(distribution.batch_shape)

    def _lowered(self):
        code = self.distribution._base_code()
        if code is None:
            raise NotImplementedError(
                "base distribution has no fused kernel (Independent(Normal(0,1)), "
                "Independent(Uniform(0,1)) and MultivariateNormalDiag(0,I) do); no CPU fallback")
        chain, direction = lower_for_log_prob(self.bijector)
        flow, event_shape = chain.lowered(code)
        if tuple(event_shape) != self.event_shape:
            raise ValueError(f"bijector event shape {event_shape} differs from the base's {self.event_shape}")
        return flow, direction

    def set_option(self, name: str, value: str) -> None:
        """nf_flow_set_option on the device flow behind this model, e.g. ("hidden_matmul", "bf16x3")
        (nothing in distrax corresponds; the environment variable NF_HIDDEN_MATMUL sets the default)."""
        flow, _ = self._lowered()
        flow.set_option(name, value)

    def log_prob(self, value):
        """x, ildj = bijector.inverse_and_log_det(value); base.log_prob(x) + ildj -- one launch."""
        from . import haiku_like
        frame = haiku_like.current_frame()
        if frame is not None and frame.mode in ("init", "plan"):
            # hk.transform(...).init: every parameter exists once the model is built; nothing is evaluated
            _, batch = _flatten_rows(np.asarray(value, np.float32), self.event_shape)
            return np.zeros(batch, np.float32)
        flow, direction = self._lowered()
        rows, batch = _flatten_rows(value, self.event_shape)
        lib = _lib.load()
        if isinstance(rows, (DeviceArray, DeviceView)):
            out = DeviceArray((rows.shape[0],))
            check(lib.nf_chain_log_prob(flow._h, direction, rows.ptr, rows.shape[0], out.ptr, _lib.MEM_DEVICE))
            return out
        out = np.empty(rows.shape[0], np.float32)
        check(lib.nf_chain_log_prob(flow._h, direction, rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                                    out.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
        return out.reshape(batch)

    def nll(self, value) -> float:
        """-mean(log_prob(value)) with a float64, fixed-order reduction on the GPU
        (loss_fn / eval_fn, src/real_nvp_toy_densities.py:139-151)."""
        from . import haiku_like
        frame = haiku_like.current_frame()
        if frame is not None and frame.mode in ("init", "plan"):
            return 0.0
        flow, direction = self._lowered()
        if direction != _lib.DIR_CHAIN_FORWARD:
            raise NotImplementedError("nll is fused for Transformed(base, Inverse(Chain(...))) only")
        rows, _ = _flatten_rows(value, self.event_shape)
        return flow.nll(rows)

    def sample(self, *, seed, sample_shape: IntLike = ()):
        """Base draw (Philox4x32-10) and bijector.forward in one device call (nf_sample); identical
        to `bijector.forward(distribution.sample(seed=seed, sample_shape=sample_shape))`."""
        from . import haiku_like
        frame = haiku_like.current_frame()
        if frame is not None and frame.mode in ("init", "plan"):
            return np.zeros(_as_shape(sample_shape) + self.event_shape, np.float32)
        flow, direction = self._lowered()
        if direction != _lib.DIR_CHAIN_FORWARD:
            return self.bijector.forward(self.distribution.sample(seed=seed, sample_shape=sample_shape))
        shape = _as_shape(sample_shape)
        rows = int(np.prod(shape)) if shape else 1
        D = int(np.prod(self.event_shape))
        out = np.empty((rows, D), np.float32)
        check(_lib.load().nf_sample(flow._h, C.c_uint64(_seed_u64(seed)), rows, out.ctypes.data_as(C.c_void_p),
                                    _lib.MEM_HOST))
        return out.reshape(shape + self.event_shape)

    def sample_and_log_prob(self, *, seed, sample_shape: IntLike = ()):
        x = self.sample(seed=seed, sample_shape=sample_shape)
        return x, self.log_prob(x)
