"""Device-side coupling chain: a thin object over the nf_flow handle of libnf_hip.

Everything numeric happens in the HIP library; this module only validates
shapes, packs parameters into the ABI's layer-major float32 layout and moves
NumPy arrays across the boundary.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import NFError, check

LayerParams = Sequence[Tuple[np.ndarray, np.ndarray]]  # [(W [in,out], b [out]), ...]


@dataclass(frozen=True)
class FlowConfig:
    """What `make_flow_model` fixes in the reference."""

    kind: int                      # _lib.KIND_*
    event_dim: int
    num_layers: int
    hidden: Tuple[int, ...]
    activation: int = _lib.ACT_RELU
    first_mask_parity: int = 1     # masked kinds: reference value 1; split kind: 0
    base: int = _lib.BASE_STD_NORMAL
    rqs_bins: int = 0
    rqs_range_min: float = 0.0
    rqs_range_max: float = 1.0
    rqs_min_bin_size: float = 1e-4
    rqs_min_knot_slope: float = 1e-4

    def to_desc(self) -> _lib.FlowDesc:
        if len(self.hidden) < 1 or len(self.hidden) > _lib.NF_MAX_HIDDEN:
            raise ValueError(f"conditioner needs 1..{_lib.NF_MAX_HIDDEN} hidden layers, got {len(self.hidden)}")
        d = _lib.FlowDesc()
        d.abi_version = _lib.NF_ABI_VERSION
        d.kind = self.kind
        d.event_dim = self.event_dim
        d.num_layers = self.num_layers
        d.num_hidden = len(self.hidden)
        for j, h in enumerate(self.hidden):
            d.hidden[j] = int(h)
        d.activation = self.activation
        d.first_mask_parity = self.first_mask_parity
        d.base = self.base
        d.rqs_bins = self.rqs_bins
        d.rqs_range_min = self.rqs_range_min
        d.rqs_range_max = self.rqs_range_max
        d.rqs_min_bin_size = self.rqs_min_bin_size
        d.rqs_min_knot_slope = self.rqs_min_knot_slope
        return d

    def linear_shapes(self) -> List[Tuple[int, int]]:
        if self.kind == _lib.KIND_AFFINE_SPLIT:
            first_in, last_out = self.event_dim // 2, self.event_dim
        elif self.kind == _lib.KIND_AFFINE_MASKED:
            first_in, last_out = self.event_dim, 2
        else:
            first_in, last_out = self.event_dim, self.event_dim * (3 * self.rqs_bins + 1)
        sizes = [first_in, *self.hidden, last_out]
        return list(zip(sizes[:-1], sizes[1:]))


def pack_weights(config: FlowConfig, layers: Sequence[LayerParams]) -> np.ndarray:
    """Layer-major float32 vector in the order nf_flow_set_weights documents."""
    shapes = config.linear_shapes()
    if len(layers) != config.num_layers:
        raise ValueError(f"expected parameters for {config.num_layers} layers, got {len(layers)}")
    chunks = []
    for l, layer in enumerate(layers):
        if len(layer) != len(shapes):
            raise ValueError(f"layer {l}: expected {len(shapes)} Linear modules, got {len(layer)}")
        for j, ((W, b), (fi, fo)) in enumerate(zip(layer, shapes)):
            W = np.asarray(W, np.float32)
            b = np.asarray(b, np.float32)
            if W.shape != (fi, fo) or b.shape != (fo,):
                raise ValueError(f"layer {l} linear {j}: expected W {(fi, fo)} b {(fo,)}, got {W.shape} {b.shape}")
            chunks.append(W.ravel())
            chunks.append(b.ravel())
    return np.ascontiguousarray(np.concatenate(chunks), np.float32)


class DeviceArray:
    """float32 array resident in HBM (row-major), owned by this object."""

    def __init__(self, shape: Sequence[int]):
        _lib.require_device()
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        p = C.c_void_p()
        check(_lib.load().nf_malloc(C.byref(p), self.size * 4))
        self._ptr = p

    @property
    def ptr(self) -> C.c_void_p:
        return self._ptr

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a, np.float32)
        out = cls(a.shape)
        check(_lib.load().nf_memcpy_h2d(out.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return out

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.shape, np.float32)
        check(_lib.load().nf_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def rows(self, begin: int, end: int) -> "DeviceView":
        return DeviceView(self, begin, end)

    def free(self) -> None:
        if self._ptr is not None and self._ptr.value:
            _lib.load().nf_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """Rows [begin, end) of a DeviceArray (no copy).  begin*row_bytes must keep
    16-byte alignment, which the library checks."""

    def __init__(self, base: DeviceArray, begin: int, end: int):
        if not (0 <= begin <= end <= base.shape[0]):
            raise ValueError("row range outside the array")
        self.base = base
        self.shape = (end - begin,) + base.shape[1:]
        row = int(np.prod(base.shape[1:])) if len(base.shape) > 1 else 1
        self.ptr = C.c_void_p(base.ptr.value + begin * row * 4)

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.shape, np.float32)
        check(_lib.load().nf_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out


ArrayLike = Union[np.ndarray, DeviceArray, DeviceView]


def _is_device(a) -> bool:
    return isinstance(a, (DeviceArray, DeviceView))


class CouplingFlow:
    """A chain of coupling layers living on one GPU (opaque nf_flow handle)."""

    def __init__(self, config: FlowConfig, layers: Optional[Sequence[LayerParams]] = None):
        _lib.require_device()
        self.config = config
        self._lib = _lib.load()
        self._h = C.c_void_p()
        desc = config.to_desc()
        check(self._lib.nf_flow_create(C.byref(desc), C.byref(self._h)))
        if layers is not None:
            self.set_weights(layers)

    # -- lifetime ------------------------------------------------------------
    def close(self) -> None:
        if self._h is not None and self._h.value:
            self._lib.nf_flow_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def kernel_name(self) -> str:
        buf = C.create_string_buffer(64)
        check(self._lib.nf_flow_kernel_name(self._h, buf, 64))
        return buf.value.decode()

    def set_weights(self, layers: Sequence[LayerParams]) -> None:
        packed = pack_weights(self.config, layers)
        check(self._lib.nf_flow_set_weights(self._h, packed.ctypes.data_as(C.c_void_p), packed.size))

    def synchronize(self) -> None:
        check(self._lib.nf_flow_synchronize(self._h))

    def set_option(self, name: str, value: str) -> None:
        """nf_flow_set_option: e.g. ("hidden_matmul", "bf16x3") for the split-bf16 hidden Linear."""
        check(self._lib.nf_flow_set_option(self._h, name.encode(), value.encode()))

    def get_option(self, name: str) -> str:
        buf = C.create_string_buffer(64)
        check(self._lib.nf_flow_get_option(self._h, name.encode(), buf, 64))
        return buf.value.decode()

    # -- helpers -------------------------------------------------------------
    def _rows_of(self, x: ArrayLike) -> int:
        D = self.config.event_dim
        if len(x.shape) != 2 or x.shape[1] != D:
            raise ValueError(f"expected a [rows, {D}] sample matrix, got shape {tuple(x.shape)}")
        return int(x.shape[0])

    @staticmethod
    def _host(x) -> np.ndarray:
        return np.ascontiguousarray(x, np.float32)

    # -- hot path --------------------------------------------------------------
    def log_prob(self, x: ArrayLike, out: Optional[ArrayLike] = None):
        """[rows, D] -> [rows].  Device in -> device out (enqueued, not synchronised)."""
        rows = self._rows_of(x)
        if _is_device(x):
            out = out if out is not None else DeviceArray((rows,))
            check(self._lib.nf_log_prob(self._h, x.ptr, rows, out.ptr, _lib.MEM_DEVICE))
            return out
        xh = self._host(x)
        res = np.empty(rows, np.float32)
        check(self._lib.nf_log_prob(self._h, xh.ctypes.data_as(C.c_void_p), rows,
                                    res.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
        return res

    def chain_apply(self, direction: int, x: ArrayLike, layer_begin: int = 0,
                    layer_end: Optional[int] = None, want_y: bool = True, want_logdet: bool = True):
        """Walk layers [layer_begin, layer_end) in `direction` (see nf_direction).
        Returns (y [rows, D] or None, logdet [rows] or None)."""
        rows = self._rows_of(x)
        D = self.config.event_dim
        layer_end = self.config.num_layers if layer_end is None else layer_end
        if _is_device(x):
            y = DeviceArray((rows, D)) if want_y else None
            ld = DeviceArray((rows,)) if want_logdet else None
            check(self._lib.nf_chain_apply(self._h, direction, layer_begin, layer_end, x.ptr, rows,
                                           y.ptr if y else None, ld.ptr if ld else None, _lib.MEM_DEVICE))
            return y, ld
        xh = self._host(x)
        y = np.empty((rows, D), np.float32) if want_y else None
        ld = np.empty(rows, np.float32) if want_logdet else None
        check(self._lib.nf_chain_apply(
            self._h, direction, layer_begin, layer_end, xh.ctypes.data_as(C.c_void_p), rows,
            y.ctypes.data_as(C.c_void_p) if want_y else None,
            ld.ctypes.data_as(C.c_void_p) if want_logdet else None, _lib.MEM_HOST))
        return y, ld

    def nll(self, x: ArrayLike, rows_global: Optional[int] = None) -> float:
        """-mean(log_prob); with a communicator attached, over all ranks' rows."""
        rows = self._rows_of(x)
        rows_global = rows if rows_global is None else int(rows_global)
        res = C.c_double()
        if _is_device(x):
            check(self._lib.nf_nll(self._h, x.ptr, rows, rows_global, None, C.byref(res), _lib.MEM_DEVICE))
        else:
            xh = self._host(x)
            check(self._lib.nf_nll(self._h, xh.ctypes.data_as(C.c_void_p), rows, rows_global, None,
                                   C.byref(res), _lib.MEM_HOST))
        return res.value

    def nll_enqueue(self, x_dev, rows_global: int, slot: int, out_log_prob=None) -> None:
        rows = self._rows_of(x_dev)
        check(self._lib.nf_nll_enqueue(self._h, x_dev.ptr, rows, int(rows_global),
                                       out_log_prob.ptr if out_log_prob is not None else None, slot))

    def nll_result(self, slot: int) -> float:
        res = C.c_double()
        check(self._lib.nf_nll_result(self._h, slot, C.byref(res)))
        return res.value

    # -- timing (HIP events on this handle's stream) ----------------------------
    def timer_start(self) -> None:
        check(self._lib.nf_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self._lib.nf_timer_stop(self._h, C.byref(ms)))
        return ms.value

    # -- training step (masked affine flows on the fused kernel) ------------------
    def unpack_weights(self, packed: np.ndarray) -> List[List[Tuple[np.ndarray, np.ndarray]]]:
        """Layer-major float32 vector (nf_flow_set_weights layout) -> per-layer [(W, b), ...]."""
        shapes = self.config.linear_shapes()
        out, pos = [], 0
        for _ in range(self.config.num_layers):
            layer = []
            for fi, fo in shapes:
                W = packed[pos:pos + fi * fo].reshape(fi, fo).copy(); pos += fi * fo
                b = packed[pos:pos + fo].copy(); pos += fo
                layer.append((W, b))
            out.append(layer)
        if pos != packed.size:
            raise ValueError("packed vector has the wrong length")
        return out

    def _weight_count(self) -> int:
        return sum(fi * fo + fo for fi, fo in self.config.linear_shapes()) * self.config.num_layers

    def get_weights(self) -> List[List[Tuple[np.ndarray, np.ndarray]]]:
        n = self._weight_count()
        packed = np.empty(n, np.float32)
        check(self._lib.nf_flow_get_weights(self._h, packed.ctypes.data_as(C.c_void_p), n))
        return self.unpack_weights(packed)

    def nll_grad(self, x: ArrayLike, rows_global: Optional[int] = None):
        """(NLL, d NLL / d weights as per-layer [(dW, db), ...]) -- jax.value_and_grad(loss)."""
        rows = self._rows_of(x)
        rows_global = rows if rows_global is None else int(rows_global)
        n = self._weight_count()
        grad = np.empty(n, np.float32)
        nll = C.c_double()
        if _is_device(x):
            gd = DeviceArray((n,))
            check(self._lib.nf_nll_grad(self._h, x.ptr, rows, rows_global, C.byref(nll), gd.ptr, _lib.MEM_DEVICE))
            grad = gd.to_numpy()
        else:
            xh = self._host(x)
            check(self._lib.nf_nll_grad(self._h, xh.ctypes.data_as(C.c_void_p), rows, rows_global, C.byref(nll),
                                        grad.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
        return nll.value, self.unpack_weights(grad)

    def train_step(self, x: ArrayLike, optimizer: "_lib.Optimizer", rows_global: Optional[int] = None,
                   want_nll: bool = True) -> Optional[float]:
        """One optimiser step on the handle's weights; returns the NLL before the step."""
        rows = self._rows_of(x)
        rows_global = rows if rows_global is None else int(rows_global)
        nll = C.c_double()
        out = C.byref(nll) if want_nll else None
        if _is_device(x):
            check(self._lib.nf_train_step(self._h, x.ptr, rows, rows_global, C.byref(optimizer), out, _lib.MEM_DEVICE))
        else:
            xh = self._host(x)
            check(self._lib.nf_train_step(self._h, xh.ctypes.data_as(C.c_void_p), rows, rows_global,
                                          C.byref(optimizer), out, _lib.MEM_HOST))
        return nll.value if want_nll else None

    def attach_comm(self, comm) -> None:
        check(self._lib.nf_flow_attach_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm

    def collective_count(self) -> int:
        """Collectives this handle has issued (nf_flow_collective_count): must agree across ranks."""
        n = C.c_uint64(0)
        check(self._lib.nf_flow_collective_count(self._h, C.byref(n)))
        return int(n.value)
