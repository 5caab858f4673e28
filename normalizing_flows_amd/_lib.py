"""ctypes binding of libnf_hip.so (C ABI: include/nf_hip.h).

The HIP library is the only compute back end of this package.  If it is missing
or no gfx950 device is visible, every compute call raises: there is no CPU
fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

NF_ABI_VERSION = 1
NF_MAX_HIDDEN = 8
NF_NLL_SLOTS = 64
NF_COMM_ID_BYTES = 128

# nf_kind
KIND_AFFINE_SPLIT, KIND_AFFINE_MASKED, KIND_RQS_MASKED = 0, 1, 2
# nf_activation
ACT_RELU, ACT_LEAKY_RELU = 0, 1
# nf_base
BASE_STD_NORMAL, BASE_UNIFORM01 = 0, 1
# nf_mem
MEM_HOST, MEM_DEVICE = 0, 1
# nf_direction
DIR_CHAIN_FORWARD, DIR_CHAIN_INVERSE, DIR_FIRST_TO_LAST_FORWARD, DIR_LAST_TO_FIRST_INVERSE = 0, 1, 2, 3

_STATUS_NAMES = {
    -1: "NF_ERR_INVALID", -2: "NF_ERR_HIP", -3: "NF_ERR_UNSUPPORTED",
    -4: "NF_ERR_NO_DEVICE", -5: "NF_ERR_COMM", -6: "NF_ERR_STATE",
}


class NFError(RuntimeError):
    """A libnf_hip call failed (HIP error, missing device, unsupported config)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{_STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class Optimizer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("learning_rate", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float)]


OPT_SGD, OPT_ADAM = 0, 1


class FlowDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("kind", C.c_int32),
        ("event_dim", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_hidden", C.c_int32),
        ("hidden", C.c_int32 * NF_MAX_HIDDEN),
        ("activation", C.c_int32),
        ("first_mask_parity", C.c_int32),
        ("base", C.c_int32),
        ("rqs_bins", C.c_int32),
        ("rqs_range_min", C.c_float),
        ("rqs_range_max", C.c_float),
        ("rqs_min_bin_size", C.c_float),
        ("rqs_min_knot_slope", C.c_float),
    ]


# nf_elementwise_kind
ELEM_AFFINE_LOG_SCALE, ELEM_AFFINE_SCALE, ELEM_RQS = 0, 1, 2


class ElementwiseDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("kind", C.c_int32),
        ("rqs_bins", C.c_int32),
        ("rqs_range_min", C.c_float),
        ("rqs_range_max", C.c_float),
        ("rqs_min_bin_size", C.c_float),
        ("rqs_min_knot_slope", C.c_float),
    ]


_c_float_p = C.POINTER(C.c_float)
_void_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks
# against the declarations in include/nf_hip.h
PROTOTYPES = {
    "nf_abi_version": (C.c_int, []),
    "nf_last_error": (C.c_char_p, []),
    "nf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nf_set_device": (C.c_int, [C.c_int]),
    "nf_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "nf_device_synchronize": (C.c_int, []),
    "nf_malloc": (C.c_int, [_void_pp, C.c_size_t]),
    "nf_free": (C.c_int, [C.c_void_p]),
    "nf_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nf_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nf_flow_weight_count": (C.c_int, [C.POINTER(FlowDesc), C.POINTER(C.c_size_t)]),
    "nf_flow_create": (C.c_int, [C.POINTER(FlowDesc), _void_pp]),
    "nf_flow_destroy": (None, [C.c_void_p]),
    "nf_flow_set_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nf_flow_kernel_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "nf_flow_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "nf_flow_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "nf_flow_synchronize": (C.c_int, [C.c_void_p]),
    "nf_log_prob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "nf_chain_log_prob": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "nf_base_log_prob": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "nf_elementwise_apply": (C.c_int, [C.POINTER(ElementwiseDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int]),
    "nf_base_sample": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_int]),
    "nf_density_sample": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_int]),
    "nf_sample": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_int]),
    "nf_chain_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_int]),
    "nf_nll": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                         C.POINTER(C.c_double), C.c_int]),
    "nf_nll_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]),
    "nf_nll_result": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "nf_nll_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_void_p,
                              C.c_int]),
    "nf_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_double),
                                C.c_int]),
    "nf_flow_get_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nf_timer_start": (C.c_int, [C.c_void_p]),
    "nf_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "nf_debug_read_f2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "nf_debug_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int]),
    "nf_comm_preflight": (C.c_int, []),
    "nf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "nf_comm_init_rank": (C.c_int, [_void_pp, C.c_int, C.c_int, C.c_void_p]),
    "nf_comm_destroy": (C.c_int, [C.c_void_p]),
    "nf_flow_attach_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nf_flow_collective_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

_LIB: Optional[C.CDLL] = None


def library_path() -> str:
    return os.environ.get(
        "NF_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnf_hip.so"))


def load() -> C.CDLL:
    """Load libnf_hip.so (once) and attach prototypes.  Raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise NFError(-4, f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
                          f"g.build()'` (or `make -C normalizing_flows_amd/csrc`); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.nf_abi_version() != NF_ABI_VERSION:
        raise NFError(-1, f"{path} speaks ABI {lib.nf_abi_version()}, this package {NF_ABI_VERSION}")
    _LIB = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().nf_last_error()
        raise NFError(status, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    check(load().nf_device_count(C.byref(n)))
    return n.value


def require_device() -> None:
    if device_count() < 1:
        raise NFError(-4, "no HIP device visible; normalizing_flows_amd computes only on MI355X (gfx950)")
