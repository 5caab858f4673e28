"""ctypes loader of oracle/libnf_oracle.so (test infrastructure, see nf_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libnf_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE], check=True)
        _LIB = C.CDLL(path)
    return _LIB


def _flat(layers):
    return np.ascontiguousarray(np.concatenate(
        [np.concatenate([np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()])
         for layer in layers for W, b in layer]), np.float32)


def affine_masked_log_prob(spec, x, threads=0):
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    hidden = np.array([W.shape[1] for W, _ in spec.layers[0][:-1]], np.int32)
    w = _flat(spec.layers)
    out = np.empty(x.shape[0], np.float32)
    parity = 1 if spec.masks[0][1 % spec.event_size] else 0
    rc = lib.nfo_affine_masked_log_prob(w.ctypes.data_as(C.c_void_p), spec.event_size, len(spec.layers),
                                        hidden.ctypes.data_as(C.c_void_p), len(hidden), parity,
                                        x.ctypes.data_as(C.c_void_p), C.c_int64(x.shape[0]),
                                        out.ctypes.data_as(C.c_void_p), threads)
    assert rc == 0
    return out


def affine_split_log_prob(params, flips, y, threads=0):
    lib = load()
    y = np.ascontiguousarray(y, np.float32)
    hidden = np.array([np.asarray(W).shape[1] for W, _ in params[0][:-1]], np.int32)
    w = _flat([[(W, b) for W, b in net] for net in params])
    out = np.empty(y.shape[0], np.float32)
    rc = lib.nfo_affine_split_log_prob(w.ctypes.data_as(C.c_void_p), y.shape[1], len(params),
                                       hidden.ctypes.data_as(C.c_void_p), len(hidden), int(flips[0]),
                                       y.ctypes.data_as(C.c_void_p), C.c_int64(y.shape[0]),
                                       out.ctypes.data_as(C.c_void_p), threads)
    assert rc == 0
    return out


def max_threads():
    return load().nfo_max_threads()
