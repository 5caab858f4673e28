"""The reference's toy-density data generators (src/densities/*.py), drawn on the GPU.

Same function names and arguments (`seed, batch_size, num_batches`); each is a generator of
[batch_size, 2] float32 batches.  The reference samples with tfp on the host, one un-jitted call
per training step; here a batch is one kernel launch
(`nf_density_sample`, Philox4x32-10; the JAX threefry stream is not reproduced).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator

import numpy as np

from . import _lib
from ._lib import check
from .distributions import _seed_u64
from .flow import DeviceArray

DENSITIES = {"banana": 0, "energy1": 1, "energy2": 2, "energy3": 3, "energy4": 4, "gaussian_blobs": 5}


def sample_density(density: str, seed, rows: int, device: bool = False):
    """One batch: NumPy [rows, 2] float32, or a DeviceArray left resident in HBM."""
    if density not in DENSITIES:
        raise ValueError(f"unknown density {density!r}; one of {sorted(DENSITIES)}")
    lib = _lib.load()
    _lib.require_device()
    if device:
        out = DeviceArray((rows, 2))
        check(lib.nf_density_sample(DENSITIES[density], C.c_uint64(_seed_u64(seed)), rows, out.ptr, _lib.MEM_DEVICE))
        check(lib.nf_device_synchronize())
        return out
    out = np.empty((rows, 2), np.float32)
    check(lib.nf_density_sample(DENSITIES[density], C.c_uint64(_seed_u64(seed)), rows,
                                out.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
    return out


def _make_dataset(density: str, seed, batch_size: int, num_batches: int, device: bool) -> Iterator:
    base = _seed_u64(seed)
    for b in range(num_batches):
        yield sample_density(density, (base + 0x9E3779B97F4A7C15 * (b + 1)) & 0xFFFFFFFFFFFFFFFF, batch_size, device)


def make_dataset_banana(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/banana.py:31-41."""
    return _make_dataset("banana", seed, batch_size, num_batches, device)


def make_dataset_energy_1(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/energies.py:47-66."""
    return _make_dataset("energy1", seed, batch_size, num_batches, device)


def make_dataset_energy_2(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/energies.py:69-82."""
    return _make_dataset("energy2", seed, batch_size, num_batches, device)


def make_dataset_energy_3(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/energies.py:85-104."""
    return _make_dataset("energy3", seed, batch_size, num_batches, device)


def make_dataset_energy_4(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/energies.py:107-127."""
    return _make_dataset("energy4", seed, batch_size, num_batches, device)


def make_dataset_gaussian_blobs(seed, batch_size: int = 8, num_batches: int = 1, device: bool = False):
    """src/densities/gaussian_blobs.py:34-59 (8 blobs, r = 2, sigma 0.3)."""
    return _make_dataset("gaussian_blobs", seed, batch_size, num_batches, device)


def get_density(density: str):
    """src/real_nvp_toy_densities.py:39-53."""
    return {"banana": make_dataset_banana, "gaussian_blobs": make_dataset_gaussian_blobs,
            "energy1": make_dataset_energy_1, "energy2": make_dataset_energy_2,
            "energy3": make_dataset_energy_3, "energy4": make_dataset_energy_4}[density]
