"""Host-side data preparation of the image flow.

`prepare_data` is the function of the same name in src/maf_mnist.py:101-106: uint8 pixels to
float32, optional uniform dequantisation noise in [0, 1), division by 256.  The reference draws the
noise with `jax.random.uniform(prng_key, ...)`; here the key seeds NumPy's Philox generator (the
values differ from jax's -- any U[0,1) noise dequantises -- the evaluation path takes no key and
is exact)."""
from typing import Mapping, Optional, Union

import numpy as np


def prepare_data(batch: Union[Mapping[str, np.ndarray], np.ndarray], prng_key: Optional[int] = None) -> np.ndarray:
    image = batch["image"] if isinstance(batch, Mapping) else batch
    data = np.asarray(image).astype(np.float32)
    if prng_key is not None:
        key = int(np.asarray(prng_key).ravel()[-1]) & 0xFFFFFFFFFFFFFFFF
        noise = np.random.Generator(np.random.Philox(key)).random(data.shape, dtype=np.float32)
        data += noise  # dequantise {0, ..., 255} with U[0, 1)
    return data / np.float32(256.0)
