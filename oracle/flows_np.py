"""CPU oracle: NumPy restatement of the reference's coupling-flow log_prob path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``normalizing_flows_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
(SURVEY.md §4), and jax / distrax / haiku / tfp are not installed in this image,
so the reference itself cannot be executed here.  This file restates

* variant A from the reference source text (``src/real_nvp.py``), function by
  function, and
* variants B / C / D from the reference call sites
  (``src/real_nvp_toy_densities.py``, ``src/maf_toy_densities.py``,
  ``src/maf_mnist.py``) plus the published semantics of the third-party
  libraries they call (distrax ~0.1.x, dm-haiku; un-pinned in
  ``environment.yml:17-18``).  The rational-quadratic-spline details are
  restated from the distrax source as published, unverified in-container.

It is pinned instead by the known-answer tests of SURVEY.md §8c
(``tests/test_oracle_kat.py``).

Every function takes a ``dtype`` (np.float32 reproduces the reference's
arithmetic type, np.float64 is the high-precision arbiter used for golden
vectors).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

LEAKY_SLOPE = 0.01  # jax.nn.leaky_relu default negative_slope (src/real_nvp.py:167)


# --------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------
def relu(v):
    return np.maximum(v, 0)


def leaky_relu(v):
    return np.where(v >= 0, v, v * v.dtype.type(LEAKY_SLOPE))


_ACT = {"relu": relu, "leaky_relu": leaky_relu}


# --------------------------------------------------------------------------
# Variant A  --  src/real_nvp.py
# --------------------------------------------------------------------------
def a_shift_and_log_scale(net_params, x1, dtype=np.float64):
    """src/real_nvp.py:161-171.  ``net_params = [[W,b], ..., [W_last,b_last]]``.

    Hidden layers use leaky_relu, the last layer is linear and its two output
    columns are (shift, log_scale).
    """
    act = np.asarray(x1, dtype)
    for W, b in net_params[:-1]:
        act = leaky_relu(act @ np.asarray(W, dtype) + np.asarray(b, dtype))
    W, b = net_params[-1]
    out = act @ np.asarray(W, dtype) + np.asarray(b, dtype)
    half = out.shape[1] // 2
    return out[:, :half], out[:, half:]


def _a_split(v, flip):
    d = v.shape[-1] // 2
    lo, hi = v[:, :d], v[:, d:]
    return (hi, lo) if flip else (lo, hi)


def _a_join(keep, moved, flip):
    return np.concatenate([moved, keep] if flip else [keep, moved], axis=-1)


def a_forward_layer(net_params, x, flip, dtype=np.float64):
    """src/real_nvp.py:71-101 (generative direction of one coupling layer)."""
    x = np.asarray(x, dtype)
    x1, x2 = _a_split(x, flip)
    shift, log_scale = a_shift_and_log_scale(net_params, x1, dtype)
    y2 = x2 * np.exp(log_scale) + shift
    return _a_join(x1, y2, flip)


def a_inverse_layer(net_params, y, flip, dtype=np.float64):
    """src/real_nvp.py:104-134; returns (x, log_scale)."""
    y = np.asarray(y, dtype)
    y1, y2 = _a_split(y, flip)
    shift, log_scale = a_shift_and_log_scale(net_params, y1, dtype)
    x2 = (y2 - shift) * np.exp(-log_scale)
    return _a_join(y1, x2, flip), log_scale


def a_base_log_prob(x, dtype=np.float64):
    """src/real_nvp.py:57-68: MultivariateNormalDiag(0, I).log_prob."""
    x = np.asarray(x, dtype)
    d = x.shape[-1]
    return -0.5 * np.sum(x * x, axis=-1) - dtype(0.5 * d * math.log(2.0 * math.pi))


def a_log_prob_chain(params, flips, y, dtype=np.float64):
    """src/real_nvp.py:137-158, 251-283.

    The closure chain evaluates the LAST layer's inverse first and the base
    density last; every layer contributes ``-sum(log_scale)``.
    """
    v = np.asarray(y, dtype)
    total = np.zeros(v.shape[0], dtype)
    for net_params, flip in zip(reversed(params), reversed(flips)):
        v, log_scale = a_inverse_layer(net_params, v, flip, dtype)
        total = total - np.sum(log_scale, axis=-1)
    return a_base_log_prob(v, dtype) + total


def a_sample_chain_from_z(params, flips, z, dtype=np.float64):
    """src/real_nvp.py:228-248 with the base sample ``z`` supplied by the host."""
    v = np.asarray(z, dtype)
    for net_params, flip in zip(params, flips):
        v = a_forward_layer(net_params, v, flip, dtype)
    return v


def a_nll(params, flips, y, dtype=np.float64):
    """src/real_nvp.py:292-293."""
    return -np.mean(a_log_prob_chain(params, flips, y, dtype))


# --------------------------------------------------------------------------
# haiku-style conditioner used by B / C / D
# --------------------------------------------------------------------------
def conditioner_apply(layer_params, masked_x, dtype=np.float64, activation="relu"):
    """hk.Sequential([Flatten, MLP(hidden, activate_final=True), Linear(P)]).

    ``layer_params = [(W0,b0), ..., (W_last,b_last)]``; all but the last are
    followed by relu (haiku default activation), the last is the zero-initialised
    hk.Linear (src/real_nvp_toy_densities.py:56-71, src/maf_mnist.py:34-49).
    ``masked_x`` is already flattened to [B, D].
    """
    fn = relu if activation == "relu" else leaky_relu   # leaky_relu: not used by the reference's B/C/D
    act = np.asarray(masked_x, dtype)                   # scripts, but the C ABI offers it for every kind
    for W, b in layer_params[:-1]:
        act = fn(act @ np.asarray(W, dtype) + np.asarray(b, dtype))
    W, b = layer_params[-1]
    return act @ np.asarray(W, dtype) + np.asarray(b, dtype)


def alternating_masks(event_size: int, num_layers: int, first_parity: int = 1):
    """src/real_nvp_toy_densities.py:80-82,102: ``arange(D) % 2`` then flipped
    (logical_not) after every layer.  True = pass-through, visible to the
    conditioner.  ``first_parity=1`` is the reference's value."""
    m = (np.arange(event_size) % 2) == first_parity
    out = []
    for _ in range(num_layers):
        out.append(m.copy())
        m = ~m
    return out


# --------------------------------------------------------------------------
# distrax.ScalarAffine
# --------------------------------------------------------------------------
def scalar_affine_fwd(x, shift, log_scale):
    return np.exp(log_scale) * x + shift, np.broadcast_to(log_scale, np.broadcast(x, log_scale).shape)


def scalar_affine_inv(y, shift, log_scale):
    return np.exp(-log_scale) * (y - shift), np.broadcast_to(-log_scale, np.broadcast(y, log_scale).shape)


# --------------------------------------------------------------------------
# distrax.RationalQuadraticSpline
# --------------------------------------------------------------------------
def _softmax(v):
    v = v - np.max(v, axis=-1, keepdims=True)
    e = np.exp(v)
    return e / np.sum(e, axis=-1, keepdims=True)


def _softplus(v):
    return np.logaddexp(v, v.dtype.type(0))


def rqs_knots(params, range_min, range_max, min_bin_size=1e-4, min_knot_slope=1e-4,
              dtype=np.float64):
    """Normalise raw conditioner outputs [..., 3K+1] into knots.

    Returns (x_pos, y_pos, slopes), each [..., K+1].  boundary_slopes =
    'unconstrained' (the distrax default, which is what the reference uses).
    """
    p = np.asarray(params, dtype)
    n = p.shape[-1]
    if n % 3 != 1 or n < 4:
        raise ValueError(f"spline needs 3K+1 parameters, got {n}")
    if not range_min < range_max:
        raise ValueError("range_min must be below range_max")
    K = (n - 1) // 3
    total = dtype(range_max - range_min)
    if K * min_bin_size > total:
        raise ValueError("bins do not fit the range")
    mb = dtype(min_bin_size)
    widths = _softmax(p[..., :K]) * (total - K * mb) + mb
    heights = _softmax(p[..., K:2 * K]) * (total - K * mb) + mb
    lo = np.full(p.shape[:-1] + (1,), range_min, dtype)
    hi = np.full(p.shape[:-1] + (1,), range_max, dtype)
    x_pos = np.concatenate([lo, dtype(range_min) + np.cumsum(widths[..., :-1], axis=-1), hi], axis=-1)
    y_pos = np.concatenate([lo, dtype(range_min) + np.cumsum(heights[..., :-1], axis=-1), hi], axis=-1)
    ms = dtype(min_knot_slope)
    offset = np.log(np.exp(dtype(1) - ms) - dtype(1))
    slopes = _softplus(p[..., 2 * K:] + offset) + ms
    return x_pos, y_pos, slopes


def _pick_bin(v, pos):
    """Index of the bin with pos[k] <= v < pos[k+1]; bin 0 when none matches."""
    inside = (v[..., None] >= pos[..., :-1]) & (v[..., None] < pos[..., 1:])
    any_in = inside.any(axis=-1)
    idx = np.argmax(inside, axis=-1)
    return np.where(any_in, idx, 0)


def _take(a, idx):
    return np.take_along_axis(a, idx[..., None], axis=-1)[..., 0]


def rqs_fwd(x, x_pos, y_pos, slopes):
    """Forward spline + log|dy/dx| with linear tails outside the range."""
    dt = x_pos.dtype
    x = np.asarray(x, dt)
    one, two = dt.type(1), dt.type(2)
    k = _pick_bin(x, x_pos)
    xl, xr = _take(x_pos, k), _take(x_pos, k + 1)
    yl, yr = _take(y_pos, k), _take(y_pos, k + 1)
    sl, sr = _take(slopes, k), _take(slopes, k + 1)
    w = xr - xl
    h = yr - yl
    delta = h / w
    z = np.clip((x - xl) / w, 0, 1)
    zz = z * z
    z1mz = z - zz
    omz2 = (one - z) ** 2
    st = sr + sl - two * delta
    num = h * (delta * zz + sl * z1mz)
    den = delta + st * z1mz
    y = yl + num / den
    logdet = two * np.log(delta) + np.log(sr * zz + two * delta * z1mz + sl * omz2) - two * np.log(den)
    below = x <= x_pos[..., 0]
    above = x >= x_pos[..., -1]
    y = np.where(below, (x - x_pos[..., 0]) * slopes[..., 0] + y_pos[..., 0], y)
    y = np.where(above, (x - x_pos[..., -1]) * slopes[..., -1] + y_pos[..., -1], y)
    logdet = np.where(below, np.log(slopes[..., 0]), logdet)
    logdet = np.where(above, np.log(slopes[..., -1]), logdet)
    return y, logdet


def rqs_inv(y, x_pos, y_pos, slopes):
    """Inverse spline (quadratic root) + log|dx/dy|."""
    dt = x_pos.dtype
    y = np.asarray(y, dt)
    one, two, four = dt.type(1), dt.type(2), dt.type(4)
    k = _pick_bin(y, y_pos)
    xl, xr = _take(x_pos, k), _take(x_pos, k + 1)
    yl, yr = _take(y_pos, k), _take(y_pos, k + 1)
    sl, sr = _take(slopes, k), _take(slopes, k + 1)
    w = xr - xl
    h = yr - yl
    delta = h / w
    u = np.clip((y - yl) / h, 0, 1)
    st = sr + sl - two * delta
    c = -delta * u
    b = sl - st * u
    a = delta - b
    z = -two * c / (b + np.sqrt(b * b - four * a * c))
    z = np.clip(z, 0, 1)
    x = w * z + xl
    zz = z * z
    z1mz = z - zz
    omz2 = (one - z) ** 2
    den = delta + st * z1mz
    logdet = -two * np.log(delta) - np.log(sr * zz + two * delta * z1mz + sl * omz2) + two * np.log(den)
    below = y <= y_pos[..., 0]
    above = y >= y_pos[..., -1]
    x = np.where(below, (y - y_pos[..., 0]) / slopes[..., 0] + x_pos[..., 0], x)
    x = np.where(above, (y - y_pos[..., -1]) / slopes[..., -1] + x_pos[..., -1], x)
    logdet = np.where(below, -np.log(slopes[..., 0]), logdet)
    logdet = np.where(above, -np.log(slopes[..., -1]), logdet)
    return x, logdet


# --------------------------------------------------------------------------
# distrax.MaskedCoupling with the two inner bijectors the reference uses
# --------------------------------------------------------------------------
class FlowSpec:
    """Plain description of a B / C / D flow (what the reference's
    ``make_flow_model`` builds).

    kind            'affine' (B: ScalarAffine, conditioner emits 2 numbers that
                    broadcast over every event dim, src/real_nvp_toy_densities.py:84-86)
                    or 'rqs' (C/D: conditioner emits D*(3K+1), reshaped [D, 3K+1])
    event_size      D (flattened event)
    layers          list of conditioner parameter lists [(W,b), ...]
    masks           list of bool[D], True = pass-through
    base            'normal' (Independent(Normal(0,1))) or 'uniform01'
    """

    def __init__(self, kind, event_size, layers, masks, base="normal",
                 num_bins=0, range_min=0.0, range_max=1.0,
                 min_bin_size=1e-4, min_knot_slope=1e-4):
        self.kind = kind
        self.event_size = int(event_size)
        self.layers = layers
        self.masks = masks
        self.base = base
        self.num_bins = int(num_bins)
        self.range_min = float(range_min)
        self.range_max = float(range_max)
        self.min_bin_size = float(min_bin_size)
        self.min_knot_slope = float(min_knot_slope)


def coupling_layer(spec: FlowSpec, l: int, v, inverse: bool, dtype=np.float64):
    """One MaskedCoupling.forward_and_log_det / inverse_and_log_det (row b3)."""
    v = np.asarray(v, dtype)
    mask = spec.masks[l]
    masked = np.where(mask, v, dtype(0))
    raw = conditioner_apply(spec.layers[l], masked, dtype, getattr(spec, "activation", "relu"))
    if spec.kind == "affine":
        shift, log_scale = raw[:, :1], raw[:, 1:2]  # split(params, 2, axis=-1)
        if inverse:
            out, ld = scalar_affine_inv(v, shift, log_scale)
        else:
            out, ld = scalar_affine_fwd(v, shift, log_scale)
    elif spec.kind == "rqs":
        P = 3 * spec.num_bins + 1
        raw = raw.reshape(v.shape[0], spec.event_size, P)
        xp, yp, sl = rqs_knots(raw, spec.range_min, spec.range_max,
                               spec.min_bin_size, spec.min_knot_slope, dtype)
        out, ld = (rqs_inv if inverse else rqs_fwd)(v, xp, yp, sl)
    else:
        raise ValueError(spec.kind)
    out = np.where(mask, v, out)
    logdet = np.sum(np.where(mask, dtype(0), ld), axis=-1)
    return out.astype(dtype), logdet.astype(dtype)


def chain_forward(spec: FlowSpec, v, dtype=np.float64):
    """distrax.Chain(layers).forward_and_log_det: LAST layer first (row b5)."""
    v = np.asarray(v, dtype)
    total = np.zeros(v.shape[0], dtype)
    for l in reversed(range(len(spec.layers))):
        v, ld = coupling_layer(spec, l, v, False, dtype)
        total = total + ld
    return v, total


def chain_inverse(spec: FlowSpec, v, dtype=np.float64):
    """distrax.Chain(layers).inverse_and_log_det: layer 0 first."""
    v = np.asarray(v, dtype)
    total = np.zeros(v.shape[0], dtype)
    for l in range(len(spec.layers)):
        v, ld = coupling_layer(spec, l, v, True, dtype)
        total = total + ld
    return v, total


def base_log_prob(spec: FlowSpec, z, dtype=np.float64):
    z = np.asarray(z, dtype)
    if spec.base == "normal":
        # Independent(Normal(0,1)): sum_d(-x^2/2 - log(2 pi)/2)
        return np.sum(-0.5 * z * z - dtype(0.5 * math.log(2.0 * math.pi)), axis=-1)
    if spec.base == "uniform01":
        inside = np.all((z >= 0) & (z <= 1), axis=-1)
        return np.where(inside, dtype(0), dtype(-np.inf))
    raise ValueError(spec.base)


def log_prob(spec: FlowSpec, x, dtype=np.float64):
    """Transformed(base, Inverse(Chain(layers))).log_prob (rows b5, b6)."""
    z, ld = chain_forward(spec, x, dtype)
    return base_log_prob(spec, z, dtype) + ld


def sample_from_z(spec: FlowSpec, z, dtype=np.float64):
    """Transformed.sample with the base draw supplied: Chain.inverse(z)."""
    return chain_inverse(spec, z, dtype)[0]


def nll(spec: FlowSpec, x, dtype=np.float64):
    """loss_fn / eval_fn: -mean(log_prob) (src/real_nvp_toy_densities.py:139-151)."""
    return -np.mean(log_prob(spec, x, dtype))


# --------------------------------------------------------------------------
# synthetic weights and inputs
# --------------------------------------------------------------------------
def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def make_conditioner_weights(layer_index, in_size, hidden, out_size, final_std=0.05, seed=534):
    """Hidden layers: TruncNormal(+-2 sigma), sigma = 1/sqrt(fan_in), b = 0 (haiku
    MLP default).  Final layer N(0, final_std^2) weights and biases (non-zero so
    the flow is not the identity; final_std=0 gives the reference's zero init).
    Seed = 534 + layer index."""
    rng = np.random.Generator(np.random.PCG64(seed + layer_index))
    sizes = [in_size] + list(hidden)
    out = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        out.append((_trunc_normal(rng, (fi, fo), 1.0 / math.sqrt(fi)), np.zeros(fo, np.float32)))
    W = (rng.standard_normal((sizes[-1], out_size)) * final_std).astype(np.float32)
    b = (rng.standard_normal(out_size) * final_std).astype(np.float32)
    out.append((W, b))
    return out


def make_affine_spec(num_layers=8, event_size=2, hidden=(64, 64), final_std=0.05,
                     first_parity=1, seed=534):
    layers = [make_conditioner_weights(l, event_size, hidden, 2, final_std, seed)
              for l in range(num_layers)]
    return FlowSpec("affine", event_size, layers,
                    alternating_masks(event_size, num_layers, first_parity), "normal")


def make_rqs_spec(num_layers=8, event_size=2, hidden=(64, 64), num_bins=12,
                  range_min=-8.0, range_max=8.0, base="normal", final_std=0.05,
                  first_parity=1, seed=534):
    P = 3 * num_bins + 1
    layers = [make_conditioner_weights(l, event_size, hidden, event_size * P, final_std, seed)
              for l in range(num_layers)]
    return FlowSpec("rqs", event_size, layers,
                    alternating_masks(event_size, num_layers, first_parity), base,
                    num_bins, range_min, range_max)


def make_a_params(num_layers=8, sizes=(1, 4, 2), seed=534):
    """Variant A weights: W ~ 0.1 N(0,1), b = 0, distinct seed per layer
    (src/real_nvp.py:187-194 reuses one key; SURVEY §8d asks for distinct)."""
    params, flips = [], []
    flip = False
    for l in range(num_layers):
        rng = np.random.Generator(np.random.PCG64(seed + l))
        net = []
        for m, n in zip(sizes[:-1], sizes[1:]):
            net.append([(rng.standard_normal((m, n)) * 0.1).astype(np.float32),
                        np.zeros(n, np.float32)])
        params.append(net)
        flips.append(flip)
        flip = not flip
    return params, flips


def _w1(z):
    return np.sin(2.0 * np.pi * z / 4.0)


def make_inputs(density: str, n: int, seed: int = 2123) -> np.ndarray:
    """Synthetic [n,2] float32 rows with the distributions of the reference's
    generators (src/densities/*.py); NumPy Philox instead of JAX threefry."""
    rng = np.random.Generator(np.random.Philox(seed))
    if density == "banana":  # src/densities/banana.py:37-41
        x2 = rng.normal(0.0, math.sqrt(3.0), n)
        x1 = rng.normal(x2 ** 2 / 3.0 - 2.0, math.sqrt(3.0) / 2.0)
        rows = np.stack([x1, x2], -1)
    elif density == "energy1":  # src/densities/energies.py:47-66
        z = rng.normal(0.0, 2.0, n)
        mu = 2.0 - z ** 2 / 32.0
        c = rng.integers(0, 2, n) * 2 - 1
        v = rng.normal(c * mu, 0.3)
        rows = np.stack([v, z], -1)
    elif density in ("energy2", "energy3", "energy4"):  # :69-127
        x1 = rng.uniform(-1.5 * np.pi, 1.5 * np.pi, n)
        m = _w1(x1)
        pick = rng.integers(0, 2, n).astype(bool)
        if density == "energy3":
            m = np.where(pick, m - 3.0 * np.exp(-0.5 * ((x1 - 1.0) / 0.6) ** 2), m)
        elif density == "energy4":
            m = np.where(pick, m + 3.0 / (1.0 + np.exp(-(x1 - 1.0) / 0.3)), m)
        rows = np.stack([x1, rng.normal(m, 0.5)], -1)
    elif density == "gaussian_blobs":  # src/densities/gaussian_blobs.py:34-59
        k = rng.integers(0, 8, n)
        th = 2.0 * np.pi * k / 8.0
        rows = np.stack([2.0 * np.cos(th), 2.0 * np.sin(th)], -1) + rng.normal(0.0, 0.3, (n, 2))
    elif density == "mnist_like":  # src/maf_mnist.py:101-106 eval path: k/256
        rows = rng.integers(0, 256, (n, 784)).astype(np.float64) / 256.0
    else:
        raise ValueError(density)
    return rows.astype(np.float32)


# --------------------------------------------------------------------------
# base sampler: restatement of libnf_hip's Philox4x32-10 draw (include/nf_hip.h nf_base_sample).
# The reference draws with JAX threefry; that stream is not reproduced.
# --------------------------------------------------------------------------
def philox4x32_10(counter_lo, counter_hi, seed):
    """Vectorised Philox4x32-10: counters (c0=counter_lo, c1=counter_hi, 0, 0), key = seed (64 bit)."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    c0 = np.asarray(counter_lo, np.uint64) & mask
    c1 = np.asarray(counter_hi, np.uint64) & mask
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def base_sample(base: str, rows: int, event_size: int, seed: int) -> np.ndarray:
    """float64 restatement of nf_base_sample: [rows, event_size]."""
    total = rows * event_size
    groups = (total + 3) // 4
    g = np.arange(groups, dtype=np.uint64)
    r = np.stack(philox4x32_10(g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), int(seed)), axis=-1)  # [groups, 4]
    m = (r >> np.uint64(8)).astype(np.float64)
    if base == "uniform01":
        v = m * 2.0 ** -24
    else:
        u = (m + 0.5) * 2.0 ** -24
        rad = np.sqrt(-2.0 * np.log(u[:, 0::2]))
        ang = 2.0 * np.pi * u[:, 1::2]
        v = np.empty_like(u)
        v[:, 0::2] = rad * np.cos(ang)
        v[:, 1::2] = rad * np.sin(ang)
    return v.reshape(-1)[:total].reshape(rows, event_size)


def density_sample(density: str, rows: int, seed: int) -> np.ndarray:
    """float64 restatement of nf_density_sample (same Philox words, same formulas):
    the reference's generators src/densities/banana.py:37-41, energies.py:47-127,
    gaussian_blobs.py:34-59 driven by libnf_hip's stream instead of JAX threefry."""
    r = np.arange(rows, dtype=np.uint64)
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    c0, c1 = r & mask, r >> np.uint64(32)
    c2 = np.full(rows, 0x44454E53, np.uint64)
    c3 = np.zeros(rows, np.uint64)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask,
                          ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    u0 = ((c0 >> np.uint64(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u1 = ((c1 >> np.uint64(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u2 = (c2 >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
    rad = np.sqrt(-2.0 * np.log(u0))
    n0, n1 = rad * np.cos(2 * np.pi * u1), rad * np.sin(2 * np.pi * u1)
    pick = (c3 & np.uint64(1)) != 0
    if density == "banana":
        b = math.sqrt(3.0) * n0
        a = b * b / 3.0 - 2.0 + math.sqrt(3.0) / 2.0 * n1
    elif density == "energy1":
        b = 2.0 * n0
        mu = 2.0 - b * b / 32.0
        a = np.where(pick, mu, -mu) + 0.3 * n1
    elif density == "gaussian_blobs":
        th = 2.0 * np.pi * (c3 & np.uint64(7)).astype(np.float64) / 8.0
        a = 2.0 * np.cos(th) + 0.3 * n0
        b = 2.0 * np.sin(th) + 0.3 * n1
    else:
        a = (u2 * 3.0 - 1.5) * np.pi
        m = _w1(a)
        if density == "energy3":
            m = np.where(pick, m - 3.0 * np.exp(-0.5 * ((a - 1.0) / 0.6) ** 2), m)
        elif density == "energy4":
            m = np.where(pick, m + 3.0 / (1.0 + np.exp(-(a - 1.0) / 0.3)), m)
        b = m + 0.5 * n0
    return np.stack([a, b], -1)


# --------------------------------------------------------------------------
# gradient of the NLL (the training step's jax.grad(loss_fn), src/real_nvp_toy_densities.py:164-173),
# restated by hand for the masked affine flow: reverse-mode through the chain, float64.
# --------------------------------------------------------------------------
def affine_nll_and_grad(spec: FlowSpec, x, dtype=np.float64):
    """Returns (nll, grads) with grads[l] = [(dW, db), ...] shaped like spec.layers[l].

    loss = -mean_i log_prob(x_i);  relu'(0) = 0 as in jax.nn.relu."""
    assert spec.kind == "affine" and spec.base == "normal"
    x = np.asarray(x, dtype)
    n = x.shape[0]
    L = len(spec.layers)
    # forward (Chain.forward: last layer first), remembering what the backward pass needs
    tape = []
    v = x
    total_ld = np.zeros(n, dtype)
    for l in reversed(range(L)):
        mask = spec.masks[l]
        masked = np.where(mask, v, dtype(0))
        acts = [masked]
        pre = []
        a = masked
        for W, b in spec.layers[l][:-1]:
            z = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
            pre.append(z)
            a = relu(z)
            acts.append(a)
        W, b = spec.layers[l][-1]
        out = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
        shift, ls = out[:, :1], out[:, 1:2]
        e = np.exp(ls)
        y = np.where(mask, v, v * e + shift)
        n_t = int((~mask).sum())
        total_ld = total_ld + n_t * ls[:, 0]
        tape.append((l, v, acts, pre, e, ls))
        v = y
    z_final = v
    nll = np.mean(0.5 * np.sum(z_final * z_final, -1) + 0.5 * spec.event_size * math.log(2 * math.pi) - total_ld)
    # backward
    grads = [None] * L
    G = z_final / n                                   # d nll / d z
    for l, v_in, acts, pre, e, ls in reversed(tape):
        mask = spec.masks[l]
        n_t = int((~mask).sum())
        Gt = np.where(mask, dtype(0), G)              # grads of the transformed dims' outputs
        g_shift = Gt.sum(-1, keepdims=True)
        g_ls = (Gt * v_in * e).sum(-1, keepdims=True) - dtype(n_t) / n
        g_out = np.concatenate([g_shift, g_ls], -1)
        lin = spec.layers[l]
        lg = []
        g = g_out
        for j in reversed(range(len(lin))):
            W = np.asarray(lin[j][0], dtype)
            lg.append((acts[j].T @ g, g.sum(0)))
            g = g @ W.T
            if j > 0:
                g = g * (pre[j - 1] > 0)
        grads[l] = lg[::-1]
        G = np.where(mask, G + g, Gt * e)             # d/d(input): pass-through dims also feed the MLP
    return float(nll), grads


def adam_reference(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """optax.adam update on flat float64 arrays (scale_by_adam + scale(-lr)); returns (params, m, v)."""
    m = b1 * m + (1 - b1) * grads
    v = b2 * v + (1 - b2) * grads * grads
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    return params - lr * mh / (np.sqrt(vh) + eps), m, v


def a_nll_and_grad(params, flips, y, dtype=np.float64):
    """jax.value_and_grad(loss)(params, batch) of src/real_nvp.py:292-297 restated by hand (float64).

    Returns (nll, grads) with grads[l][k] = [dW, db] like params.  leaky_relu'(x) = 1 for x >= 0."""
    v = np.asarray(y, dtype)
    n = v.shape[0]
    d = v.shape[1] // 2
    tape = []
    total = np.zeros(n, dtype)
    for l in reversed(range(len(params))):          # log_prob: last layer's inverse first
        net, flip = params[l], flips[l]
        y1, y2 = _a_split(v, flip)
        acts, pres = [y1], []
        a = y1
        for W, b in net[:-1]:
            z = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
            pres.append(z)
            a = leaky_relu(z)
            acts.append(a)
        W, b = net[-1]
        out = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
        shift, ls = out[:, :d], out[:, d:]
        x2 = (y2 - shift) * np.exp(-ls)
        total = total - ls.sum(-1)
        tape.append((l, flip, acts, pres, ls, x2))
        v = _a_join(y1, x2, flip)
    nll = np.mean(0.5 * np.sum(v * v, -1) + 0.5 * v.shape[1] * math.log(2 * math.pi) - total)
    grads = [None] * len(params)
    G = v / n
    for l, flip, acts, pres, ls, x2 in reversed(tape):
        G1, G2 = _a_split(G, flip)
        einv = np.exp(-ls)
        g_shift = -G2 * einv
        g_ls = -G2 * x2 + dtype(1.0) / n
        g = np.concatenate([g_shift, g_ls], -1)
        net = params[l]
        lg = []
        for j in reversed(range(len(net))):
            W = np.asarray(net[j][0], dtype)
            lg.append([acts[j].T @ g, g.sum(0)])
            g = g @ W.T
            if j > 0:
                g = g * np.where(pres[j - 1] >= 0, dtype(1), dtype(LEAKY_SLOPE))
        grads[l] = lg[::-1]
        G = _a_join(G1 + g, G2 * einv, flip)
    return float(nll), grads


# --------------------------------------------------------------------------
# gradient of the NLL for the masked kinds in general (any D, widths, affine or spline) --
# what jax.grad(loss_fn) returns in src/real_nvp_toy_densities.py:164-173,
# src/maf_toy_densities.py:167-176, src/maf_mnist.py:159-168.  The spline's own Jacobian is taken
# by complex-step differentiation of the rational-quadratic formula (exact to rounding in float64,
# and independent of the hand-derived partials in the HIP kernel).
# --------------------------------------------------------------------------
def _rqs_bin_formula(x, xl, xr, yl, yr, sl, sr):
    w = xr - xl
    h = yr - yl
    delta = h / w
    z = (x - xl) / w
    zz = z * z
    z1mz = z - zz
    omz2 = (1.0 - z) ** 2
    st = sr + sl - 2.0 * delta
    y = yl + h * (delta * zz + sl * z1mz) / (delta + st * z1mz)
    ld = 2.0 * np.log(delta) + np.log(sr * zz + 2.0 * delta * z1mz + sl * omz2) - 2.0 * np.log(delta + st * z1mz)
    return y, ld


def rqs_param_grads(x, raw, gy, gld, spec: FlowSpec):
    """x [n, T] inputs of the transformed dims, raw [n, T, 3K+1] conditioner outputs, gy [n, T] =
    dL/dy, gld (scalar) = dL/d(logdet).  Returns (dL/draw [n, T, 3K+1], dL/dx [n, T])."""
    K = spec.num_bins
    xp, yp, sl = rqs_knots(raw, spec.range_min, spec.range_max, spec.min_bin_size, spec.min_knot_slope, np.float64)
    x = np.asarray(x, np.float64)
    k = _pick_bin(x, xp)
    ins = [x, _take(xp, k), _take(xp, k + 1), _take(yp, k), _take(yp, k + 1), _take(sl, k), _take(sl, k + 1)]
    h = 1e-30
    g7 = []
    for j in range(7):
        pert = [a.astype(np.complex128) for a in ins]
        pert[j] = pert[j] + 1j * h
        y, ld = _rqs_bin_formula(*pert)
        g7.append(gy * (y.imag / h) + gld * (ld.imag / h))
    below = x <= xp[..., 0]
    above = x >= xp[..., -1]
    tail = below | above
    s_tail = np.where(below, sl[..., 0], sl[..., -1])
    edge = np.where(below, xp[..., 0], xp[..., -1])
    gx = np.where(tail, gy * s_tail, g7[0])
    g_xp = np.zeros_like(xp)
    g_yp = np.zeros_like(yp)
    g_sl = np.zeros_like(sl)
    inside = (~tail).astype(np.float64)

    def scatter(dst, idx, val):
        np.put_along_axis(dst, idx[..., None], np.take_along_axis(dst, idx[..., None], -1) + val[..., None], -1)

    scatter(g_xp, k, g7[1] * inside)
    scatter(g_xp, k + 1, g7[2] * inside)
    scatter(g_yp, k, g7[3] * inside)
    scatter(g_yp, k + 1, g7[4] * inside)
    scatter(g_sl, k, g7[5] * inside)
    scatter(g_sl, k + 1, g7[6] * inside)
    g_tail = (gy * (x - edge) + gld / s_tail) * tail
    scatter(g_sl, np.where(below, 0, K) * np.ones_like(k), g_tail)
    # knots -> bin sizes (the first and last knot are constants) -> softmax logits
    c = (spec.range_max - spec.range_min) - K * spec.min_bin_size

    def sizes_to_logits(g_pos, logits):
        g_size = np.zeros(g_pos.shape[:-1] + (K,))
        g_size[..., :K - 1] = np.cumsum(g_pos[..., 1:K][..., ::-1], -1)[..., ::-1]
        s = _softmax(np.asarray(logits, np.float64))
        return c * s * (g_size - np.sum(g_size * s, -1, keepdims=True))

    raw = np.asarray(raw, np.float64)
    offset = np.log(np.exp(1.0 - spec.min_knot_slope) - 1.0)
    g_raw = np.concatenate([sizes_to_logits(g_xp, raw[..., :K]), sizes_to_logits(g_yp, raw[..., K:2 * K]),
                            g_sl / (1.0 + np.exp(-(raw[..., 2 * K:] + offset)))], -1)
    return g_raw, gx


def flow_nll_and_grad(spec: FlowSpec, x):
    """(nll, grads) of -mean(log_prob) for a masked-affine or masked-spline FlowSpec in float64;
    grads[l] = [(dW, db), ...] shaped like spec.layers[l].  relu'(0) = 0."""
    dtype = np.float64
    x = np.asarray(x, dtype)
    n, D = x.shape
    L = len(spec.layers)
    P = 3 * spec.num_bins + 1
    tape = []
    v = x
    total_ld = np.zeros(n, dtype)
    for l in reversed(range(L)):
        mask = spec.masks[l]
        masked = np.where(mask, v, 0.0)
        acts, pre, a = [masked], [], masked
        leaky = getattr(spec, "activation", "relu") != "relu"
        for W, b in spec.layers[l][:-1]:
            z = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
            pre.append(z)
            a = leaky_relu(z) if leaky else relu(z)
            acts.append(a)
        W, b = spec.layers[l][-1]
        out = a @ np.asarray(W, dtype) + np.asarray(b, dtype)
        y, ld = coupling_layer(spec, l, v, False, dtype)
        tape.append((l, v, acts, pre, out))
        total_ld = total_ld + ld
        v = y
    z_final = v
    nll = float(-np.mean(base_log_prob(spec, z_final, dtype) + total_ld))
    grads = [None] * L
    G = z_final / n if spec.base == "normal" else np.zeros_like(z_final)
    gld = -1.0 / n
    for l, v_in, acts, pre, out in reversed(tape):
        mask = spec.masks[l]
        t_idx = np.nonzero(~mask)[0]
        if spec.kind == "affine":
            ls = out[:, 1:2]
            e = np.exp(ls)
            Gt = G[:, t_idx]
            g_out = np.concatenate([Gt.sum(-1, keepdims=True),
                                    (Gt * v_in[:, t_idx] * e).sum(-1, keepdims=True) + gld * len(t_idx)], -1)
            gx_t = Gt * e
        else:
            raw = out.reshape(n, D, P)[:, t_idx, :]
            g_raw, gx_t = rqs_param_grads(v_in[:, t_idx], raw, G[:, t_idx], gld, spec)
            g_out = np.zeros((n, D, P))
            g_out[:, t_idx, :] = g_raw
            g_out = g_out.reshape(n, D * P)
        lin = spec.layers[l]
        lg, g = [], g_out
        for j in reversed(range(len(lin))):
            W = np.asarray(lin[j][0], dtype)
            lg.append((acts[j].T @ g, g.sum(0)))
            g = g @ W.T
            if j > 0:
                g = g * np.where(pre[j - 1] > 0, 1.0, 0.01 if leaky else 0.0)
        grads[l] = lg[::-1]
        Gn = np.where(mask, G + g, 0.0)
        Gn[:, t_idx] = gx_t
        G = Gn
    return nll, grads
