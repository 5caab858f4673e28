"""NumPy float64 transcription of the hand-derived spline gradient of csrc/nf_rqs2d_bwd.hip (step 3 of its header),
checked against the oracle's complex-step Jacobian (oracle.flows_np.rqs_param_grads).  Development aid:
    python tests/tools/rqs_grad_formulas.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import flows_np as O  # noqa: E402


def kernel_formulas(x, raw, gy, gl, spec):
    K = spec.num_bins
    lo, hi, mb, ms = spec.range_min, spec.range_max, spec.min_bin_size, spec.min_knot_slope
    xp, yp, slopes = O.rqs_knots(raw, lo, hi, mb, ms, np.float64)
    kb = O._pick_bin(x, xp)
    kb = np.where(x >= hi, K - 1, kb)      # the kernel's bin search counts knots <= x: K - 1 above the range
    take = O._take
    xl, yl = take(xp, kb), take(yp, kb)
    bw, bh = take(xp, kb + 1) - xl, take(yp, kb + 1) - yl
    off = np.log(np.exp(1.0 - ms) - 1.0)
    vl, vr = take(raw[..., 2 * K:], kb) + off, take(raw[..., 2 * K:], kb + 1) + off
    sl, sr = np.logaddexp(vl, 0) + ms, np.logaddexp(vr, 0) + ms
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    rbw = 1.0 / bw
    delta = bh * rbw
    z = np.clip((x - xl) * rbw, 0, 1)
    q, u, omz = z * z, z - z * z, 1 - z
    omz2, tz, tu = omz * omz, 1 - 2 * z, 1 - 2 * (z - z * z)
    st = sr + sl - 2 * delta
    A = delta * q + sl * u
    den = delta + st * u
    inner = sr * q + 2 * delta * u + sl * omz2
    rden, rin, rdel = 1 / den, 1 / inner, 1 / delta
    hr2 = bh * rden * rden
    y_z = hr2 * delta * inner
    l_z = 2 * ((sr * z + delta * tz - sl * omz) * rin - st * tz * rden)
    y_d = hr2 * (q * den - A * tu)
    l_d = 2 * (rdel + u * rin - tu * rden)
    y_sl, l_sl = hr2 * u * (den - A), omz2 * rin - 2 * u * rden
    y_sr, l_sr = -hr2 * A * u, q * rin - 2 * u * rden
    t_z, t_d = gy * y_z + gl * l_z, gy * y_d + gl * l_d
    t_sl, t_sr = gy * y_sl + gl * l_sl, gy * y_sr + gl * l_sr
    g_x = t_z * rbw
    t_w = -(t_z * z + t_d * delta) * rbw
    t_hh = gy * A * rden + t_d * rbw
    kl = np.stack([-g_x - t_w, gy - t_hh])      # [axis]
    kr = np.stack([t_w, t_hh])
    below, above = x <= lo, x >= hi
    tail = below | above
    vb = np.where(below, raw[..., 2 * K], raw[..., 3 * K]) + off
    s = np.logaddexp(vb, 0) + ms
    t_s = (gy * (x - np.where(below, lo, hi)) + gl / s) * sig(vb)
    g_x = np.where(tail, gy * s, g_x)
    kl, kr = np.where(tail, 0, kl), np.where(tail, 0, kr)
    d_l = np.where(tail, np.where(below, t_s, 0), t_sl * sig(vl))
    d_r = np.where(tail, np.where(below, 0, t_s), t_sr * sig(vr))
    g_kl = np.where(kb >= 1, kl, 0)
    g_kr = np.where(kb + 1 <= K - 1, kr, 0)
    cs = (hi - lo) - K * mb
    g_raw = np.zeros_like(raw)
    for axis, (pos, first) in enumerate(((xp, 0), (yp, K))):
        size = pos[..., 1:] - pos[..., :-1]
        lo_k, sz = take(pos, kb), take(size, kb)
        s_lt = (lo_k - lo - kb * mb) / cs
        s_le = s_lt + (sz - mb) / cs
        dot = g_kl[axis] * s_lt + g_kr[axis] * s_le
        for k in range(K):
            gk = np.where(k < kb, g_kl[axis], 0) + np.where(k <= kb, g_kr[axis], 0)
            g_raw[..., first + k] = (size[..., k] - mb) * (gk - dot)
    for k in range(K + 1):
        g_raw[..., 2 * K + k] = np.where(k == kb, d_l, 0) + np.where(k == kb + 1, d_r, 0)
    return g_raw, g_x


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for K, lo, hi in ((12, -8.0, 8.0), (4, -3.0, 4.0), (8, 0.0, 1.0)):
        spec = O.FlowSpec("rqs", 2, [], [], "normal", K, lo, hi)
        n = 4000
        raw = rng.standard_normal((n, 1, 3 * K + 1)) * 1.2
        x = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 1))
        gy = rng.standard_normal((n, 1))
        ref_raw, ref_x = O.rqs_param_grads(x, raw, gy, -1.0, spec)
        got_raw, got_x = kernel_formulas(x, raw, gy, -1.0, spec)
        print(K, "d raw", np.abs(got_raw - ref_raw).max() / np.abs(ref_raw).max(), "d x", np.abs(got_x - ref_x).max() / np.abs(ref_x).max())
        for name, sl_ in (("w", slice(0, K)), ("h", slice(K, 2 * K)), ("s", slice(2 * K, 3 * K + 1))):
            e = np.abs(got_raw[..., sl_] - ref_raw[..., sl_])
            wi = np.unravel_index(np.argmax(e), e.shape)
            print("   ", name, e.max(), "at sample", wi[0], "x", x[wi[0], 0], "kb", O._pick_bin(x, O.rqs_knots(raw, lo, hi)[0])[wi[0], 0],
                  got_raw[wi[0], 0, sl_][wi[2]], ref_raw[wi[0], 0, sl_][wi[2]])
