"""Where does a float32 rational-quadratic spline lose its digits?  (CPU only; development aid behind the design of
normalizing_flows_amd/csrc/nf_rqs_math.h.  Lives under tests/: it uses the oracle as the checker.)

BASELINE.json configs[2] (8 spline layers, K = 12 on [-8, 8]) on 16,384 rows per density.  The flow state is kept
in float32 throughout; ONE stage of the spline at a time is evaluated in float32 while everything else runs in
float64, and the scaled log_prob error |f - f64| / max(1, |f64|) is reported (max and rms).

Finding (the table this prints): the conditioner in float32 costs ~1e-7 rms, the rational-quadratic formula 1.3e-7,
slopes and log-det < 1e-7 -- the error of the all-float32 chain (1e-6 rms, 6e-6 ... 1.4e-5 worst) comes from the
KNOTS: softmax normalisation (8e-7) and the running sum (7e-7) over a 16-wide range.  Accurate knots rounded once to
float32 cost 9e-8.  Hardware-quality float32 exponentials under an otherwise float64 normalisation leave
2.4e-7 (0.5 ulp) ... 3.8e-7 (1 more ulp of noise).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import flows_np as O  # noqa: E402

f4, f8 = np.float32, np.float64


def spline(raw, x, spec, st, exp_noise_ulp=0.0, rng=None):
    """One forward spline with per-stage dtypes st[stage]; raw [B, 3K+1] float32, x [B] float32."""
    K = spec.num_bins
    d = st["softmax"]
    total, mb = d(spec.range_max - spec.range_min), d(spec.min_bin_size)

    def weights(lg):
        t = lg.astype(d)
        t = t - t.max(-1, keepdims=True)
        e = np.exp(t)
        if exp_noise_ulp:
            e = (e * (1 + rng.uniform(-1, 1, e.shape) * exp_noise_ulp * 2.0 ** -24)).astype(d)
        e = e.astype(st["normalise"])
        return e / e.sum(-1, keepdims=True) * st["normalise"](total - K * mb) + st["normalise"](mb)

    def knots(lg):
        w = weights(lg).astype(st["cumsum"])
        lo = np.full((lg.shape[0], 1), spec.range_min, st["cumsum"])
        hi = np.full((lg.shape[0], 1), spec.range_max, st["cumsum"])
        return np.concatenate([lo, st["cumsum"](spec.range_min) + np.cumsum(w[:, :-1], -1), hi], -1)

    xp, yp = knots(raw[:, :K]), knots(raw[:, K:2 * K])
    d3 = st["slopes"]
    ms = d3(spec.min_knot_slope)
    off = np.log(np.exp(d3(1) - ms) - d3(1))
    sl_all = np.logaddexp(raw[:, 2 * K:].astype(d3) + off, d3(0)) + ms
    d4 = st["formula"]
    y, ld = O.rqs_fwd(x.astype(d4), xp.astype(d4), yp.astype(d4), sl_all.astype(d4))
    return y.astype(f4), ld.astype(f4)


def log_prob(spec, x, st, **kw):
    v = np.asarray(x, f4)
    tot = np.zeros(v.shape[0], f4)
    P = 3 * spec.num_bins + 1
    rng = np.random.default_rng(1)
    for l in reversed(range(len(spec.layers))):
        mask = spec.masks[l]
        masked = np.where(mask, v, 0).astype(f4)
        raw = O.conditioner_apply(spec.layers[l], masked, st["cond"]).astype(f4).reshape(v.shape[0], 2, P)
        t = int(np.where(~mask)[0][0])
        y, ld = spline(raw[:, t, :], v[:, t], spec, st, rng=rng, **kw)
        v = v.copy()
        v[:, t] = y
        tot = tot + ld
    return O.base_log_prob(spec, v.astype(f8), f8) + tot


def main():
    spec = O.make_rqs_spec(8, final_std=0.05)
    all64 = dict(cond=f8, softmax=f8, normalise=f8, cumsum=f8, slopes=f8, formula=f8)
    rows = [("everything float64 (state float32)", {}, {}),
            ("conditioner float32", dict(cond=f4), {}),
            ("softmax + normalisation float32", dict(softmax=f4, normalise=f4), {}),
            ("running sum float32", dict(cumsum=f4), {}),
            ("slopes float32", dict(slopes=f4), {}),
            ("formula + log-det float32 (knots rounded once)", dict(formula=f4), {}),
            ("everything float32", dict(cond=f4, softmax=f4, normalise=f4, cumsum=f4, slopes=f4, formula=f4), {}),
            ("float32, but normalisation + running sum float64 (exp 0.5 ulp)",
             dict(cond=f4, softmax=f4, slopes=f4, formula=f4), {}),
            ("  the same with 1 more ulp of noise on every exp", dict(cond=f4, softmax=f4, slopes=f4, formula=f4),
             dict(exp_noise_ulp=1.0)),
            ("float32, but all of the knots float64", dict(cond=f4, slopes=f4, formula=f4), {})]
    for density in ["gaussian_blobs", "banana", "energy2"]:
        x = O.make_inputs(density, 16384)
        ref = O.log_prob(spec, x)
        scale = np.maximum(1, np.abs(ref))
        print(density)
        for name, upd, kw in rows:
            st = dict(all64)
            st.update(upd)
            e = np.abs(log_prob(spec, x, st, **kw) - ref) / scale
            print(f"  {name:66s} max {e.max():.2e}  rms {np.sqrt((e * e).mean()):.2e}")


if __name__ == "__main__":
    main()
