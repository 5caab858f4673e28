"""NumPy float32 emulation of the two forms of the src/real_nvp.py conditioner (MLP 1-H-2, leaky_relu 0.01) in
csrc/nf_split2d_small.hip: the reference's order (act = max-form, then the second Linear) and the kernel's
alpha a + beta |a| form folded into B + A c + sum C |a_k| -- against float64, on weights where inactive units have
large |w a| (the folded form cancels two O(|w a|) terms).   python tests/tools/split2d_abs_form_study.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import flows_np as O  # noqa: E402

f32 = np.float32


def big_params(L, H, seed, w0_scale, b0_shift, w1_scale):
    rng = np.random.default_rng(seed)
    params, flips = [], []
    for l in range(L):
        W0 = (rng.standard_normal((1, H)) * w0_scale).astype(f32)
        b0 = (-np.abs(rng.standard_normal(H)) * b0_shift - b0_shift).astype(f32)      # strongly negative pre-activations
        W1 = (rng.standard_normal((H, 2)) * w1_scale).astype(f32)
        W1[:, 1] *= 0.05                                                               # keep exp(log_scale) finite
        b1 = (rng.standard_normal(2) * 0.1).astype(f32)
        params.append([[W0, b0], [W1, b1]])
        flips.append(bool(l & 1))
    return params, flips


def layer_out(net, c, form):
    (W0, b0), (W1, b1) = net
    a = (c[:, None].astype(f32) * W0[0] + b0).astype(f32)
    if form == "ref":
        act = np.where(a >= 0, a, f32(0.01) * a).astype(f32)
        out = np.zeros((len(c), 2), f32)
        for k in range(W0.shape[1]):
            out = (out + act[:, k:k + 1] * W1[k]).astype(f32)
        return (out + b1).astype(f32)
    al, be = 0.505, 0.495
    A = (al * (W1.astype(np.float64) * W0[0].astype(np.float64)[:, None]).sum(0)).astype(f32)
    B = (b1.astype(np.float64) + al * (W1.astype(np.float64) * b0.astype(np.float64)[:, None]).sum(0)).astype(f32)
    C = (be * W1.astype(np.float64)).astype(f32)
    out = (B + A * c[:, None].astype(f32)).astype(f32)
    for k in range(W0.shape[1]):
        out = (out + np.abs(a[:, k:k + 1]) * C[k]).astype(f32)
    return out


def chain_log_prob(params, flips, y, form):
    v = y.astype(f32)
    total = np.zeros(len(v), f32)
    for net, flip in zip(reversed(params), reversed(flips)):
        y1, y2 = (v[:, 1], v[:, 0]) if flip else (v[:, 0], v[:, 1])
        o = layer_out(net, y1, form)
        x2 = ((y2 - o[:, 0]) * np.exp(-o[:, 1]).astype(f32)).astype(f32)
        v = np.stack([x2, y1], 1) if flip else np.stack([y1, x2], 1)
        total = (total - o[:, 1]).astype(f32)
    return (-0.5 * (v.astype(np.float64) ** 2).sum(1) - np.log(2 * np.pi) + total).astype(f32)


if __name__ == "__main__":
    x = O.make_inputs("energy2", 4096)
    for name, args in (("bench-like", None), ("w0 3, b0 -10, w1 5", (3.0, 10.0, 5.0)), ("w0 5, b0 -20, w1 10", (5.0, 20.0, 10.0)),
                       ("w0 1, b0 -30, w1 10", (1.0, 30.0, 10.0))):
        params, flips = O.make_a_params(8, sizes=(1, 16, 2)) if args is None else big_params(8, 16, 1, *args)
        ref = O.a_log_prob_chain(params, flips, x)
        fin = np.isfinite(ref)
        for form in ("ref", "abs"):
            with np.errstate(all="ignore"):
                got = chain_log_prob(params, flips, x, form).astype(np.float64)
            e = np.abs(got - ref)[fin] / np.maximum(1, np.abs(ref[fin]))
            print(f"{name:24s} {form}: scaled error max {np.nanmax(e):.3e} rms {np.sqrt(np.nanmean(e ** 2)):.3e}  (|log_prob| up to {np.abs(ref[fin]).max():.0f}, {fin.mean():.2f} finite)")
