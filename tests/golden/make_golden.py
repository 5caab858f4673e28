#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the float64 oracle (oracle/flows_np.py).

The reference itself cannot run in this image (jax / distrax / haiku / tfp are not
installed), so these vectors pin the ORACLE, not the reference:
"parity unpinned" stands.  Each file holds the float32 weights, the float32 inputs
and the float64 expected outputs, so a fixture is self-contained data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flows_np as O  # noqa: E402


def flat(layers):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in layers for W, b in layer])


def main():
    # variant B, BASELINE configs[0] shape: 4 layers, banana; and the 8-layer headline flow
    for name, L, dens, rows in [("affine_l4_banana", 4, "banana", 256), ("affine_l8_energy3", 8, "energy3", 256)]:
        spec = O.make_affine_spec(L, final_std=0.02)
        x = O.make_inputs(dens, rows, seed=2235)
        z, ld = O.chain_forward(spec, x)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), weights=flat(spec.layers), x=x,
                            log_prob=O.log_prob(spec, x), z=z, logdet=ld, nll=O.nll(spec, x),
                            num_layers=L, hidden=np.array([64, 64]), first_parity=1)
    # variant A
    params, flips = O.make_a_params(8)
    x = O.make_inputs("energy2", 256, seed=2235)
    np.savez_compressed(os.path.join(HERE, "split_l8_energy2.npz"),
                        weights=flat([[(W, b) for W, b in net] for net in params]), x=x,
                        log_prob=O.a_log_prob_chain(params, flips, x), nll=O.a_nll(params, flips, x),
                        sample_from_x=O.a_sample_chain_from_z(params, flips, x), num_layers=8)
    # variant C
    spec = O.make_rqs_spec(8, final_std=0.05)
    x = O.make_inputs("gaussian_blobs", 256, seed=2235)
    z, ld = O.chain_forward(spec, x)
    np.savez_compressed(os.path.join(HERE, "rqs_l8_blobs.npz"), weights=flat(spec.layers), x=x,
                        log_prob=O.log_prob(spec, x), z=z, logdet=ld, num_layers=8, num_bins=12,
                        range_min=-8.0, range_max=8.0)
    # gradients of the NLL (the training step's known answers): headline flow, 4 layers, and a 3-layer spline flow
    spec = O.make_affine_spec(4, final_std=0.03)
    x = O.make_inputs("banana", 256, seed=2235)
    nll, grads = O.affine_nll_and_grad(spec, x)
    np.savez_compressed(os.path.join(HERE, "affine_l4_grad.npz"), weights=flat(spec.layers), x=x, nll=nll,
                        grad=flat(grads), num_layers=4)
    spec = O.make_rqs_spec(3, num_bins=8, range_min=-6.0, range_max=6.0, final_std=0.05)
    x = O.make_inputs("gaussian_blobs", 256, seed=2235)
    nll, grads = O.flow_nll_and_grad(spec, x)
    np.savez_compressed(os.path.join(HERE, "rqs_l3_grad.npz"), weights=flat(spec.layers), x=x, nll=nll,
                        grad=flat(grads), num_layers=3, num_bins=8, range_min=-6.0, range_max=6.0)
    # hand-computed case: 1 layer, H = (2, 2), numbers worked by hand below
    np.savez_compressed(os.path.join(HERE, "affine_hand_case.npz"), **hand_case())
    # the same for a spline layer: expected values from rational arithmetic + math.log, NOT from the oracle
    np.savez_compressed(os.path.join(HERE, "rqs_hand_case.npz"), **hand_case_rqs())


def hand_case():
    """One masked affine layer, mask [F, T], conditioner 2 -> 2 -> 2 -> 2, x = (0.5, 2.0).

    masked = (0, 2);  h1 = relu((0,2) @ [[9,9],[0.5,-1]] + (0.25, 0.5)) = relu(1.25, -1.5) = (1.25, 0)
    h2 = relu((1.25,0) @ [[2,-1],[7,7]] + (-0.5, 3)) = relu(2.0, 1.75) = (2, 1.75)
    out = (2,1.75) @ [[0.5,0.1],[-1,0.2]] + (0.25,-0.05) = (1-1.75+0.25, 0.2+0.35-0.05) = (-0.5, 0.5)
    shift = -0.5, log_scale = 0.5;  y0 = 0.5*exp(0.5) - 0.5, y1 = 2;  logdet = 0.5
    """
    W1 = np.array([[9, 9], [0.5, -1]], np.float32); b1 = np.array([0.25, 0.5], np.float32)
    W2 = np.array([[2, -1], [7, 7]], np.float32); b2 = np.array([-0.5, 3], np.float32)
    W3 = np.array([[0.5, 0.1], [-1, 0.2]], np.float32); b3 = np.array([0.25, -0.05], np.float32)
    y0 = 0.5 * np.exp(0.5) - 0.5
    lp = -0.5 * (y0 ** 2 + 4.0) - np.log(2 * np.pi) + 0.5
    return dict(W1=W1, b1=b1, W2=W2, b2=b2, W3=W3, b3=b3, x=np.array([[0.5, 2.0]], np.float32),
                y=np.array([[y0, 2.0]]), logdet=np.array([0.5]), log_prob=np.array([lp]))


def hand_case_rqs():
    """One masked spline layer worked by hand (no oracle call): mask [F, T], K = 2 bins on [-2, 2],
    min_bin_size = min_knot_slope = 0.5 (so that the numbers are short), conditioner 2 -> 2 -> 2 -> 14 with the
    hidden layers of hand_case(): every row has x1 = 2, hence h2 = (2, 1.75), and the last Linear is chosen as
    W3[0, k] = r_k / 4, W3[1, k] = r_k / 7, b3[k] = r_k / 4, so that out_k = r_k (2/4 + 1.75/7 + 1/4) = r_k for the
    transformed dim 0 (dim 1's 7 parameters are never read by MaskedCoupling: filled with junk).

    r = [0, ln 3 | ln 3, 0 | r(1), r(2), r(1.5)],  r(s) = ln((e^(s - 1/2) - 1) / (e^(1/2) - 1))  <=>  slope s:
      widths  = softmax(0, ln 3) * (4 - 2 * 0.5) + 0.5 = (1/4, 3/4) * 3 + 1/2 = (5/4, 11/4): x knots -2, -3/4, 2
      heights = (3/4, 1/4) * 3 + 1/2 = (11/4, 5/4):                                          y knots -2,  3/4, 2
      slopes  = softplus(r + ln(e^(1/2) - 1)) + 1/2 = (1, 2, 3/2)
    x0 = -11/8 (middle of bin 0: z = 1/2, delta = 11/5, sl = 1, sr = 2):
      y = -2 + (11/4)(11/5 * 1/4 + 1 * 1/4) / (11/5 + (3 - 22/5) / 4) = -2 + (11/5) / (37/20) = -2 + 44/37
      logdet = 2 ln(11/5) + ln(2/4 + 2 (11/5)/4 + 1/4) - 2 ln(37/20) = 2 ln(11/5) - ln(37/20)
    x0 = 5/8 (middle of bin 1: z = 1/2, delta = 5/11, sl = 2, sr = 3/2):
      den = 5/11 + (7/2 - 10/11)/4 = 97/88;  y = 3/4 + (5/4)(5/44 + 1/2) / (97/88) = 3/4 + (135/176)(88/97)
      logdet = 2 ln(5/11) + ln(3/8 + 5/22 + 1/2) - 2 ln(97/88) = 2 ln(5/11) + ln(97/88) - 2 ln(97/88)
    x0 = -3/4 (the inner knot itself: bin 1, z = 0): y = 3/4, logdet = ln(sl) = ln 2
    x0 = 3 (upper tail): y = 2 + (3 - 2) * 3/2 = 7/2, logdet = ln(3/2);  x0 = -5/2 (lower tail): y = -5/2, logdet = 0.
    log_prob = -(y0^2 + x1^2)/2 - ln(2 pi) + logdet."""
    import math
    from fractions import Fraction as F

    cases = [  # x0, y, logdet -- the numbers of the docstring
        (F(-11, 8), -2 + F(44, 37), 2 * math.log(11 / 5) - math.log(37 / 20)),
        (F(5, 8), F(3, 4) + F(135, 176) * F(88, 97), 2 * math.log(5 / 11) - math.log(97 / 88)),
        (F(-3, 4), F(3, 4), math.log(2.0)),
        (F(3), F(7, 2), math.log(1.5)),
        (F(-5, 2), F(-5, 2), 0.0),
    ]
    # the rational helper below reproduces them (it then works the 4-bin case the fused kernel can take)
    xk, yk, sk = [F(-2), F(-3, 4), F(2)], [F(-2), F(3, 4), F(2)], [F(1), F(2), F(3, 2)]
    for x0, y0, ld0 in cases:
        y1, ld1 = _rq_exact(x0, xk, yk, sk)
        assert y1 == y0 and abs(ld1 - ld0) < 1e-14, (x0, y1, y0, ld1, ld0)
    out = _hand_layer([0.0, math.log(3.0)], [math.log(3.0), 0.0], [1.0, 2.0, 1.5], 0.5, [c[0] for c in cases],
                      [c[1] for c in cases], [c[2] for c in cases])
    # K = 4, min_bin_size 1/4: widths softmax(0, ln 3, 0, ln 3) * 3 + 1/4 = (5/8, 11/8, 5/8, 11/8), heights the
    # mirror image, slopes (1, 2, 3/2, 2, 5/4); rows: inside every bin, both inner-most knots, both tails
    xk4 = [F(-2), F(-11, 8), F(0), F(5, 8), F(2)]
    yk4 = [F(-2), F(-5, 8), F(0), F(11, 8), F(2)]
    sk4 = [F(1), F(2), F(3, 2), F(2), F(5, 4)]
    xs = [F(-7, 4), F(-1), F(1, 4), F(3, 2), F(-11, 8), F(0), F(5, 2), F(-9, 4)]
    ys, lds = zip(*[_rq_exact(x0, xk4, yk4, sk4) for x0 in xs])
    out4 = _hand_layer([0.0, math.log(3.0), 0.0, math.log(3.0)], [math.log(3.0), 0.0, math.log(3.0), 0.0],
                       [float(v) for v in sk4], 0.25, xs, ys, lds)
    out.update({k + "_k4": v for k, v in out4.items()})
    return out


def _rq_exact(x, xk, yk, sk):
    """The rational-quadratic map of SURVEY.md 8a row c2 in exact rational arithmetic (logs in float64): knots
    xk / yk, slopes sk (Fractions).  Written for the hand cases; does not import the oracle."""
    import math
    from fractions import Fraction as F
    if x <= xk[0]:
        return (x - xk[0]) * sk[0] + yk[0], math.log(sk[0])
    if x >= xk[-1]:
        return (x - xk[-1]) * sk[-1] + yk[-1], math.log(sk[-1])
    k = max(i for i in range(len(xk) - 1) if xk[i] <= x)
    w, h = xk[k + 1] - xk[k], yk[k + 1] - yk[k]
    d, z, sl, sr = h / w, (x - xk[k]) / w, sk[k], sk[k + 1]
    den = d + (sr + sl - 2 * d) * z * (1 - z)
    y = yk[k] + h * (d * z * z + sl * z * (1 - z)) / den
    inner = sr * z * z + 2 * d * z * (1 - z) + sl * (1 - z) ** 2
    return y, math.log(d * d * inner / (den * den))


def _hand_layer(width_logits, height_logits, slopes, min_size, xs, ys, lds):
    """Weights of the one-layer flow whose transformed dim 0 receives exactly these spline parameters at x1 = 2
    (see hand_case_rqs), the rows (x0, 2) and the expected outputs."""
    import math

    def r(s):  # slope logit that yields slope s under min_knot_slope = 1/2
        return math.log((math.exp(s - 0.5) - 1.0) / (math.exp(0.5) - 1.0))

    raw = list(width_logits) + list(height_logits) + [r(s) for s in slopes]
    junk = [0.3, -0.2, 0.1, 0.4, -0.5, 0.2, 0.6, -0.1, 0.25, -0.35, 0.15, 0.05, -0.45][:len(raw)]
    W1 = np.array([[9, 9], [0.5, -1]], np.float32); b1 = np.array([0.25, 0.5], np.float32)
    W2 = np.array([[2, -1], [7, 7]], np.float32); b2 = np.array([-0.5, 3], np.float32)
    W3 = np.array([[v / 4 for v in raw] + junk, [v / 7 for v in raw] + junk[::-1]], np.float32)
    b3 = np.array([v / 4 for v in raw] + junk, np.float32)
    x = np.array([[float(v), 2.0] for v in xs], np.float32)
    y = np.array([[float(v), 2.0] for v in ys])
    logdet = np.array([float(v) for v in lds])
    lp = -0.5 * (y[:, 0] ** 2 + 4.0) - math.log(2 * math.pi) + logdet
    return dict(W1=W1, b1=b1, W2=W2, b2=b2, W3=W3, b3=b3, x=x, y=y, logdet=logdet, log_prob=lp,
                num_bins=len(width_logits), range_min=-2.0, range_max=2.0, min_bin_size=min_size, min_knot_slope=0.5)


if __name__ == "__main__":
    main()
