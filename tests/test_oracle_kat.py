"""Known-answer tests that pin the CPU oracle.

The reference has no tests and cannot run here, so these are what stands between the
oracle and a silent mistake in its restatement.  All on CPU, float64 unless noted.
"""
import os

import numpy as np
import pytest

from oracle import flows_np as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RNG = np.random.default_rng(0)


def unflat_affine(w, L, hidden=(64, 64), D=2, out=2):
    sizes = [D, *hidden, out]
    layers, p = [], 0
    for _ in range(L):
        layer = []
        for fi, fo in zip(sizes[:-1], sizes[1:]):
            W = w[p:p + fi * fo].reshape(fi, fo); p += fi * fo
            b = w[p:p + fo]; p += fo
            layer.append((W, b))
        layers.append(layer)
    assert p == w.size
    return layers


# 1 -- identity at init -----------------------------------------------------------
def test_identity_at_init():
    x = O.make_inputs("banana", 512)
    for spec in (O.make_affine_spec(8, final_std=0.0), O.make_rqs_spec(8, final_std=0.0)):
        z, ld = O.chain_forward(spec, x)
        assert np.abs(z - x).max() < 1e-12 and np.abs(ld).max() < 1e-12
        assert np.abs(O.log_prob(spec, x) - O.base_log_prob(spec, x)).max() < 1e-12
    params, flips = O.make_a_params(8)
    for net in params:
        net[-1][0][:] = 0
    assert np.abs(O.a_log_prob_chain(params, flips, x) - O.a_base_log_prob(x)).max() < 1e-12


# 2 -- round trip -----------------------------------------------------------------
@pytest.mark.parametrize("maker", [lambda: O.make_affine_spec(8, final_std=0.02),
                                   lambda: O.make_rqs_spec(8, final_std=0.05),
                                   lambda: O.make_rqs_spec(3, event_size=6, hidden=(16,), num_bins=4,
                                                           range_min=-3, range_max=3, final_std=0.2)])
def test_round_trip(maker):
    spec = maker()
    x = RNG.standard_normal((300, spec.event_size)) * 2
    z, ld = O.chain_forward(spec, x)
    xb, ldb = O.chain_inverse(spec, z)
    assert np.abs(xb - x).max() < 1e-9
    assert np.abs(ld + ldb).max() < 1e-9


# 3 -- log-det against finite differences --------------------------------------------
def _fd_logdet(fn, x, eps=1e-6):
    n, D = x.shape
    J = np.zeros((n, D, D))
    for d in range(D):
        e = np.zeros(D); e[d] = eps
        J[:, :, d] = (fn(x + e) - fn(x - e)) / (2 * eps)
    return np.log(np.abs(np.linalg.det(J)))


def test_jacobian_finite_differences():
    x = RNG.standard_normal((64, 2)) * 1.5
    for spec in (O.make_affine_spec(4, final_std=0.05), O.make_rqs_spec(3, final_std=0.1)):
        _, ld = O.chain_forward(spec, x)
        assert np.abs(_fd_logdet(lambda v: O.chain_forward(spec, v)[0], x) - ld).max() < 1e-5
    params, flips = O.make_a_params(4)
    v, tot = x, 0.0
    for net, fl in zip(reversed(params), reversed(flips)):
        v, ls = O.a_inverse_layer(net, v, fl)
        tot = tot - ls.sum(-1)

    def inv(u):
        for net, fl in zip(reversed(params), reversed(flips)):
            u, _ = O.a_inverse_layer(net, u, fl)
        return u
    assert np.abs(_fd_logdet(inv, x) - tot).max() < 1e-6


# 4 -- the density integrates to one ----------------------------------------------------
def test_normalisation():
    g = np.linspace(-12, 12, 481)
    X, Y = np.meshgrid(g, g)
    pts = np.stack([X.ravel(), Y.ravel()], -1)
    h = g[1] - g[0]
    for lp in (O.log_prob(O.make_affine_spec(4, final_std=0.01), pts),
               O.a_log_prob_chain(*O.make_a_params(8), pts)):
        assert abs(np.exp(lp).sum() * h * h - 1.0) < 2e-3


# 5 -- a case worked by hand ------------------------------------------------------------
def test_hand_computed_case():
    g = np.load(os.path.join(GOLD, "affine_hand_case.npz"))
    spec = O.FlowSpec("affine", 2, [[(g["W1"], g["b1"]), (g["W2"], g["b2"]), (g["W3"], g["b3"])]],
                      [np.array([False, True])])
    y, ld = O.coupling_layer(spec, 0, g["x"], False)
    assert np.allclose(y, [[0.5 * np.exp(0.5) - 0.5, 2.0]], atol=1e-12)
    assert np.allclose(ld, [0.5], atol=1e-12)
    assert np.allclose(O.log_prob(spec, g["x"]), g["log_prob"], atol=1e-12)
    xb, ldb = O.coupling_layer(spec, 0, y, True)
    assert np.allclose(xb, g["x"], atol=1e-12) and np.allclose(ldb, [-0.5], atol=1e-12)


def test_hand_computed_rqs_case():
    """A spline layer worked by hand (tests/golden/make_golden.py:hand_case_rqs -- rational arithmetic + math.log,
    no oracle call): both bins, the inner knot itself, both linear tails.  Pins the oracle's reading of
    distrax.RationalQuadraticSpline by an independent derivation.  The float32 rounding of
    the weights (r_k / 4, r_k / 7) moves the results by ~1e-7."""
    g = np.load(os.path.join(GOLD, "rqs_hand_case.npz"))
    spec = O.FlowSpec("rqs", 2, [[(g["W1"], g["b1"]), (g["W2"], g["b2"]), (g["W3"], g["b3"])]],
                      [np.array([False, True])], num_bins=int(g["num_bins"]), range_min=float(g["range_min"]),
                      range_max=float(g["range_max"]), min_bin_size=float(g["min_bin_size"]),
                      min_knot_slope=float(g["min_knot_slope"]))
    raw = O.conditioner_apply(spec.layers[0], np.array([[0.0, 2.0]]))[0, :7]
    xp, yp, sl = O.rqs_knots(raw[None], -2.0, 2.0, 0.5, 0.5)
    assert np.allclose(xp, [[-2, -0.75, 2]], atol=1e-6) and np.allclose(yp, [[-2, 0.75, 2]], atol=1e-6)
    assert np.allclose(sl, [[1.0, 2.0, 1.5]], atol=1e-6)
    y, ld = O.coupling_layer(spec, 0, g["x"], False)
    assert np.allclose(y, g["y"], atol=2e-6) and np.allclose(ld, g["logdet"], atol=2e-6)
    assert np.allclose(O.log_prob(spec, g["x"]), g["log_prob"], atol=4e-6)
    xb, ldb = O.coupling_layer(spec, 0, y, True)
    assert np.allclose(xb, g["x"], atol=1e-9) and np.allclose(ldb, -ld, atol=1e-9)
    # the 4-bin case of the same generator (a bin count the fused kernel takes)
    spec = O.FlowSpec("rqs", 2, [[(g["W1_k4"], g["b1_k4"]), (g["W2_k4"], g["b2_k4"]), (g["W3_k4"], g["b3_k4"])]],
                      [np.array([False, True])], num_bins=4, range_min=-2.0, range_max=2.0,
                      min_bin_size=float(g["min_bin_size_k4"]), min_knot_slope=0.5)
    y, ld = O.coupling_layer(spec, 0, g["x_k4"], False)
    assert np.allclose(y, g["y_k4"], atol=2e-6) and np.allclose(ld, g["logdet_k4"], atol=2e-6)
    assert np.allclose(O.log_prob(spec, g["x_k4"]), g["log_prob_k4"], atol=4e-6)


def test_variant_a_hand_case():
    """src/real_nvp.py by hand: MLP 1->1->2 on y=(1,3), no flip.
    act = leaky(1*2-3) = leaky(-1) = -0.01; out = -0.01*(10, 20) + (0.3, 0.2) = (0.2, 0.0)
    shift 0.2, log_scale 0: x2 = (3-0.2)*1 = 2.8."""
    net = [[np.array([[2.0]]), np.array([-3.0])], [np.array([[10.0, 20.0]]), np.array([0.3, 0.2])]]
    x, ls = O.a_inverse_layer(net, np.array([[1.0, 3.0]]), False)
    assert np.allclose(x, [[1.0, 2.8]]) and np.allclose(ls, [[0.0]])
    x, ls = O.a_inverse_layer(net, np.array([[3.0, 1.0]]), True)   # flipped: second half conditions
    assert np.allclose(x, [[2.8, 1.0]])
    y = O.a_forward_layer(net, np.array([[1.0, 2.8]]), False)
    assert np.allclose(y, [[1.0, 3.0]])


# 6 -- chain order -----------------------------------------------------------------------
def test_chain_order():
    spec = O.make_affine_spec(2, final_std=0.1)
    x = RNG.standard_normal((50, 2))
    z, _ = O.chain_forward(spec, x)
    a, _ = O.coupling_layer(spec, 1, x, False)     # Chain.forward applies the LAST layer first
    b, _ = O.coupling_layer(spec, 0, a, False)
    assert np.abs(z - b).max() < 1e-14
    wrong, _ = O.coupling_layer(spec, 1, O.coupling_layer(spec, 0, x, False)[0], False)
    assert np.abs(z - wrong).max() > 1e-3
    params, flips = O.make_a_params(2)
    lp = O.a_log_prob_chain(params, flips, x)
    v, ls1 = O.a_inverse_layer(params[1], x, flips[1])   # last layer's inverse first
    v, ls0 = O.a_inverse_layer(params[0], v, flips[0])
    assert np.abs(lp - (O.a_base_log_prob(v) - ls1.sum(-1) - ls0.sum(-1))).max() < 1e-14


# 7 -- mask semantics ---------------------------------------------------------------------
def test_mask_semantics():
    spec = O.make_affine_spec(1, final_std=0.1)
    assert list(spec.masks[0]) == [False, True]          # arange(2) % 2: dim 1 passes through
    x = RNG.standard_normal((20, 2))
    x2 = x.copy(); x2[:, 0] += 5.0                       # perturb the TRANSFORMED coordinate
    y, ld = O.coupling_layer(spec, 0, x, False)
    y2, ld2 = O.coupling_layer(spec, 0, x2, False)
    assert np.array_equal(ld, ld2)                        # (shift, log_scale) do not see it
    assert np.array_equal(y[:, 1], x[:, 1])               # masked dim unchanged
    m = O.alternating_masks(784, 3)
    assert m[0].reshape(28, 28)[:, 0].sum() == 0 and m[0].reshape(28, 28)[:, 1].all()  # column stripes
    assert np.array_equal(m[1], ~m[0]) and np.array_equal(m[2], m[0])


# 8 -- spline properties ---------------------------------------------------------------------
def test_rqs_properties():
    K = 6
    p = RNG.standard_normal((40, 3 * K + 1)) * 1.5
    xp, yp, sl = O.rqs_knots(p, -2.0, 3.0)
    assert (np.diff(xp, axis=-1) > 0).all() and (np.diff(yp, axis=-1) > 0).all() and (sl > 0).all()
    assert np.allclose(xp[:, 0], -2) and np.allclose(xp[:, -1], 3) and np.allclose(yp[:, 0], -2) and np.allclose(yp[:, -1], 3)
    assert (np.diff(xp, axis=-1) >= 1e-4 - 1e-12).all()
    for edge in (-2.0, 3.0):
        y, _ = O.rqs_fwd(np.full(40, edge), xp, yp, sl)
        assert np.allclose(y, edge, atol=1e-12)
    # knots map to knots, C1 across interior knots
    for k in range(1, K):
        xk = xp[:, k]
        y, _ = O.rqs_fwd(xk, xp, yp, sl)
        assert np.allclose(y, yp[:, k], atol=1e-10)
        eps = 1e-9 * np.minimum(xp[:, k] - xp[:, k - 1], xp[:, k + 1] - xp[:, k])
        _, l_lo = O.rqs_fwd(xk - eps, xp, yp, sl)
        _, l_hi = O.rqs_fwd(xk + eps, xp, yp, sl)
        assert np.abs(np.exp(l_lo) / sl[:, k] - 1).max() < 1e-4 and np.abs(np.exp(l_hi) / sl[:, k] - 1).max() < 1e-4
    # linear tails
    y, ld = O.rqs_fwd(np.full(40, 5.0), xp, yp, sl)
    assert np.allclose(y, 3 + 2 * sl[:, -1]) and np.allclose(ld, np.log(sl[:, -1]))
    y, ld = O.rqs_fwd(np.full(40, -4.5), xp, yp, sl)
    assert np.allclose(y, -2 - 2.5 * sl[:, 0]) and np.allclose(ld, np.log(sl[:, 0]))
    # inverse and finite differences
    x = RNG.uniform(-4, 5, 40)
    y, ld = O.rqs_fwd(x, xp, yp, sl)
    xb, ldb = O.rqs_inv(y, xp, yp, sl)
    assert np.abs(xb - x).max() < 1e-9 and np.abs(ld + ldb).max() < 1e-9
    fd = (O.rqs_fwd(x + 1e-6, xp, yp, sl)[0] - O.rqs_fwd(x - 1e-6, xp, yp, sl)[0]) / 2e-6
    assert np.abs(np.log(fd) - ld).max() < 1e-5
    # zero parameters: equal bins and unit slopes -> identity (offset makes softplus give 1 - 1e-4 + 1e-4)
    xp0, yp0, sl0 = O.rqs_knots(np.zeros((1, 3 * K + 1)), 0.0, 1.0)
    assert np.allclose(sl0, 1.0, atol=1e-12) and np.allclose(np.diff(xp0), 1 / K)
    with pytest.raises(ValueError):
        O.rqs_knots(np.zeros((1, 6)), 0, 1)
    with pytest.raises(ValueError):
        O.rqs_knots(np.zeros((1, 7)), 1, 0)


def test_uniform_base_and_normal_base():
    spec = O.make_rqs_spec(1, base="uniform01", range_min=0, range_max=1, num_bins=4)
    assert list(O.base_log_prob(spec, np.array([[0.0, 1.0], [0.5, 1.01], [-1e-9, 0.5]]))) == [0.0, -np.inf, -np.inf]
    z = RNG.standard_normal((10, 2))
    assert np.allclose(O.base_log_prob(O.make_affine_spec(1), z), O.a_base_log_prob(z))
    from scipy.stats import norm
    assert np.allclose(O.a_base_log_prob(z), norm.logpdf(z).sum(-1))


# 9 -- NLL -----------------------------------------------------------------------------------
def test_nll_is_minus_mean():
    spec = O.make_affine_spec(3, final_std=0.05)
    x = O.make_inputs("energy1", 1000)
    assert abs(O.nll(spec, x) + O.log_prob(spec, x).sum() / 1000) < 1e-12


# golden vectors ------------------------------------------------------------------------------
def test_golden_affine():
    for name in ("affine_l4_banana", "affine_l8_energy3"):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        L = int(g["num_layers"])
        spec = O.FlowSpec("affine", 2, unflat_affine(g["weights"], L), O.alternating_masks(2, L, 1))
        assert np.array_equal(O.log_prob(spec, g["x"]), g["log_prob"])
        z, ld = O.chain_forward(spec, g["x"])
        assert np.array_equal(z, g["z"]) and np.array_equal(ld, g["logdet"])
        lp32 = O.log_prob(spec, g["x"], np.float32)
        assert (np.abs(lp32 - g["log_prob"]) <= 3e-5 * np.maximum(1, np.abs(g["log_prob"]))).all()


def test_golden_split_and_rqs():
    g = np.load(os.path.join(GOLD, "split_l8_energy2.npz"))
    layers = unflat_affine(g["weights"], 8, hidden=(4,), D=1, out=2)
    params = [[[W, b] for W, b in layer] for layer in layers]
    flips = [False, True] * 4
    assert np.array_equal(O.a_log_prob_chain(params, flips, g["x"]), g["log_prob"])
    assert np.array_equal(O.a_sample_chain_from_z(params, flips, g["x"]), g["sample_from_x"])
    g = np.load(os.path.join(GOLD, "rqs_l8_blobs.npz"))
    spec = O.FlowSpec("rqs", 2, unflat_affine(g["weights"], 8, out=74), O.alternating_masks(2, 8, 1),
                      "normal", 12, -8.0, 8.0)
    assert np.array_equal(O.log_prob(spec, g["x"]), g["log_prob"])


def test_golden_gradients():
    """The training step's committed known answers (tests/golden/*_grad.npz)."""
    g = np.load(os.path.join(GOLD, "affine_l4_grad.npz"))
    spec = O.FlowSpec("affine", 2, unflat_affine(g["weights"], 4), O.alternating_masks(2, 4, 1))
    nll, grads = O.affine_nll_and_grad(spec, g["x"])
    flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in grads for W, b in layer])
    assert nll == float(g["nll"]) and np.array_equal(flat, g["grad"])
    nll2, grads2 = O.flow_nll_and_grad(spec, g["x"])          # the general restatement agrees with the golden too
    flat2 = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in grads2 for W, b in layer])
    assert abs(nll2 - nll) < 1e-12 and np.abs(flat2 - g["grad"]).max() < 1e-12
    g = np.load(os.path.join(GOLD, "rqs_l3_grad.npz"))
    spec = O.FlowSpec("rqs", 2, unflat_affine(g["weights"], 3, out=50), O.alternating_masks(2, 3, 1), "normal", 8,
                      -6.0, 6.0)
    nll, grads = O.flow_nll_and_grad(spec, g["x"])
    flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in grads for W, b in layer])
    assert nll == float(g["nll"]) and np.array_equal(flat, g["grad"])


def test_synthetic_inputs_match_generator_statistics():
    """make_inputs restates src/densities/*.py: check the moments those generators imply."""
    b = O.make_inputs("banana", 200000).astype(np.float64)
    assert abs(b[:, 1].std() - np.sqrt(3)) < 0.02 and abs(b[:, 0].mean() - (1.0 - 2.0)) < 0.02
    g = O.make_inputs("gaussian_blobs", 200000).astype(np.float64)
    assert abs(np.hypot(g[:, 0], g[:, 1]).mean() - 2.0) < 0.05
    e = O.make_inputs("energy2", 200000).astype(np.float64)
    assert abs(e[:, 0].min() + 1.5 * np.pi) < 1e-3 and abs((e[:, 1] - np.sin(np.pi * e[:, 0] / 2)).std() - 0.5) < 0.01
    m = O.make_inputs("mnist_like", 10)
    assert m.shape == (10, 784) and m.min() >= 0 and m.max() < 1 and np.all(m * 256 == np.round(m * 256))


def test_c_restatement_matches_numpy_oracle():
    """oracle/nf_oracle.c (the cpu_baseline "port") against oracle/flows_np.py."""
    from oracle import c_oracle as CO
    spec = O.make_affine_spec(8, final_std=0.02)
    x = O.make_inputs("energy3", 5000)
    ref = O.log_prob(spec, x)
    got = CO.affine_masked_log_prob(spec, x, 2)
    assert (np.abs(got - ref) <= 3e-5 * np.maximum(1, np.abs(ref))).all()
    spec5 = O.make_affine_spec(3, event_size=5, hidden=(33,), final_std=0.03, first_parity=0)
    x5 = RNG.standard_normal((300, 5)).astype(np.float32)
    assert np.abs(CO.affine_masked_log_prob(spec5, x5) - O.log_prob(spec5, x5)).max() < 2e-5
    params, flips = O.make_a_params(8)
    y = O.make_inputs("energy2", 3000)
    assert np.abs(CO.affine_split_log_prob(params, flips, y) - O.a_log_prob_chain(params, flips, y)).max() < 3e-5


def test_philox_known_answers():
    """Random123's published Philox4x32-10 vectors pin the sampler's restatement."""
    kat = [((0, 0), 0, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8))]
    for (lo, hi), key, want in kat:
        got = O.philox4x32_10(np.array([lo]), np.array([hi]), key)
        assert tuple(int(g[0]) for g in got) == want
    # all-ones vector: counter ffffffff x4, key ffffffff x2 -> 408f276d 41c83b0e a20bc7c6 6d5451fd;
    # the sampler keeps c2 = c3 = 0, so only structural properties are checked beyond the vector above
    z = O.base_sample("normal", 100000, 2, 11)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs(np.corrcoef(z.T)[0, 1]) < 0.01
    u = O.base_sample("uniform01", 1000, 784, 5)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.002
    assert not np.array_equal(O.base_sample("normal", 8, 2, 1), O.base_sample("normal", 8, 2, 2))


def test_density_restatement_statistics():
    """oracle density_sample (the Philox-driven restatement of src/densities/*.py) has the moments
    the reference's generators imply, and agrees in distribution with make_inputs."""
    b = O.density_sample("banana", 200000, 3)
    assert abs(b[:, 1].std() - np.sqrt(3)) < 0.02 and abs(b[:, 0].mean() + 1.0) < 0.02
    g = O.density_sample("gaussian_blobs", 200000, 3)
    assert abs(np.hypot(g[:, 0], g[:, 1]).mean() - 2.0) < 0.05
    ang = np.round(np.arctan2(g[:, 1], g[:, 0]) / (np.pi / 4)) % 8
    assert np.abs(np.bincount(ang.astype(int), minlength=8) / 200000 - 0.125).max() < 0.01
    e1 = O.density_sample("energy1", 200000, 3)
    assert abs(e1[:, 1].std() - 2.0) < 0.02 and abs(np.abs(e1[:, 0]).mean() - (2 - 4 / 32)) < 0.05
    for name in ("energy2", "energy3", "energy4"):
        e = O.density_sample(name, 200000, 3)
        ref = O.make_inputs(name, 200000).astype(np.float64)
        assert abs(e[:, 0].min() + 1.5 * np.pi) < 1e-3 and abs(e[:, 0].max() - 1.5 * np.pi) < 1e-3
        assert abs(e[:, 1].mean() - ref[:, 1].mean()) < 0.02 and abs(e[:, 1].std() - ref[:, 1].std()) < 0.02


def test_gradient_restatement_against_finite_differences():
    """oracle affine_nll_and_grad (reverse mode written by hand) vs central differences in float64."""
    spec = O.make_affine_spec(3, final_std=0.05)
    x = O.make_inputs("gaussian_blobs", 64)
    nll, grads = O.affine_nll_and_grad(spec, x)
    assert abs(nll - O.nll(spec, x)) < 1e-12
    rng = np.random.default_rng(0)
    for l in range(3):
        for j in range(3):
            for which in (0, 1):
                arr = spec.layers[l][j][which]
                for _ in range(4):
                    idx = tuple(rng.integers(0, s) for s in arr.shape)

                    def nll_with(val):
                        lay = [[(W.astype(np.float64).copy(), b.astype(np.float64).copy()) for W, b in layer]
                               for layer in spec.layers]
                        lay[l][j][which][idx] = val
                        return O.nll(O.FlowSpec("affine", 2, lay, spec.masks), x)
                    old, eps = float(arr[idx]), 1e-5
                    fd = (nll_with(old + eps) - nll_with(old - eps)) / (2 * eps)
                    assert abs(fd - grads[l][j][which][idx]) <= 1e-6 * max(1.0, abs(fd))
    # one Adam step from zero moments moves every weight by lr * sign(g) (bias-corrected)
    p = np.array([1.0, -2.0, 3.0]); g = np.array([0.5, -4.0, 1e-3])
    p2, m, v = O.adam_reference(p, g, np.zeros(3), np.zeros(3), 1, 1e-2)
    assert np.allclose(p2, p - 1e-2 * np.sign(g), atol=1e-6)


def test_variant_a_gradient_restatement_against_finite_differences():
    params, flips = O.make_a_params(3)
    rng = np.random.default_rng(1)
    for net in params:
        for j in range(2):
            net[j][1] = (rng.standard_normal(net[j][1].shape) * 0.1).astype(np.float32)
    y = O.make_inputs("energy2", 50)
    nll, grads = O.a_nll_and_grad(params, flips, y)
    assert abs(nll - O.a_nll(params, flips, y)) < 1e-12
    for l in range(3):
        for j in range(2):
            for which in range(2):
                arr = params[l][j][which]
                for idx in np.ndindex(arr.shape):
                    def f(val):
                        p2 = [[[W.astype(np.float64).copy(), b.astype(np.float64).copy()] for W, b in net] for net in params]
                        p2[l][j][which][idx] = val
                        return O.a_nll(p2, flips, y)
                    old, eps = float(arr[idx]), 1e-6
                    fd = (f(old + eps) - f(old - eps)) / (2 * eps)
                    assert abs(fd - grads[l][j][which][idx]) <= 1e-7 * max(1.0, abs(fd))


@pytest.mark.parametrize("kind,D,hidden,K", [("rqs", 2, (16, 16), 4), ("rqs", 5, (12,), 3), ("affine", 4, (8, 8), 0),
                                             ("rqs", 2, (64, 64), 12)])
def test_flow_gradient_matches_finite_differences(kind, D, hidden, K):
    """KAT 10: flow_nll_and_grad (what jax.grad(loss_fn) returns for the masked flows, spline
    Jacobian by complex step) against central differences of the oracle NLL in float64.
    Inputs reach beyond the spline range, so the linear tails take part."""
    rng = np.random.default_rng(D * 10 + K)
    if kind == "rqs":
        spec = O.make_rqs_spec(3, event_size=D, hidden=hidden, num_bins=K, range_min=-3.0, range_max=3.0,
                               final_std=0.3)
    else:
        spec = O.make_affine_spec(3, event_size=D, hidden=hidden, final_std=0.2)
    spec.layers = [[(np.asarray(W, np.float64), np.asarray(b, np.float64) + 0.01 * rng.standard_normal(b.shape))
                    for W, b in lay] for lay in spec.layers]
    x = rng.standard_normal((200, D)) * 2.0
    nll, g = O.flow_nll_and_grad(spec, x)
    assert abs(nll - O.nll(spec, x)) < 1e-12
    for l in range(3):
        for j, (W, b) in enumerate(spec.layers[l]):
            for which, arr in ((0, W), (1, b)):
                scale = np.abs(g[l][j][which]).max()
                for _ in range(4):
                    idx = tuple(rng.integers(0, s) for s in arr.shape)
                    old, eps = arr[idx], 1e-6
                    arr[idx] = old + eps
                    fp = O.nll(spec, x)
                    arr[idx] = old - eps
                    fm = O.nll(spec, x)
                    arr[idx] = old
                    assert abs((fp - fm) / (2 * eps) - g[l][j][which][idx]) <= 1e-6 * max(scale, 1e-3), (l, j, which, idx)


def test_flow_gradient_agrees_with_the_affine_restatement():
    spec = O.make_affine_spec(4, final_std=0.05)
    x = O.make_inputs("banana", 500)
    nll_a, g_a = O.affine_nll_and_grad(spec, x)
    nll_b, g_b = O.flow_nll_and_grad(spec, x)
    assert abs(nll_a - nll_b) < 1e-12
    for la, lb in zip(g_a, g_b):
        for (wa, ba), (wb, bb) in zip(la, lb):
            assert np.allclose(wa, wb, rtol=1e-10, atol=1e-14) and np.allclose(ba, bb, rtol=1e-10, atol=1e-14)
