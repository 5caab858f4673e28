"""GPU parity of the optional "bf16x3" hidden matmul (nf_flow_set_option) against the CPU oracle.

Same tolerance as tests/test_parity_gpu.py: |hip - oracle_f64| <= 1e-5 * max(1, |oracle_f64|).
The mode must hold it wherever the exact-float32 kernel does; it is an alternative evaluation
order of the same float32 arithmetic (split operands, float32 accumulate), not a precision level.
"""
import numpy as np
import pytest

from oracle import flows_np as O
from test_parity_gpu import DENSITIES, affine_flow, assert_close, rqs_flow, split_flow

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nfa(nf_lib):
    from normalizing_flows_amd import _lib, flow
    _lib.require_device()
    return flow


def x3_flow(nfa, spec, parity=1):
    f = affine_flow(nfa, spec, parity)
    f.set_option("hidden_matmul", "bf16x3")
    return f


@pytest.mark.parametrize("density", DENSITIES)
def test_x3_log_prob_l8(nfa, density):
    spec = O.make_affine_spec(8, final_std=0.02)
    f = x3_flow(nfa, spec)
    x = O.make_inputs(density, 4096)
    ref = O.log_prob(spec, x)
    got = f.log_prob(x)
    assert_close(got, ref, f"bf16x3 log_prob {density}")
    # and statistically no further from float64 than the float32 restatement is
    # (measured: rms 1.9e-6 against 2.7e-6 for NumPy float32 and 1.8e-6 for the exact-float32 kernel)
    e32 = O.log_prob(spec, x, np.float32).astype(np.float64) - ref
    err = got.astype(np.float64) - ref
    scale = np.maximum(1.0, np.abs(ref))
    rms = lambda e: float(np.sqrt(np.mean((e / scale) ** 2)))
    assert rms(err) <= max(2 * rms(e32), 2e-7)
    assert (np.abs(err) / scale).max() <= max(2 * (np.abs(e32) / scale).max(), 2e-6)


@pytest.mark.parametrize("L,parity", [(1, 1), (1, 0), (2, 1), (3, 0), (5, 1), (9, 1), (10, 1), (12, 0), (19, 1)])
def test_x3_layer_counts_and_parity(nfa, L, parity):
    spec = O.make_affine_spec(L, final_std=0.02, first_parity=parity)
    f = x3_flow(nfa, spec, parity)
    x = O.make_inputs("energy2", 1500)
    assert_close(f.log_prob(x), O.log_prob(spec, x), f"L={L} parity={parity}")


@pytest.mark.parametrize("rows", [1, 2, 31, 32, 33, 63, 64, 65, 127, 1000, 4097, 70001, 300000, 1 << 20])
def test_x3_ragged_rows(nfa, rows):
    spec = O.make_affine_spec(8, final_std=0.02)
    f = x3_flow(nfa, spec)
    x = O.make_inputs("gaussian_blobs", rows, seed=rows)
    assert_close(f.log_prob(x), O.log_prob(spec, x), f"rows={rows}")


@pytest.mark.parametrize("tune", ["1,8", "2,8", "4,8", "1,16", "1,4", "2,4"])
def test_x3_every_instantiation(nfa, monkeypatch, tune):
    monkeypatch.setenv("NF_AFFINE2D_X3_TUNE", tune)
    spec = O.make_affine_spec(8, final_std=0.02)
    f = x3_flow(nfa, spec)
    x = O.make_inputs("banana", 1 << 19, seed=5)
    assert_close(f.log_prob(x), O.log_prob(spec, x), f"tune {tune}")
    assert abs(f.nll(x) - O.nll(spec, x)) <= 1e-5


def test_x3_identity_at_init(nfa):
    from normalizing_flows_amd import _lib
    spec = O.make_affine_spec(8, final_std=0.0)
    f = x3_flow(nfa, spec)
    x = O.make_inputs("banana", 2048)
    y, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    assert np.array_equal(y, x)
    assert np.all(ld == 0)


def test_x3_chain_directions_and_sampling(nfa):
    from normalizing_flows_amd import _lib
    spec = O.make_affine_spec(8, final_std=0.02)
    f = x3_flow(nfa, spec)
    x = O.make_inputs("energy4", 3000)
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    zr, ldr = O.chain_forward(spec, x)
    assert_close(z, zr, "chain forward y")
    assert_close(ld, ldr, "chain forward logdet")
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    xr, ldr2 = O.chain_inverse(spec, z)
    assert_close(xb, xr, "chain inverse x")
    assert_close(ldb, ldr2, "chain inverse logdet")
    assert np.abs(xb - x).max() <= 2e-5 * max(1, np.abs(x).max())
    for direction in (_lib.DIR_FIRST_TO_LAST_FORWARD, _lib.DIR_LAST_TO_FIRST_INVERSE):
        y2, ld2 = f.chain_apply(direction, x, layer_begin=2, layer_end=7)
        g = affine_flow(nfa, spec)
        y3, ld3 = g.chain_apply(direction, x, layer_begin=2, layer_end=7)
        assert np.abs(y2 - y3).max() <= 1e-5 * max(1, np.abs(y3).max())
        assert np.abs(ld2 - ld3).max() <= 1e-5


def test_x3_option_switches_and_errors(nfa):
    from normalizing_flows_amd._lib import NFError
    spec = O.make_affine_spec(8, final_std=0.02)
    x = O.make_inputs("banana", 5000)
    f = affine_flow(nfa, spec)
    f.set_option("hidden_matmul", "f32")   # whatever NF_HIDDEN_MATMUL made the default
    lp_f32 = f.log_prob(x)
    f.set_option("hidden_matmul", "bf16x3")
    lp_x3 = f.log_prob(x)
    f.set_option("hidden_matmul", "f32")
    assert np.array_equal(f.log_prob(x), lp_f32)      # back on the exact kernel, bit for bit
    assert not np.array_equal(lp_x3, lp_f32)            # and the option did select another kernel
    d = np.abs(lp_x3.astype(np.float64) - lp_f32) / np.maximum(1.0, np.abs(lp_f32))
    assert d.max() <= 4e-6          # the two modes agree to a few float32 ulps
    assert (d == 0).mean() > 0.5    # and mostly bit for bit
    with pytest.raises(NFError):
        f.set_option("hidden_matmul", "fp8")
    with pytest.raises(NFError):
        f.set_option("no_such_option", "1")
    # flows on other kernels refuse the mode instead of ignoring it
    g = split_flow(nfa, O.make_a_params(4)[0])
    with pytest.raises(NFError):
        g.set_option("hidden_matmul", "bf16x3")
    g.set_option("hidden_matmul", "f32")   # the default is accepted everywhere


def test_x3_follows_weight_updates(nfa):
    """set_weights and nf_train_step both invalidate the split image."""
    from normalizing_flows_amd import training
    spec_a = O.make_affine_spec(4, final_std=0.02)
    spec_b = O.make_affine_spec(4, final_std=0.02, seed=99)
    x = O.make_inputs("banana", 4096)
    f = x3_flow(nfa, spec_a)
    assert_close(f.log_prob(x), O.log_prob(spec_a, x), "before")
    f.set_weights(spec_b.layers)
    assert_close(f.log_prob(x), O.log_prob(spec_b, x), "after set_weights")
    f.train_step(x, training.adam(1e-3))
    spec_c = O.FlowSpec("affine", 2, f.get_weights(), spec_b.masks, "normal")
    assert_close(f.log_prob(x), O.log_prob(spec_c, x), "after a training step")


def test_x3_full_size_properties(nfa):
    """BASELINE.json configs[1] batch (2^20): invertibility and the additive log-det."""
    from normalizing_flows_amd import _lib
    spec = O.make_affine_spec(8, final_std=0.02)
    f = x3_flow(nfa, spec)
    x = O.make_inputs("banana", 1 << 20, seed=11)
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    assert np.abs(xb - x).max() <= 5e-5 * max(1, np.abs(x).max())
    assert np.abs(ld + ldb).max() <= 2e-5
    lp = f.log_prob(x)
    base = -0.5 * (z.astype(np.float64) ** 2).sum(-1) - np.log(2 * np.pi)
    assert np.abs(lp - (base + ld)).max() <= 1e-5 * max(1, np.abs(lp).max())
    assert abs(f.nll(x) + float(np.mean(lp.astype(np.float64)))) <= 1e-6


# ------------------------------------------------------------------ general path (nf_gemm_x3.hip)
@pytest.mark.parametrize("D,hidden,L,parity", [(2, (64, 64, 64), 2, 1), (4, (64, 64), 4, 1), (40, (128, 96), 3, 0),
                                               (64, (100, 260), 2, 1), (36, (64,), 3, 1)])
def test_x3_generic_affine(nfa, D, hidden, L, parity):
    """Every Linear with K >= 32, N >= 64 (multiples of 4) runs the split-bf16 GEMM, the others the
    float32 one; the chain must hold the same tolerance either way."""
    from normalizing_flows_amd import _lib
    spec = O.make_affine_spec(L, event_size=D, hidden=hidden, final_std=0.03, first_parity=parity)
    f = affine_flow(nfa, spec, parity)
    assert f.kernel_name == "generic_gemm"
    x = np.random.default_rng(D * 100 + L).standard_normal((3001, D)).astype(np.float32) * 1.5
    f.set_option("hidden_matmul", "f32")   # whatever NF_HIDDEN_MATMUL made the default
    lp32 = f.log_prob(x)
    f.set_option("hidden_matmul", "bf16x3")
    lp = f.log_prob(x)
    ref = O.log_prob(spec, x)
    assert_close(lp, ref, "generic bf16x3 log_prob")
    assert not np.array_equal(lp, lp32)  # the option did switch kernels
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    zr, ldr = O.chain_forward(spec, x)
    assert_close(z, zr, "generic bf16x3 z")
    assert_close(ld, ldr, "generic bf16x3 logdet")
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    xr, ldr2 = O.chain_inverse(spec, z)
    assert_close(xb, xr, "generic bf16x3 inverse")
    assert_close(ldb, ldr2, "generic bf16x3 inverse logdet")
    f.set_option("hidden_matmul", "f32")
    assert np.array_equal(f.log_prob(x), lp32)
    # new weights under the option
    spec2 = O.make_affine_spec(L, event_size=D, hidden=hidden, final_std=0.03, first_parity=parity, seed=77)
    f.set_option("hidden_matmul", "bf16x3")
    f.set_weights(spec2.layers)
    assert_close(f.log_prob(x), O.log_prob(spec2, x), "generic bf16x3 after set_weights")


@pytest.mark.parametrize("rows", [1, 127, 128, 129, 1000])
def test_x3_generic_ragged_rows(nfa, rows):
    spec = O.make_affine_spec(3, event_size=8, hidden=(96, 160), final_std=0.03)
    f = affine_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    x = np.random.default_rng(rows).standard_normal((rows, 8)).astype(np.float32)
    assert_close(f.log_prob(x), O.log_prob(spec, x), f"rows={rows}")


def test_x3_generic_rqs(nfa):
    spec = O.make_rqs_spec(3, event_size=6, hidden=(64, 64), num_bins=8, range_min=-4.0, range_max=4.0,
                           final_std=0.1, first_parity=0)
    f = rqs_flow(nfa, spec, 0)
    assert f.kernel_name == "generic_gemm"
    f.set_option("hidden_matmul", "bf16x3")
    x = np.random.default_rng(6).standard_normal((1500, 6)).astype(np.float32) * 2.0
    assert_close(f.log_prob(x), O.log_prob(spec, x), "rqs D=6 bf16x3")


@pytest.mark.parametrize("small_batch_form", [True, False])
def test_x3_mnist_shape(nfa, monkeypatch, small_batch_form):
    """BASELINE.json configs[4] shapes under the option; the two statements of
    tests/test_parity_gpu.py::test_rqs_mnist_shape_config5."""
    if not small_batch_form:
        monkeypatch.setenv("NF_NO_SPLITK", "1")  # the large-batch kernels (gemm_x3, fused spline epilogue) on 96 rows
    x = O.make_inputs("mnist_like", 96)
    spec = O.make_rqs_spec(2, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0,
                           range_max=1.0, base="uniform01", final_std=0.2)
    f = rqs_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    ref = O.log_prob(spec, x)
    assert_close(f.log_prob(x), ref, "mnist-shaped rqs bf16x3 (i)")
    spec = O.make_rqs_spec(2, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0,
                           range_max=1.0, base="uniform01", final_std=0.02)
    f = rqs_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    ref = O.log_prob(spec, x)
    e32 = np.abs(O.log_prob(spec, x, np.float32).astype(np.float64) - ref).max()
    got = f.log_prob(x)
    assert np.abs(got - ref).max() <= 2 * e32 + 1e-6, (np.abs(got - ref).max(), e32)


# ------------------------------------------------------------------ fused spline kernel (nf_rqs2d_x3.hip)
# The parity tests of the exact-float32 kernel, unchanged, with the mode as the process default.
import test_parity_gpu as P  # noqa: E402


@pytest.fixture
def x3_default(monkeypatch, nfa):
    """The spline tests of the exact-float32 kernel re-run under the mode, same per-sample bound
    (test_parity_gpu.SPLINE_F32_FACTOR): the knots are float64 in both kernels, which is where the digits were."""
    monkeypatch.setenv("NF_HIDDEN_MATMUL", "bf16x3")
    spec = O.make_rqs_spec(1, final_std=0.05)
    f = rqs_flow(nfa, spec)
    assert f.kernel_name == "rqs2d_h64_mfma" and f.get_option("hidden_matmul") == "bf16x3"
    f.close()


@pytest.mark.parametrize("density", ["gaussian_blobs", "banana", "energy2"])
def test_x3_rqs_toy_config3(nfa, x3_default, density):
    P.test_rqs_toy_config3(nfa, density)


@pytest.mark.parametrize("K,L,parity,rows,lo,hi", [(12, 8, 1, 4096, -8.0, 8.0), (12, 1, 0, 33, -8.0, 8.0),
                                                   (8, 3, 1, 1000, -5.0, 5.0), (4, 5, 0, 70001, -3.0, 4.0),
                                                   (12, 11, 1, 513, -8.0, 8.0), (6, 4, 1, 777, -6.0, 6.0),
                                                   (10, 2, 0, 2049, -8.0, 8.0)])
def test_x3_rqs_fused_kernel_shapes(nfa, x3_default, K, L, parity, rows, lo, hi):
    P.test_rqs_fused_kernel_shapes(nfa, K, L, parity, rows, lo, hi)


def test_x3_rqs_fused_identity_and_edges(nfa, x3_default):
    P.test_rqs_fused_identity_and_edges(nfa)


@pytest.mark.parametrize("tune", ["1,16", "1,8", "2,8"])
def test_x3_rqs_every_instantiation(nfa, monkeypatch, tune):
    monkeypatch.setenv("NF_RQS2D_X3_TUNE", tune)
    spec = O.make_rqs_spec(8, final_std=0.05)
    f = rqs_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    x = O.make_inputs("gaussian_blobs", 1 << 17, seed=9)
    ref = P.assert_spline_log_prob(f.log_prob(x), spec, x, f"rqs bf16x3 tune {tune}")
    assert abs(f.nll(x) + ref.mean()) <= 1e-5 * max(1.0, abs(ref.mean()))


def test_x3_env_default_covers_the_other_kernels(nfa, monkeypatch):
    monkeypatch.setenv("NF_HIDDEN_MATMUL", "bf16x3")
    f = affine_flow(nfa, O.make_affine_spec(2, final_std=0.02))
    assert f.get_option("hidden_matmul") == "bf16x3"
    g = affine_flow(nfa, O.make_affine_spec(2, event_size=8, hidden=(96, 64), final_std=0.02))
    assert g.kernel_name == "generic_gemm" and g.get_option("hidden_matmul") == "bf16x3"
    s = split_flow(nfa, O.make_a_params(4)[0])
    assert s.get_option("hidden_matmul") == "f32"   # no MFMA in that kernel: the default does not apply
    monkeypatch.delenv("NF_HIDDEN_MATMUL")
    assert affine_flow(nfa, O.make_affine_spec(2, final_std=0.02)).get_option("hidden_matmul") == "f32"
    from normalizing_flows_amd._lib import NFError
    monkeypatch.setenv("NF_HIDDEN_MATMUL", "fp8")       # a typo must not silently mean float32
    with pytest.raises(NFError):
        affine_flow(nfa, O.make_affine_spec(2, final_std=0.02))


def test_x3_spline_flow_follows_training(nfa):
    """The mode on a fused spline handle while it trains (general backward, float32): the split
    image is rebuilt from the trained weights, like the float32 one."""
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(3, final_std=0.05)
    f = rqs_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    x = O.make_inputs("gaussian_blobs", 2048, seed=3)
    for _ in range(3):
        f.train_step(x, training.adam(1e-3), want_nll=False)
    new_spec = O.make_rqs_spec(3, final_std=0.05)
    new_spec.layers = f.get_weights()
    assert np.abs(np.asarray(new_spec.layers[0][0][0]) - np.asarray(spec.layers[0][0][0])).max() > 1e-4
    ref = O.log_prob(new_spec, x)
    e32 = np.abs(O.log_prob(new_spec, x, np.float32) - ref).max()
    assert_close(f.log_prob(x), ref, "bf16x3 spline log_prob after training", 6 * e32)
    f.set_option("hidden_matmul", "f32")
    assert_close(f.log_prob(x), ref, "f32 spline log_prob after training", 4 * e32)


def test_x3_general_path_training_keeps_the_split_weights_current(nfa):
    from normalizing_flows_amd import training
    spec = O.make_affine_spec(2, event_size=8, hidden=(96, 64), final_std=0.05)
    f = affine_flow(nfa, spec)
    f.set_option("hidden_matmul", "bf16x3")
    x = np.random.default_rng(2).standard_normal((1000, 8)).astype(np.float32)
    for _ in range(2):
        f.train_step(x, training.sgd(1e-2), want_nll=False)
    new_spec = O.make_affine_spec(2, event_size=8, hidden=(96, 64), final_std=0.05)
    new_spec.layers = f.get_weights()
    assert f.get_option("hidden_matmul") == "bf16x3"
    assert_close(f.log_prob(x), O.log_prob(new_spec, x), "general bf16x3 log_prob after training")


@pytest.mark.parametrize("L,parity,rows,density", [(8, 1, 4096, "banana"), (3, 0, 1000, "energy2"), (1, 1, 33, "gaussian_blobs")])
def test_x3_fused_affine_backward(nfa, L, parity, rows, density):
    """With the mode on, the fused backward runs its two operand-feeding products (recompute of a2,
    g1 = W2 g2) as bf16x3 too; the gradient holds the same 2e-4-of-the-tensor's-maximum bound."""
    from normalizing_flows_amd import training
    from test_parity_gpu import _flat
    spec = O.make_affine_spec(L, final_std=0.03, first_parity=parity)
    f = x3_flow(nfa, spec, parity)
    x = O.make_inputs(density, rows, seed=L)
    nll, grads = f.nll_grad(x)
    rnll, rgrads = O.affine_nll_and_grad(spec, x)
    assert abs(nll - rnll) <= 1e-5 * max(1.0, abs(rnll))
    for l in range(L):
        for j in range(3):
            for which in (0, 1):
                got, ref = grads[l][j][which], rgrads[l][j][which]
                scale = max(np.abs(ref).max(), 1e-6)
                assert np.abs(got - ref).max() <= 2e-4 * scale, (l, j, which, np.abs(got - ref).max(), scale)
    # and the float32 form gives (slightly) different numbers: the mode did switch kernels
    f.set_option("hidden_matmul", "f32")
    _, g32 = f.nll_grad(x)
    assert not np.array_equal(_flat(g32), _flat(grads))
    # two optimiser steps under the mode follow the float64 Adam restatement
    f.set_option("hidden_matmul", "bf16x3")
    p = _flat([[(W.astype(np.float64), b.astype(np.float64)) for W, b in layer] for layer in spec.layers])
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in (1, 2):
        f.train_step(x, training.adam(1e-3), want_nll=False)
        cur = O.FlowSpec("affine", 2, f.unpack_weights(p), spec.masks)
        _, rg = O.affine_nll_and_grad(cur, x)
        p, m, v = O.adam_reference(p, _flat(rg), m, v, step, 1e-3)
    assert np.abs(_flat(f.get_weights()) - p).max() <= 0.05 * 2 * 1e-3


def test_x3_after_training_in_f32_uses_the_trained_weights(nfa):
    """General path: the split weight images are built on the host; switching the mode on after optimiser steps
    that ran on the device must pack the TRAINED weights (and the fused last Linear must follow them too)."""
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(2, event_size=6, hidden=(96, 128), num_bins=8, range_min=-4.0, range_max=4.0, final_std=0.05)
    f = rqs_flow(nfa, spec)
    assert f.kernel_name == "generic_gemm"
    x = (np.random.default_rng(5).standard_normal((700, 6)) * 1.5).astype(np.float32)
    opt = training.adam(5e-3)
    for _ in range(3):
        f.train_step(x, opt)
    trained = O.make_rqs_spec(2, event_size=6, hidden=(96, 128), num_bins=8, range_min=-4.0, range_max=4.0, final_std=0.05)
    trained.layers = f.get_weights()
    ref = O.log_prob(trained, x)
    assert np.abs(ref - O.log_prob(spec, x)).max() > 1e-2     # the steps did change the density
    e32 = np.abs(O.log_prob(trained, x, np.float32) - ref).max()
    assert_close(f.log_prob(x), ref, "f32 after training", 4 * e32)
    f.set_option("hidden_matmul", "bf16x3")
    assert_close(f.log_prob(x), ref, "bf16x3 after training in f32", 6 * e32)
    f.train_step(x, opt)   # and one more step under the mode
    trained.layers = f.get_weights()
    ref = O.log_prob(trained, x)
    assert_close(f.log_prob(x), ref, "bf16x3 after a step under the mode", 6 * np.abs(O.log_prob(trained, x, np.float32) - ref).max())


@pytest.mark.parametrize("shape", ["padded_2d_spline", "fused_2d_affine", "general_k5"])
def test_mode_switches_interleaved_with_training(nfa, shape):
    """State across option changes, optimiser steps and weight uploads: after every transition log_prob must be the
    oracle's for the weights the handle reports (fused images, split images, the re-laid last Linear and the padded
    layout all have to follow)."""
    from normalizing_flows_amd import training
    if shape == "padded_2d_spline":
        mk = lambda: O.make_rqs_spec(4, hidden=(50, 50), num_bins=4, final_std=0.05)
        f = rqs_flow(nfa, mk(), pad=True)
        x = O.make_inputs("gaussian_blobs", 1500, seed=3)
    elif shape == "fused_2d_affine":
        mk = lambda: O.make_affine_spec(4, final_std=0.03)
        f = affine_flow(nfa, mk())
        x = O.make_inputs("banana", 1500, seed=3)
    else:
        mk = lambda: O.make_rqs_spec(2, event_size=20, hidden=(128, 128), num_bins=8, range_min=-4.0, range_max=4.0,
                                     final_std=0.03)
        f = rqs_flow(nfa, mk())
        x = (np.random.default_rng(9).standard_normal((600, 20)) * 1.5).astype(np.float32)
    spec0 = mk()
    opt = training.adam(2e-3)

    def check(what):
        cur = mk()
        cur.layers = f.get_weights()
        ref = O.log_prob(cur, x)
        e32 = np.abs(O.log_prob(cur, x, np.float32) - ref).max()
        assert_close(f.log_prob(x), ref, what, 6 * e32)
        assert abs(f.nll(x) + ref.mean()) <= 2e-5 * max(1.0, abs(ref.mean())), what
        return cur

    check("fresh")
    f.set_option("hidden_matmul", "bf16x3")
    check("bf16x3 on")
    f.train_step(x, opt)
    f.train_step(x, opt)
    check("two steps under bf16x3")
    f.set_option("hidden_matmul", "f32")
    cur = check("back to f32")
    f.train_step(x, opt)
    check("a step in f32")
    f.set_option("hidden_matmul", "bf16x3")
    check("bf16x3 on again after an f32 step")
    f.set_weights(spec0.layers)
    after = check("weights uploaded under bf16x3")
    assert np.array_equal(_flat_layers(after.layers), _flat_layers(spec0.layers))
    f.set_option("hidden_matmul", "f32")
    check("f32 after the upload")


def _flat_layers(layers):
    return np.concatenate([np.concatenate([np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()])
                           for layer in layers for W, b in layer])
