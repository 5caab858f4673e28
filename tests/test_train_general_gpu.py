"""GPU parity of the general training step (nf_generic_bwd.hip): d NLL / d weights of the masked
coupling kinds -- affine and rational-quadratic spline, any event size and conditioner widths,
including the fused spline kernel's flows -- against the oracle's float64 reverse mode
(oracle.flows_np.flow_nll_and_grad, itself checked against finite differences in
tests/test_oracle_kat.py).  Tolerance per tensor: 2e-4 of its largest entry (float32 accumulation
over the batch), as for the fused affine backward in tests/test_parity_gpu.py.

Reference: jax.grad(loss_fn) + optax.adam in `update` (src/maf_toy_densities.py:167-176,
src/maf_mnist.py:159-168, src/real_nvp_toy_densities.py:164-173)."""
import numpy as np
import pytest

from oracle import flows_np as O
from test_parity_gpu import (_flat, affine_flow, assert_close, assert_spline_forward, assert_spline_inverse,
                             assert_spline_log_prob, rqs_flow)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nfa(nf_lib):
    from normalizing_flows_amd import _lib, flow
    _lib.require_device()
    return flow


def check_grads(f, spec, x, tol=2e-4):
    nll, grads = f.nll_grad(x)
    rnll, rgrads = O.flow_nll_and_grad(spec, x)
    assert abs(nll - rnll) <= 2e-5 * max(1.0, abs(rnll)), (nll, rnll)
    for l in range(len(spec.layers)):
        for j in range(len(spec.layers[l])):
            for which in (0, 1):
                got, ref = grads[l][j][which], rgrads[l][j][which]
                assert got.shape == ref.shape
                scale = max(np.abs(ref).max(), 1e-6)
                assert np.abs(got - ref).max() <= tol * scale, (l, j, which, np.abs(got - ref).max(), scale)
    return grads, rgrads


@pytest.mark.parametrize("K,L,parity,rows,density", [(12, 8, 1, 4096, "banana"), (12, 2, 0, 1000, "gaussian_blobs"),
                                                     (8, 3, 1, 33, "energy2"), (4, 5, 0, 2000, "banana"),
                                                     # every bin count rqs2d_bwd::chain_kernel is instantiated for;
                                                     # 70001 rows: 2188 tiles, more than one per wave, a ragged last one
                                                     (10, 3, 1, 500, "banana"), (6, 4, 0, 777, "energy2"),
                                                     (12, 2, 1, 70001, "gaussian_blobs")])
def test_spline_2d_gradient(nfa, K, L, parity, rows, density):
    """BASELINE.json configs[2]'s flow: since round 4 the fused, activation-free backward (nf_rqs2d_bwd.hip; the
    forward kernel records each layer's input, one backward launch recomputes everything else).
    banana leaves [-8, 8]: the linear tails' gradients take part."""
    spec = O.make_rqs_spec(L, num_bins=K, final_std=0.05, first_parity=parity)
    f = rqs_flow(nfa, spec, parity)
    assert f.kernel_name == "rqs2d_h64_mfma"
    x = O.make_inputs(density, rows, seed=K + L)
    grads, _ = check_grads(f, spec, x)
    P = 3 * K + 1
    for l in range(L):
        # the masked input's row of the first Linear and the pass-through dim's parameter columns get exactly zero
        masked_row = 0 if spec.masks[l][1] else 1
        assert np.all(grads[l][0][0][masked_row] == 0)
        passthrough = 0 if spec.masks[l][0] else 1
        assert np.all(grads[l][-1][0][:, passthrough * P:(passthrough + 1) * P] == 0)
        assert np.all(grads[l][-1][1][passthrough * P:(passthrough + 1) * P] == 0)


@pytest.mark.parametrize("D,hidden,K,L,parity,lo,hi", [(6, (32, 32), 4, 3, 0, -3.0, 3.0), (5, (24,), 8, 2, 1, -4.0, 4.0),
                                                       (3, (40, 20, 36), 3, 2, 1, -2.0, 2.0), (4, (16,), 16, 2, 0, -5.0, 5.0)])
def test_spline_general_gradient(nfa, D, hidden, K, L, parity, lo, hi):
    spec = O.make_rqs_spec(L, event_size=D, hidden=hidden, num_bins=K, range_min=lo, range_max=hi, final_std=0.2,
                           first_parity=parity)
    f = rqs_flow(nfa, spec, parity)
    assert f.kernel_name == "generic_gemm"
    x = np.random.default_rng(D + K).standard_normal((1500, D)).astype(np.float32) * 2.0
    check_grads(f, spec, x)


@pytest.mark.parametrize("D,hidden,L,parity", [(4, (64, 64), 4, 1), (7, (16, 16), 2, 1), (2, (32, 32), 4, 0), (5, (33,), 3, 0)])
def test_affine_general_gradient(nfa, D, hidden, L, parity):
    spec = O.make_affine_spec(L, event_size=D, hidden=hidden, final_std=0.05, first_parity=parity)
    f = affine_flow(nfa, spec, parity)
    assert f.kernel_name == "generic_gemm"
    x = np.random.default_rng(D * 10 + L).standard_normal((3001, D)).astype(np.float32) * 1.5
    check_grads(f, spec, x)


def test_general_gradient_matches_the_fused_backward(nfa):
    """Same model through both backward implementations: D = 2 / (64, 64) is the fused kernel's,
    the general path is reached with a third hidden layer of identity-free shape -- not the same
    model; so both are held to the oracle on their own models, and here only the oracle's two
    restatements are tied together on the fused kernel's model."""
    spec = O.make_affine_spec(3, final_std=0.05)
    x = O.make_inputs("banana", 2000)
    _, g_a = O.affine_nll_and_grad(spec, x)
    _, g_b = O.flow_nll_and_grad(spec, x)
    assert np.abs(_flat(g_a) - _flat(g_b)).max() <= 1e-12


@pytest.mark.parametrize("rows", [64, 73])
def test_mnist_shape_gradient(nfa, monkeypatch, rows):
    """BASELINE.json configs[4] shapes (784-D, MLP 512x3, K = 8 on [0, 1], Uniform base), 2 layers, 64 rows; 73 rows: a
    ragged second pass of the last Linear's LDS-staged weight gradient (tn_few_rows_kernel), and the same gradient
    through the 32 x 32 tiles (NF_NO_TN_FEW_ROWS)."""
    if rows == 73:
        x = O.make_inputs("mnist_like", rows)
        spec = O.make_rqs_spec(2, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                               base="uniform01", final_std=0.1)
        _, g_wide = rqs_flow(nfa, spec).nll_grad(x)
        monkeypatch.setenv("NF_NO_TN_FEW_ROWS", "1")
        _, g_tiles = rqs_flow(nfa, spec).nll_grad(x)
        scale = np.abs(_flat(g_tiles)).max()
        assert np.abs(_flat(g_wide) - _flat(g_tiles)).max() <= 2e-5 * scale
        assert not np.array_equal(_flat(g_wide), _flat(g_tiles))   # the switch does select another kernel
        monkeypatch.delenv("NF_NO_TN_FEW_ROWS")
    x = O.make_inputs("mnist_like", rows)
    spec = O.make_rqs_spec(2, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                           base="uniform01", final_std=0.1)
    f = rqs_flow(nfa, spec)
    check_grads(f, spec, x, tol=5e-4)   # 392 transformed dims feed every conditioner weight


def test_small_batch_gradient_leaves_nothing_behind(nfa):
    """Below 512 rows the general backward walk does not zero its gradient buffer: every weight-gradient product stores
    instead of adding (on a second stream, beside the data-gradient chain).  A second call on other rows must not see
    the first one's gradient, a larger call (which zeroes and adds) in between neither."""
    spec = O.make_rqs_spec(3, event_size=40, hidden=(96, 80), num_bins=6, range_min=-3.0, range_max=3.0, final_std=0.05)
    rng = np.random.default_rng(3)
    x1 = (rng.standard_normal((200, 40)) * 1.2).astype(np.float32)
    x2 = (rng.standard_normal((77, 40)) * 1.2).astype(np.float32)
    x3 = (rng.standard_normal((700, 40)) * 1.2).astype(np.float32)
    f = rqs_flow(nfa, spec)
    assert f.kernel_name == "generic_gemm"
    f.nll_grad(x1)
    _, g2 = f.nll_grad(x2)
    _, g3 = f.nll_grad(x3)
    _, g2_again = f.nll_grad(x2)
    _, g2_fresh = rqs_flow(nfa, spec).nll_grad(x2)
    _, g3_fresh = rqs_flow(nfa, spec).nll_grad(x3)
    assert np.array_equal(_flat(g2), _flat(g2_fresh)) and np.array_equal(_flat(g2_again), _flat(g2_fresh))
    assert np.array_equal(_flat(g3), _flat(g3_fresh))
    check_grads(f, spec, x2, tol=5e-4)


@pytest.mark.parametrize("D,hidden,K,L,sizes", [(300, (64, 96), 4, 3, (128, 77, 200, 128, 33, 128)),
                                                 # the 784-D shape: the LDS-staged wide weight gradient, 64 split-K slabs
                                                 (784, (512, 512, 512), 8, 2, (128, 64, 33))])
def test_two_stream_backward_is_the_one_stream_backward(nfa, monkeypatch, D, hidden, K, L, sizes):
    """Below 512 rows the weight-gradient products run on a second stream beside the data-gradient chain and store
    into a gradient buffer that is not zeroed.  Same kernels, same order of additions per entry: six optimiser steps
    must leave bit-identical weights and losses with and without it (NF_NO_BWD_OVERLAP) -- any race between the two
    streams would show here."""
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(L, event_size=D, hidden=hidden, num_bins=K, range_min=-3.0, range_max=3.0, final_std=0.05)
    rng = np.random.default_rng(11)
    batches = [(rng.standard_normal((rows, D)) * 1.2).astype(np.float32) for rows in sizes]

    def run():
        f = rqs_flow(nfa, spec)
        opt = training.adam(1e-3)
        losses = [f.train_step(x, opt) for x in batches]
        return losses, _flat(f.get_weights())

    l_two, w_two = run()
    monkeypatch.setenv("NF_NO_BWD_OVERLAP", "1")
    l_one, w_one = run()
    assert l_two == l_one
    assert np.array_equal(w_two, w_one)


def test_spline_adam_steps_and_fused_kernel_follows(nfa):
    """Three `update` steps on the configs[2] flow against optax.adam restated in float64; afterwards
    the fused kernel (log_prob) and the general path (training) must agree on the new weights."""
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(4, final_std=0.05)
    f = rqs_flow(nfa, spec)
    opt = training.adam(1e-3)
    p = _flat([[(W.astype(np.float64), b.astype(np.float64)) for W, b in layer] for layer in spec.layers])
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in range(1, 4):
        x = O.make_inputs("gaussian_blobs", 2048, seed=50 + step)
        loss = f.train_step(x, opt)
        cur = O.make_rqs_spec(4, final_std=0.05)
        cur.layers = f.unpack_weights(p.astype(np.float64))
        rl, rg = O.flow_nll_and_grad(cur, x)
        assert abs(loss - rl) <= 2e-5 * max(1.0, abs(rl)), (step, loss, rl)
        p, m, v = O.adam_reference(p, _flat(rg), m, v, step, 1e-3)
    got = _flat(f.get_weights())
    assert np.abs(got - p).max() <= 0.05 * 3 * 1e-3, np.abs(got - p).max()
    assert np.abs(got - _flat(spec.layers)).max() > 1e-3
    new_spec = O.make_rqs_spec(4, final_std=0.05)
    new_spec.layers = f.get_weights()
    x = O.make_inputs("gaussian_blobs", 3000, seed=7)
    ref = O.log_prob(new_spec, x)
    assert f.kernel_name == "rqs2d_h64_mfma"
    assert_spline_log_prob(f.log_prob(x), new_spec, x, "fused spline log_prob after training")
    assert abs(f.nll(x) + ref.mean()) <= 1e-5 * max(1.0, abs(ref.mean()))
    # new weights from the host restart the optimiser state and reach both representations
    f.set_weights(spec.layers)
    assert_spline_log_prob(f.log_prob(x), spec, x, "after set_weights")
    check_grads(f, spec, x)
    # ... and the restarted optimiser (moments, step count, both weight images refreshed on the device after every step)
    # walks exactly like a fresh handle's
    fresh = rqs_flow(nfa, spec)
    for step in range(2):
        xb = O.make_inputs("gaussian_blobs", 2048, seed=60 + step)
        f.train_step(xb, opt)
        fresh.train_step(xb, opt)
    wf, wn = _flat(f.get_weights()), _flat(fresh.get_weights())
    assert np.abs(wf - wn).max() <= 1e-7 * max(1.0, np.abs(wn).max())
    assert np.array_equal(f.log_prob(x), fresh.log_prob(x))


def test_spline_training_reduces_nll(nfa):
    """The reference's maf_toy_densities.py loop in miniature: zero-initialised final Linear
    (identity flow), Adam, gaussian_blobs."""
    from normalizing_flows_amd import training
    spec = O.make_rqs_spec(4, final_std=0.0)
    f = rqs_flow(nfa, spec)
    opt = training.adam(2e-3)
    x_eval = O.make_inputs("gaussian_blobs", 8192, seed=999)
    before = f.nll(x_eval)
    for step in range(80):
        f.train_step(O.make_inputs("gaussian_blobs", 2048, seed=step), opt, want_nll=False)
    after = f.nll(x_eval)
    assert np.isfinite(after) and after < before - 0.3, (before, after)


def test_split_kind_on_the_general_path_still_refuses(nfa):
    from normalizing_flows_amd import _lib
    from normalizing_flows_amd._lib import NFError
    params, _ = O.make_a_params(2, sizes=(2, 8, 4))
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=4, num_layers=2, hidden=(8,),
                         activation=_lib.ACT_LEAKY_RELU)
    f = nfa.CouplingFlow(cfg, [[(W, b) for W, b in net] for net in params])
    with pytest.raises(NFError):
        f.nll_grad(np.zeros((8, 4), np.float32))


def test_gradient_all_reduce_single_rank(nfa):
    """With a communicator attached the gradient goes through the RCCL all-reduce (one rank: identity)."""
    from normalizing_flows_amd import sharding
    spec = O.make_rqs_spec(2, final_std=0.05)
    f = rqs_flow(nfa, spec)
    x = O.make_inputs("gaussian_blobs", 1000, seed=4)
    _, g0 = f.nll_grad(x)
    comm = sharding.RcclComm(1, 0, sharding.RcclComm.new_unique_id())
    f.attach_comm(comm)
    _, g1 = f.nll_grad(x)
    f.attach_comm(None)
    comm.close()
    assert np.array_equal(_flat(g0), _flat(g1))


def test_golden_gradients(nfa):
    """HIP against the committed gradient vectors (tests/golden/*_grad.npz): the fused affine backward
    and the general spline backward."""
    import os
    from normalizing_flows_amd import _lib
    from test_oracle_kat import GOLD, unflat_affine
    g = np.load(os.path.join(GOLD, "affine_l4_grad.npz"))
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
    f = nfa.CouplingFlow(cfg, unflat_affine(g["weights"], 4))
    nll, grads = f.nll_grad(g["x"])
    assert abs(nll - float(g["nll"])) <= 1e-5
    ref = unflat_affine(g["grad"], 4)
    for l in range(4):
        for j in range(3):
            for which in (0, 1):
                r = ref[l][j][which]
                assert np.abs(grads[l][j][which] - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-6), (l, j, which)
    g = np.load(os.path.join(GOLD, "rqs_l3_grad.npz"))
    cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=3, hidden=(64, 64), rqs_bins=8,
                         rqs_range_min=-6.0, rqs_range_max=6.0)
    f = nfa.CouplingFlow(cfg, unflat_affine(g["weights"], 3, out=50))
    nll, grads = f.nll_grad(g["x"])
    assert abs(nll - float(g["nll"])) <= 1e-5 * max(1.0, abs(float(g["nll"])))
    ref = unflat_affine(g["grad"], 3, out=50)
    for l in range(3):
        for j in range(3):
            for which in (0, 1):
                r = ref[l][j][which]
                assert np.abs(grads[l][j][which] - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-6), (l, j, which)



@pytest.mark.parametrize("case", range(12))
def test_random_shapes_forward_and_gradient(nfa, case):
    """Seeded random masked flows on the general path: event size 2..11, one to three hidden layers of
    width 8..96, affine or spline (K 2..14, random range), either mask parity, ragged row counts --
    log_prob, both chain walks and the gradient against the oracle."""
    from normalizing_flows_amd import _lib
    rng = np.random.default_rng(1000 + case)
    D = int(rng.integers(2, 12))
    hidden = tuple(int(rng.integers(2, 25)) * 4 for _ in range(int(rng.integers(1, 4))))
    L = int(rng.integers(1, 5))
    parity = int(rng.integers(0, 2))
    rows = int(rng.integers(1, 700))
    if D == 2 and hidden == (64, 64):
        hidden = (64, 60)   # keep it on the general path
    if case % 2 == 0:
        K = int(rng.integers(2, 15))
        lo = -float(rng.uniform(2.0, 6.0))
        hi = float(rng.uniform(2.0, 6.0))
        spec = O.make_rqs_spec(L, event_size=D, hidden=hidden, num_bins=K, range_min=lo, range_max=hi,
                               final_std=0.15, first_parity=parity, seed=case)
        f = rqs_flow(nfa, spec, parity)
    else:
        spec = O.make_affine_spec(L, event_size=D, hidden=hidden, final_std=0.02, first_parity=parity, seed=case)
        f = affine_flow(nfa, spec, parity)
    assert f.kernel_name == "generic_gemm"
    x = (rng.standard_normal((rows, D)) * 2.0).astype(np.float32)
    _, z = assert_spline_forward(f, spec, x, f"case {case}")
    assert_spline_inverse(f, spec, z, f"case {case} inverse")
    check_grads(f, spec, x, tol=3e-4)


@pytest.mark.parametrize("kind", ["affine", "rqs"])
def test_leaky_relu_conditioner_forward_and_gradient(nfa, kind):
    """The C ABI offers leaky_relu (slope 0.01, jax.nn.leaky_relu's default as in src/real_nvp.py:166)
    for the masked kinds too; the reference's B/C/D scripts use relu, so this is the library's own
    extension, held to the oracle like everything else."""
    from normalizing_flows_amd import _lib
    if kind == "affine":
        spec = O.make_affine_spec(3, event_size=6, hidden=(40, 24), final_std=0.05)
        kid = _lib.KIND_AFFINE_MASKED
        extra = {}
    else:
        spec = O.make_rqs_spec(2, event_size=4, hidden=(32, 32), num_bins=6, range_min=-4.0, range_max=4.0,
                               final_std=0.15)
        kid = _lib.KIND_RQS_MASKED
        extra = dict(rqs_bins=6, rqs_range_min=-4.0, rqs_range_max=4.0)
    spec.activation = "leaky_relu"
    rng = np.random.default_rng(8)
    for layer in spec.layers:       # biases off zero so that negative pre-activations are common
        for j in range(len(layer) - 1):
            W, b = layer[j]
            layer[j] = (W, (b + 0.3 * rng.standard_normal(b.shape)).astype(np.float32))
    cfg = nfa.FlowConfig(kind=kid, event_dim=spec.event_size, num_layers=len(spec.layers),
                         hidden=tuple(w.shape[1] for w, _ in spec.layers[0][:-1]), first_mask_parity=1,
                         activation=_lib.ACT_LEAKY_RELU, **extra)
    f = nfa.CouplingFlow(cfg, spec.layers)
    x = (rng.standard_normal((900, spec.event_size)) * 1.5).astype(np.float32)
    ref = O.log_prob(spec, x)
    assert_spline_log_prob(f.log_prob(x), spec, x, "leaky_relu log_prob")
    relu_spec_lp = O.log_prob(type(spec)(spec.kind, spec.event_size, spec.layers, spec.masks, spec.base,
                                         spec.num_bins, spec.range_min, spec.range_max), x)
    assert np.abs(relu_spec_lp - ref).max() > 1e-3      # the activation matters on this data
    check_grads(f, spec, x)


@pytest.mark.parametrize("kind,D,hidden,K,rows", [("rqs", 2, (64, 64), 12, 5000), ("rqs", 5, (24,), 8, 33),
                                                  ("rqs", 3, (48, 40, 56), 4, 1), ("affine", 2, (50, 50), 0, 4097),
                                                  ("affine", 6, (64, 64, 64), 0, 777), ("affine", 7, (16,), 0, 130)])
def test_narrow_backward_both_forms(nfa, monkeypatch, kind, D, hidden, K, rows):
    """Narrow conditioners (every Linear <= 64 wide): the one-launch MLP backward (nf_mlp_small_bwd.hip: one, two
    and three hidden layers, odd D, one to many workgroups, ragged last tile) and the generic per-Linear form
    (NF_NO_MLP_SMALL_BWD) against the oracle and against each other."""
    if kind == "rqs":
        spec = O.make_rqs_spec(3, event_size=D, hidden=hidden, num_bins=K, range_min=-4.0, range_max=4.0,
                               final_std=0.05, first_parity=1)
        make = lambda: rqs_flow(nfa, spec, 1)
    else:
        spec = O.make_affine_spec(3, event_size=D, hidden=hidden, final_std=0.05, first_parity=1)
        make = lambda: affine_flow(nfa, spec, 1)
    x = (np.random.default_rng(D * 7 + rows).standard_normal((rows, D)) * 1.5).astype(np.float32)
    f = make()
    fused, ref = check_grads(f, spec, x, tol=3e-4 if rows < 64 else 2e-4)
    monkeypatch.setenv("NF_NO_MLP_SMALL_BWD", "1")
    g = make()
    generic, _ = check_grads(g, spec, x, tol=3e-4 if rows < 64 else 2e-4)
    for l in range(len(spec.layers)):
        for j in range(len(spec.layers[l])):
            for which in (0, 1):
                scale = max(np.abs(ref[l][j][which]).max(), 1e-6)
                assert np.abs(fused[l][j][which] - generic[l][j][which]).max() <= 4e-4 * scale


@pytest.mark.parametrize("kind", ["affine", "rqs"])
def test_padded_narrow_flow_trains_in_the_callers_layout(nfa, kind):
    """hidden 50/50 (the reference scripts' default) on the fused kernels through zero padding: gradients arrive in
    the caller's layout and match the oracle, three Adam steps follow optax.adam restated in float64 (the padding
    units receive zero gradients and stay zero: the flow after training is still the caller's 50/50 flow), SGD
    moves the weights by exactly -lr * gradient."""
    from normalizing_flows_amd import training
    if kind == "affine":
        spec = O.make_affine_spec(4, hidden=(50, 50), final_std=0.03)
        f = affine_flow(nfa, spec, pad=True)
        assert f.kernel_name == "affine2d_h64_mfma"
        fresh = lambda: O.make_affine_spec(4, hidden=(50, 50), final_std=0.03)
    else:
        spec = O.make_rqs_spec(4, hidden=(50, 50), num_bins=4, final_std=0.05)
        f = rqs_flow(nfa, spec, pad=True)
        assert f.kernel_name == "rqs2d_h64_mfma"
        fresh = lambda: O.make_rqs_spec(4, hidden=(50, 50), num_bins=4, final_std=0.05)
    x = O.make_inputs("banana", 3000, seed=11)
    check_grads(f, spec, x)
    opt = training.adam(1e-3)
    p = _flat([[(W.astype(np.float64), b.astype(np.float64)) for W, b in layer] for layer in spec.layers])
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in range(1, 4):
        xb = O.make_inputs("gaussian_blobs", 2048, seed=70 + step)
        loss = f.train_step(xb, opt)
        cur = fresh()
        cur.layers = f.unpack_weights(p.astype(np.float64))
        rl, rg = O.flow_nll_and_grad(cur, xb)
        assert abs(loss - rl) <= 2e-5 * max(1.0, abs(rl)), (step, loss, rl)
        p, m, v = O.adam_reference(p, _flat(rg), m, v, step, 1e-3)
    got = _flat(f.get_weights())
    assert got.shape == p.shape
    assert np.abs(got - p).max() <= 0.05 * 3 * 1e-3, np.abs(got - p).max()
    new_spec = fresh()
    new_spec.layers = f.get_weights()
    ref = O.log_prob(new_spec, x)
    if kind == "rqs":
        assert_spline_log_prob(f.log_prob(x), new_spec, x, "padded flow after training")
    else:
        assert_close(f.log_prob(x), ref, "padded flow after training")
    g_before = f.nll_grad(x)[1]
    w_before = _flat(f.get_weights())
    f.train_step(x, training.sgd(0.01))
    assert np.abs(_flat(f.get_weights()) - (w_before - 0.01 * _flat(g_before))).max() <= 1e-6


@pytest.mark.parametrize("kind", ["affine_fused", "rqs_general"])
def test_skip_nonfinite_updates(nfa, kind):
    """Option "skip_nonfinite_updates" (optax.apply_if_finite): a batch whose loss overflows float32 leaves weights
    and optimiser moments untouched; without it the step makes every weight NaN, which is what the reference's
    unwrapped optax.adam does with a non-finite gradient."""
    from normalizing_flows_amd import training
    if kind == "affine_fused":
        spec = O.make_affine_spec(4, final_std=0.05)
        make = lambda: affine_flow(nfa, spec)
    else:
        spec = O.make_rqs_spec(3, event_size=4, hidden=(32, 32), num_bins=8, range_min=-4.0, range_max=4.0, final_std=0.05)
        make = lambda: rqs_flow(nfa, spec)
    D = spec.event_size
    good = (np.random.default_rng(3).standard_normal((256, D)) * 1.5).astype(np.float32)
    bad = good.copy()
    bad[17, 0] = 3e38   # the base density's square overflows
    opt = training.adam(1e-3)
    f = make()
    f.set_option("skip_nonfinite_updates", "1")
    assert f.get_option("skip_nonfinite_updates") == "1"
    w0 = _flat(f.get_weights())
    loss = f.train_step(bad, opt)
    assert not np.isfinite(loss)
    assert np.array_equal(_flat(f.get_weights()), w0)
    loss = f.train_step(good, opt)
    w1 = _flat(f.get_weights())
    assert np.isfinite(loss) and np.isfinite(w1).all() and np.abs(w1 - w0).max() > 1e-4
    # the skipped step does not count: Adam's bias correction on the applied step is that of step 1, so the
    # weights equal those of a fresh handle after its first step on the same batch (optax.apply_if_finite keeps
    # the inner optimiser's count where it was); with the count advanced the update would be 0.53x / 0.71x as large
    h = make()
    h.set_option("skip_nonfinite_updates", "1")
    h.train_step(good, opt)
    assert np.abs(_flat(h.get_weights()) - w1).max() <= 1e-7 * max(1.0, np.abs(w1).max())
    g = make()
    g.train_step(bad, opt)
    assert not np.isfinite(_flat(g.get_weights())).all()
    # Toggling the option keeps ONE count of applied steps: on (one skipped, one applied) -> off (one applied) ->
    # on (one applied) must be the plain three-step run on the same good batches.  (The device-side count used
    # under the option was once seeded a single time: switched back on it missed the steps applied meanwhile,
    # and the host's count used while it was off had counted the skipped one.)
    b2 = (np.random.default_rng(4).standard_normal((256, D)) * 1.5).astype(np.float32)
    b3 = (np.random.default_rng(5).standard_normal((256, D)) * 1.5).astype(np.float32)
    plain = make()
    for xb in (good, b2, b3):
        plain.train_step(xb, opt)
    t = make()
    t.set_option("skip_nonfinite_updates", "1")
    t.train_step(bad, opt)
    t.train_step(good, opt)
    t.set_option("skip_nonfinite_updates", "0")
    t.train_step(b2, opt)
    t.set_option("skip_nonfinite_updates", "1")
    t.train_step(b3, opt)
    wp, wt = _flat(plain.get_weights()), _flat(t.get_weights())
    assert np.abs(wt - wp).max() <= 2e-6 * max(1.0, np.abs(wp).max()), np.abs(wt - wp).max()
    # set_weights on a trained handle restarts the optimiser -- moments, host count AND the device-side count of applied
    # steps the option keeps: five steps, new weights, two more steps == a fresh handle's two steps (left at t = 5 the
    # bias correction would make the first step after the restart ~3x too large)
    r = make()
    r.set_option("skip_nonfinite_updates", "1")
    for xb in (good, b2, b3, good, b2):
        r.train_step(xb, opt)
    r.set_weights(spec.layers)
    fresh = make()
    fresh.set_option("skip_nonfinite_updates", "1")
    for xb in (b3, good):
        r.train_step(xb, opt)
        fresh.train_step(xb, opt)
    wr, wf = _flat(r.get_weights()), _flat(fresh.get_weights())
    assert np.abs(wr - wf).max() <= 1e-7 * max(1.0, np.abs(wf).max()), np.abs(wr - wf).max()
    # ... and switching the option off afterwards hands over the restarted count, not the old one
    r.set_option("skip_nonfinite_updates", "0")
    fresh.set_option("skip_nonfinite_updates", "0")
    r.train_step(b2, opt)
    fresh.train_step(b2, opt)
    wr, wf = _flat(r.get_weights()), _flat(fresh.get_weights())
    assert np.abs(wr - wf).max() <= 1e-7 * max(1.0, np.abs(wf).max()), np.abs(wr - wf).max()


def test_spline_2d_gradient_fused_against_plan_and_uniform_base(nfa, monkeypatch):
    """The fused backward of the 2-D spline flow (i) against the general plan's gradient of the SAME handle shape
    (NF_NO_RQS2D_BWD=1: the round-3 route, dual-number spline Jacobian, stored activations) -- two independent
    derivations of the same numbers -- and (ii) with the Uniform(0,1) base of src/maf_mnist.py on [0, 1]^2: the base
    contributes no gradient inside the cube (G starts at zero), the log-dets all of it."""
    spec = O.make_rqs_spec(4, num_bins=8, range_min=-6.0, range_max=6.0, final_std=0.08)
    x = O.make_inputs("energy2", 3000, seed=5)
    f = rqs_flow(nfa, spec)
    fused, ref = check_grads(f, spec, x)
    monkeypatch.setenv("NF_NO_RQS2D_BWD", "1")
    g = rqs_flow(nfa, spec)
    monkeypatch.delenv("NF_NO_RQS2D_BWD")
    plan, _ = check_grads(g, spec, x)
    for l in range(4):
        for j in range(3):
            for which in (0, 1):
                scale = max(np.abs(ref[l][j][which]).max(), 1e-6)
                assert np.abs(fused[l][j][which] - plan[l][j][which]).max() <= 4e-4 * scale
    spec_u = O.make_rqs_spec(3, num_bins=8, range_min=0.0, range_max=1.0, base="uniform01", final_std=0.3)
    xu = np.random.default_rng(8).uniform(0.02, 0.98, (2000, 2)).astype(np.float32)
    fu = rqs_flow(nfa, spec_u)
    assert fu.kernel_name == "rqs2d_h64_mfma"
    check_grads(fu, spec_u, xu)


@pytest.mark.parametrize("K", [21, 32])
def test_spline_gradient_many_bins(nfa, K):
    """Largest bin counts of the general backward (K <= 32): the spline's backward kernel stages 3K+1 parameters and
    their gradient per item through LDS and takes fewer items per pass when 256 would not fit."""
    spec = O.make_rqs_spec(2, event_size=3, hidden=(24,), num_bins=K, range_min=-4.0, range_max=4.0, final_std=0.05)
    f = rqs_flow(nfa, spec)
    x = (np.random.default_rng(K).standard_normal((700, 3)) * 1.5).astype(np.float32)
    check_grads(f, spec, x)
