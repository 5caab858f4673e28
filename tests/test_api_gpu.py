"""The reference's scripts, re-typed against this package (imports swapped, nothing else),
checked against the oracle.  Each helper below mirrors the named reference function."""
import numpy as np
import pytest

from oracle import flows_np as O

pytestmark = pytest.mark.gpu


def tol(ref):
    return 1e-5 * np.maximum(1.0, np.abs(ref))


@pytest.fixture(scope="module")
def distrax(nf_lib):
    import normalizing_flows_amd as pkg
    from normalizing_flows_amd import _lib
    _lib.require_device()
    return pkg


def make_flow_model_affine(distrax, event_shape, num_layers, hidden_sizes, rng, final_std):
    """src/real_nvp_toy_densities.py:74-110 with `jnp` -> `np` and haiku -> make_conditioner."""
    mask = np.arange(0, np.prod(event_shape)) % 2
    mask = np.reshape(mask, event_shape).astype(bool)

    def bijector_fn(params):
        shift, log_scale = np.split(params, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, log_scale=log_scale)

    layers = []
    for _ in range(num_layers):
        layers.append(distrax.MaskedCoupling(
            mask=mask, bijector=bijector_fn,
            conditioner=distrax.make_conditioner(event_shape, hidden_sizes, 2, rng=rng, final_std=final_std)))
        mask = np.logical_not(mask)
    flow = distrax.Inverse(distrax.Chain(layers))
    base = distrax.Independent(distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape)),
                               reinterpreted_batch_ndims=len(event_shape))
    return distrax.Transformed(base, flow), layers


def make_flow_model_rqs(distrax, event_shape, num_layers, hidden_sizes, bins, lo, hi, base_kind, rng, final_std):
    """src/maf_toy_densities.py:74-111 / src/maf_mnist.py:52-89."""
    mask = np.arange(0, np.prod(event_shape)) % 2
    mask = np.reshape(mask, event_shape).astype(bool)
    P = 3 * bins + 1

    def bijector_fn(params):
        return distrax.RationalQuadraticSpline(params, range_min=lo, range_max=hi)

    layers = []
    for _ in range(num_layers):
        layers.append(distrax.MaskedCoupling(
            mask=mask, bijector=bijector_fn,
            conditioner=distrax.make_conditioner(event_shape, hidden_sizes, P, per_dim=True, rng=rng,
                                                 final_std=final_std)))
        mask = np.logical_not(mask)
    flow = distrax.Inverse(distrax.Chain(layers))
    if base_kind == "normal":
        inner = distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape))
    else:
        inner = distrax.Uniform(low=np.zeros(event_shape), high=np.ones(event_shape))
    return distrax.Transformed(distrax.Independent(inner, reinterpreted_batch_ndims=len(event_shape)), flow), layers


def spec_of(layers, kind, event_size, **kw):
    masks = [l.mask.reshape(-1) for l in layers]
    return O.FlowSpec(kind, event_size, [l.conditioner.linears for l in layers], masks, **kw)


def test_real_nvp_toy_script(distrax):
    rng = np.random.default_rng(42)
    model, layers = make_flow_model_affine(distrax, (2,), 8, [64, 64], rng, 0.02)
    spec = spec_of(layers, "affine", 2)
    x = O.make_inputs("banana", 4096)
    lp = model.log_prob(x)                      # log_prob.apply(params, batch)
    ref = O.log_prob(spec, x)
    assert lp.shape == (4096,)
    assert (np.abs(lp - ref) <= tol(ref)).all()
    assert abs(model.nll(x) - O.nll(spec, x)) <= 1e-5     # eval_fn
    assert abs(-np.mean(lp.astype(np.float64)) - model.nll(x)) <= 1e-9
    # sample.apply: base draw on the host, Chain.inverse on the GPU
    s = model.sample(seed=7, sample_shape=(1000,))
    z = model.distribution.sample(seed=7, sample_shape=(1000,))
    zr = O.base_sample("normal", 1000, 2, 7)              # the oracle's restatement of the Philox draw
    assert s.shape == (1000, 2) and z.shape == (1000, 2)
    assert np.abs(z - zr).max() <= 2e-6
    assert (np.abs(s - O.sample_from_z(spec, zr)) <= 2 * tol(O.sample_from_z(spec, zr))).all()
    assert np.array_equal(s, model.bijector.forward(z))   # one device call == draw, then walk
    big = model.sample(seed=np.array([3, 4], np.uint32), sample_shape=(1 << 20,))   # N = 10^6 in the reference
    assert big.shape == (1 << 20, 2) and np.isfinite(big).all() and abs(big.mean()) < 5
    # bijector protocol on the pieces
    chain = model.bijector.bijector
    y, ld = chain.forward_and_log_det(x)
    yr, ldr = O.chain_forward(spec, x)
    assert (np.abs(y - yr) <= tol(yr)).all() and (np.abs(ld - ldr) <= tol(ldr)).all()
    xi, ldi = model.bijector.forward_and_log_det(y)       # Inverse swaps directions
    assert np.abs(xi - x).max() <= 2e-5 * np.abs(x).max()
    y1, ld1 = layers[3].forward_and_log_det(x)            # one MaskedCoupling
    y1r, ld1r = O.coupling_layer(spec, 3, x, False)
    assert (np.abs(y1 - y1r) <= tol(y1r)).all() and (np.abs(ld1 - ld1r) <= tol(ld1r)).all()
    assert np.array_equal(layers[3].forward(x), y1)
    assert np.array_equal(layers[3].forward_log_det_jacobian(x), ld1)
    # leading batch dims are kept
    lp3 = model.log_prob(x.reshape(64, 64, 2))
    assert lp3.shape == (64, 64) and np.array_equal(lp3.reshape(-1), lp)


def test_identity_at_init_known_answer(distrax):
    """Freshly initialised flow (zero final Linear) is the identity: log_prob == base log_prob."""
    model, _ = make_flow_model_affine(distrax, (2,), 8, [64, 64], np.random.default_rng(1), 0.0)
    x = O.make_inputs("energy1", 1024)
    lp = model.log_prob(x)
    base = model.distribution.log_prob(x)
    assert np.abs(lp - base).max() <= 4e-6   # two kernels, one ulp apart at |log_prob| ~ 16
    ref = -0.5 * (x.astype(np.float64) ** 2).sum(-1) - np.log(2 * np.pi)
    assert (np.abs(base - ref) <= tol(ref)).all()


def test_swapped_split_and_param_updates(distrax):
    rng = np.random.default_rng(3)
    event_shape = (2,)
    mask = (np.arange(2) % 2).astype(bool)

    def swapped_fn(params):
        log_scale, shift = np.split(params, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, log_scale=log_scale)

    cond = distrax.make_conditioner(event_shape, [64, 64], 2, rng=rng, final_std=0.05)
    layer = distrax.MaskedCoupling(mask=mask, bijector=swapped_fn, conditioner=cond)
    x = O.make_inputs("energy2", 512)
    W, b = cond.linears[-1]
    swapped = list(cond.linears[:-1]) + [(W[:, ::-1], b[::-1])]
    spec = O.FlowSpec("affine", 2, [swapped], [mask])
    y, ld = layer.forward_and_log_det(x)
    yr, ldr = O.coupling_layer(spec, 0, x, False)
    assert (np.abs(y - yr) <= tol(yr)).all() and (np.abs(ld - ldr) <= tol(ldr)).all()
    # new parameters are picked up (handle rebuilt on version change)
    cond.set_params([(w * 0.5, bb) for w, bb in cond.linears])
    y2, _ = layer.forward_and_log_det(x)
    assert np.abs(y2 - y).max() > 1e-4


def test_unsupported_structures_fail_loudly(distrax):
    mask = (np.arange(2) % 2).astype(bool)
    with pytest.raises(NotImplementedError):
        distrax.MaskedCoupling(mask=mask, bijector=lambda p: None, conditioner=lambda x: x)

    def tanh_fn(params):
        shift, log_scale = np.split(params, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, log_scale=np.tanh(log_scale))

    cond = distrax.make_conditioner((2,), [64, 64], 2)
    layer = distrax.MaskedCoupling(mask=mask, bijector=tanh_fn, conditioner=cond)
    with pytest.raises(NotImplementedError):
        layer.forward_and_log_det(np.zeros((4, 2), np.float32))
    # (stand-alone inner bijectors evaluate since round 4: test_standalone_inner_bijectors)
    y0, l0 = distrax.ScalarAffine(shift=0.0, log_scale=0.0).forward_and_log_det(np.zeros(3))
    assert y0.shape == (3,) and np.all(y0 == 0) and np.all(l0 == 0)

    def scale_fn(params):   # the scale= form has no coupling kernel (the reference uses log_scale)
        shift, scale = np.split(params, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, scale=scale)
    with pytest.raises(NotImplementedError):
        distrax.MaskedCoupling(mask=mask, bijector=scale_fn, conditioner=cond).forward_and_log_det(np.zeros((4, 2), np.float32))
    with pytest.raises(ValueError):
        distrax.RationalQuadraticSpline(np.zeros(5), 0.0, 1.0)
    with pytest.raises(ValueError):
        distrax.RationalQuadraticSpline(np.zeros(7), 1.0, 0.0)


def test_maf_toy_script(distrax):
    rng = np.random.default_rng(5)
    model, layers = make_flow_model_rqs(distrax, (2,), 8, [64, 64], 12, -8.0, 8.0, "normal", rng, 0.05)
    spec = spec_of(layers, "rqs", 2, num_bins=12, range_min=-8.0, range_max=8.0)
    x = O.make_inputs("gaussian_blobs", 4096)
    from test_parity_gpu import assert_spline_log_prob, sample_noise, spline_ok
    lp = model.log_prob(x)
    # per-sample bound of the spline tests (test_parity_gpu.SPLINE_F32_FACTOR): no batch-wide floor
    ref = assert_spline_log_prob(lp, spec, x, "maf_toy log_prob")
    assert abs(model.nll(x) - O.nll(spec, x)) <= 1e-5 * max(1.0, abs(O.nll(spec, x)))
    s = model.sample(seed=11, sample_shape=500)
    z = O.base_sample("normal", 500, 2, 11)
    sr = O.sample_from_z(spec, z)
    # the generative walk is the ill-conditioned direction: forward error per sample, or backward error
    # (the float64 density walk of the returned sample lands on z)
    fwd_ok = spline_ok(s, sr, sample_noise(spec, z)).all(axis=-1)
    zb, _ = O.chain_forward(spec, s)
    bwd_ok = (np.abs(zb - z) <= tol(z)).all(axis=-1)
    assert (fwd_ok | bwd_ok).all()


def test_maf_mnist_script_shapes(distrax):
    rng = np.random.default_rng(6)
    shape = (28, 28, 1)
    model, layers = make_flow_model_rqs(distrax, shape, 2, [512, 512, 512], 8, 0.0, 1.0, "uniform", rng, 0.2)
    spec = spec_of(layers, "rqs", 784, num_bins=8, range_min=0.0, range_max=1.0, base="uniform01")
    x = O.make_inputs("mnist_like", 64).reshape(64, 28, 28, 1)   # prepare_data(batch): k / 256
    lp = model.log_prob(x)
    ref = O.log_prob(spec, x.reshape(64, 784))
    assert lp.shape == (64,)
    assert (np.abs(lp - ref) <= tol(ref)).all()


def test_real_nvp_py_free_functions(distrax):
    from normalizing_flows_amd import real_nvp as R
    params, fns_config = R.init_nvp_chain(n=8)
    flips = [flip for _, flip in fns_config]
    assert flips == [False, True] * 4
    # distinct layers so that order matters
    rng = np.random.default_rng(534)
    params = [[[(rng.standard_normal(W.shape) * 0.1).astype(np.float32), (rng.standard_normal(b.shape) * 0.05).astype(np.float32)]
               for W, b in net] for net in params]
    y = O.make_inputs("energy2", 4096)
    lp = R.log_prob_nvp_chains(params, fns_config, R.log_prob_N01, y)
    ref = O.a_log_prob_chain(params, flips, y)
    assert (np.abs(lp - ref) <= tol(ref)).all()
    assert abs(R.loss(params, fns_config, y) - O.a_nll(params, flips, y)) <= 1e-5
    x = R.sample_nvp_chain(params, fns_config, R.sample_N01, 1000, 7809)
    z = R.sample_N01(1000, 7809)
    assert np.abs(z - O.base_sample("normal", 1000, 2, 7809)).max() <= 2e-6
    xr = O.a_sample_chain_from_z(params, flips, z)
    assert (np.abs(x - xr) <= tol(xr)).all()
    # single-layer functions
    f1 = R.forward_nvp(params[2], R.shift_and_log_scale_fn, y, flip=False)
    assert (np.abs(f1 - O.a_forward_layer(params[2], y, False)) <= 1e-5 * np.maximum(1, np.abs(f1))).all()
    xi, ls = R.inverse_nvp(params[3], R.shift_and_log_scale_fn, y, flip=True)
    xr, lsr = O.a_inverse_layer(params[3], y, True)
    assert (np.abs(xi - xr) <= tol(xr)).all() and (np.abs(ls - lsr) <= tol(lsr)).all()
    shift, log_scale = R.shift_and_log_scale_fn(params[0], y[:, :1])
    sr, lr = O.a_shift_and_log_scale(params[0], y[:, :1])
    assert (np.abs(shift - sr) <= tol(sr)).all() and (np.abs(log_scale - lr) <= tol(lr)).all()
    lp1 = R.log_prob_nvp(params[0], R.shift_and_log_scale_fn, R.log_prob_N01, y, flip=False)
    x0, ls0 = O.a_inverse_layer(params[0], y, False)
    ref1 = O.a_base_log_prob(x0) - ls0.sum(-1)
    assert (np.abs(lp1 - ref1) <= tol(ref1)).all()
    with pytest.raises(NotImplementedError):
        R.log_prob_nvp_chains(params, fns_config, lambda v: v.sum(-1), y)


def test_mnist_shape_sampling(distrax):
    """Transformed.sample on the 784-D flow: Uniform base draw + spline inverse walk on the GPU."""
    rng = np.random.default_rng(9)
    model, layers = make_flow_model_rqs(distrax, (28, 28, 1), 2, [512, 512, 512], 8, 0.0, 1.0, "uniform", rng, 0.1)
    spec = spec_of(layers, "rqs", 784, num_bins=8, range_min=0.0, range_max=1.0, base="uniform01")
    s = model.sample(seed=21, sample_shape=(48,))
    assert s.shape == (48, 28, 28, 1)
    z = O.base_sample("uniform01", 48, 784, 21)
    sr = O.sample_from_z(spec, z)
    from test_parity_gpu import sample_noise, spline_ok
    fwd_ok = spline_ok(s.reshape(48, 784), sr, sample_noise(spec, z)).all(axis=-1)
    zb, _ = O.chain_forward(spec, s.reshape(48, 784))
    assert (fwd_ok | (np.abs(zb - z) <= tol(z)).all(axis=-1)).all()


def test_density_generators_on_device(distrax):
    from normalizing_flows_amd import densities
    from normalizing_flows_amd.distributions import _seed_u64
    for name in densities.DENSITIES:
        got = densities.sample_density(name, 2123, 50000)
        ref = O.density_sample(name, 50000, 2123)
        assert got.shape == (50000, 2)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name
    ds = densities.make_dataset_banana(seed=2123, batch_size=128, num_batches=3)     # src/real_nvp_toy_densities.py:179
    batches = list(ds)
    assert len(batches) == 3 and batches[0].shape == (128, 2) and not np.array_equal(batches[0], batches[1])
    assert np.array_equal(next(densities.get_density("banana")(2123, 128, 1)), batches[0])
    dev = next(densities.make_dataset_energy_2(seed=5, batch_size=4096, num_batches=1, device=True))
    assert np.array_equal(dev.to_numpy(), next(densities.make_dataset_energy_2(seed=5, batch_size=4096, num_batches=1)))
    with pytest.raises(ValueError):
        densities.sample_density("moons", 1, 8)


def test_training_step_through_the_distrax_surface(distrax):
    """`update` of src/real_nvp_toy_densities.py:164-173 with the model built as the script builds it."""
    from normalizing_flows_amd import densities, training
    rng = np.random.default_rng(42)
    model, layers = make_flow_model_affine(distrax, (2,), 8, [64, 64], rng, 0.0)   # zero-init final Linear
    optimizer = training.adam(1e-3)
    train_ds = densities.make_dataset_banana(seed=2123, batch_size=4096, num_batches=80)
    valid = next(densities.make_dataset_banana(seed=2235, batch_size=8192, num_batches=1))
    before = model.nll(valid)
    spec0 = spec_of(layers, "affine", 2)
    loss0, grads = training.value_and_grad(model, valid)
    rl, rg = O.affine_nll_and_grad(spec0, valid)
    assert abs(loss0 - rl) <= 1e-5 * max(1.0, abs(rl))
    assert sorted(grads) == sorted(distrax.layers_to_haiku_params([l.conditioner.linears for l in layers]))
    g_last = grads["linear_3"]["w"]
    assert np.abs(g_last - rg[3][2][0]).max() <= 2e-4 * max(np.abs(rg[3][2][0]).max(), 1e-6)
    for batch in train_ds:
        training.update(model, optimizer, batch, want_loss=False)
    after = model.nll(valid)
    assert after < before - 0.2
    training.sync_params(model)          # conditioners now hold the trained weights
    spec1 = spec_of(layers, "affine", 2)
    assert abs(O.nll(spec1, valid) - after) <= 1e-5 * max(1.0, abs(after))
    assert np.abs(layers[0].conditioner.linears[-1][0]).max() > 0


def test_real_nvp_py_training_loop(distrax):
    """The __main__ loop of src/real_nvp.py:286-315 (sgd_update on batches of the chosen density)."""
    from normalizing_flows_amd import densities, real_nvp as R
    params, fns_config = R.init_nvp_chain(n=8)
    rng = np.random.default_rng(5)
    params = [[[(rng.standard_normal(W.shape) * 0.1).astype(np.float32), b] for W, b in net] for net in params]
    flips = [flip for _, flip in fns_config]
    valid = next(densities.make_dataset_energy_2(seed=231, batch_size=8192, num_batches=1))
    first = R.loss(params, fns_config, valid)
    nll0, grads = R.value_and_grad_loss(params, fns_config, valid)
    rn, rg = O.a_nll_and_grad(params, flips, valid)
    assert abs(nll0 - rn) <= 1e-5 * max(1.0, abs(rn))
    assert np.abs(grads[7][1][0] - rg[7][1][0]).max() <= 2e-4 * np.abs(rg[7][1][0]).max()
    for batch in densities.make_dataset_energy_2(seed=231, batch_size=128, num_batches=300):
        nll, params = R.sgd_update(params, fns_config, batch, learning_rate=1e-2)
    assert R.loss(params, fns_config, valid) < first - 0.05


def test_maf_toy_training_through_the_distrax_surface(distrax):
    """`update` of src/maf_toy_densities.py:167-176 (jax.grad(loss_fn) + optax.adam) with the spline
    model built as the script builds it; gradients keyed like the haiku parameter dict."""
    from normalizing_flows_amd import densities, training
    rng = np.random.default_rng(43)
    model, layers = make_flow_model_rqs(distrax, (2,), 4, [64, 64], 12, -8.0, 8.0, "normal", rng, 0.02)
    spec0 = spec_of(layers, "rqs", 2, num_bins=12, range_min=-8.0, range_max=8.0)
    valid = next(densities.make_dataset_gaussian_blobs(seed=2235, batch_size=4096, num_batches=1))
    loss0, grads = training.value_and_grad(model, valid)
    rl, rg = O.flow_nll_and_grad(spec0, valid)
    assert abs(loss0 - rl) <= 2e-5 * max(1.0, abs(rl))
    assert sorted(grads) == sorted(distrax.layers_to_haiku_params([l.conditioner.linears for l in layers]))
    for name, (l, j) in {"mlp/~/linear_0": (0, 0), "mlp_2/~/linear_1": (2, 1), "linear_3": (3, 2)}.items():
        for key, which in (("w", 0), ("b", 1)):
            ref = rg[l][j][which]
            assert np.abs(grads[name][key] - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), (name, key)
    optimizer = training.adam(2e-3)
    before = model.nll(valid)
    for batch in densities.make_dataset_gaussian_blobs(seed=2123, batch_size=2048, num_batches=60):
        training.update(model, optimizer, batch, want_loss=False)
    after = model.nll(valid)
    assert np.isfinite(after) and after < before - 0.2, (before, after)
    training.sync_params(model)
    spec1 = spec_of(layers, "rqs", 2, num_bins=12, range_min=-8.0, range_max=8.0)
    assert abs(O.nll(spec1, valid) - after) <= 2e-5 * max(1.0, abs(after))
    s = model.sample(seed=3, sample_shape=2000)        # the trained flow samples through the fused kernel
    assert np.isfinite(s).all()


def test_maf_mnist_training_step_shapes(distrax):
    """One `update` of src/maf_mnist.py:159-168 at its shapes (28x28x1 events, Uniform base)."""
    from normalizing_flows_amd import training
    rng = np.random.default_rng(44)
    model, layers = make_flow_model_rqs(distrax, (28, 28, 1), 2, [512, 512, 512], 8, 0.0, 1.0, "uniform", rng, 0.05)
    x = O.make_inputs("mnist_like", 64).reshape(64, 28, 28, 1)
    spec = spec_of(layers, "rqs", 784, num_bins=8, range_min=0.0, range_max=1.0, base="uniform01")
    loss, grads = training.value_and_grad(model, x)
    rl, rg = O.flow_nll_and_grad(spec, x.reshape(64, 784))
    assert abs(loss - rl) <= 2e-5 * max(1.0, abs(rl))
    ref = rg[1][3][0]
    assert grads["linear_1"]["w"].shape == ref.shape == (512, 784 * 25)
    assert np.abs(grads["linear_1"]["w"] - ref).max() <= 5e-4 * np.abs(ref).max()
    before = model.nll(x)
    for _ in range(5):
        training.update(model, training.adam(1e-4), x, want_loss=False)
    assert model.nll(x) < before


def test_option_through_the_distrax_surface(distrax):
    rng = np.random.default_rng(45)
    model, layers = make_flow_model_affine(distrax, (2,), 4, [64, 64], rng, 0.02)
    x = O.make_inputs("banana", 3000)
    spec = spec_of(layers, "affine", 2)
    ref = O.log_prob(spec, x)
    model.set_option("hidden_matmul", "f32")   # whatever NF_HIDDEN_MATMUL made the default
    lp32 = model.log_prob(x)
    model.set_option("hidden_matmul", "bf16x3")
    lpx3 = model.log_prob(x)
    assert (np.abs(lpx3 - ref) <= tol(ref)).all() and (np.abs(lp32 - ref) <= tol(ref)).all()
    assert not np.array_equal(lp32, lpx3)
    model.set_option("hidden_matmul", "f32")
    assert np.array_equal(model.log_prob(x), lp32)



def test_non_inverted_chain_and_single_coupling(distrax):
    """Transformed(base, Chain(...)) without the Inverse wrapper (the generative direction is then
    chain.forward and log_prob walks chain.inverse), and a single MaskedCoupling as the bijector."""
    rng = np.random.default_rng(46)
    model, layers = make_flow_model_affine(distrax, (2,), 5, [64, 64], rng, 0.03)
    spec = spec_of(layers, "affine", 2)
    base = model.distribution
    plain = distrax.Transformed(base, distrax.Chain(layers))
    x = O.make_inputs("energy2", 2000)
    z, ild = O.chain_inverse(spec, x)
    ref = O.base_log_prob(spec, z) + ild
    got = plain.log_prob(x)
    assert (np.abs(got - ref) <= tol(ref)).all()
    s = plain.sample(seed=5, sample_shape=300)
    sr, _ = O.chain_forward(spec, O.base_sample("normal", 300, 2, 5))
    assert (np.abs(s - sr) <= 2 * tol(sr)).all()
    one = distrax.Transformed(base, layers[2])
    y, ld = O.coupling_layer(spec, 2, x, True)            # log_prob of Transformed(base, b) uses b.inverse
    ref1 = O.base_log_prob(spec, y) + ld
    assert (np.abs(one.log_prob(x) - ref1) <= tol(ref1)).all()
    yf, ldf = layers[2].forward_and_log_det(x)
    yr, ldr = O.coupling_layer(spec, 2, x, False)
    assert (np.abs(yf - yr) <= tol(yr)).all() and (np.abs(ldf - ldr) <= tol(ldr)).all()


def test_two_threads_two_handles_first_launches(distrax):
    """include/nf_hip.h: handles are not thread-safe, distinct handles are usable from distinct threads.  The kernels
    that need up to 160 KB of dynamic LDS set that attribute once per (device, kernel) behind a lock that is held until
    it IS set (ensure_dynamic_lds; a process-wide `static bool` let a second thread launch early: VERDICT r03 weak 10).
    A fresh process (tests/_threads_worker.py) lets two threads meet at the very first launch of the affine and the
    spline chain kernels, three rounds each, different batch sizes (different kernel instantiations): every result
    against the oracle."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_threads_worker.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res.pop("errors") == []
    assert len(res) == 12 and all(v <= 1e-5 for v in res.values()), res


def test_standalone_inner_bijectors(distrax):
    """The distrax.Bijector protocol (README.md:32-57 of the reference) on the INNER bijectors themselves, outside any
    MaskedCoupling: ScalarAffine(shift, log_scale | scale) (src/real_nvp_toy_densities.py:86) and
    RationalQuadraticSpline(params, range_min, range_max) (src/maf_toy_densities.py:87) evaluate explicit parameter
    arrays through nf_elementwise_apply -- against oracle.scalar_affine_* / rqs_fwd / rqs_inv, tails, range edges and
    the hand-worked spline case included; event_ndims_in = 0, so the log-det has x's shape."""
    import os
    rng = np.random.default_rng(42)
    # --- ScalarAffine, log_scale form, distrax-style broadcasting ([5,1] against [1,3])
    x = rng.standard_normal((5, 3)).astype(np.float32) * 3
    shift = rng.standard_normal((1, 3)).astype(np.float32)
    ls = (rng.standard_normal((5, 1)) * 0.7).astype(np.float32)
    b = distrax.ScalarAffine(shift=shift, log_scale=ls)
    y, ld = b.forward_and_log_det(x)
    yr, ldr = O.scalar_affine_fwd(x.astype(np.float64), shift.astype(np.float64), ls.astype(np.float64))
    assert y.shape == (5, 3) and ld.shape == (5, 3)
    assert (np.abs(y - yr) <= tol(yr)).all() and (np.abs(ld - ldr) <= tol(ldr)).all()
    xb, ldb = b.inverse_and_log_det(y)
    xr, ldbr = O.scalar_affine_inv(y.astype(np.float64), shift.astype(np.float64), ls.astype(np.float64))
    assert (np.abs(xb - xr) <= tol(xr)).all() and (np.abs(ldb - ldbr) <= tol(ldbr)).all()
    assert (np.abs(b.forward(x) - y) == 0).all() and (np.abs(b.inverse_log_det_jacobian(y) - ldb) == 0).all()
    # the hand-worked affine layer (tests/golden/affine_hand_case.npz): shift and log_scale from its 2-2-2-2 relu MLP
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "affine_hand_case.npz"))
    xm = g["x"].astype(np.float64) * np.array([0.0, 1.0])          # layer 0 passes dim 1 through (mask [F, T])
    h = np.maximum(xm @ g["W1"] + g["b1"], 0)
    h = np.maximum(h @ g["W2"] + g["b2"], 0)
    p = (h @ g["W3"] + g["b3"]).astype(np.float32)
    y0, l0 = distrax.ScalarAffine(shift=p[:, 0], log_scale=p[:, 1]).forward_and_log_det(g["x"][:, 0])
    assert abs(y0[0] - g["y"][0, 0]) <= 1e-6 and abs(l0[0] - g["logdet"][0]) <= 1e-6
    # --- scale form (negative scales included): y = scale x + shift, logdet = log|scale|
    scale = np.array([[2.0], [-0.5], [1e-3], [-7.0], [1.0]], np.float32)
    b = distrax.ScalarAffine(shift=shift, scale=scale)
    y, ld = b.forward_and_log_det(x)
    assert (np.abs(y - (scale.astype(np.float64) * x + shift)) <= tol(y)).all()
    assert (np.abs(ld - np.broadcast_to(np.log(np.abs(scale.astype(np.float64))), (5, 3))) <= 1e-6).all()
    xb, ldb = b.inverse_and_log_det(y)
    assert (np.abs(xb - x) <= 1e-4 * np.maximum(1, np.abs(x))).all() and (np.abs(ldb + ld) <= 1e-6).all()
    with pytest.raises(ValueError):
        distrax.ScalarAffine(shift=shift, scale=scale, log_scale=ls)
    # --- RationalQuadraticSpline: register kernels (K = 4, 12), the run-time-K kernel (K = 5), ragged sizes
    for K, rows, D, lo, hi in ((12, 777, 2, -8.0, 8.0), (4, 33, 3, -3.0, 4.0), (5, 1000, 1, -2.0, 2.0), (8, 1, 1, 0.0, 1.0)):
        P = 3 * K + 1
        params = (rng.standard_normal((rows, D, P)) * 1.5).astype(np.float32)
        x = (rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), (rows, D))).astype(np.float32)
        x.reshape(-1)[:2] = (lo, hi) if x.size >= 2 else lo                         # exact range edges
        b = distrax.RationalQuadraticSpline(params, range_min=lo, range_max=hi)
        xp, yp, sl = O.rqs_knots(params, lo, hi)
        yr, ldr = O.rqs_fwd(x, xp, yp, sl)
        y, ld = b.forward_and_log_det(x)
        assert y.shape == x.shape and ld.shape == x.shape
        # bound per element: 1e-5 scaled, or -- random logits of scale 1.5 make bins as narrow as min_bin_size, where no
        # float32 evaluation resolves float64 -- 8 x what the NumPy float32 restatement itself misses float64 by there
        # (tests/test_parity_gpu.py's per-sample float32-noise bound); at least 99 % within the plain 1e-5
        y32, ld32 = O.rqs_fwd(x, *O.rqs_knots(params, lo, hi, dtype=np.float32))
        ny, nl = np.abs(y32.astype(np.float64) - yr), np.abs(ld32.astype(np.float64) - ldr)
        ok = (np.abs(y - yr) <= tol(yr)) & (np.abs(ld - ldr) <= tol(ldr))
        assert ok.mean() >= 0.99, (K, ok.mean())
        assert (np.abs(y - yr) <= np.maximum(tol(yr), 8 * ny)).all() and (np.abs(ld - ldr) <= np.maximum(tol(ldr), 8 * nl)).all(), \
            (K, ok.mean(), np.abs(y - yr).max(), np.abs(ld - ldr).max())
        # inverse: by its forward error, or -- where dx/dy is large -- by its backward error
        yq = yr.astype(np.float32)
        xb, ldb = b.inverse_and_log_det(yq)
        xr, ldbr = O.rqs_inv(yq, xp, yp, sl)
        fwd_ok = (np.abs(xb - xr) <= tol(xr)) & (np.abs(ldb - ldbr) <= tol(ldbr))
        yb, ldf = O.rqs_fwd(xb.astype(np.float64), xp, yp, sl)
        bwd_ok = (np.abs(yb - yq) <= tol(yq)) & (np.abs(ldb + ldf) <= 10 * tol(ldf))
        assert (fwd_ok | bwd_ok).all(), (K, (~(fwd_ok | bwd_ok)).sum())
        # one parameter set broadcast over a batch of inputs (distrax broadcasts params' batch dims)
        one = distrax.RationalQuadraticSpline(params[0], range_min=lo, range_max=hi)
        yy, ll = one.forward_and_log_det(x)
        yr1, ldr1 = O.rqs_fwd(x, *(np.broadcast_to(a[0], a.shape) for a in (xp, yp, sl)))
        assert ((np.abs(yy - yr1) <= tol(yr1)) & (np.abs(ll - ldr1) <= tol(ldr1))).mean() >= 0.99
        assert (np.abs(yy - yr1) <= 30 * tol(yr1)).all() and (np.abs(ll - ldr1) <= 30 * tol(ldr1)).all()
    # the spline worked by hand in exact rational arithmetic (tests/golden/rqs_hand_case.npz): parameters from its MLP
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rqs_hand_case.npz"))
    for sfx in ("", "_k4"):
        K = int(g["num_bins" + sfx])
        xm = g["x" + sfx].astype(np.float64) * np.array([0.0, 1.0])
        h = np.maximum(xm @ g["W1" + sfx] + g["b1" + sfx], 0)
        h = np.maximum(h @ g["W2" + sfx] + g["b2" + sfx], 0)
        p = (h @ g["W3" + sfx] + g["b3" + sfx]).astype(np.float32)[:, :3 * K + 1]   # dim 0 is the transformed one
        b = distrax.RationalQuadraticSpline(p, range_min=-2.0, range_max=2.0, min_bin_size=float(g["min_bin_size" + sfx]),
                                            min_knot_slope=0.5)
        y, ld = b.forward_and_log_det(g["x" + sfx][:, 0])
        assert (np.abs(y - g["y" + sfx][:, 0]) <= tol(g["y" + sfx][:, 0])).all()
        assert (np.abs(ld - g["logdet" + sfx]) <= tol(g["logdet" + sfx])).all()
        xb, ldb = b.inverse_and_log_det(g["y" + sfx][:, 0].astype(np.float32))
        assert (np.abs(xb - g["x" + sfx][:, 0]) <= tol(g["x" + sfx][:, 0])).all() and (np.abs(ldb + g["logdet" + sfx]) <= 1e-5).all()
    with pytest.raises(ValueError):
        distrax.RationalQuadraticSpline(np.zeros((3, 12), np.float32), range_min=-1.0, range_max=1.0)


def test_elementwise_apply_abi(distrax):
    """nf_elementwise_apply through raw ctypes: device pointers (no staging), either output NULL, n = 0, and the
    argument errors (the reference raises ValueError: NF_ERR_INVALID here) -- unknown kind, 3K+1 mismatch caught by the
    bin / range checks of distrax.RationalQuadraticSpline's constructor, NULL inputs."""
    import ctypes as C
    from normalizing_flows_amd import _lib
    from normalizing_flows_amd.flow import DeviceArray
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, K = 1000, 6
    x = rng.uniform(-4, 4, n).astype(np.float32)
    params = (rng.standard_normal((n, 3 * K + 1)) * 0.8).astype(np.float32)
    desc = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION, kind=_lib.ELEM_RQS, rqs_bins=K, rqs_range_min=-3.0,
                                rqs_range_max=3.0, rqs_min_bin_size=1e-4, rqs_min_knot_slope=1e-4)
    xd, pd = DeviceArray.from_numpy(x), DeviceArray.from_numpy(params)
    yd, ld = DeviceArray((n,)), DeviceArray((n,))
    _lib.check(lib.nf_elementwise_apply(C.byref(desc), 0, xd.ptr, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE))
    yr, lr = O.rqs_fwd(x, *O.rqs_knots(params, -3.0, 3.0))
    y, l = yd.to_numpy(), ld.to_numpy()
    assert (np.abs(y - yr) <= 3 * tol(yr)).all() and (np.abs(l - lr) <= 3 * tol(lr)).all()
    assert np.array_equal(xd.to_numpy(), x)                 # the input is not the kernels' in-place state
    # only one output wanted: the other pointer NULL; same bits
    y2 = DeviceArray((n,))
    _lib.check(lib.nf_elementwise_apply(C.byref(desc), 0, xd.ptr, pd.ptr, n, y2.ptr, None, _lib.MEM_DEVICE))
    assert np.array_equal(y2.to_numpy(), y)
    l2 = np.empty(n, np.float32)
    _lib.check(lib.nf_elementwise_apply(C.byref(desc), 0, x.ctypes.data_as(C.c_void_p), params.ctypes.data_as(C.c_void_p), n,
                                        None, l2.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
    assert np.array_equal(l2, l)
    # inverse on the device round-trips
    xb = DeviceArray((n,))
    _lib.check(lib.nf_elementwise_apply(C.byref(desc), 1, yd.ptr, pd.ptr, n, xb.ptr, None, _lib.MEM_DEVICE))
    inside = np.abs(x) < 2.9
    assert np.abs(xb.to_numpy() - x)[inside].max() <= 2e-3
    # n = 0 succeeds without touching anything; errors are NF_ERR_INVALID with a message
    assert lib.nf_elementwise_apply(C.byref(desc), 0, None, None, 0, None, None, _lib.MEM_HOST) == 0
    bad = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION, kind=7)
    assert lib.nf_elementwise_apply(C.byref(bad), 0, xd.ptr, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE) == -1
    bad = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION, kind=_lib.ELEM_RQS, rqs_bins=0, rqs_range_min=0.0, rqs_range_max=1.0,
                               rqs_min_bin_size=1e-4, rqs_min_knot_slope=1e-4)
    assert lib.nf_elementwise_apply(C.byref(bad), 0, xd.ptr, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE) == -1
    bad = _lib.ElementwiseDesc(abi_version=_lib.NF_ABI_VERSION, kind=_lib.ELEM_RQS, rqs_bins=4, rqs_range_min=1.0, rqs_range_max=1.0,
                               rqs_min_bin_size=1e-4, rqs_min_knot_slope=1e-4)
    assert lib.nf_elementwise_apply(C.byref(bad), 0, xd.ptr, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE) == -1
    assert b"range" in lib.nf_last_error()
    assert lib.nf_elementwise_apply(C.byref(desc), 0, None, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE) == -1
    assert lib.nf_elementwise_apply(None, 0, xd.ptr, pd.ptr, n, yd.ptr, ld.ptr, _lib.MEM_DEVICE) == -1
