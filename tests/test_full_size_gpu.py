"""BASELINE.json configs[2] and configs[4] at their FULL sizes, the weight-file path on the GPU, and two host-side
hazards (stream ordering of device arrays, in-place parameter updates).

At 2^20 x 2-D and 2^16 x 784-D the NumPy oracle cannot walk every row in a test's time, so the whole batch is checked
through size-independent properties (slice-vs-whole bit equality, log_prob = base(z) + logdet, round trip, NLL = mean,
fused-vs-unfused agreement) and the oracle is run on a strided subset of the same batch."""
import os
import tempfile

import numpy as np
import pytest

from oracle import flows_np as O
from test_parity_gpu import (assert_close, assert_spline_close, assert_spline_inverse, assert_spline_log_prob, f32_noise,
                             rqs_flow, tol)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nfa(nf_lib):
    from normalizing_flows_amd import _lib, flow
    _lib.require_device()
    return flow


# ------------------------------------------------------------------ configs[2]: src/maf_toy_densities.py:32-35 shape
def test_rqs_config3_full_size_properties(nfa):
    """8 spline coupling layers, K = 12 on [-8, 8], 2^20 gaussian_blobs rows (the fused kernel: every CU walks four
    passes of 16 tiles, the layer images cycle through the LDS double buffer 32 times)."""
    from normalizing_flows_amd import _lib
    spec = O.make_rqs_spec(8, final_std=0.05)
    f = rqs_flow(nfa, spec)
    assert f.kernel_name == "rqs2d_h64_mfma"
    n = 1 << 20
    x = O.make_inputs("gaussian_blobs", n)
    lp = f.log_prob(x)
    assert np.isfinite(lp).all()
    # (a) a slice evaluated alone gives the same bits (no cross-sample coupling, no dependence on the grid)
    for s, m in ((0, 4096), (4096 * 17 + 3 * 32, 1000), (n - 4099, 4099)):
        assert np.array_equal(f.log_prob(np.ascontiguousarray(x[s:s + m])), lp[s:s + m])
    # (b) log_prob == base(z) + logdet with z, logdet from the chain walk
    z, ld = f.chain_apply(_lib.DIR_CHAIN_FORWARD, x)
    recon = -0.5 * (z.astype(np.float64) ** 2).sum(-1) - np.log(2 * np.pi) + ld
    assert_close(lp, recon, "log_prob = base + logdet")
    # (c) round trip.  The inverse walk is ill-conditioned on isolated rows (where a layer is nearly flat), so the
    # whole batch is held to a quantile and the strided subset below to the forward-or-backward error bound
    xb, ldb = f.chain_apply(_lib.DIR_CHAIN_INVERSE, z)
    rt = np.abs(xb - x).max(axis=-1)
    assert np.quantile(rt, 0.9999) <= 1e-4 and np.isfinite(rt).all(), (np.quantile(rt, 0.9999), rt.max())
    assert np.quantile(np.abs(ld + ldb), 0.9999) <= 1e-4
    # (d) NLL equals the float64 mean of the per-sample values
    assert abs(f.nll(x) + lp.astype(np.float64).mean()) <= 1e-9
    # (e) the oracle on a strided subset of the big batch: log_prob, z, logdet per-sample bound; inverse walk
    idx = np.arange(0, n, 257)
    xs = np.ascontiguousarray(x[idx])
    from test_parity_gpu import SPLINE_ESCAPE_FRACTION
    ref = assert_spline_log_prob(lp[idx], spec, xs, "strided oracle check, log_prob", SPLINE_ESCAPE_FRACTION)
    zr, ldr = O.chain_forward(spec, xs)
    nz, nld = f32_noise(spec, lambda c, v: O.chain_forward(c, v, np.float32), (zr, ldr), xs)
    assert_spline_close(z[idx], zr, nz, "strided oracle check, z", SPLINE_ESCAPE_FRACTION)
    assert_spline_close(ld[idx], ldr, nld, "strided oracle check, logdet", SPLINE_ESCAPE_FRACTION)
    assert_spline_inverse(f, spec, np.ascontiguousarray(z[idx]), "strided oracle check, inverse")
    scaled = np.abs(lp[idx] - ref) / np.maximum(1.0, np.abs(ref))
    print(f"configs[2] at 2^20 rows: strided subset of {idx.size}: worst scaled log_prob error {scaled.max():.2e}, "
          f"rms {np.sqrt((scaled ** 2).mean()):.2e}")


# ------------------------------------------------------------------ configs[4]: src/maf_mnist.py:28-31,34-89 shape
def test_mnist_config5_full_size(nfa, monkeypatch):
    """6 spline layers on 784-D rows, MLP 512x3, K = 8 on [0, 1], Uniform base; 2^16 - 37 rows (not a multiple of the
    128-row GEMM tile: the last row block and the last band of 8 row blocks are ragged), default kernels (whole-K GEMMs,
    spline in the last GEMM's epilogue, 512 row blocks x 77 column blocks in the banded XCD order, ld_reduce over 77
    column blocks)."""
    spec = O.make_rqs_spec(6, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                           base="uniform01", final_std=0.05)
    n = (1 << 16) - 37
    x = O.make_inputs("mnist_like", n)
    outside = [5, 4099, n - 1]
    x[outside[0], 100] = 1.5      # leaves the Uniform(0, 1) support: -inf
    x[outside[1], 783] = -0.25
    x[outside[2], 0] = 1.0001
    f = rqs_flow(nfa, spec)
    assert f.kernel_name == "generic_gemm"
    lp = f.log_prob(x)
    inf_rows = np.zeros(n, bool)
    inf_rows[outside] = True
    assert np.array_equal(np.isneginf(lp), inf_rows) and np.isfinite(lp[~inf_rows]).all()
    # (a) the oracle on 96 strided rows (+ the three rows outside the support)
    idx = np.unique(np.concatenate([np.arange(0, n, n // 93), outside]))
    xs = np.ascontiguousarray(x[idx])
    ref = O.log_prob(spec, xs)
    assert np.array_equal(np.isneginf(ref), inf_rows[idx])
    fin = np.isfinite(ref)
    with np.errstate(all="ignore"):
        noise = f32_noise(spec, lambda c, v: O.log_prob(c, v, np.float32), ref, xs)
    assert_spline_close(lp[idx][fin], ref[fin], noise[fin], "strided oracle check (784-D, L = 6)")
    rel = np.abs(lp[idx][fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))
    print(f"configs[4] at {n} rows: {int(fin.sum())} strided rows, |log_prob| up to {np.abs(ref[fin]).max():.0f}: worst "
          f"relative error {rel.max():.2e}")
    # (b) a slice evaluated alone agrees with the whole (different row-block alignment, same arithmetic per row)
    s = 128 * 301 + 17
    part = f.log_prob(np.ascontiguousarray(x[s:s + 1000]))
    assert np.array_equal(np.isneginf(part), np.isneginf(lp[s:s + 1000]))
    m = np.isfinite(part)
    assert np.abs(part[m].astype(np.float64) - lp[s:s + 1000][m]).max() <= 2e-5 * np.abs(lp[s:s + 1000][m]).max()
    # (c) NLL of the finite rows equals their float64 mean (the -inf rows make the full NLL +inf)
    keep = np.ascontiguousarray(x[~inf_rows])
    assert abs(f.nll(keep) + lp[~inf_rows].astype(np.float64).mean()) <= 1e-5 * abs(lp[~inf_rows].astype(np.float64).mean())
    assert f.nll(x) == np.inf
    # (d) the unfused form (parameter tensor in HBM + transform kernel) agrees on every row
    monkeypatch.setenv("NF_NO_K5", "1")
    g = rqs_flow(nfa, spec)
    plain = g.log_prob(x)
    assert np.array_equal(np.isneginf(plain), inf_rows)
    d = np.abs(plain[~inf_rows].astype(np.float64) - lp[~inf_rows])
    assert (d <= 2 * tol(lp[~inf_rows].astype(np.float64))).all(), d.max()


# ------------------------------------------------------------------ weight files with haiku module names
def _haiku_dict(layers):
    """What `log_prob.init(next(prng_seq), np.zeros((1, *event_shape)))` returns for the reference's flows
    (src/real_nvp_toy_densities.py:175-177): module names in haiku's creation order, 'w' [in, out], 'b' [out]."""
    out = {}
    for k, layer in enumerate(layers):
        sfx = "" if k == 0 else f"_{k}"
        for i, (W, b) in enumerate(layer[:-1]):
            out[f"mlp{sfx}/~/linear_{i}"] = {"w": np.array(W, np.float32), "b": np.array(b, np.float32)}
        out[f"linear{sfx}"] = {"w": np.array(layer[-1][0], np.float32), "b": np.array(layer[-1][1], np.float32)}
    return out


@pytest.mark.parametrize("variant", ["B", "D"])
def test_haiku_named_weight_file_on_the_gpu(nf_lib, variant):
    """A haiku-keyed parameter dict -> save_params -> load_params -> layers_from_haiku_params -> the flow the
    reference's make_flow_model builds (src/real_nvp_toy_densities.py:74-110 / src/maf_mnist.py:52-89) ->
    Transformed.log_prob, against the oracle on the same weights.  D has four Linears per conditioner
    ('mlp_k/~/linear_2' exists)."""
    import normalizing_flows_amd as distrax
    from normalizing_flows_amd import _lib
    _lib.require_device()
    if variant == "B":
        spec = O.make_affine_spec(8, final_std=0.03, seed=77)
        event_shape, hidden, x = (2,), [64, 64], O.make_inputs("energy3", 4096)
        nparams, per_dim = 2, False
    else:
        spec = O.make_rqs_spec(2, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                               base="uniform01", final_std=0.1, seed=78)
        event_shape, hidden, x = (28, 28, 1), [512, 512, 512], O.make_inputs("mnist_like", 64).reshape(64, 28, 28, 1)
        nparams, per_dim = 25, True
    params = _haiku_dict(spec.layers)
    assert ("mlp_1/~/linear_2" in params) == (variant == "D") and "linear_1" in params and "mlp/~/linear_0" in params
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "flow.npz")
        distrax.save_params(path, params)
        loaded = distrax.load_params(path)
    layers_w = distrax.layers_from_haiku_params(loaded)
    assert len(layers_w) == len(spec.layers)

    # make_flow_model of the reference script, conditioners created empty and filled from the file
    mask = (np.arange(0, np.prod(event_shape)) % 2).reshape(event_shape).astype(bool)
    if variant == "B":
        def bijector_fn(p):
            shift, log_scale = np.split(p, 2, axis=-1)
            return distrax.ScalarAffine(shift=shift, log_scale=log_scale)
        base = distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape))
    else:
        def bijector_fn(p):
            return distrax.RationalQuadraticSpline(p, range_min=0.0, range_max=1.0)
        base = distrax.Uniform(low=np.zeros(event_shape), high=np.ones(event_shape))
    couplings = []
    for k in range(len(layers_w)):
        cond = distrax.make_conditioner(event_shape, hidden, nparams, per_dim=per_dim)
        cond.set_params(layers_w[k])
        couplings.append(distrax.MaskedCoupling(mask=mask, bijector=bijector_fn, conditioner=cond))
        mask = np.logical_not(mask)
    model = distrax.Transformed(distrax.Independent(base, reinterpreted_batch_ndims=len(event_shape)),
                                distrax.Inverse(distrax.Chain(couplings)))
    lp = model.log_prob(x)
    xf = x.reshape(x.shape[0], -1)
    ref = O.log_prob(spec, xf)
    assert lp.shape == (x.shape[0],)
    if variant == "B":
        assert_close(lp, ref, "log_prob from a haiku-named weight file (B)")
    else:
        assert_spline_log_prob(lp, spec, xf, "log_prob from a haiku-named weight file (D)")

    # failure cases raise ValueError, as distrax / haiku do on a bad tree
    bad = dict(params)
    del bad["linear_1"]
    with pytest.raises(ValueError):
        distrax.layers_from_haiku_params(bad)
    bad = dict(params)
    bad["conv2_d"] = {"w": np.zeros((1, 1)), "b": np.zeros(1)}
    with pytest.raises(ValueError):
        distrax.layers_from_haiku_params(bad)
    wrong = [list(l) for l in layers_w]
    W, b = wrong[0][1]
    wrong[0][1] = (W[:, :-1], b[:-1])      # hidden width no longer follows
    with pytest.raises(ValueError):
        cond = distrax.make_conditioner(event_shape, hidden, nparams, per_dim=per_dim)
        cond.set_params(wrong[0])
    with pytest.raises(ValueError):        # a file whose module lacks its bias
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "broken.npz")
            np.savez(path, **{"mlp/~/linear_0/w": np.zeros((2, 4), np.float32)})
            distrax.load_params(path)


def test_reference_call_sites_unchanged_through_haiku_like(nf_lib):
    """`log_prob.init(next(prng_seq), np.zeros((1, *event_shape)))` and `log_prob.apply(params, batch)` exactly as
    src/real_nvp_toy_densities.py:113-123,143,175-177 write them (hk = normalizing_flows_amd.haiku_like):
    identity at init (zero final Linear: log_prob == base log-density, SURVEY 8c KAT 1), the oracle on trained-like
    weights, one device handle for a loop of apply calls, in-place parameter updates seen."""
    import normalizing_flows_amd as distrax
    import normalizing_flows_amd.haiku_like as hk
    from normalizing_flows_amd import _lib
    from test_abi_and_host import _reference_style_log_prob
    _lib.require_device()
    log_prob = _reference_style_log_prob(distrax, hk, num_layers=8)
    prng_seq = hk.PRNGSequence(42)
    params = log_prob.init(next(prng_seq), np.zeros((1, 2)))
    batch = O.make_inputs("banana", 4096)
    lp0 = log_prob.apply(params, batch)
    assert_close(lp0, O.a_base_log_prob(batch), "identity at init through init/apply")
    # the oracle's weights in haiku's tree
    spec = O.make_affine_spec(8, final_std=0.03, seed=91)
    params = _haiku_dict(spec.layers)
    loss = -np.mean(log_prob.apply(params, batch))                       # loss_fn, :139-145
    ref = O.log_prob(spec, batch)
    assert_close(log_prob.apply(params, batch), ref, "log_prob.apply(params, batch)")
    assert abs(loss + ref.mean()) <= 1e-5
    handles = {id(v[1]) for v in log_prob.flow_cache.values()}
    for _ in range(3):
        log_prob.apply(params, batch)
    assert {id(v[1]) for v in log_prob.flow_cache.values()} == handles and len(handles) == 1   # no handle per call
    params["linear_3"]["b"] += np.float32(0.125)                         # optimiser-style in-place update
    spec.layers[3][-1] = (spec.layers[3][-1][0], params["linear_3"]["b"])
    assert_close(log_prob.apply(params, batch), O.log_prob(spec, batch), "apply after an in-place update")
    assert {id(v[1]) for v in log_prob.flow_cache.values()} == handles   # same handle, new weights


def test_haiku_like_two_models_of_one_architecture(nf_lib):
    """One transformed function, two flows of the same architecture on different modules of `params`: each gets its
    own device handle and its own weights, in both orders of building and evaluating (a cache slot per
    architecture alone would hand the second model the first one's handle without an upload)."""
    import normalizing_flows_amd as distrax
    import normalizing_flows_amd.haiku_like as hk
    from normalizing_flows_amd import _lib
    _lib.require_device()

    def build(event_shape, n):
        mask = (np.arange(2) % 2).astype(bool)
        layers = []
        for _ in range(n):
            def fn(p):
                shift, log_scale = np.split(p, 2, axis=-1)
                return distrax.ScalarAffine(shift=shift, log_scale=log_scale)
            layers.append(distrax.MaskedCoupling(mask=mask, bijector=fn,
                                                 conditioner=distrax.make_conditioner(event_shape, [64, 64], 2)))
            mask = np.logical_not(mask)
        base = distrax.Independent(distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape)), 1)
        return distrax.Transformed(base, distrax.Inverse(distrax.Chain(layers)))

    @hk.without_apply_rng
    @hk.transform
    def build_both_then_evaluate(data):
        m1, m2 = build(data.shape[-1:], 4), build(data.shape[-1:], 4)
        return np.stack([m1.log_prob(data), m2.log_prob(data)])

    @hk.without_apply_rng
    @hk.transform
    def evaluate_between(data):
        a = build(data.shape[-1:], 4).log_prob(data)
        b = build(data.shape[-1:], 4).log_prob(data)
        return np.stack([a, b])

    s1 = O.make_affine_spec(4, final_std=0.03, seed=17)
    s2 = O.make_affine_spec(4, final_std=0.03, seed=18)
    params = _haiku_dict(s1.layers + s2.layers)
    batch = O.make_inputs("banana", 2048, seed=5)
    r1, r2 = O.log_prob(s1, batch), O.log_prob(s2, batch)
    assert np.abs(r1 - r2).max() > 1e-2
    for fn in (build_both_then_evaluate, evaluate_between):
        for _ in range(2):                                   # second call: both handles come from the cache
            got = fn.apply(params, batch)
            assert_close(got[0], r1, "first model of two")
            assert_close(got[1], r2, "second model of two")
        assert len({id(v[1]) for v in fn.flow_cache.values()}) == 2
    # an update of the SECOND model's parameters re-uploads that handle only
    keys0 = {slot: v[0] for slot, v in evaluate_between.flow_cache.items()}
    params["linear_6"]["b"] += np.float32(0.25)
    s2.layers[2][-1] = (s2.layers[2][-1][0], params["linear_6"]["b"])
    got = evaluate_between.apply(params, batch)
    assert_close(got[0], r1, "first model after the second one's update")
    assert_close(got[1], O.log_prob(s2, batch), "second model after its update")
    changed = [slot for slot, v in evaluate_between.flow_cache.items() if v[0] != keys0[slot]]
    assert len(changed) == 1 and changed[0][2][0][0] == "mlp_4/~/linear_0"


@pytest.mark.parametrize("kind", ["affine", "rqs", "split"])
def test_large_host_buffer_calls_are_pipelined_with_the_same_bits(nfa, monkeypatch, kind):
    """log_prob / sample on large NumPy arrays (>= 2^19 rows) of the fused 2-D kernels run in chunks of 2^18 rows with the
    copies on the side stream beside the kernels; the results are the unchunked call's, bit for bit (a row's arithmetic
    does not depend on the batch; the base draw is one launch either way).  A ragged last chunk included."""
    from normalizing_flows_amd import _lib
    import ctypes as C
    if kind == "affine":
        spec = O.make_affine_spec(8, final_std=0.03)
        cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64))
        layers = spec.layers
    elif kind == "rqs":
        spec = O.make_rqs_spec(4, final_std=0.05)
        cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=4, hidden=(64, 64), rqs_bins=12,
                             rqs_range_min=-8.0, rqs_range_max=8.0)
        layers = spec.layers
    else:
        params, flips = O.make_a_params(8)
        cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,),
                             activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
        layers = [[(W, b) for W, b in net] for net in params]
    f = nfa.CouplingFlow(cfg, layers)
    rows = (1 << 20) + 12345
    x = O.make_inputs("banana", rows, seed=3)
    lib = _lib.load()

    def sample(seed, h=None):
        out = np.empty((rows, 2), np.float32)
        _lib.check(lib.nf_sample((h or f)._h, C.c_uint64(seed), rows, out.ctypes.data_as(C.c_void_p), _lib.MEM_HOST))
        return out
    lp_piped, s_piped = f.log_prob(x), sample(11)
    monkeypatch.setenv("NF_NO_HOST_PIPELINE", "1")   # switches are read when a handle is created
    g = nfa.CouplingFlow(cfg, layers)
    monkeypatch.delenv("NF_NO_HOST_PIPELINE")
    lp_plain, s_plain = g.log_prob(x), sample(11, g)
    assert np.array_equal(lp_piped, lp_plain) and np.isfinite(lp_piped).all()
    assert np.array_equal(s_piped, s_plain)
    assert not np.array_equal(s_piped, sample(12))


# ------------------------------------------------------------------ host-side hazards
def test_device_results_need_no_manual_synchronize(nfa):
    """Handles run on non-blocking streams; DeviceArray.to_numpy (nf_memcpy_d2h) and the handle-less device-mode
    entry points wait for them.  Device in -> device out -> host, with NO synchronize call in between, must give
    the finished result every time (a 2^22-row call keeps the kernel running for ~2 ms after the call returns)."""
    from normalizing_flows_amd import _lib
    from normalizing_flows_amd import distributions
    spec = O.make_affine_spec(8, final_std=0.02)
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64))
    f = nfa.CouplingFlow(cfg, spec.layers)
    n = 1 << 22
    x = O.make_inputs("banana", n, seed=3)
    ref = f.log_prob(x)                       # host in -> host out: synchronised by the call itself
    xd = nfa.DeviceArray.from_numpy(x)
    for _ in range(5):
        out = f.log_prob(xd)                  # enqueued only
        got = out.to_numpy()                  # no f.synchronize()
        assert np.array_equal(got, ref)
        out.free()
    # chain walk on the handle's stream, then the handle-less base density on its output
    zd, ldd = f.chain_apply(_lib.DIR_CHAIN_FORWARD, xd)
    base = distributions.Independent(distributions.Normal(loc=np.zeros(2), scale=np.ones(2)), 1)
    blp = base.log_prob(zd)
    got = (blp.to_numpy() if hasattr(blp, "to_numpy") else np.asarray(blp)).astype(np.float64) + ldd.to_numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_real_nvp_cache_sees_in_place_updates_and_old_params(nf_lib):
    """normalizing_flows_amd/real_nvp.py caches handles by parameter CONTENT: an in-place NumPy update must be
    seen, sgd_update must not leave the old parameters pointing at the stepped handle, and a long loop keeps the
    cache bounded (src/real_nvp.py:295-315 runs 50,001 steps)."""
    from normalizing_flows_amd import _lib
    from normalizing_flows_amd import real_nvp as R
    _lib.require_device()
    params, fns_config = R.init_nvp_chain(n=4)
    rng = np.random.default_rng(3)
    params = [[[(rng.standard_normal(W.shape) * 0.2).astype(np.float32), (rng.standard_normal(b.shape) * 0.1).astype(np.float32)]
               for W, b in net] for net in params]
    flips = [flip for _, flip in fns_config]
    y = O.make_inputs("energy2", 2048)
    before = R.loss(params, fns_config, y)
    assert abs(before - O.a_nll(params, flips, y)) <= 1e-5
    params[1][0][0] += np.float32(0.25)          # in place: same array objects, new values
    after = R.loss(params, fns_config, y)
    assert abs(after - O.a_nll(params, flips, y)) <= 1e-5 and abs(after - before) > 1e-4
    # sgd_update returns new params; the OLD ones still evaluate to the old loss
    old_loss = R.loss(params, fns_config, y)
    nll, new_params = R.sgd_update(params, fns_config, y, learning_rate=0.05)
    assert abs(nll - old_loss) <= 1e-6
    assert abs(R.loss(params, fns_config, y) - old_loss) <= 1e-12
    new_loss = R.loss(new_params, fns_config, y)
    assert abs(new_loss - O.a_nll(new_params, flips, y)) <= 1e-5 and new_loss < old_loss
    p = new_params
    for _ in range(40):
        _, p = R.sgd_update(p, fns_config, y, learning_rate=0.01)
    assert len(R._CACHE) <= R._CACHE_MAX
