"""CPU-side checks: the C ABI library loads and exports what include/nf_hip.h declares,
host-side logic (packing, lowering, sharding), and that the product has no CPU fallback."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "nf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nf_lib):
    from normalizing_flows_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(nf_lib, n), f"{n} is declared in include/nf_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes table and header disagree"
    assert nf_lib.nf_abi_version() == 1


def test_desc_struct_layout_matches_header():
    from normalizing_flows_amd import _lib
    # 6 + 8 int32, then 3 int32, then 4 floats: no padding expected
    assert C.sizeof(_lib.FlowDesc) == 4 * (5 + 8 + 4 + 4)
    assert _lib.FlowDesc.hidden.offset == 20 and _lib.FlowDesc.rqs_range_min.offset == 68


def test_no_device_means_loud_failure(nf_lib):
    """Without a GPU nothing computes: create fails with NF_ERR_NO_DEVICE, never a CPU path."""
    from normalizing_flows_amd import _lib, flow
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=2, hidden=(64, 64))
    with pytest.raises(_lib.NFError) as e:
        flow.CouplingFlow(cfg)
    assert e.value.status == -4
    h = C.c_void_p()
    d = cfg.to_desc()
    assert nf_lib.nf_flow_create(C.byref(d), C.byref(h)) == -4 and not h.value
    import normalizing_flows_amd as distrax
    base = distrax.Independent(distrax.Normal(np.zeros(2), np.ones(2)), 1)
    with pytest.raises(_lib.NFError):
        base.log_prob(np.zeros((3, 2), np.float32))


def test_desc_validation_and_weight_count(nf_lib):
    from normalizing_flows_amd import _lib, flow
    n = C.c_size_t()
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64))
    d = cfg.to_desc()
    assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == 0 and n.value == 8 * 4482
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=6, hidden=(512, 512, 512),
                          rqs_bins=8, base=_lib.BASE_UNIFORM01)
    d = cfg.to_desc()
    assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == 0 and n.value == 6 * 10982032
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,), first_mask_parity=0)
    d = cfg.to_desc()
    assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == 0 and n.value == 8 * 18
    for bad in (dict(event_dim=1), dict(num_layers=0), dict(kind=7), dict(activation=5), dict(base=9),
                dict(first_mask_parity=2)):
        base = dict(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=2, hidden=(64, 64))
        base.update(bad)
        d = flow.FlowConfig(**base).to_desc()
        assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == -1, bad
        assert nf_lib.nf_last_error()
    d = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=3, num_layers=2, hidden=(4,)).to_desc()
    assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == -1
    for bad in (dict(rqs_bins=0), dict(rqs_range_min=1.0, rqs_range_max=0.0), dict(rqs_min_knot_slope=1.5),
                dict(rqs_bins=4, rqs_min_bin_size=0.5)):
        base = dict(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=2, hidden=(64, 64), rqs_bins=4)
        base.update(bad)
        d = flow.FlowConfig(**base).to_desc()
        assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == -1, bad
    d = cfg.to_desc()
    d.abi_version = 99
    assert nf_lib.nf_flow_weight_count(C.byref(d), C.byref(n)) == -1


def test_pack_weights_order_and_checks():
    from normalizing_flows_amd import _lib, flow
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=2, hidden=(3,))
    layers = []
    v = 0
    for _ in range(2):
        W0 = np.arange(v, v + 6, dtype=np.float32).reshape(2, 3); v += 6
        b0 = np.arange(v, v + 3, dtype=np.float32); v += 3
        W1 = np.arange(v, v + 6, dtype=np.float32).reshape(3, 2); v += 6
        b1 = np.arange(v, v + 2, dtype=np.float32); v += 2
        layers.append([(W0, b0), (W1, b1)])
    packed = flow.pack_weights(cfg, layers)
    assert np.array_equal(packed, np.arange(34, dtype=np.float32))
    with pytest.raises(ValueError):
        flow.pack_weights(cfg, layers[:1])
    with pytest.raises(ValueError):
        flow.pack_weights(cfg, [[(np.zeros((2, 4)), np.zeros(4)), (np.zeros((4, 2)), np.zeros(2))]] * 2)


def test_conditioner_and_haiku_param_mapping():
    import normalizing_flows_amd as distrax
    c = distrax.make_conditioner((2,), [64, 64], 2, rng=np.random.default_rng(0))
    assert c.hidden_sizes == (64, 64) and c.in_size == 2 and c.out_size == 2
    assert np.all(c.linears[-1][0] == 0) and np.all(c.linears[-1][1] == 0)       # zero-init final Linear
    assert abs(c.linears[1][0].std() - 1 / 8 * 0.88) < 0.02 and np.abs(c.linears[1][0]).max() <= 2 / 8 + 1e-6
    c = distrax.make_conditioner((28, 28, 1), [16], 25, per_dim=True)
    assert c.out_size == 19600 and c.params_per_dim == 25
    with pytest.raises(NotImplementedError):
        c(np.zeros((1, 784)))
    hk = {}
    for k in range(3):
        pre = "mlp" if k == 0 else f"mlp_{k}"
        lin = "linear" if k == 0 else f"linear_{k}"
        hk[f"{pre}/~/linear_0"] = {"w": np.full((2, 4), k), "b": np.zeros(4)}
        hk[f"{pre}/~/linear_1"] = {"w": np.full((4, 4), k + 10), "b": np.zeros(4)}
        hk[lin] = {"w": np.full((4, 2), k + 20), "b": np.zeros(2)}
    layers = distrax.layers_from_haiku_params(hk)
    assert len(layers) == 3 and [len(l) for l in layers] == [3, 3, 3]
    assert layers[2][0][0][0, 0] == 2 and layers[2][1][0][0, 0] == 12 and layers[1][2][0][0, 0] == 21
    with pytest.raises(ValueError):
        distrax.layers_from_haiku_params({"conv": {"w": 0, "b": 0}})
    # weight file round trip (npz with haiku module names)
    import tempfile
    back = distrax.layers_to_haiku_params(layers)
    assert sorted(back) == sorted(hk)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "flow.npz")
        distrax.save_params(path, back)
        again = distrax.layers_from_haiku_params(distrax.load_params(path))
    for la, lb in zip(layers, again):
        for (Wa, ba), (Wb, bb) in zip(la, lb):
            assert np.array_equal(Wa, Wb) and np.array_equal(ba, bb)


def _reference_style_log_prob(distrax, hk, num_layers=3, hidden=(64, 64), per_dim_bins=None):
    """src/real_nvp_toy_densities.py:74-123 with the imports swapped (spline variant: src/maf_toy_densities.py)."""
    def make_flow_model(event_shape, num_layers, hidden_sizes):
        mask = np.arange(0, np.prod(event_shape)) % 2
        mask = np.reshape(mask, event_shape).astype(bool)
        if per_dim_bins is None:
            def bijector_fn(params):
                shift, log_scale = np.split(params, 2, axis=-1)
                return distrax.ScalarAffine(shift=shift, log_scale=log_scale)
            nparams, per_dim = 2, False
        else:
            def bijector_fn(params):
                return distrax.RationalQuadraticSpline(params, range_min=-8.0, range_max=8.0)
            nparams, per_dim = 3 * per_dim_bins + 1, True
        layers = []
        for _ in range(num_layers):
            layer = distrax.MaskedCoupling(
                mask=mask, bijector=bijector_fn,
                conditioner=distrax.make_conditioner(event_shape, hidden_sizes, nparams, per_dim=per_dim))
            layers.append(layer)
            mask = np.logical_not(mask)
        flow = distrax.Inverse(distrax.Chain(layers))
        base_distribution = distrax.Independent(
            distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape)),
            reinterpreted_batch_ndims=len(event_shape))
        return distrax.Transformed(base_distribution, flow)

    @hk.without_apply_rng
    @hk.transform
    def log_prob(data):
        model = make_flow_model(event_shape=data.shape[-1:], num_layers=num_layers, hidden_sizes=list(hidden))
        return model.log_prob(data)

    return log_prob


def test_haiku_like_transform_init_names_and_errors():
    """hk.transform / without_apply_rng / PRNGSequence around the hot path (src/real_nvp_toy_densities.py:113-123,
    175-177): init returns haiku's names in creation order with haiku's initialisers; apply checks the tree."""
    import normalizing_flows_amd as distrax
    import normalizing_flows_amd.haiku_like as hk
    log_prob = _reference_style_log_prob(distrax, hk, num_layers=3)
    prng_seq = hk.PRNGSequence(42)
    params = log_prob.init(next(prng_seq), np.zeros((1, 2)))
    assert list(params) == ["mlp/~/linear_0", "mlp/~/linear_1", "linear", "mlp_1/~/linear_0", "mlp_1/~/linear_1",
                            "linear_1", "mlp_2/~/linear_0", "mlp_2/~/linear_1", "linear_2"]
    assert params["mlp/~/linear_0"]["w"].shape == (2, 64) and params["mlp_2/~/linear_1"]["w"].shape == (64, 64)
    assert params["linear_1"]["w"].shape == (64, 2) and params["linear_1"]["b"].shape == (2,)
    assert np.all(params["linear"]["w"] == 0) and np.all(params["mlp/~/linear_0"]["b"] == 0)   # hk defaults + zeros
    w = params["mlp_1/~/linear_1"]["w"]
    assert np.abs(w).max() <= 2 / 8 + 1e-6 and abs(w.std() - 0.88 / 8) < 0.02                  # TruncatedNormal(1/sqrt(64))
    assert not np.array_equal(params["mlp/~/linear_0"]["w"], params["mlp_1/~/linear_0"]["w"])
    again = log_prob.init(next(prng_seq), np.zeros((1, 2)))
    assert not np.array_equal(again["mlp/~/linear_0"]["w"], params["mlp/~/linear_0"]["w"])     # another key
    assert distrax.layers_from_haiku_params(params)[2][1][0] is not None                         # the loader's names
    # the spline script's tree: 74 = 2 * 37 outputs per conditioner
    lp_rqs = _reference_style_log_prob(distrax, hk, num_layers=2, per_dim_bins=12)
    p2 = lp_rqs.init(7, np.zeros((1, 2)))
    assert p2["linear_1"]["w"].shape == (64, 74)
    # apply: missing module, wrong shape, wrong layer count -> ValueError before anything touches the GPU
    bad = {k: v for k, v in params.items() if k != "linear_1"}
    with pytest.raises(ValueError):
        log_prob.apply(bad, np.zeros((4, 2), np.float32))
    bad = dict(params)
    bad["mlp_1/~/linear_0"] = {"w": np.zeros((2, 32), np.float32), "b": np.zeros(32, np.float32)}
    with pytest.raises(ValueError):
        log_prob.apply(bad, np.zeros((4, 2), np.float32))
    more = dict(params)
    more["linear_3"] = params["linear_2"]
    with pytest.raises(ValueError):
        log_prob.apply(more, np.zeros((4, 2), np.float32))
    with pytest.raises(TypeError):
        hk.transform(lambda d: d).apply(params)       # rng argument missing on the un-wrapped form
    # the cache key sees an in-place update and a re-assignment, not a no-op
    k0 = hk.params_key(params)
    assert hk.params_key(params) == k0
    params["linear"]["b"][0] = 0.5
    assert hk.params_key(params) != k0
    # ... of ONE entry of a large array too (every byte is hashed; a strided sample would miss index 12345)
    big = {"linear": {"w": np.zeros((512, 19600), np.float32), "b": np.zeros(19600, np.float32)}}
    k0 = hk.params_key(big)
    big["linear"]["w"].reshape(-1)[12345] = 1e-3
    assert hk.params_key(big) != k0


def test_haiku_like_hashes_immutable_arrays_once(monkeypatch):
    """The content key of a parameter array is a digest of every byte, taken on every apply -- except for arrays that
    nothing can write to (read-only all the way down the view chain: np.asarray of the reference's jax parameters):
    those are hashed once per array object.  A read-only VIEW of a writable array is not such an array."""
    import normalizing_flows_amd.haiku_like as hk
    calls = []
    real = hk._digest
    monkeypatch.setattr(hk, "_digest", lambda buf: (calls.append(1), real(buf))[1])
    frozen = np.arange(1000, dtype=np.float32)
    frozen.flags.writeable = False
    k0 = hk._array_key(frozen)
    assert hk._array_key(frozen) == k0 and hk._array_key(frozen) == k0 and len(calls) == 1
    other = np.arange(1000, dtype=np.float32) + 1
    other.flags.writeable = False
    assert hk._array_key(other) != k0 and len(calls) == 2          # another object: its own digest
    live = np.arange(1000, dtype=np.float32)
    k1 = hk._array_key(live)
    live[3] = -1.0
    assert hk._array_key(live) != k1 and len(calls) == 4           # writable: hashed every time, the edit is seen
    view = live[:500]
    view.flags.writeable = False                                   # read-only view, writable base
    k2 = hk._array_key(view)
    live[7] = 9.0
    assert hk._array_key(view) != k2 and len(calls) == 6
    n = len(hk._FROZEN_DIGESTS)
    del frozen, other
    import gc
    gc.collect()
    assert len(hk._FROZEN_DIGESTS) <= n - 2                        # entries leave with their arrays


def test_haiku_like_function_building_two_models():
    """A transformed function may build two flows of ONE architecture (and evaluate the first before it builds the
    second).  The tree is validated by a dry run that evaluates nothing, so a bad tree raises ValueError without a
    GPU, the layer count is checked when the function returns and not in the middle, and each model's conditioners
    know which modules they read (the device-handle cache is keyed on them: tests/test_full_size_gpu.py)."""
    import normalizing_flows_amd as distrax
    import normalizing_flows_amd.haiku_like as hk
    seen = []

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
        seen.append([l.conditioner.hk_names for l in layers])
        base = distrax.Independent(distrax.Normal(loc=np.zeros(event_shape), scale=np.ones(event_shape)), 1)
        return distrax.Transformed(base, distrax.Inverse(distrax.Chain(layers)))

    @hk.without_apply_rng
    @hk.transform
    def two(data):
        a = build(data.shape[-1:], 2).log_prob(data)      # evaluated before the second model exists
        b = build(data.shape[-1:], 2).log_prob(data)
        return a - b

    params = two.init(3, np.zeros((1, 2)))
    assert list(params)[-1] == "linear_3" and len(params) == 12
    assert seen[0] == [("mlp/~/linear_0", "mlp/~/linear_1", "linear"), ("mlp_1/~/linear_0", "mlp_1/~/linear_1", "linear_1")]
    assert seen[1] == [("mlp_2/~/linear_0", "mlp_2/~/linear_1", "linear_2"), ("mlp_3/~/linear_0", "mlp_3/~/linear_1", "linear_3")]
    x = np.zeros((4, 2), np.float32)
    for drop in ("linear_3", "mlp_2/~/linear_1"):
        with pytest.raises(ValueError):
            two.apply({k: v for k, v in params.items() if k != drop}, x)
    more = dict(params)
    more["mlp_4/~/linear_0"], more["mlp_4/~/linear_1"], more["linear_4"] = (params[k] for k in list(params)[:3])
    with pytest.raises(ValueError, match="5 coupling layers"):
        two.apply(more, x)
    assert two._validated is None            # nothing was accepted, nothing touched the GPU


def test_layer_signature_cache_distinguishes_closures():
    """MaskedCoupling caches what its probe of `bijector_fn` finds per (code object, closure contents, defaults, mask,
    conditioner shape): haiku_like's apply rebuilds the model on every call and must not pay 16 probe calls each time.
    The same source with other captured values is another key; mutable captured content is not cached at all."""
    import normalizing_flows_amd as distrax
    from normalizing_flows_amd import bijectors
    mask = (np.arange(2) % 2).astype(bool)

    def make(hi):
        def rqs_fn(p):
            return distrax.RationalQuadraticSpline(p, range_min=-hi, range_max=hi)
        return distrax.MaskedCoupling(mask, rqs_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))

    bijectors._SIGNATURES.clear()
    a, b, c = make(8.0), make(4.0), make(8.0)
    assert a.signature().rqs[1:3] == (-8.0, 8.0) and b.signature().rqs[1:3] == (-4.0, 4.0)
    assert len(bijectors._SIGNATURES) == 2
    assert c.signature() is a.signature()                      # same code, same captured value: one probe
    # other shapes under the same function: another key
    d = distrax.MaskedCoupling(mask, a._bijector_fn, distrax.make_conditioner((2,), [64, 64], 25, per_dim=True))
    assert d.signature().rqs[0] == 8 and len(bijectors._SIGNATURES) == 3
    # a captured array is mutable: no caching, the probe runs
    box = np.array([8.0])

    def arr_fn(p):
        return distrax.RationalQuadraticSpline(p, range_min=-float(box[0]), range_max=float(box[0]))
    e = distrax.MaskedCoupling(mask, arr_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    assert e._signature_cache_key() is None and e.signature().rqs[2] == 8.0
    box[0] = 6.0
    f2 = distrax.MaskedCoupling(mask, arr_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    assert f2.signature().rqs[2] == 6.0 and len(bijectors._SIGNATURES) == 3
    # affine: the column order the probe found travels with the cached signature
    def swapped(p):
        log_scale, shift = np.split(p, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, log_scale=log_scale)
    g1 = distrax.MaskedCoupling(mask, swapped, distrax.make_conditioner((2,), [64, 64], 2))
    g2 = distrax.MaskedCoupling(mask, swapped, distrax.make_conditioner((2,), [64, 64], 2))
    g1.signature()
    g2.signature()
    assert g1._perm == (1, 0) and g2._perm == (1, 0)
    # NumPy scalars are keyed by VALUE (a fresh np.float32 per rebuild was keyed by its address, which
    # CPython recycles -- a changed value could land on the old id and silently reuse the old spline range)
    def make_np(hi):
        lo, up = np.float32(-hi), np.float32(hi)

        def rqs_fn(p):
            return distrax.RationalQuadraticSpline(p, range_min=float(lo), range_max=float(up))
        return distrax.MaskedCoupling(mask, rqs_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    n0 = len(bijectors._SIGNATURES)
    for hi in (8.0, 5.0, 8.0, 3.0, 5.0) * 20:      # rebuilt over and over, as haiku_like's apply does
        assert make_np(hi).signature().rqs[1:3] == (-hi, hi)
    assert len(bijectors._SIGNATURES) == n0 + 3
    # identity-keyed objects are held by the cache entry (their ids cannot be recycled while it lives) ...
    assert all(len(v) == 3 and any(o is distrax for o in v[2]) for k, v in bijectors._SIGNATURES.items())
    # ... a module global the function reads is part of the key ...
    g = {"distrax": distrax, "HI": 8.0}
    exec("def glob_fn(p):\n    return distrax.RationalQuadraticSpline(p, range_min=-HI, range_max=HI)\n", g)
    mk = lambda: distrax.MaskedCoupling(mask, g["glob_fn"], distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))  # noqa: E731
    assert mk().signature().rqs[2] == 8.0
    g["HI"] = 2.0
    assert mk().signature().rqs[2] == 2.0
    # ... and any other captured object makes the layer uncacheable instead of being keyed by its address
    class Cfg:
        hi = 8.0
    cfg = Cfg()

    def cfg_fn(p):
        return distrax.RationalQuadraticSpline(p, range_min=-cfg.hi, range_max=cfg.hi)
    h1 = distrax.MaskedCoupling(mask, cfg_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    assert h1._signature_cache_key() is None and h1.signature().rqs[2] == 8.0
    cfg.hi = 7.0
    h2 = distrax.MaskedCoupling(mask, cfg_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    assert h2.signature().rqs[2] == 7.0


def test_lowering_recognises_reference_structures_without_a_gpu():
    import normalizing_flows_amd as distrax
    from normalizing_flows_amd import _lib
    from normalizing_flows_amd.bijectors import lower_for_log_prob
    mask = (np.arange(2) % 2).astype(bool)

    def affine_fn(p):
        shift, log_scale = np.split(p, 2, axis=-1)
        return distrax.ScalarAffine(shift=shift, log_scale=log_scale)

    layer = distrax.MaskedCoupling(mask, affine_fn, distrax.make_conditioner((2,), [64, 64], 2))
    sig = layer.signature()
    assert sig.kind == _lib.KIND_AFFINE_MASKED and sig.parity == 1 and sig.hidden == (64, 64)
    layer0 = distrax.MaskedCoupling(~mask, affine_fn, distrax.make_conditioner((2,), [64, 64], 2))
    assert layer0.signature().parity == 0

    def rqs_fn(p):
        return distrax.RationalQuadraticSpline(p, range_min=-8.0, range_max=8.0)

    rl = distrax.MaskedCoupling(mask, rqs_fn, distrax.make_conditioner((2,), [64, 64], 37, per_dim=True))
    s = rl.signature()
    assert s.kind == _lib.KIND_RQS_MASKED and s.rqs[:3] == (12, -8.0, 8.0)
    chain = distrax.Chain([layer, layer0])
    inv = distrax.Inverse(chain)
    assert lower_for_log_prob(inv) == (chain, _lib.DIR_CHAIN_FORWARD)
    assert lower_for_log_prob(chain) == (chain, _lib.DIR_CHAIN_INVERSE)
    assert lower_for_log_prob(distrax.Inverse(inv))[1] == _lib.DIR_CHAIN_INVERSE
    with pytest.raises(NotImplementedError):                      # masks must alternate
        distrax.Chain([layer, layer]).lowered()
    with pytest.raises(NotImplementedError):
        distrax.MaskedCoupling(np.array([True, True]), affine_fn,
                               distrax.make_conditioner((2,), [64, 64], 2)).signature()

    class Rotation(distrax.Bijector):        # the reference's README example subclasses Bijector
        def __init__(self):
            super().__init__(event_ndims_in=1, event_ndims_out=1)

        def forward_and_log_det(self, x):
            return x, 0.0

    r = Rotation()
    assert r.event_ndims_in == 1 and r.forward(3.0) == 3.0
    with pytest.raises(NotImplementedError):
        r.inverse(1.0)
    with pytest.raises(NotImplementedError):
        lower_for_log_prob(r)
    with pytest.raises(ValueError):
        distrax.Chain([])
    base = distrax.Independent(distrax.Normal(np.zeros(2), np.ones(2)), reinterpreted_batch_ndims=1)
    assert base.event_shape == (2,) and base._base_code() == _lib.BASE_STD_NORMAL
    assert distrax.Independent(distrax.Uniform(np.zeros((28, 28, 1)), np.ones((28, 28, 1))), 3)._base_code() == 1
    assert distrax.Independent(distrax.Normal(np.zeros(2), 2 * np.ones(2)), 1)._base_code() is None
    assert distrax.MultivariateNormalDiag(np.zeros(2), np.ones(2))._base_code() == 0
    from normalizing_flows_amd.distributions import _seed_u64
    assert _seed_u64(7) == 7 and _seed_u64(np.array([1, 2], np.uint32)) == (0x9E3779B97F4A7C15 + 2) % 2 ** 64
    with pytest.raises(ValueError):
        distrax.Transformed(distrax.Normal(0.0, 1.0), chain)


@pytest.mark.parametrize("mode", ["ok", "fail", "hang"])
def test_bench_spawns_its_own_ranks(tmp_path, mode):
    """`python bench.py --gpus N` as typed (WORLD_SIZE unset): the parent starts N rank processes with the launcher's
    environment, relays rank 0's line when all succeed, and ends everybody + exits non-zero when one rank fails or the
    limit passes (the ranks here are tests/_bench_stub_worker.py -- no GPU, no library)."""
    import json
    import time
    root, HERE = ROOT, os.path.join(ROOT, "tests")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NF_BENCH_WORKER=os.path.join(HERE, "_bench_stub_worker.py"), NF_STUB_DIR=str(tmp_path), NF_STUB_MODE=mode,
               NF_BENCH_SPAWN_TIMEOUT="2" if mode == "hang" else "60")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "7", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(3)]
    assert [rec["RANK"] for rec in recs] == ["0", "1", "2"] and [rec["LOCAL_RANK"] for rec in recs] == ["0", "1", "2"]
    assert all(rec["WORLD_SIZE"] == "3" and rec["MASTER_ADDR"] == "127.0.0.1" for rec in recs)
    assert len({rec["MASTER_PORT"] for rec in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert all(rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for rec in recs)
    assert all(rec["argv"] == ["--gpus", "3", "--steps", "7", "--warmup", "2"] for rec in recs)
    assert len({rec["sid"] for rec in recs}) == 3 and all(rec["sid"] == rec["pid"] for rec in recs)  # own groups
    if mode == "ok":
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip().splitlines() == [json.dumps({"metric": "stub", "n_gpus": 3})]
        assert "rank 1 chatter" in r.stderr
    else:
        assert r.returncode == (3 if mode == "fail" else 124), (r.returncode, r.stderr)
        assert r.stdout == ""                       # no result line, not even rank 0's partial output
        assert took < 30                            # nobody waited for the sleepers
        assert ("rank 1 of 3 exited with 3" in r.stderr) if mode == "fail" else ("still running" in r.stderr)
    for rec in recs:                                # every rank process is gone either way
        with pytest.raises(ProcessLookupError):
            os.kill(rec["pid"], 0)


def test_shard_rows():
    from normalizing_flows_amd import shard_rows
    for rows, world in [(1 << 23, 8), (4096, 2), (4097, 2), (7, 4), (0, 3), (1, 2), (1000003, 8)]:
        spans = [shard_rows(rows, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == rows
        for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
            assert b == c and a % 2 == 0 and c % 2 == 0
        sizes = [b - a for a, b in spans]
        assert max(sizes[:-1] or [0]) - min(sizes[:-1] or [0]) <= 2
    assert [shard_rows(1 << 23, 8, r)[1] - shard_rows(1 << 23, 8, r)[0] for r in range(8)] == [1 << 20] * 8
    with pytest.raises(ValueError):
        shard_rows(10, 2, 2)


def test_product_never_touches_the_oracle_or_a_cpu_path():
    pkg = os.path.join(ROOT, "normalizing_flows_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                # comments may cite the oracle; code may not import, include, open or load it
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f"{f} imports the oracle"
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", text), f"{f} includes the oracle"
                assert "nf_oracle" not in text and "libnf_oracle" not in text and "flows_np import" not in text, f
                assert "import torch" not in text and "import jax" not in text and "import numpy.linalg" not in text, f


def test_sharded_nll_gloo_world2():
    """N > 1 host path on CPU: two gloo ranks, each sums ITS rows (here via the oracle, which
    stands in for the GPU's local sum), one all-reduce of one float64, same NLL as one rank."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    vals = [float(o.strip().splitlines()[-1].split()[-1]) for o in outs]
    assert vals[0] == vals[1]
    from oracle import flows_np as O
    spec = O.make_affine_spec(4, final_std=0.02)
    x = O.make_inputs("banana", 4097)
    assert abs(vals[0] - O.nll(spec, x)) < 1e-12


@pytest.mark.parametrize("mode,port", [("onesided", "29641"), ("counts", "29642"), ("preflight", "29643")])
def test_multirank_failures_raise_on_every_rank(mode, port):
    """What keeps a one-sided failure from hanging the peers inside ncclCommInitRank / the next all-reduce
    (the mean the collective implements: /root/reference src/real_nvp_toy_densities.py:143,150):
    every step that can fail on one rank is agreed on over the CPU group first, so ALL ranks raise.
    `preflight`: no GPU in this container, so nf_comm_preflight fails on both ranks and
    RcclComm.from_torch_distributed must stop before any rank calls ncclCommInitRank (skipped on a GPU box)."""
    from normalizing_flows_amd import _lib
    if mode == "preflight" and _lib.device_count() > 0:
        pytest.skip("a GPU is visible: the preflight succeeds here")
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", mode], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for r, o in enumerate(outs):
        assert f"rank {r} RanksDisagree" in o, o
    if mode == "onesided":
        assert all("rank(s) 1 (librccl missing (simulated))" in o for o in outs), outs
    if mode == "counts":
        assert all("rank 0: 7, rank 1: 8" in o for o in outs), outs


def test_prepare_data_matches_the_reference_arithmetic():
    """src/maf_mnist.py:101-106: float32(image) / 256, noise in [0, 1) only with a key."""
    from normalizing_flows_amd import prepare_data
    img = np.arange(2 * 28 * 28, dtype=np.uint32).reshape(2, 28, 28, 1).astype(np.uint8)
    out = prepare_data({"image": img})
    assert out.dtype == np.float32 and out.shape == img.shape
    assert np.array_equal(out, img.astype(np.float32) / 256.0)
    assert out.min() >= 0.0 and out.max() < 1.0
    noisy = prepare_data({"image": img}, prng_key=42)
    d = noisy * 256.0 - img
    assert (d >= 0).all() and (d < 1.0 + 1e-6).all() and d.std() > 0.2
    assert np.array_equal(noisy, prepare_data(img, prng_key=42))          # reproducible, accepts the array itself
    assert not np.array_equal(noisy, prepare_data(img, prng_key=43))



def test_oracle_is_used_only_where_it_may_be():
    """oracle/ is test infrastructure: outside tests/ only __graft_entry__.smoke() and bench.py's
    cpu_baseline() / parity_check() (the checker of the measured kernel's output) may import it -- not the package, not the examples, not the measurement scripts."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        found = []
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                found.append(node.lineno)
        return tree, found

    for pattern in ("normalizing_flows_amd/*.py", "examples/*.py", "profiles/*.py", "profiles/scratch/*.py"):
        for path in glob.glob(os.path.join(root, pattern)):
            assert oracle_imports(path)[1] == [], path
    for path, allowed in ((os.path.join(root, "bench.py"), ("cpu_baseline", "parity_check")),
                          (os.path.join(root, "__graft_entry__.py"), ("smoke",))):
        tree, lines = oracle_imports(path)
        fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in allowed]
        assert lines and all(any(fn.lineno <= ln <= fn.end_lineno for fn in fns) for ln in lines), (path, lines)
