"""GPU tests of the drop-in model classes, mirroring the reference's own model tests
(tests/models/single/test_eof.py, tests/models/cross/test_mca.py) on its mock fixtures
(tests/conftest.py:225-278), plus value parity against the oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def mock_values():
    rng = np.random.default_rng(7)
    noise = rng.normal(5, 3, size=(25, 5, 4))
    signal = 2 * np.sin(np.linspace(0, 2 * np.pi, 25))[:, None, None]
    return signal + noise


def mock_data_array(values=None):
    import xeofs_amd as xe

    v = mock_values() if values is None else values
    return xe.DataArray(v, dims=("time", "lat", "lon"),
                        coords={"time": np.arange(2001, 2026), "lat": [20.0, 30.0, 40.0, 50.0, 60.0],
                                "lon": [-10.0, 0.0, 10.0, 20.0]}, name="t2m")


@pytest.fixture(autouse=True)
def _ctx(ctx):
    return ctx


@pytest.mark.parametrize("dim", [("time",), ("lat", "lon"), ("lon", "lat")])
def test_eof_fit_dims(dim):  # test_eof.py:19-31, 126-181
    import xeofs_amd as xe

    X = mock_data_array()
    m = xe.single.EOF(n_modes=3, random_state=1).fit(X, dim)
    comps, scores = m.components(), m.scores()
    fdims = tuple(d for d in X.dims if d not in dim)
    assert comps.dims == ("mode",) + fdims and scores.dims == ("mode",) + tuple(dim)
    assert comps.shape == (3,) + tuple(X.sizes[d] for d in fdims)
    assert not np.isnan(comps.values).any() and not np.isnan(scores.values).any()
    ev, evr = m.explained_variance(), m.explained_variance_ratio()
    assert (ev.values > 0).all()                                   # test_eof.py:65-74
    assert evr.values.sum() <= 1 + 1e-5                            # test_eof.py:85-100
    assert list(m.singular_values().coords["mode"]) == [1, 2, 3]
    assert comps.attrs["model"] == "EOF analysis" and comps.attrs["center"] == "True"


def test_eof_values_vs_oracle():
    import xeofs_amd as xe

    X = mock_data_array()
    m = xe.single.EOF(n_modes=4, use_coslat=True, random_state=3, solver="randomized").fit(X, "time")
    w = np.repeat(orc.sqrt_cos_lat_weights(X.coords["lat"]), 4)
    ref = orc.eof_fit(mock_values().reshape(25, 20), 4, feature_weights=w, random_state=3, solver="randomized")
    s = m.singular_values().values
    assert np.all(np.abs(s - ref["norms"]) <= 1e-5 * ref["norms"][0])
    c = m.components().values.reshape(4, 20)
    for j in range(4):
        assert np.dot(c[j], ref["components"][:, j]) >= 1 - 1e-5
    assert np.allclose(m.explained_variance_ratio().values, ref["explained_variance_ratio"], rtol=2e-5)
    sc = m.scores().values
    assert np.allclose(sc, ref["scores"].T, rtol=1e-3, atol=2e-4 * np.abs(ref["scores"]).max())
    # normalized accessors (base_model_single_set.py:316-336)
    assert np.allclose(m.scores(normalized=True).values, sc / s[:, None], rtol=1e-6)
    assert np.allclose(m.components(normalized=False).values.reshape(4, 20), c * s[:, None], rtol=1e-6)


def test_eof_isolated_nan_raises():  # test_eof.py:111-115
    import xeofs_amd as xe

    v = mock_values()
    v[0, 1, 0] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        xe.single.EOF().fit(mock_data_array(v), "time")


@pytest.mark.parametrize("kind", ["full_dimensional", "boundary"])
def test_eof_nan_fixtures(kind):  # conftest.py:265-278, test_eof.py:148-181, 267-298
    import xeofs_amd as xe

    v = mock_values()
    v[:, 1, :] = np.nan
    v[1 if kind == "full_dimensional" else 0] = np.nan
    m = xe.single.EOF(n_modes=3, random_state=0).fit(mock_data_array(v), "time")
    c, s = m.components().values, m.scores().values
    assert c.shape == (3, 5, 4) and s.shape == (3, 25)
    assert np.isnan(c[:, 1, :]).all() and not np.isnan(np.delete(c, 1, axis=1)).any()
    row = 1 if kind == "full_dimensional" else 0
    assert np.isnan(s[:, row]).all() and not np.isnan(np.delete(s, row, axis=1)).any()


def test_eof_transform_equals_scores_and_inverse_roundtrip():  # test_eof.py:364-391, 455-488
    import xeofs_amd as xe

    X = mock_data_array()
    m = xe.single.EOF(n_modes=20, solver="full", standardize=True).fit(X, "time")
    sc = m.scores()
    tr = m.transform(X)
    assert tr.dims == sc.dims
    assert np.allclose(tr.values, sc.values, rtol=1e-3, atol=1e-3 * np.abs(sc.values).max())
    rec = m.inverse_transform(sc)
    assert rec.dims == X.dims
    assert np.allclose(rec.values, X.values, rtol=1e-4, atol=2e-4)
    # unseen data -> no NaN (test_eof.py:393-408)
    new = mock_data_array(mock_values()[:7] + 1.0)
    new.coords["time"] = np.arange(7)
    out = m.transform(new)
    assert out.shape == (20, 7) and not np.isnan(out.values).any()
    # NaN feature in new data -> error (test_eof.py:419-441)
    bad = mock_values()
    bad[:, 2, 1] = np.nan
    with pytest.raises(ValueError, match="different locations"):
        m.transform(mock_data_array(bad))


def test_eof_list_input_and_weights():
    import xeofs_amd as xe

    X = mock_data_array()
    X2 = mock_data_array(mock_values() ** 2)
    wts = xe.DataArray(np.linspace(0.5, 1.5, 5), dims=("lat",), coords={"lat": X.coords["lat"]})
    m = xe.single.EOF(n_modes=3, random_state=2).fit([X, X2], "time", weights=wts)
    comps = m.components()
    assert isinstance(comps, list) and len(comps) == 2 and comps[0].dims == ("mode", "lat", "lon")
    M = np.concatenate([mock_values().reshape(25, 20), (mock_values() ** 2).reshape(25, 20)], axis=1)
    w = np.tile(np.repeat(np.linspace(0.5, 1.5, 5), 4), 2)
    ref = orc.eof_fit(M, 3, feature_weights=w, random_state=2, solver="randomized")
    assert np.allclose(m.singular_values().values, ref["norms"], rtol=1e-5)


def test_mca_against_oracle_and_invariants():  # test_mca.py:20-119, test_cpcca.py:152-164
    import xeofs_amd as xe

    rng = np.random.default_rng(5)
    T = rng.standard_normal((60, 4)) * (3.0 * 0.6 ** np.arange(4))
    A = (T @ rng.standard_normal((4, 30)) + 0.3 * rng.standard_normal((60, 30))).reshape(60, 5, 6)
    B = (T @ rng.standard_normal((4, 28)) + 0.3 * rng.standard_normal((60, 28))).reshape(60, 4, 7)
    X = xe.DataArray(A, dims=("time", "lat", "lon"))
    Y = xe.DataArray(B, dims=("time", "y", "x"))
    m = xe.cross.MCA(n_modes=3, random_state=7, solver="randomized", use_pca=False).fit(X, Y, "time")
    ref = orc.mca_fit(A.reshape(60, 30), B.reshape(60, 28), 3, random_state=7, solver="randomized")
    s = m.singular_values().values
    assert np.allclose(s, ref["singular_values"], rtol=2e-5)
    c1, c2 = m.components()
    assert c1.dims == ("mode", "lat", "lon") and c2.dims == ("mode", "y", "x")
    for j in range(3):
        assert np.dot(c1.values.reshape(3, -1)[j], ref["components1"][:, j]) >= 1 - 1e-5
        assert np.dot(c2.values.reshape(3, -1)[j], ref["components2"][:, j]) >= 1 - 1e-5
    s1, s2 = m.scores()
    assert s1.dims == ("mode", "time") and np.allclose(s1.values, ref["scores1"].T, rtol=1e-3, atol=1e-3)
    assert np.isclose(m.total_squared_covariance(), ref["total_squared_covariance"], rtol=1e-5)
    assert m.squared_covariance_fraction().values.sum() <= 1 + 1e-5
    with pytest.warns(UserWarning, match="sensitive to the number of modes"):     # mca.py:127-189, 3 modes of 28
        cf = m.covariance_fraction_CD95().values
    assert np.allclose(cf, ref["singular_values"] / ref["singular_values"].sum(), rtol=2e-5)
    t1, t2 = m.transform(X=X, Y=Y)
    assert np.allclose(t1.values, s1.values, rtol=1e-3, atol=1e-3) and not np.isnan(t2.values).any()
    with pytest.raises(ValueError, match="same number of samples"):
        xe.cross.MCA(n_modes=2, use_pca=False).fit(X, xe.DataArray(B[:50], dims=("time", "y", "x")), "time")


def test_decomposer_mirror(ctx):  # tests/linalg/test_decomposer.py
    import warnings

    from xeofs_amd.linalg import Decomposer

    X = mock_values().reshape(25, 20)
    X = (X - X.mean(0)).astype(np.float32)
    d = Decomposer(n_modes=5, random_state=3).fit(X)
    assert d.U_.shape == (25, 5) and d.s_.shape == (5,) and d.V_.shape == (20, 5)
    d2 = Decomposer(n_modes=5, random_state=3).fit(X)
    assert np.array_equal(d.U_, d2.U_) and np.array_equal(d.V_, d2.V_)        # bitwise determinism
    with pytest.raises(ValueError, match="rank"):
        Decomposer(n_modes=21).fit(X)
    with pytest.raises(ValueError, match="Unrecognized solver"):
        Decomposer(n_modes=2, solver="nope").fit(X)
    with pytest.raises(ValueError, match="init_rank_reduction"):
        Decomposer(n_modes=0.5, init_rank_reduction=0.0)
    tv = float(np.var(X.astype(np.float64), axis=0, ddof=1).sum())
    d3 = Decomposer(n_modes=0.6, init_rank_reduction=1.0, solver="full").fit(X, total_variance=tv)
    sv = np.linalg.svd(X.astype(np.float64), compute_uv=False)
    assert (d3.s_.astype(np.float64) ** 2).sum() / (sv ** 2).sum() >= 0.6 and d3.s_.size < 20
    Uo, so, Vo = orc.decomposer_fit(X.astype(np.float64), 0.6, init_rank_reduction=1.0, solver="full")
    assert d3.s_.size == so.size and np.allclose(d3.s_, so, rtol=1e-5)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Decomposer(n_modes=0.99, init_rank_reduction=0.1, solver="full").fit(X, total_variance=tv)
        assert any("init_rank_reduction" in str(x.message) for x in w)


def test_eof_dask_branch_policy(ctx):
    """a chunked (dask-backed) input selects the reference's dask branch
    (linalg/decomposer.py:104, 163-171: svd_compressed(k, seed, n_power_iter=4), sketch width
    max(20, k + 10)).  The engine materialises the data but keeps the branch's parameters; the result
    matches the oracle's restatement of svd_compressed (and the exact SVD) on a gap-separated spectrum."""
    import xeofs_amd as xe
    from xeofs_amd.linalg import Decomposer

    rng = np.random.default_rng(1)
    vals = ((rng.standard_normal((300, 7)) * (9.0 * 0.65 ** np.arange(7))) @ rng.standard_normal((7, 40 * 30))
            + 0.05 * rng.standard_normal((300, 1200))).reshape(300, 40, 30).astype(np.float32)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"), chunks=((100, 100, 100), (40,), (30,)))
    assert xe.labelled.is_lazy(X) and not xe.labelled.is_lazy(xe.DataArray(vals, dims=("time", "lat", "lon")))
    m = xe.single.EOF(n_modes=5, random_state=3).fit(X, "time")
    Xc = vals.reshape(300, -1).astype(np.float64)
    Xc -= Xc.mean(0)
    u, s, vt = orc.svd_compressed(Xc, 5, seed=3)
    assert np.allclose(m.singular_values().values, s, rtol=2e-5)
    comps = m.components().values.reshape(5, -1)
    for j in range(5):
        assert abs(np.dot(comps[j], vt[j])) > 1 - 1e-5
    # policy: width max(20, k + 10) and 4 power passes; never the exact solver for lazy input
    d = Decomposer(n_modes=3, lazy_input=True, ctx=ctx, solver_kwargs={"n_power_iter": 2, "compute": False})
    d.fit(Xc.astype(np.float32))
    assert np.allclose(d.s_, s[:3], rtol=2e-5)


def test_compute_false_defers_the_fit(ctx):
    """Row R10, `compute=False` (base_model.py:57-82, base_model_single_set.py:157-159): with a chunked input the fit is
    deferred -- `fit` returns without touching the data, `compute()` (or the first read of a fitted quantity, like
    touching a lazy DataArray in the reference) runs it; results equal the eager fit bitwise.  An in-memory input is
    fitted at once whatever `compute` says (the reference's numpy branch has nothing lazy)."""
    import xeofs_amd as xe

    rng = np.random.default_rng(1)
    vals = ((rng.standard_normal((300, 7)) * (9.0 * 0.65 ** np.arange(7))) @ rng.standard_normal((7, 40 * 30))
            + 0.05 * rng.standard_normal((300, 1200))).reshape(300, 40, 30).astype(np.float32)
    lazy = xe.DataArray(vals, dims=("time", "lat", "lon"), chunks=((100, 100, 100), (40,), (30,)))
    eager = xe.single.EOF(n_modes=5, random_state=3).fit(lazy, "time")
    assert not eager.is_deferred
    m = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(lazy, "time")
    assert m.is_deferred and m._data == {}
    assert m.compute() is m and not m.is_deferred
    assert np.array_equal(m.singular_values().values, eager.singular_values().values)
    m2 = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(lazy, "time")
    assert m2.is_deferred
    assert np.array_equal(m2.components().values, eager.components().values) and not m2.is_deferred   # first read computes
    m3 = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(xe.DataArray(vals, dims=("time", "lat", "lon")), "time")
    assert not m3.is_deferred
    lz2 = xe.DataArray(vals[:, :20], dims=("time", "lat", "lon"), chunks=((300,), (20,), (30,)))
    c = xe.cross.MCA(n_modes=3, random_state=1, compute=False).fit(lazy, lz2, "time")
    assert c.is_deferred
    ce = xe.cross.MCA(n_modes=3, random_state=1).fit(lazy, lz2, "time")
    assert np.allclose(c.singular_values().values, ce.singular_values().values, rtol=1e-6) and not c.is_deferred
    # consumers that read the fitted state before any accessor run the pending fit themselves
    m4 = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(lazy, "time")
    assert m4.is_deferred
    assert np.allclose(m4.transform(lazy).values, eager.transform(lazy).values, atol=1e-5) and not m4.is_deferred
    m5 = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(lazy, "time")
    r5 = xe.single.EOFRotator(n_modes=3).fit(m5)
    re = xe.single.EOFRotator(n_modes=3).fit(eager)
    assert not m5.is_deferred and np.array_equal(r5.components().values, re.components().values)
    m6 = xe.single.EOF(n_modes=5, random_state=3, compute=False).fit(lazy, "time")
    b6 = xe.validation.EOFBootstrapper(n_bootstraps=2, seed=1).fit(m6, random_state=0)
    be = xe.validation.EOFBootstrapper(n_bootstraps=2, seed=1).fit(eager, random_state=0)
    assert not m6.is_deferred and np.array_equal(b6.explained_variance().values, be.explained_variance().values)
    c2 = xe.cross.MCA(n_modes=3, random_state=1, compute=False).fit(lazy, lz2, "time")
    t2 = c2.transform(lazy, lz2)
    te = ce.transform(lazy, lz2)
    assert not c2.is_deferred and np.allclose(t2[0].values, te[0].values, atol=1e-5)


def test_dataset_in_dataset_out(ctx):
    """Dataset in -> Dataset out with the same data_vars (preprocessing/stacker.py:203-206,
    271-275); the README quickstart (config 1) feeds a Dataset.  Two variables on different grids are
    concatenated along the feature axis (concatenator.py:58-81)."""
    import xeofs_amd as xe

    v = mock_values()
    a = xe.DataArray(v, dims=("time", "lat", "lon"), coords={"lat": [20.0, 30.0, 40.0, 50.0, 60.0]})
    b = xe.DataArray((v ** 2)[:, :3, :], dims=("time", "y", "x"))
    ds = xe.Dataset({"air": a, "sq": b})
    m = xe.single.EOF(n_modes=3, random_state=2, solver="randomized").fit(ds, "time")
    comps = m.components()
    assert isinstance(comps, xe.Dataset) and list(comps.data_vars) == ["air", "sq"]
    assert comps["air"].dims == ("mode", "lat", "lon") and comps["sq"].dims == ("mode", "y", "x")
    assert comps["air"].shape == (3, 5, 4) and comps["sq"].shape == (3, 3, 4)
    M = np.concatenate([v.reshape(25, 20), (v ** 2)[:, :3, :].reshape(25, 12)], axis=1)
    ref = orc.eof_fit(M, 3, random_state=2, solver="randomized")
    assert np.allclose(m.singular_values().values, ref["norms"], rtol=1e-5)
    flat = np.concatenate([comps["air"].values.reshape(3, -1), comps["sq"].values.reshape(3, -1)], axis=1)
    for j in range(3):
        assert np.dot(flat[j], ref["components"][:, j]) > 1 - 1e-5
    sc = m.scores()
    assert sc.dims == ("mode", "time")                       # scores are a single DataArray
    rec = m.inverse_transform(sc)
    assert isinstance(rec, xe.Dataset) and rec["sq"].dims == ("time", "y", "x")
    tr = m.transform(ds)
    assert np.allclose(tr.values, sc.values, atol=1e-3 * np.abs(sc.values).max())


def test_lazy_input_error_conventions(ctx):
    """linalg/decomposer.py:172-177 and :191-193 (tests/linalg/test_decomposer.py:255-268): complex + dask is
    not implemented; a variance-based number of modes cannot be combined with a dask-backed input."""
    import xeofs_amd as xe

    v = mock_values().astype(np.float32)
    lazy = xe.DataArray(v, dims=("time", "lat", "lon"), chunks=((25,), (5,), (4,)))
    with pytest.raises(NotImplementedError, match="Complex data together with dask"):
        xe.single.HilbertEOF(n_modes=2).fit(lazy, "time")
    with pytest.raises(ValueError, match="not supported with dask arrays"):
        xe.single.EOF(n_modes=0.9).fit(lazy, "time")


def test_resident_torch_input(ctx):
    """A field that already lives in HBM (torch CUDA tensor inside a DataArray) is read in place: same result,
    bit for bit, as the same values handed over as a numpy array."""
    import torch
    import xeofs_amd as xe

    v = mock_values().astype(np.float32)
    a = xe.DataArray(v, dims=("time", "lat", "lon"), coords={"lat": [20.0, 30.0, 40.0, 50.0, 60.0]})
    b = xe.DataArray(torch.as_tensor(v, device="cuda"), dims=("time", "lat", "lon"), coords={"lat": [20.0, 30.0, 40.0, 50.0, 60.0]})
    ma = xe.single.EOF(n_modes=3, use_coslat=True, random_state=4).fit(a, ("time",))
    mb = xe.single.EOF(n_modes=3, use_coslat=True, random_state=4).fit(b, ("time",))
    assert np.array_equal(ma.components().values, mb.components().values)
    assert np.array_equal(ma.scores().values, mb.scores().values)
    # non-leading sample dimension: the permuted view is materialised on the device
    mc = xe.single.EOF(n_modes=2, random_state=1).fit(b, ("lat", "lon"))
    md = xe.single.EOF(n_modes=2, random_state=1).fit(a, ("lat", "lon"))
    assert np.array_equal(mc.singular_values().values, md.singular_values().values)
    c1, c2 = xe.cross.MCA(n_modes=2, random_state=3, use_pca=False).fit(b, b, "time").components()
    assert c1.dims == ("mode", "lat", "lon") and not np.isnan(c2.values).any()


@pytest.mark.parametrize("solver_kw", [dict(solver="full"), dict(n_modes=0.95)])
def test_eof_wide_branch_on_a_land_mask(solver_kw):
    """A land / sea mask keeps the field in place (layout mode 3) and solver='full' with more than 256 samples -- or a float
    n_modes whose int(0.3 rank) + 10 exceeds the sketch kernels -- sends the decomposition through the exact Gram route
    (xeofs_amd/pca.py), which has to compact / scatter the feature axis like every other consumer of a masked matrix:
    components aligned with the grid, NaN exactly at the masked points, values against the oracle."""
    import warnings

    import xeofs_amd as xe

    n, nlat, nlon = (900, 30, 50) if "n_modes" in solver_kw else (300, 24, 40)
    vals, lat = orc.synthetic_field(n, nlat, nlon, rank=12, seed=3)
    vals = vals.reshape(n, nlat, nlon)
    land = np.zeros((nlat, nlon), bool)
    land[3:9, 5:17] = True
    land[15:20, 25:38] = True
    land[0, 0] = land[-1, -1] = True
    vals[:, land] = np.nan
    kw = dict(n_modes=6, random_state=1)
    kw.update(solver_kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = xe.single.EOF(**kw).fit(xe.DataArray(vals, dims=("time", "lat", "lon"), coords={"lat": lat}), "time")
    assert m.data["input_data"].masked
    c = m.components().values
    k = c.shape[0]
    assert np.isnan(c[:, land]).all() and not np.isnan(c[:, ~land]).any()
    flat = vals.reshape(n, -1)[:, ~land.reshape(-1)].astype(np.float64)
    ref = orc.eof_fit(flat, k, solver="full", random_state=1)
    s = np.asarray(m.singular_values().values, dtype=np.float64)
    assert np.all(np.abs(s - ref["norms"][:k]) <= 1e-5 * ref["norms"][0])
    C = c[:, ~land].T.astype(np.float64)
    for j in range(min(k, 6)):
        assert np.dot(C[:, j], ref["components"][:, j]) > 1 - 1e-5
    sc = m.scores().values
    t = m.transform(xe.DataArray(vals, dims=("time", "lat", "lon"), coords={"lat": lat})).values
    assert np.allclose(t, sc, atol=1e-4 * np.abs(sc).max())
