"""GPU parity of the CPCCA family: whitener, CCA / RDA / CPCCA(alpha), transform /
inverse_transform / predict and the Swenson (2015) diagnostics against the oracle restatement of
xeofs/cross/cpcca.py, xeofs/preprocessing/whitener.py and xeofs/utils/optional/statistics.py."""

import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def _pair(n=200, shape1=(12, 15), shape2=(10, 14), seed=1, noise=0.3):
    rng = np.random.default_rng(seed)
    p1, p2 = int(np.prod(shape1)), int(np.prod(shape2))
    T = rng.standard_normal((n, 5)) * (3.0 * 0.75 ** np.arange(5))
    A = T @ rng.standard_normal((5, p1)) + noise * rng.standard_normal((n, p1))
    B = T @ rng.standard_normal((5, p2)) + noise * rng.standard_normal((n, p2))
    return A.reshape((n,) + shape1), B.reshape((n,) + shape2)


def _models(alpha, use_pca, n_pca_modes=8, k=3, cls=None, **kw):
    import xeofs_amd as xe

    A, B = _pair()
    X = xe.DataArray(A, dims=("time", "lat", "lon"))
    Y = xe.DataArray(B, dims=("time", "y", "x"))
    if cls is None:
        m = xe.cross.CPCCA(n_modes=k, alpha=alpha, use_pca=use_pca, n_pca_modes=n_pca_modes, random_state=3, **kw)
    else:
        m = cls(n_modes=k, use_pca=use_pca, n_pca_modes=n_pca_modes, random_state=3, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(X, Y, "time")
        ref = orc.cpcca_fit(A.reshape(200, -1), B.reshape(200, -1), k, alpha=alpha, use_pca=use_pca,
                            n_pca_modes=n_pca_modes, pca_solver="full", random_state=3)
    return m, ref, X, Y, A, B


def _check_fit(m, ref, k=3, tol=2e-4):
    assert np.allclose(m.singular_values().values, ref["singular_values"], rtol=tol)
    c1, c2 = m.components()
    C1, C2 = c1.values.reshape(k, -1).T.astype(np.float64), c2.values.reshape(k, -1).T.astype(np.float64)
    for j in range(k):
        for C, key in ((C1, "components1"), (C2, "components2")):
            r = ref[key][:, j]
            assert abs(np.dot(C[:, j], r)) / np.linalg.norm(C[:, j]) / np.linalg.norm(r) > 1 - 20 * tol, (key, j)
    sgn = np.sign(np.sum(C1 * ref["components1"], axis=0))
    s1, s2 = m.scores()
    assert np.allclose(s1.values.T * sgn, ref["scores1"], atol=50 * tol * np.abs(ref["scores1"]).max())
    assert np.allclose(s2.values.T * sgn, ref["scores2"], atol=50 * tol * np.abs(ref["scores2"]).max())
    assert np.isclose(m.total_squared_covariance(), ref["total_squared_covariance"], rtol=10 * tol)
    return sgn


@pytest.mark.parametrize("alpha,use_pca", [(0.0, True), (0.5, True), ([0.0, 1.0], True), (1.0, True), (0.3, False),
                                           ([1.0, 0.2], [False, True])])
def test_cpcca_fit_vs_oracle(ctx, alpha, use_pca):
    if isinstance(use_pca, list):        # mixed: the oracle takes one flag -> compare invariants only
        m, _, X, Y, A, B = _models(alpha, use_pca)
        s1, s2 = m.scores()
        t1, t2 = m.transform(X=X, Y=Y)
        assert np.allclose(t1.values, s1.values, atol=2e-3 * np.abs(s1.values).max())
        assert np.allclose(t2.values, s2.values, atol=2e-3 * np.abs(s2.values).max())
        assert (m.squared_covariance_fraction().values >= 0).all()
        return
    m, ref, X, Y, A, B = _models(alpha, use_pca)
    _check_fit(m, ref)


def test_mca_matrix_free_sketch_drawn_ahead(ctx):
    """MCA(use_pca=False) on fields wide enough for the sketch to be drawn ahead (cpcca.py::_sketch_ahead): the draw is as
    tall as the RAW feature count of the narrower field and its leading rows are used -- numpy fills row by row, so they
    are exactly what `RandomState(seed).normal(size=(valid features, l))` gives the oracle -- here with all-NaN grid
    points dropped from the narrower field."""
    import xeofs_amd as xe

    rng = np.random.default_rng(4)
    n, k = 60, 3
    T = rng.standard_normal((n, 5)) * np.array([9.0, 6.0, 4.0, 1.0, 0.5])
    A = (T @ rng.standard_normal((5, 64 * 60)) + 0.3 * rng.standard_normal((n, 64 * 60))).astype(np.float32)
    B = (T @ rng.standard_normal((5, 50 * 70)) + 0.3 * rng.standard_normal((n, 50 * 70))).astype(np.float32)
    B[:, rng.choice(B.shape[1], 100, replace=False)] = np.nan          # the narrower field loses 100 grid points
    X = xe.DataArray(A.reshape(n, 64, 60), dims=("time", "lat", "lon"))
    Y = xe.DataArray(B.reshape(n, 50, 70), dims=("time", "y", "x"))
    m = xe.cross.MCA(n_modes=k, use_pca=False, random_state=11)
    m._SKETCH_AHEAD_MIN = 2000                                          # (production: 20 000 features and more)
    fut = m._sketch_ahead(X, Y, "time")
    assert fut is not None and fut[0].result().shape == (3500, k + 10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(X, Y, "time")
        ref = orc.cpcca_fit(A.astype(np.float64), B.astype(np.float64), k, alpha=1.0, use_pca=False, random_state=11)
    assert np.allclose(m.singular_values().values, ref["singular_values"], rtol=2e-5)
    c1, c2 = m.components()
    C2 = c2.values.reshape(k, -1).T
    vf = ~np.isnan(B).all(axis=0)
    assert np.isnan(C2[~vf]).all()
    for j in range(k):
        r = ref["components2"][:, j]
        assert abs(np.dot(C2[vf, j].astype(np.float64), r)) / np.linalg.norm(C2[vf, j]) / np.linalg.norm(r) > 1 - 1e-5


@pytest.mark.parametrize("cls_name,alpha", [("CCA", 0.0), ("RDA", [0.0, 1.0]), ("MCA", 1.0)])
def test_named_models(ctx, cls_name, alpha):
    import xeofs_amd as xe

    m, ref, X, Y, A, B = _models(alpha, True, cls=getattr(xe.cross, cls_name))
    assert "alpha" not in m.get_params()
    _check_fit(m, ref)
    if cls_name == "CCA":      # canonical correlations: scores of each field are uncorrelated
        cx = m.correlation_coefficients_X().values
        assert np.abs(cx - np.diag(np.diag(cx))).max() < 1e-3


@pytest.mark.parametrize("alpha,use_pca", [(0.0, True), (0.6, True), (1.0, True), (1.0, False)])
def test_cpcca_diagnostics_vs_oracle(ctx, alpha, use_pca):
    m, ref, X, Y, A, B = _models(alpha, use_pca)
    d = orc.cpcca_diagnostics(ref)
    tol = dict(rtol=2e-3, atol=2e-4)
    # (the sign of a mode pair is a free choice of the SVD: compare magnitudes)
    assert np.allclose(np.abs(m.cross_correlation_coefficients().values), np.abs(d["cross_correlation_coefficients"]), **tol)
    assert np.allclose(np.abs(m.correlation_coefficients_X().values), np.abs(d["correlation_coefficients_X"]), **tol)
    assert np.allclose(np.abs(m.correlation_coefficients_Y().values), np.abs(d["correlation_coefficients_Y"]), **tol)
    assert np.allclose(m.squared_covariance_fraction().values, d["squared_covariance_fraction"], **tol)
    assert np.allclose(m.fraction_variance_X_explained_by_X().values, d["fraction_variance_X_explained_by_X"], **tol)
    assert np.allclose(m.fraction_variance_Y_explained_by_Y().values, d["fraction_variance_Y_explained_by_Y"], **tol)
    if use_pca:
        assert np.allclose(m.fraction_variance_Y_explained_by_X().values, d["fraction_variance_Y_explained_by_X"], **tol)
    else:
        with pytest.raises(NotImplementedError):
            m.fraction_variance_Y_explained_by_X()
    if np.isclose(alpha, 1.0):    # MCA: the residual formula reduces to s^2 / TSC
        assert np.allclose(m.squared_covariance_fraction().values,
                           m.squared_covariance().values / m.total_squared_covariance(), rtol=1e-3)


@pytest.mark.parametrize("alpha,use_pca", [(0.2, True), (1.0, False)])
@pytest.mark.parametrize("kind", ["homogeneous", "heterogeneous"])
def test_correlation_patterns_vs_oracle(ctx, alpha, use_pca, kind):
    m, ref, X, Y, A, B = _models(alpha, use_pca)
    (p1, p2), (v1, v2) = getattr(m, f"{kind}_patterns")()
    (r1, q1), (r2, q2) = orc.cpcca_patterns(ref, kind)
    assert p1.dims == ("mode", "lat", "lon") and v2.dims == ("mode", "y", "x")
    P1, P2 = p1.values.reshape(3, -1).T, p2.values.reshape(3, -1).T
    sgn = np.sign(np.sum(P1 * r1, axis=0))
    assert np.allclose(P1 * sgn, r1, atol=2e-3) and np.allclose(P2 * sgn, r2, atol=2e-3)
    assert np.abs(P1).max() <= 1 + 1e-4
    big = q1 > 1e-6      # p-values: compared where they are not astronomically small
    assert np.allclose(v1.values.reshape(3, -1).T[big], q1[big], rtol=5e-2, atol=1e-6)
    # `correction=`: the reference validates the name and then calls statsmodels' multipletests with ITS defaults
    # (utils/optional/statistics.py:150-157 forwards neither method nor alpha): Holm-Sidak step-down per mode
    (_, _), (c1, c2) = getattr(m, f"{kind}_patterns")(correction="fdr_bh")
    raw, adj = v1.values.reshape(3, -1).T.astype(np.float64), c1.values.reshape(3, -1).T.astype(np.float64)
    assert np.allclose(adj, orc.holm_sidak(raw), rtol=1e-5, atol=1e-12)
    assert np.all(adj >= raw * (1 - 1e-6)) and np.all(adj <= 1.0)
    for j in range(3):        # step-down: adjusted values are monotone in the raw ones
        o = np.argsort(raw[:, j], kind="stable")
        assert np.all(np.diff(adj[o, j]) >= -1e-7)
    with pytest.raises(ValueError, match="is not in the accepted methods"):
        m.homogeneous_patterns(correction="benjamini")


@pytest.mark.parametrize("alpha,use_pca", [(0.2, True), (1.0, False), (1.0, True)])
def test_transform_inverse_predict(ctx, alpha, use_pca):
    import xeofs_amd as xe

    m, ref, X, Y, A, B = _models(alpha, use_pca)
    s1, s2 = m.scores()
    t1, t2 = m.transform(X=X, Y=Y)
    assert np.allclose(t1.values, s1.values, atol=2e-3 * np.abs(s1.values).max())
    assert np.allclose(t2.values, s2.values, atol=2e-3 * np.abs(s2.values).max())
    tn = m.transform(X=X, normalized=True)
    assert np.allclose(np.linalg.norm(tn.values, axis=1), 1.0, atol=1e-3)
    # predict: oracle restatement on the same centred data
    Xc = A.reshape(200, -1) - A.reshape(200, -1).mean(0)
    pred = m.predict(X)
    sgn = np.sign(np.sum(s1.values.T * ref["scores1"], axis=0))
    refp = orc.cpcca_predict(ref, Xc)
    assert pred.dims == ("mode", "time")
    assert np.allclose(pred.values.T * sgn, refp, atol=5e-3 * np.abs(refp).max())
    # inverse_transform of the first two modes
    sub = xe.DataArray(s1.values[:2], dims=s1.dims, coords={"mode": [1, 2], "time": np.arange(200)})
    rec = m.inverse_transform(X=sub)
    assert rec.dims == ("time", "lat", "lon")
    refrec = orc.cpcca_inverse_transform(ref, ref["scores1"][:, :2], 1) + A.reshape(200, -1).mean(0)
    assert np.allclose(rec.values.reshape(200, -1), refrec, atol=5e-3 * np.abs(refrec).max())
    recs = m.inverse_transform(X=sub, Y=xe.DataArray(s2.values[:1], dims=s2.dims, coords={"mode": [1], "time": np.arange(200)}))
    assert isinstance(recs, list) and recs[1].dims == ("time", "y", "x")


@pytest.mark.parametrize("alpha,use_pca,power", [(1.0, True, 1), (1.0, False, 1), (0.3, True, 1), (1.0, True, 2),
                                                 (0.0, True, 2)])
def test_cpcca_rotator_vs_oracle(ctx, alpha, use_pca, power):
    """cross/cpcca_rotator.py:122-372 (N2, cross models)."""
    import xeofs_amd as xe

    m, ref, X, Y, A, B = _models(alpha, use_pca, k=4)
    rot = xe.cross.MCARotator(n_modes=3, power=power).fit(m) if np.isclose(alpha, 1.0) else \
        xe.cross.CPCCARotator(n_modes=3, power=power).fit(m)
    # align the oracle model's mode signs with the GPU model's before rotating (a free choice of the SVD)
    C1 = m.components()[0].values.reshape(4, -1).T
    sgn = np.sign(np.sum(C1 * ref["components1"], axis=0))
    for key in ("Q1", "Q2", "components1", "components2", "scores1", "scores2"):
        ref[key] = ref[key] * sgn
    rr = orc.cpcca_rotator_fit(ref, 3, power=power)
    assert np.allclose(rot.squared_covariance().values, rr["squared_covariance"], rtol=2e-3)
    assert (np.diff(rot.squared_covariance().values) <= 0).all()
    assert np.allclose(rot.data["norm1"], rr["norm1"], rtol=2e-3) and np.allclose(rot.data["norm2"], rr["norm2"], rtol=2e-3)
    c1, c2 = rot.components()
    F1, F2 = c1.values.reshape(3, -1).T, c2.values.reshape(3, -1).T
    assert np.allclose(F1, rr["components1"], atol=3e-3 * np.abs(rr["components1"]).max())
    assert np.allclose(F2, rr["components2"], atol=3e-3 * np.abs(rr["components2"]).max())
    s1, s2 = rot.scores()
    assert np.allclose(s1.values.T, rr["scores1"], atol=3e-3 * np.abs(rr["scores1"]).max())
    assert np.allclose(s2.values.T, rr["scores2"], atol=3e-3 * np.abs(rr["scores2"]).max())
    t1, t2 = rot.transform(X=X, Y=Y)
    Xc, Yc = A.reshape(200, -1) - A.reshape(200, -1).mean(0), B.reshape(200, -1) - B.reshape(200, -1).mean(0)
    r1 = orc.cpcca_rotator_transform(ref, rr, Xc, 1, 3, power)
    r2 = orc.cpcca_rotator_transform(ref, rr, Yc, 2, 3, power)
    assert np.allclose(t1.values.T, r1, atol=5e-3 * np.abs(r1).max())
    assert np.allclose(t2.values.T, r2, atol=5e-3 * np.abs(r2).max())
    if np.isclose(alpha, 1.0):     # without whitening the projection reproduces the fitted scores
        assert np.allclose(t1.values, s1.values, atol=5e-3 * np.abs(s1.values).max())
    if power == 1:   # orthogonal rotation conserves the squared covariance of the rotated modes' subspace... of MCA
        assert np.abs(rot.rotation_matrix().T @ rot.rotation_matrix() - np.eye(3)).max() < 1e-8
    assert rot.squared_covariance_fraction().values.sum() <= 1 + 1e-5


def test_whitener_identity_rule(ctx):
    """preprocessing/whitener.py:54-60: the whitener is the identity iff (1 - alpha) < eps -- every alpha >= 1 is, alpha =
    1 - 1e-9 is not (np.isclose would say it is)."""
    from xeofs_amd.cross.cpcca import _whitener_is_identity

    assert _whitener_is_identity(1.0) and _whitener_is_identity(1.5) and _whitener_is_identity(1.0 - 1e-17)
    assert not _whitener_is_identity(1.0 - 1e-9) and not _whitener_is_identity(0.0)
    m_one, ref, *_ = _models(1.0, True)
    m_big, *_ = _models(1.5, True)
    m_near, *_ = _models(1.0 - 1e-9, True)
    assert all(sd.T is None for sd in m_one.side) and all(sd.T is None for sd in m_big.side)
    assert all(sd.T is not None for sd in m_near.side)          # whitened (by an almost-identity matrix)
    assert np.allclose(m_big.singular_values().values, m_one.singular_values().values, rtol=1e-6)
    assert np.allclose(m_near.singular_values().values, ref["singular_values"], rtol=2e-4)


@pytest.mark.parametrize("alpha", [0.0, 0.5, [0.0, 1.0]])
def test_whitener_without_eigendecomposition_equals_the_eigh_route(ctx, alpha):
    """Round 5: the covariance of PC scores is diagonal up to rounding, so its matrix powers (whitener.py:106-123) follow from
    first divided differences (near_diagonal_powers) instead of the order-m eigen-decomposition.  Both routes on the same
    model: same whitener, same fit."""
    from xeofs_amd.cross import cpcca as mod

    m_new, ref, *_ = _models(alpha, True)
    routes = [sd.whitener_route for sd in m_new.side if sd.T is not None]
    assert routes and all(r == "near-diagonal" for r in routes), routes
    mod._Side._near_diagonal_ok = False
    try:
        m_old, *_ = _models(alpha, True)
    finally:
        mod._Side._near_diagonal_ok = True
    assert all(sd.whitener_route == "eigh" for sd in m_old.side if sd.T is not None)
    for a, b in zip(m_new.side, m_old.side):
        if a.T is None:
            assert b.T is None
            continue
        assert np.abs(a.T - b.T).max() <= 1e-9 * np.abs(b.T).max()
        assert np.abs(a.Tinv - b.Tinv).max() <= 1e-9 * np.abs(b.Tinv).max()
    assert np.allclose(m_new.singular_values().values, m_old.singular_values().values, rtol=1e-6)
    _check_fit(m_new, ref)


@pytest.mark.parametrize("swap", [False, True])
def test_mca_land_masks_in_place(ctx, swap):
    """MCA(use_pca=False) on two fields with land / sea masks: both stay in place (layout mode 3: all-NaN grid points are
    zero columns of the engine's matrices), the Gram route computes the total squared covariance and carries the power
    iterations.  swap: the field that is narrower BY VALID features is the wider one physically -- since round 6 the engine orients
    C by the valid feature counts like the reference (sklearn transposes when rows < cols), so that pair stays in place too
    (before, the model had to compact the fields and go again)."""
    import xeofs_amd as xe

    rng = np.random.default_rng(9)
    n, k = 80, 4
    T = rng.standard_normal((n, 6)) * np.array([9.0, 7.0, 5.0, 3.0, 1.0, 0.5])
    A = (T @ rng.standard_normal((6, 40 * 50)) + 0.3 * rng.standard_normal((n, 40 * 50))).astype(np.float32)
    B = (T @ rng.standard_normal((6, 30 * 60)) + 0.3 * rng.standard_normal((n, 30 * 60))).astype(np.float32)
    A[:, rng.choice(A.shape[1], 500 if swap else 150, replace=False)] = np.nan      # swap: 1500 valid < B's 1600 valid, 2000 > 1800 physical
    B[:, rng.choice(B.shape[1], 200, replace=False)] = np.nan
    X = xe.DataArray(A.reshape(n, 40, 50), dims=("time", "lat", "lon"))
    Y = xe.DataArray(B.reshape(n, 30, 60), dims=("time", "y", "x"))
    m = xe.cross.MCA(n_modes=k, use_pca=False, random_state=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(X, Y, "time")
        ref = orc.cpcca_fit(A.astype(np.float64), B.astype(np.float64), k, alpha=1.0, use_pca=False, random_state=2)
    assert m.data["input_data1"].masked and m.data["input_data2"].masked
    assert np.allclose(m.singular_values().values, ref["singular_values"], rtol=2e-5)
    assert np.isclose(m.total_squared_covariance(), ref["total_squared_covariance"], rtol=1e-5)
    for c, F, key in zip(m.components(), (A, B), ("components1", "components2")):
        C = c.values.reshape(k, -1).T
        vf = ~np.isnan(F).all(axis=0)
        assert np.isnan(C[~vf]).all() and not np.isnan(C[vf]).any()
        for j in range(k):
            r = ref[key][:, j]
            assert abs(np.dot(C[vf, j].astype(np.float64), r)) / np.linalg.norm(C[vf, j]) / np.linalg.norm(r) > 1 - 1e-5
    s1, s2 = m.scores()
    t1, t2 = m.transform(X=X, Y=Y)
    assert np.allclose(t1.values, s1.values, atol=1e-4 * np.abs(s1.values).max())
    assert np.allclose(t2.values, s2.values, atol=1e-4 * np.abs(s2.values).max())
