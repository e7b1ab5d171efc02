"""GPU parity of the rotation path: `eofx_panel_row_normalize_f32`,
`eofx_panel_rot_step_f64`, `xeofs_amd.rotation.promax`, `xeofs_amd.single.EOFRotator`
against the oracle restatement of xeofs/linalg/_numpy/_rotation.py and xeofs/single/eof_rotator.py."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def _loadings(seed, p, m):
    """simple-structure loadings (each feature loads mainly on one factor) hidden by a random
    orthogonal mixing: the varimax optimum is well defined and reached in a few iterations"""
    rng = np.random.default_rng(seed)
    S = 0.1 * rng.standard_normal((p, m))
    S[np.arange(p), rng.integers(0, m, p)] += rng.uniform(1, 3, p)
    Q = np.linalg.qr(rng.standard_normal((m, m)))[0]
    L = S @ np.diag(np.linspace(2, 1, m)) @ Q
    L[rng.integers(0, p, 3)] = 0.0          # zero rows (Kaiser normalisation guards them with eps)
    return L.astype(np.float32)


@pytest.mark.parametrize("p,m", [(700, 3), (5000, 10), (70000, 33), (9000, 100), (40000, 200), (777, 256)])
def test_rot_step_matches_numpy(ctx, p, m):
    """the step kernel (<= 64 columns) and its column-blocked variant (128 / 256-wide panels) against float64 numpy"""
    import torch
    from xeofs_amd import engine, rotation

    Lh = _loadings(0, p, m)
    L = rotation._rot_width(m)
    rows_pad = (p + 511) // 512 * 512
    P = engine.panel_import(ctx, Lh, rows_pad, L)
    Xn = engine.panel_row_normalize(ctx, P)
    X64 = Lh.astype(np.float64)
    h = np.sqrt((X64 ** 2).sum(1))
    Xref = X64 / (h + np.finfo(np.float32).eps)[:, None]
    got = engine.panel_export(ctx, Xn, p, m)
    assert np.abs(got - Xref).max() < 2e-6
    rng = np.random.default_rng(1)
    R = np.linalg.qr(rng.standard_normal((m, m)))[0]
    Rp = np.zeros((L, L)); Rp[:m, :m] = R
    Xf = got.astype(np.float64)
    b = Xf @ R
    aux = np.zeros(L); aux[:m] = (b ** 2).sum(0) / p
    G = engine.panel_rot_step(ctx, Xn, torch.as_tensor(Rp, device="cuda"), torch.as_tensor(aux, device="cuda"), 0)
    ref = Xf.T @ (b * (b ** 2 - aux[:m]))
    assert np.abs(G.cpu().numpy()[:m, :m] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    aux1 = np.ones(L); aux1[:m] = np.abs(b).max(0)
    G1 = engine.panel_rot_step(ctx, Xn, torch.as_tensor(Rp, device="cuda"), torch.as_tensor(aux1, device="cuda"), 1, 3.0)
    z = b / aux1[:m]
    ref1 = b.T @ (z * np.abs(z) ** 2)
    assert np.abs(G1.cpu().numpy()[:m, :m] - ref1).max() <= 1e-11 * max(1.0, np.abs(ref1).max())


@pytest.mark.parametrize("power", [1, 2, 4])
@pytest.mark.parametrize("p,m", [(900, 4), (30000, 12), (4000, 70), (2500, 130)])   # > 64 modes: the wide (library GEMM) step
def test_promax_matches_oracle(ctx, p, m, power):
    from xeofs_amd import rotation

    Lh = _loadings(2, p, m)
    Xr, R, phi = rotation.promax(ctx, Lh, power=power, rtol=1e-10)
    Xo, Ro, phio = orc.promax(Lh.astype(np.float64), power=power, rtol=1e-10)
    scale = np.abs(Xo).max()
    # float32 loadings storage bounds the agreement (tolerance: 2e-5 of the largest loading)
    assert np.abs(R - Ro).max() < 2e-5 * max(1.0, np.abs(Ro).max())
    assert np.abs(phi - phio).max() < 2e-5
    assert np.abs(Xr - Xo).max() < 2e-5 * scale
    if power == 1:
        assert np.abs(R.T @ R - np.eye(m)).max() < 1e-10


@pytest.mark.parametrize("case", ["real_a", "real_b"])
@pytest.mark.parametrize("power", [1, 2, 4])
def test_promax_matches_reference_golden(ctx, case, power):
    """HIP rotation path vs outputs of the reference's own `_promax` (tests/golden/g8_rotation.npz,
    oracle/make_golden_rotation.py).  Tolerance: float32 loadings storage, 2e-5 of the largest loading."""
    import os
    from xeofs_amd import rotation

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_rotation.npz"))
    X = g[f"{case}_X"]
    Xr, R, phi = rotation.promax(ctx, X.astype(np.float32), power=power)
    assert np.abs(R - g[f"{case}_p{power}_R"]).max() < 2e-5 * max(1.0, np.abs(g[f"{case}_p{power}_R"]).max())
    assert np.abs(phi - g[f"{case}_p{power}_phi"]).max() < 2e-5
    assert np.abs(Xr - g[f"{case}_p{power}_Xrot"]).max() < 2e-5 * np.abs(X).max()


def test_promax_errors(ctx):
    from xeofs_amd import rotation

    with pytest.raises(ValueError, match="Cannot rotate 1 modes"):
        rotation.promax(ctx, np.ones((10, 1), np.float32))
    with pytest.raises(RuntimeError, match="did not converge"):
        rotation.promax(ctx, _loadings(3, 500, 6), max_iter=2, rtol=1e-15)


@pytest.mark.parametrize("power", [1, 2])
def test_eof_rotator_model(ctx, power):
    """tests/models/single/test_eof_rotator.py:43-170 + value parity with the oracle."""
    import xeofs_amd as xe

    vals = orc.synthetic_field(80, 9, 12, rank=8, seed=4)[0].reshape(80, 9, 12)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"),
                     coords={"time": np.arange(80), "lat": np.linspace(-40, 40, 9), "lon": np.arange(12) * 30.0})
    model = xe.single.EOF(n_modes=6, random_state=2, solver="full").fit(X, "time")
    rot = xe.single.EOFRotator(n_modes=4, power=power).fit(model)
    assert rot._params["max_iter"] == 1000
    Xs = vals.reshape(80, -1).astype(np.float64)
    eof = orc.eof_fit(Xs, 6, random_state=2, solver="full")
    eof["input_data"] = Xs
    ref = orc.eof_rotator_fit(eof, 4, power=power)
    ev = rot.explained_variance().values
    assert ev.shape == (4,) and (ev > 0).all()
    assert np.allclose(ev, ref["explained_variance"], rtol=2e-4)
    if power == 1:
        assert np.isclose(ev.sum(), model.explained_variance().values[:4].sum(), rtol=1e-5)
    assert rot.explained_variance_ratio().values.sum() <= 1
    comps = rot.components().values.reshape(4, -1).T
    for j in range(4):
        assert abs(np.dot(comps[:, j], ref["components"][:, j])) / np.linalg.norm(comps[:, j]) / np.linalg.norm(ref["components"][:, j]) > 1 - 1e-4
    assert np.allclose(comps, ref["components"], atol=2e-3 * np.abs(ref["components"]).max())
    sc = rot.scores().values.reshape(4, -1).T
    assert np.allclose(sc, ref["scores"], atol=2e-3 * np.abs(ref["scores"]).max())
    # transform on the training data reproduces the scores (eof_rotator.py:220-257)
    tr = rot.transform(X).values.reshape(4, -1).T
    assert np.allclose(tr, sc, atol=2e-3 * np.abs(sc).max())
    # reconstruction from rotated scores/components (orthogonal case: identical to the unrotated one)
    rec = rot.inverse_transform(rot.scores()).values
    assert rec.shape == vals.shape
    if power == 1:
        sc0 = model.scores()
        sub = xe.DataArray(sc0.values[:4], dims=sc0.dims, coords={"mode": [1, 2, 3, 4], "time": np.arange(80)})
        rec0 = model.inverse_transform(sub).values
        assert np.allclose(rec, rec0, atol=1e-3 * np.abs(rec0).max())
    with pytest.raises(NotImplementedError):
        rot.fit_transform(model)


def test_eof_rotator_more_than_64_modes(ctx):
    """the whole rotator path (device-side finish, scores, transform) on 128-wide panels"""
    import xeofs_amd as xe

    n, nlat, nlon, k = 300, 15, 20, 72
    vals = orc.synthetic_field(n, nlat, nlon, rank=90, seed=6)[0].reshape(n, nlat, nlon)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"))
    model = xe.single.EOF(n_modes=k + 4, random_state=2, solver="full").fit(X, "time")
    rot = xe.single.EOFRotator(n_modes=k, power=1, rtol=1e-10).fit(model)
    Xs = vals.reshape(n, -1).astype(np.float64)
    eof = orc.eof_fit(Xs, k + 4, random_state=2, solver="full")
    eof["input_data"] = Xs
    ref = orc.eof_rotator_fit(eof, k, power=1, rtol=1e-10)
    ev = rot.explained_variance().values
    assert np.allclose(ev, ref["explained_variance"], rtol=5e-4)
    assert np.isclose(ev.sum(), model.explained_variance().values[:k].sum(), rtol=1e-5)
    comps = rot.components().values.reshape(k, -1).T
    cos = np.abs((comps * ref["components"]).sum(0)) / np.linalg.norm(comps, axis=0) / np.linalg.norm(ref["components"], axis=0)
    assert cos.min() > 1 - 1e-3, cos.min()
    sc = rot.scores().values.reshape(k, -1).T
    tr = rot.transform(X).values.reshape(k, -1).T
    assert np.allclose(tr, sc, atol=2e-3 * np.abs(sc).max())


@pytest.mark.parametrize("power", [1, 2, 4])
@pytest.mark.parametrize("m", [2, 5, 17, 40, 70])      # > 32 complex modes: the wide (library GEMM) step
def test_complex_promax_vs_oracle(ctx, power, m):
    """`rotation.cpromax_panel`: complex loadings as a [Re | Im] panel, the reference loop with the complex m x m matrices
    in their real embedding (`eofx_panel_rot_step_f64` modes 2 / 3) against the oracle's complex `promax` (pinned to the
    reference's own `_rotation.py` on complex inputs, fixture G8)."""
    from xeofs_amd import engine, rotation

    rng = np.random.default_rng(11 * m + power)
    p = 3000 + 37 * m
    decay = 0.8 if m <= 20 else 0.97             # (0.8^70 would leave columns of 1e-7: no defined optimum to compare)
    base = (rng.standard_normal((p, m)) + 1j * rng.standard_normal((p, m))) * (2.0 * decay ** np.arange(m))
    base[rng.integers(0, p, p // 3), rng.integers(0, m, p // 3)] *= 4.0          # some simple structure to rotate towards
    Xo, Ro, phio = orc.promax(base, power=power)
    Xrot, pp, mm, R, phi = rotation.cpromax_panel(ctx, base.astype(np.complex64), power=power)
    out = Xrot[:p].cpu().numpy()
    ch = Xrot.shape[1] // 2
    Xg = out[:, :m] + 1j * out[:, ch:ch + m]
    assert (pp, mm) == (p, m)
    assert np.abs(R - Ro).max() < 2e-4 * np.abs(Ro).max()
    assert np.abs(phi - phio).max() < 5e-4 * np.abs(phio).max()
    assert np.abs(Xg - Xo).max() < 5e-4 * np.abs(Xo).max()


@pytest.mark.parametrize("power", [1, 2])
@pytest.mark.parametrize("kind", ["complex", "hilbert"])
def test_complex_eof_rotator_model(ctx, power, kind):
    """ComplexEOFRotator / HilbertEOFRotator (eof_rotator.py:294-400) on a fitted model against the oracle's
    `eof_rotator_fit` fed with the SAME unrotated solution (the phases of a complex SVD are arbitrary, so the unrotated
    model, not the data, is the common starting point)."""
    import xeofs_amd as xe

    rng = np.random.default_rng(7)
    n, shape, k, mrot = 120, (14, 18), 6, 4
    p = int(np.prod(shape))
    t = np.arange(n)[:, None]
    T = np.sin(2 * np.pi * t * np.arange(1, 6)[None, :] / 31.0 + rng.uniform(0, 6, 5)) * (3.0 * 0.75 ** np.arange(5))
    A = T @ rng.standard_normal((5, p)) + 0.1 * rng.standard_normal((n, p))
    if kind == "complex":
        data = (A + 1j * (np.roll(T, 4, axis=0) @ rng.standard_normal((5, p)))).reshape((n,) + shape)
        model = xe.single.ComplexEOF(n_modes=k, random_state=3)
        Rot = xe.single.ComplexEOFRotator
    else:
        data = A.reshape((n,) + shape)
        model = xe.single.HilbertEOF(n_modes=k, random_state=3)
        Rot = xe.single.HilbertEOFRotator
    X = xe.DataArray(data, dims=("time", "lat", "lon"))
    model.fit(X, "time")
    with pytest.raises(TypeError):
        xe.single.EOFRotator(n_modes=mrot).fit(model)
    rot = Rot(n_modes=mrot, power=power).fit(model)
    eof = dict(components=np.asarray(model.data["components"]).astype(np.complex128),
               scores=np.asarray(model.data["scores"]).astype(np.complex128),
               norms=np.asarray(model.data["norms"], dtype=np.float64),
               explained_variance=np.asarray(model.data["explained_variance"], dtype=np.float64),
               total_variance=model.data["total_variance"], input_data=np.zeros((n, 1)))
    ref = orc.eof_rotator_fit(eof, mrot, power=power)
    c = rot.components().values.reshape(mrot, -1).T
    s = rot.scores().values.reshape(mrot, -1).T
    assert np.iscomplexobj(c) and rot.components().dims == ("mode", "lat", "lon")
    assert np.allclose(rot.explained_variance().values, ref["explained_variance"], rtol=2e-4)
    assert np.array_equal(rot.data["idx_modes_sorted"], ref["idx_modes_sorted"])
    assert np.abs(c - ref["components"]).max() < 1e-3 * np.abs(ref["components"]).max()
    assert np.abs(s - ref["scores"]).max() < 1e-3 * np.abs(ref["scores"]).max()
    assert np.abs(rot.rotation_matrix() - ref["rotation_matrix"]).max() < 5e-4
    assert np.abs(rot.phi_matrix() - ref["phi_matrix"]).max() < 1e-3
    amp = rot.components_amplitude().values.reshape(mrot, -1).T
    assert np.allclose(amp, np.abs(ref["components"]), atol=1e-3 * np.abs(ref["components"]).max())
    with pytest.raises(NotImplementedError):
        rot.transform(X)
