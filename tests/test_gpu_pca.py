"""GPU parity of the PCA pre-reduction path: `eofx_mat_gram_f32`,
`xeofs_amd.pca.ResidentPCA`, `MCA(use_pca=True)` against the oracle restatement of
xeofs/preprocessing/pca.py + xeofs/linalg/_numpy/_svd.py (exact SVD here: the reference's PCA is an unseeded
randomized SVD, so its own output is only defined to that solver's convergence)."""

import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def _field(n, p, seed, rank=12, noise=0.3):
    rng = np.random.default_rng(seed)
    T = rng.standard_normal((n, rank)) * (5.0 * 0.8 ** np.arange(rank))
    X = T @ rng.standard_normal((rank, p)) + noise * rng.standard_normal((n, p))
    return (X - X.mean(0)).astype(np.float32)


@pytest.mark.parametrize("n,p", [(300, 2000), (700, 5200), (900, 260)])
def test_mat_gram_both_sides(ctx, n, p):
    from xeofs_amd import engine

    X = _field(n, p, 0)
    mat = engine.from_dense(ctx, X)
    X64 = X.astype(np.float64)
    for side, ref in ((0, X64 @ X64.T), (1, X64.T @ X64)):
        G = mat.gram(side).cpu().numpy()
        d = ref.shape[0]
        assert np.abs(G[:d, :d] - ref).max() <= 2e-6 * np.abs(ref).max()     # float32-class products
        assert not G[d:].any() and not G[:, d:].any()
    mat.free()


@pytest.mark.parametrize("n,p,n_modes", [(400, 3000, 0.999), (400, 3000, 0.9), (400, 3000, 25), (1000, 300, 0.99),
                                         (120, 900, "all")])
def test_resident_pca_vs_exact_svd(ctx, n, p, n_modes):
    from xeofs_amd import engine
    from xeofs_amd.pca import ResidentPCA

    X = _field(n, p, 1)
    mat = engine.from_dense(ctx, X)
    X64 = X.astype(np.float64)
    tv = orc.total_variance(X64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pca = ResidentPCA(ctx, n_modes).fit(mat, tv)
        # the reference's policy with an exact solver: same n_modes_precompute, same truncation rule
        nm = min(n, p) if n_modes == "all" else n_modes
        U, s, V = orc.decomposer_fit(X64, nm, solver="full")
    m = len(s)
    assert pca.m == m
    keep = s > 1e-3 * s[0]          # ("all" reaches the numerically zero tail of centred data)
    assert np.allclose(pca.s[keep], s[keep], rtol=2e-5)
    Vg = pca.components().astype(np.float64)
    assert np.abs(Vg[:, keep].T @ Vg[:, keep] - np.eye(keep.sum())).max() < 2e-5
    gaps = np.minimum(np.abs(np.diff(s, prepend=np.inf)), np.abs(np.diff(s, append=0.0))) / s[0]
    for j in np.nonzero(keep & (gaps > 1e-3))[0]:
        assert np.dot(Vg[:, j], V[:, j]) > 1 - 1e-4, j                       # same sign convention too
    sc = pca.scores()
    assert np.allclose(sc[:, keep], (X64 @ Vg)[:, keep], atol=2e-4 * s[0])
    # transform on the training matrix reproduces the scores; back-projection is V Q
    assert np.allclose(pca.transform(mat)[:, keep], sc[:, keep], atol=2e-4 * s[0])
    Q = np.linalg.qr(np.random.default_rng(3).standard_normal((m, 3)))[0]
    assert np.allclose(pca.back_project(Q), Vg @ Q, atol=1e-5)
    mat.free()


def test_pca_variance_warning_and_errors(ctx):
    from xeofs_amd import engine
    from xeofs_amd.pca import ResidentPCA

    X = _field(200, 1500, 2, rank=150, noise=1.0)
    mat = engine.from_dense(ctx, X)
    with pytest.warns(UserWarning, match="Please consider increasing `init_rank_reduction`"):
        p = ResidentPCA(ctx, 0.9999, init_rank_reduction=0.05).fit(mat)
    assert p.m == int(200 * 0.05)
    with pytest.raises(ValueError, match="rank of the dataset"):
        ResidentPCA(ctx, 201).fit(mat)
    with pytest.raises(ValueError, match="init_rank_reduction"):
        ResidentPCA(ctx, 0.9, init_rank_reduction=0.0)
    mat.free()


@pytest.mark.parametrize("use_pca", [True, [True, False]])
def test_mca_default_pca_route_vs_oracle(ctx, use_pca):
    """`xe.cross.MCA()` with the reference's default arguments (use_pca=True, n_pca_modes=0.999)."""
    import xeofs_amd as xe

    rng = np.random.default_rng(5)
    T = rng.standard_normal((150, 6)) * (3.0 * 0.7 ** np.arange(6))
    A = (T @ rng.standard_normal((6, 24 * 30)) + 0.05 * rng.standard_normal((150, 720))).reshape(150, 24, 30)
    B = (T @ rng.standard_normal((6, 20 * 28)) + 0.05 * rng.standard_normal((150, 560))).reshape(150, 20, 28)
    X = xe.DataArray(A, dims=("time", "lat", "lon"))
    Y = xe.DataArray(B, dims=("time", "y", "x"))
    m = xe.cross.MCA(n_modes=4, random_state=7, use_pca=use_pca).fit(X, Y, "time")
    ref = orc.mca_fit(A.reshape(150, -1), B.reshape(150, -1), 4, random_state=7, use_pca=True, pca_random_state=0)
    if use_pca is True:
        assert (m.pca[0].m, m.pca[1].m) == ref["pca_modes"]
        tol = 2e-4
    else:
        ref = orc.mca_fit(A.reshape(150, -1), B.reshape(150, -1), 4, random_state=7, use_pca=False)
        assert m.pca[1] is None
        tol = 2e-3        # one field truncated at 99.9 % of its variance, the other not
    assert np.allclose(m.singular_values().values, ref["singular_values"], rtol=tol)
    c1, c2 = m.components()
    C1, C2 = c1.values.reshape(4, -1).T, c2.values.reshape(4, -1).T
    assert np.abs(C1.T @ C1 - np.eye(4)).max() < 1e-4
    for j in range(4):
        assert abs(np.dot(C1[:, j], ref["components1"][:, j])) > 1 - 10 * tol
        assert abs(np.dot(C2[:, j], ref["components2"][:, j])) > 1 - 10 * tol
    s1, s2 = m.scores()
    t1, t2 = m.transform(X=X, Y=Y)
    assert np.allclose(t1.values, s1.values, atol=2e-3 * np.abs(s1.values).max())
    assert np.allclose(t2.values, s2.values, atol=2e-3 * np.abs(s2.values).max())
    if use_pca is True:
        sgn = np.sign(np.sum(C1 * ref["components1"], axis=0))
        assert np.allclose(s1.values.T * sgn, ref["scores1"], atol=5e-3 * np.abs(ref["scores1"]).max())
        assert np.isclose(m.total_squared_covariance(), ref["total_squared_covariance"], rtol=1e-4)


@pytest.mark.parametrize("n_modes,solver", [(0.95, "auto"), (300, "randomized"), (40, "full")])
def test_decomposer_wide_route(ctx, n_modes, solver):
    """Decomposer requests beyond the sketch kernels' width (float n_modes -> int(0.3 * rank) modes,
    hundreds of modes, solver='full' at rank > 256) take the exact Gram route; policy, truncation and
    sign rule are the reference's (linalg/decomposer.py:76-226)."""
    from xeofs_amd.linalg import Decomposer

    X = _field(1000, 4000, 7, rank=30, noise=0.2)
    X64 = X.astype(np.float64)
    tv = orc.total_variance(X64)
    d = Decomposer(n_modes=n_modes, solver=solver, ctx=ctx).fit(X, total_variance=tv)
    U, s, V = orc.decomposer_fit(X64, n_modes, solver="full")
    assert d.s_.shape == s.shape and d.V_.shape == V.shape and d.U_.shape == U.shape
    assert np.allclose(d.s_, s, rtol=2e-5)
    for j in range(min(20, len(s))):
        assert np.dot(d.V_[:, j].astype(np.float64), V[:, j]) > 1 - 1e-4
        assert np.dot(d.U_[:, j].astype(np.float64), U[:, j]) > 1 - 1e-4


def test_eof_many_modes(ctx):
    import xeofs_amd as xe

    vals = _field(600, 40 * 50, 8, rank=20, noise=0.2).reshape(600, 40, 50)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"))
    m = xe.single.EOF(n_modes=400, random_state=1).fit(X, "time")
    ref = orc.eof_fit(vals.reshape(600, -1).astype(np.float64), 400, solver="full")
    assert np.allclose(m.singular_values().values, ref["norms"], rtol=2e-5)
    assert m.explained_variance_ratio().values.sum() <= 1 + 1e-6
    rec = m.inverse_transform(m.scores()).values
    full = ref["scores"] @ ref["components"].T + vals.reshape(600, -1).mean(0)
    assert np.allclose(rec.reshape(600, -1), full, atol=2e-4 * np.abs(full).max())


@pytest.mark.parametrize("n,p,n_modes", [(1200, 4000, 0.999), (1200, 4000, 60), (700, 3000, 0.95)])
def test_resident_pca_randomized_route(ctx, n, p, n_modes):
    """The reference's own PCA solver -- scikit-learn's randomized SVD of width int(0.3 rank) + 10 with 4 re-normalised power
    iterations (linalg/_numpy/_svd.py:170-186, unseeded) -- carried out on the resident sample-space Gram matrix: blocked
    device Cholesky-QR of the n x l panels, one library eigen-decomposition of order l at the end, no order-n one.  Gate
    "to the reference solver's convergence": modes that stand clear of the rest of the spectrum match the EXACT SVD to
    1e-5; every singular value is at least as close to the exact one as the seeded scikit-learn restatement gets (plus
    float32 rounding); same number of kept modes; orthonormal factors; scores = X V."""
    from xeofs_amd import engine
    from xeofs_amd.pca import ResidentPCA

    X = _field(n, p, 7)
    mat = engine.from_dense(ctx, X)
    X64 = X.astype(np.float64)
    tv = orc.total_variance(X64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pca = ResidentPCA(ctx, n_modes, solver="randomized", random_state=3).fit(mat, tv)
        pca2 = ResidentPCA(ctx, n_modes, solver="randomized", random_state=3).fit(mat, tv)
        exact = ResidentPCA(ctx, n_modes, solver="exact").fit(mat, tv)
        nm = n_modes
        k_pre = int(min(n, p) * 0.3) if isinstance(n_modes, float) else n_modes
        Ur, sr, Vtr = orc.randomized_svd(X64, k_pre, n_iter=4, random_state=3)      # the reference's solver, seeded
    assert pca.solver_used == "randomized" and exact.solver_used == "exact"
    assert np.array_equal(pca.s, pca2.s) and np.array_equal(pca.U, pca2.U)          # reproducible bit for bit
    se = np.linalg.svd(X64, compute_uv=False)
    m = pca.m
    assert abs(m - exact.m) <= max(1, int(0.02 * exact.m)), (m, exact.m)             # truncation rule on nearly the same spectrum
    err_ours = np.abs(pca.s - se[:m]) / se[0]
    err_ref = np.abs(sr[:m] - se[:m]) / se[0]
    assert np.all(err_ours <= 2.0 * err_ref + 1e-5), (err_ours.max(), err_ref.max())
    lead = [j for j in range(min(m, 10)) if (se[j] - se[j + 1]) / se[j] > 0.05]
    assert len(lead) >= 3 and np.all(err_ours[lead] <= 1e-5)
    V = pca.components().astype(np.float64)
    # V is orthonormal; U = X V / s is orthonormal only as far as the subspace is invariant (the reference keeps V and
    # defines the scores as X V, preprocessing/pca.py:120-131 -- so does this route)
    assert np.abs(V.T @ V - np.eye(m)).max() < 5e-5
    Ve = exact.components().astype(np.float64)
    for j in lead:
        assert abs(np.dot(V[:, j], Ve[:, j])) > 1 - 1e-5
    Z = X64 @ V
    assert np.allclose(Z, pca.scores(), atol=2e-4 * np.abs(Z).max())
    mat.free()


@pytest.mark.parametrize("l", [100, 333, 700])
def test_blocked_device_cholesky_qr(ctx, l):
    """eofx_panel_cholqr_f32 / eofx_panel_rinv_f64 beyond one wavefront's 64 columns: the blocked device factorisation
    (launch_rinv_blocked) against numpy, with an exactly dependent column (zero column out, as in the 64-column kernel)."""
    import torch
    from xeofs_amd import engine

    rows = 2560
    rng = np.random.default_rng(l)
    L = (l + 31) // 32 * 32
    P = np.zeros((rows, L), np.float32)
    P[:, :l] = rng.standard_normal((rows, l)) * (1.0 + 9.0 * rng.random(l))
    P[:, 77] = P[:, 3] + P[:, 5]
    Pd = torch.as_tensor(P, device="cuda")
    G = engine.panel_gram(ctx, Pd)
    R = engine.panel_rinv(ctx, G, l).cpu().numpy()
    assert not np.tril(R, -1).any() and not R[:, 77].any() and not R[l:].any() and not R[:, l:].any()
    Q = engine.panel_cholqr(ctx, Pd, l, G).double().cpu().numpy()[:, :l]
    keep = np.setdiff1d(np.arange(l), [77])
    assert not Q[:, 77].any()
    assert np.abs(Q[:, keep].T @ Q[:, keep] - np.eye(l - 1)).max() < 1e-6
    P64 = P[:, keep].astype(np.float64)
    assert np.abs(P64 - Q[:, keep] @ (Q[:, keep].T @ P64)).max() < 1e-4 * np.abs(P64).max()
    # the triangular factor itself against numpy's Cholesky of the Gram matrix without the dependent column
    Gh = G.cpu().numpy()[np.ix_(keep, keep)]
    Rn = np.linalg.inv(np.linalg.cholesky(Gh).T)
    assert np.allclose(R[np.ix_(keep, keep)], Rn, rtol=1e-6, atol=1e-9 * np.abs(Rn).max())


def test_basis_only_fast_path_against_the_eigh_route(ctx):
    """the `basis_only` fast path (the default of every MCA / CPCCA fit with alpha = 1 whose variance target is out of
    reach) drops the bottom of the sketch's spectrum through an inverse subspace iteration instead of the order-ell
    eigen-decomposition.  A/B against the eigh route on the SAME sketch: the iteration's own residual check passes, the kept
    subspaces agree (largest principal angle, captured variance), the downstream cross-covariance decomposition agrees, and the
    path does not pretend to have a spectrum (`s` raises, `singular_values_all` is None)."""
    import torch

    from xeofs_amd import engine
    from xeofs_amd.pca import ResidentPCA

    n, p = 1280, 6000
    rng = np.random.default_rng(3)
    T = rng.standard_normal((n, 30)) * (6.0 * 0.85 ** np.arange(30))
    X = (T @ rng.standard_normal((30, p)) + 0.5 * rng.standard_normal((n, p))).astype(np.float32)
    Y = (T[:, :20] @ rng.standard_normal((20, 4000)) + 0.5 * rng.standard_normal((n, 4000))).astype(np.float32)
    X -= X.mean(0)
    Y -= Y.mean(0)
    mx, my = engine.from_dense(ctx, X), engine.from_dense(ctx, Y)
    tvx, tvy = orc.total_variance(X.astype(np.float64)), orc.total_variance(Y.astype(np.float64))
    fits = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # "Dataset has 384 components, explaining ..." (both routes warn, as the reference)
        for fast in (True, False):
            px = ResidentPCA(ctx, 0.999, solver="randomized", random_state=11, basis_only=fast).fit(mx, tvx)
            py = ResidentPCA(ctx, 0.999, solver="randomized", random_state=12, basis_only=fast).fit(my, tvy)
            fits[fast] = (px, py)
    pf, pe = fits[True][0], fits[False][0]
    assert pf.solver_used == pe.solver_used == "randomized"
    assert not pf.spectrum_known and pe.spectrum_known and pf.m == pe.m == int(0.3 * n)
    assert pf.fast_path_residual is not None and pf.fast_path_residual <= 5e-2
    with pytest.raises(RuntimeError):
        pf.s
    assert pf.singular_values_all is None and pe.singular_values_all is not None
    # kept subspaces: both orthonormal bases of (nearly) the same space
    Vf = torch.as_tensor(pf.components(), device="cuda").double()
    Ve = torch.as_tensor(pe.components(), device="cuda").double()
    eye = torch.eye(pf.m, device="cuda", dtype=torch.float64)
    assert float((Vf.T @ Vf - eye).abs().max()) < 2e-5 and float((Ve.T @ Ve - eye).abs().max()) < 2e-5
    cosines = torch.linalg.svdvals(Vf.T @ Ve)
    # The dropped directions are Ritz directions to 5e-2 of their value.  Where the sketch's spectrum is flat at the cut (here:
    # a noise tail) WHICH of several directions of equal variance is dropped is not determined at that tolerance -- nor by the
    # reference, whose sketch is unseeded --, so single principal angles next to the cut may be large; what must hold: the two
    # bases share all but a sliver (at most n_oversamples directions turned by more than 8 degrees, less than 2.5 dimensions
    # exchanged in total) and capture the same variance.
    assert int((cosines < 0.99).sum()) <= pf.n_oversamples, cosines[-16:]
    assert float((1.0 - cosines ** 2).sum()) <= 2.5, cosines[-16:]
    assert int((cosines < 1 - 1e-6).sum()) <= 4 * (pf.n_oversamples + 22)
    # captured variance: the fast path keeps at least as much as exact bottom directions would lose (within 0.3 %)
    Xd = torch.as_tensor(X, device="cuda").double()
    cap_f, cap_e = float(((Xd @ Vf) ** 2).sum()), float(((Xd @ Ve) ** 2).sum())
    assert abs(cap_f - cap_e) <= 3e-3 * cap_e
    # scores stay X V on both routes
    assert np.allclose(pf.scores(), (Xd @ Vf).cpu().numpy() * 1.0, rtol=0, atol=2e-4 * np.abs(pe.scores()).max())
    # downstream: the cross-covariance decomposition on the two pairs of PC scores (what MCA does with them)
    outs = {}
    for fast in (True, False):
        wx = engine.from_dense(ctx, fits[fast][0].scores().astype(np.float32))
        wy = engine.from_dense(ctx, fits[fast][1].scores().astype(np.float32))
        r = engine.crosscov_rsvd(ctx, wx, wy, 10, random_state=5)
        r["c1"] = fits[fast][0].back_project(r["Q1"])
        outs[fast] = r
        wx.free(); wy.free()
    sf, se = outs[True]["s"].astype(np.float64), outs[False]["s"].astype(np.float64)
    assert np.all(np.abs(sf - se) <= 1e-4 * se[0]), (sf, se)
    c = np.abs(np.sum(outs[True]["c1"].astype(np.float64) * outs[False]["c1"].astype(np.float64), axis=0))
    assert np.all(c[:5] >= 1 - 1e-4), c
    assert abs(outs[True]["total_squared_covariance"] / outs[False]["total_squared_covariance"] - 1) <= 2e-3
    mx.free(); my.free()
