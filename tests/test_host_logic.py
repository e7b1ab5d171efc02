"""Host-side logic of the path that needs no GPU (runs in the CPU suite): the rules that decide which engine route a call
takes, and the pure-numpy algebra the model classes do on the host."""
import numpy as np
import pytest

from oracle import eof_oracle as orc  # noqa: F401  (checker only)


def test_sketch_leading_rows_are_the_shorter_draw():
    """What `CPCCA._sketch_ahead` relies on: numpy fills `RandomState(seed).normal(size=(rows, l))` row by row, so the
    sketch of a field that lost features to its NaN mask is the leading part of the draw for the raw feature count --
    for the native generator as for numpy itself."""
    from xeofs_amd import engine

    for seed, rows, small, l in ((11, 3500, 3400, 13), (5, 20001, 17777, 30), (0, 700, 1, 64)):
        tall = engine.sketch_matrix(rows, l, seed)
        short = engine.sketch_matrix(small, l, seed)
        assert np.array_equal(tall[:small], short)
        assert np.array_equal(short, np.random.RandomState(seed).normal(size=(small, l)).astype(np.float32))
        fut = engine.SketchFuture(rows, l, seed)
        assert np.array_equal(fut.result(), tall)


def test_fused_plan_and_lean_rules():
    """Which calls may take the one-call fit (`Decomposer.fused_plan`: sketch narrower than its 32 / 64-column panel with a
    spare column, n < P, randomized policy) and which complex models ask for the lean layout (sketches of at most 32
    complex columns)."""
    import xeofs_amd as xe
    from xeofs_amd.linalg.decomposer import Decomposer

    d = Decomposer(n_modes=50, random_state=1)
    assert d.fused_plan(10000, 1036800) == (50, 10, "auto")           # (the engine resolves sklearn's 7 / 4 rule)
    assert Decomposer(n_modes=22).fused_plan(5000, 20000) is None          # 32 columns: no spare column for the ones
    assert Decomposer(n_modes=54).fused_plan(5000, 20000) is None          # 64 columns
    assert Decomposer(n_modes=55).fused_plan(5000, 20000) is None          # wider than the fused panels
    assert Decomposer(n_modes=10).fused_plan(3000, 2000) is None           # sketch on the feature side
    assert Decomposer(n_modes=10).fused_plan(300, 400) is None             # small problem: the full solver
    assert Decomposer(n_modes=0.9).fused_plan(5000, 20000) is None         # variance-based n_modes
    assert Decomposer(n_modes=10, solver="full").fused_plan(5000, 20000) is None
    plan = Decomposer(n_modes=800, random_state=1).fused_plan(10000, 1036800)
    assert plan is None or plan[0] + plan[1] < 64
    assert xe.single.HilbertEOF(n_modes=20)._lean_ok() and xe.single.ComplexEOF(n_modes=22)._lean_ok()
    assert not xe.single.HilbertEOF(n_modes=23)._lean_ok()                 # 33 complex columns: written layouts
    assert not xe.single.ComplexEOF(n_modes=0.9)._lean_ok()
    assert xe.single.HilbertEOF(n_modes=28, solver_kwargs={"n_oversamples": 4})._lean_ok()


def test_complex_deflated_norms_formula():
    """`ComplexCPCCA._deflated_norms`: ||(Sx - r1 b1^H)^H (Sy - r2 b2^H)||_F^2 per mode from inner products of n-vectors,
    against the dense expression (the residual form of the squared covariance fraction, cpcca.py:418-512)."""
    from xeofs_amd.cross.complex_mca import ComplexCPCCA

    rng = np.random.default_rng(0)
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    for n, m1, m2, k in ((50, 7, 9, 3), (31, 12, 4, 4)):
        Sx, Sy, R1, R2, B1, B2 = c(n, m1), c(n, m2), c(n, k), c(n, k), c(m1, k), c(m2, k)
        M = Sx.conj().T @ Sy
        got = ComplexCPCCA._deflated_norms(Sx, Sy, R1, R2, B1, B2, (np.abs(M) ** 2).sum())
        ref = [np.linalg.norm((Sx - np.outer(R1[:, j], B1[:, j].conj())).conj().T @ (Sy - np.outer(R2[:, j], B2[:, j].conj()))) ** 2
               for j in range(k)]
        assert np.allclose(got, ref, rtol=1e-10)
    # real data is the special case the real CPCCA uses
    Sx, Sy = rng.standard_normal((40, 6)), rng.standard_normal((40, 5))
    U, s, Vt = np.linalg.svd(Sx.T @ Sy, full_matrices=False)
    R1, R2 = Sx @ U[:, :2], Sy @ Vt[:2].T
    got = ComplexCPCCA._deflated_norms(Sx, Sy, R1, R2, U[:, :2], Vt[:2].T, (s ** 2).sum())
    assert np.allclose(got, (s ** 2).sum() - s[:2] ** 2, rtol=1e-10)       # alpha = 1: deflation removes sigma_i^2


def test_cross_model_sketch_ahead_gate():
    """`CPCCA._sketch_ahead`: only for the matrix-free path with an integer seed and wide fields; the draw is as tall as the
    narrower RAW field."""
    import xeofs_amd as xe

    rng = np.random.default_rng(1)
    X = xe.DataArray(rng.standard_normal((12, 40, 60)).astype(np.float32), dims=("time", "lat", "lon"))
    Y = xe.DataArray(rng.standard_normal((12, 30, 70)).astype(np.float32), dims=("time", "y", "x"))
    m = xe.cross.MCA(n_modes=3, use_pca=False, random_state=4)
    assert m._sketch_ahead(X, Y, "time") is None                           # 2100 features: below the production gate
    m._SKETCH_AHEAD_MIN = 1000
    fut, l = m._sketch_ahead(X, Y, "time")
    assert l == 13 and fut.result().shape == (2100, 13)
    assert xe.cross.MCA(n_modes=3, use_pca=True, random_state=4)._sketch_ahead(X, Y, "time") is None
    unseeded = xe.cross.MCA(n_modes=3, use_pca=False)
    unseeded._SKETCH_AHEAD_MIN = 1000
    assert unseeded._sketch_ahead(X, Y, "time") is None


@pytest.mark.parametrize("m,nev", [(1, 1), (2, 2), (7, 3), (60, 30), (150, 30), (240, 30), (256, 32), (96, 96)])
def test_host_hermitian_top_eigensolver(m, nev):
    """eofx_host_zheigh_top_f64 (the Rayleigh-Ritz step of the block-Krylov complex decomposition; Householder reduction to a
    real tridiagonal matrix + QL values + inverse iteration + back-transformation, csrc/eofx_hosteig.hpp) against numpy:
    values to 1e-12 of the norm, residual and orthonormality of the returned basis -- on a random Hermitian matrix, on
    one with exactly repeated and nearly repeated leading eigenvalues, and on a block-tridiagonal one with zero rows
    (dead Lanczos columns)."""
    from xeofs_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(m + nev)

    def run(A):
        Hr, Hi = np.ascontiguousarray(A.real), np.ascontiguousarray(A.imag)
        w = np.zeros(nev)
        Xr, Xi = np.zeros((m, nev)), np.zeros((m, nev))
        rc = lib.eofx_host_zheigh_top_f64(Hr.ctypes.data, Hi.ctypes.data, m, nev, w.ctypes.data, Xr.ctypes.data, Xi.ctypes.data)
        assert rc == 0
        X = Xr + 1j * Xi
        we = np.linalg.eigvalsh(A)[::-1][:nev]
        scale = max(np.abs(np.linalg.eigvalsh(A)).max(), 1e-300)
        assert np.abs(w - we).max() <= 1e-12 * scale
        assert np.abs(X.conj().T @ X - np.eye(nev)).max() <= 1e-11
        # the basis spans the leading invariant subspace: the projected matrix reproduces the values
        assert np.abs(np.linalg.eigvalsh(X.conj().T @ A @ X)[::-1] - we).max() <= 1e-10 * scale
        return w, X

    A = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
    A = A + A.conj().T
    w, X = run(A)
    assert np.abs(A @ X - X * w).max() <= 1e-10 * np.abs(w).max()
    if m >= 60:
        Q, _ = np.linalg.qr(A)
        lam = np.concatenate([np.full(5, 9.0), np.full(5, 9.0 - 1e-13), np.linspace(8, -3, m - 10)])
        run((Q * lam) @ Q.conj().T)
        # block tridiagonal with dead columns (zero rows / columns)
        b = 30
        T = np.zeros((m, m), complex)
        for i in range(0, m, b):
            D = rng.standard_normal((min(b, m - i),) * 2) + 1j * rng.standard_normal((min(b, m - i),) * 2)
            T[i:i + b, i:i + b] = D @ D.conj().T
            if i + b < m:
                R = np.triu(rng.standard_normal((b, min(b, m - i - b))) + 1j * rng.standard_normal((b, min(b, m - i - b))))
                T[i:i + b, i + b:i + 2 * b] = R
                T[i + b:i + 2 * b, i:i + b] = R.conj().T
        dead = rng.choice(m, size=m // 10, replace=False)
        T[dead, :] = 0
        T[:, dead] = 0
        run(T)


@pytest.mark.parametrize("scale,accepted", [(1e-8, True), (1e-6, True), (1e-3, False)])
def test_near_diagonal_matrix_powers(scale, accepted):
    """The whitener of PC scores without an eigen-decomposition (cross/cpcca.py near_diagonal_powers) against the
    eigen-decomposition route (linalg/_numpy/_utils.py:6-33), with repeated and nearly repeated diagonal entries; a
    covariance that is not nearly diagonal is refused."""
    import torch

    from xeofs_amd.cross.cpcca import fractional_matrix_power, near_diagonal_powers

    rng = np.random.default_rng(0)
    m = 300
    d = np.sort(rng.uniform(0.5, 2000.0, m))[::-1].copy()
    d[10] = d[9] * (1 + 1e-9)
    d[20] = d[19]
    E = rng.standard_normal((m, m))
    E = 0.5 * (E + E.T)
    np.fill_diagonal(E, 0.0)
    C = np.diag(d) + scale * E * np.sqrt(np.outer(d, d))
    powers = [-0.5, 0.5, -0.25, 0.0 - 0.35]
    out = near_diagonal_powers(torch.as_tensor(C), powers)
    if not accepted:
        assert out is None
        return
    for pw, T in zip(powers, out):
        ref = fractional_matrix_power(C, pw)
        assert np.abs(T.numpy() - ref).max() <= 1e-9 * np.abs(ref).max(), pw
    assert np.abs(out[0].numpy() @ out[1].numpy() - np.eye(m)).max() < 1e-9       # T Tinv = I (whitener.py:117-123)
    # refused: a non-positive or tiny diagonal entry (the reference's `s > eps` cut would act), non-finite input
    Cz = C.copy()
    Cz[5, 5] = 1e-13
    assert near_diagonal_powers(torch.as_tensor(Cz), powers) is None
    Cz[5, 5] = np.nan
    assert near_diagonal_powers(torch.as_tensor(Cz), powers) is None
