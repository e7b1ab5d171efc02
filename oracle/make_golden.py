"""Generate tests/golden/*.npz: outputs of the REAL third-party solvers the reference calls
(scikit-learn randomized_svd, scipy svds(lobpcg), scipy.signal.hilbert, LAPACK svd), run in the
build container on inputs built from the reference's own test recipes.  The reference package
itself cannot be imported here (no xarray/dask), see oracle/eof_oracle.py header.

Run:  python oracle/make_golden.py      (writes small .npz fixtures, < 1 MB total)
Versions at generation time are stored in every file.
"""
import os
import sys

import numpy as np
import scipy
import sklearn
from scipy.signal import hilbert
from scipy.sparse.linalg import svds
from sklearn.utils.extmath import randomized_svd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
VERS = dict(numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__)


def mock_data_array():
    """reference tests/conftest.py:225-240 (values only)."""
    rng = np.random.default_rng(7)
    noise = rng.normal(5, 3, size=(25, 5, 4))
    signal = 2 * np.sin(np.linspace(0, 2 * np.pi, 25))[:, None, None]
    return signal + noise


def mock_complex():
    """reference tests/conftest.py:286-308 (values only)."""
    x = np.linspace(-5, 5, 128)
    t = np.linspace(0, 4 * np.pi, 256)
    f1 = 1.0 / np.cosh(x[None, :] + 3) * np.exp(2.3j * t[:, None])
    f2 = 2.0 / np.cosh(x[None, :]) * np.tanh(x) * np.exp(2.8j * t[:, None])
    return f1 + f2


def lowrank(n, p, r, seed, dtype):
    rng = np.random.default_rng(seed)
    amp = 5.0 * 0.8 ** np.arange(r)
    X = (rng.standard_normal((n, r)) * amp) @ rng.standard_normal((r, p)) + rng.standard_normal((n, p))
    return (X - X.mean(axis=0)).astype(dtype)


def main():
    os.makedirs(OUT, exist_ok=True)
    # G1: reference mock_data_array (25 x 20), exact LAPACK SVD + sklearn rSVD k=2 seed 42
    X = mock_data_array().reshape(25, 20)
    Xc = X - X.mean(axis=0)
    U, s, Vt = np.linalg.svd(Xc, full_matrices=False)
    Ur, sr, Vtr = randomized_svd(Xc, n_components=2, random_state=42)
    np.savez_compressed(os.path.join(OUT, "g1_mock_data_array.npz"), X=X, Xc=Xc, U=U, s=s, Vt=Vt,
                        rs_U=Ur, rs_s=sr, rs_Vt=Vtr, rs_seed=42, **VERS)
    # G3/G4: sklearn randomized_svd on low-rank + noise matrices: both n_iter branches, both orientations
    cases = [("g4_rsvd_wide_f32_iter7", 96, 700, 6, 3, np.float32, 0),
             ("g4_rsvd_tall_f64_iter7", 600, 90, 5, 5, np.float64, 1),
             ("g4_rsvd_wide_f32_iter4", 64, 500, 12, 42, np.float32, 2)]
    for name, n, p, k, seed, dt, ds in cases:
        X = lowrank(n, p, 8, ds, dt)
        U, s, Vt = randomized_svd(X, n_components=k, random_state=seed)
        se = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:k]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, k=k, seed=seed, U=U, s=s, Vt=Vt,
                            s_exact=se, **VERS)
    # G5: MCA: two 120 x (6x8)/(5x9) fields sharing time series; C and sklearn rSVD of C, k=4
    rng = np.random.default_rng(11)
    T = rng.standard_normal((120, 5)) * (3.0 * 0.7 ** np.arange(5))
    X = T @ rng.standard_normal((5, 48)) + 0.5 * rng.standard_normal((120, 48))
    Y = T @ rng.standard_normal((5, 45)) + 0.5 * rng.standard_normal((120, 45))
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    C = Xc.T @ Yc / (120 - 1)
    U, s, Vt = randomized_svd(C, n_components=4, random_state=7)
    np.savez_compressed(os.path.join(OUT, "g5_mca.npz"), X=X, Y=Y, C=C, U=U, s=s, Vt=Vt, seed=7,
                        tsc=(np.abs(C) ** 2).sum(), cov2=(np.cov(Xc.T, Yc.T)[:48, 48:] ** 2).sum(), **VERS)
    # G6: complex: reference mock_complex_data_array through scipy svds(lobpcg) (decomposer.py:149-160)
    Z = mock_complex()
    Uc, sc, Vtc = svds(Z, k=3, solver="lobpcg", random_state=5)
    idx = np.argsort(sc)[::-1]
    se = np.linalg.svd(Z, compute_uv=False)[:3]
    np.savez_compressed(os.path.join(OUT, "g6_complex_svds.npz"), Z=Z, U=Uc[:, idx], s=sc[idx], Vt=Vtc[idx],
                        s_exact=se, seed=5, **VERS)
    # G6b: Hilbert transform: scipy.signal.hilbert on even/odd lengths
    y1 = np.random.default_rng(3).standard_normal((64, 5))
    y2 = np.random.default_rng(4).standard_normal((51, 4))
    np.savez_compressed(os.path.join(OUT, "g6_hilbert.npz"), y_even=y1, h_even=hilbert(y1, axis=0), y_odd=y2,
                        h_odd=hilbert(y2, axis=0), **VERS)
    # G2: the reference's NaN fixtures (tests/conftest.py:265-278: "isolated" / "boundary" / full-dimensional
    # masks on mock_data_array) through numpy's own nan-aware reductions: what Scaler + Sanitizer must produce
    base = mock_data_array()
    full = base.copy()
    full[:, 0, 0] = np.nan          # a grid point missing at all times (full-dimensional along time)
    full[3, :, :] = np.nan          # a time step missing everywhere
    fl = full.reshape(25, 20)
    vf, vs = ~np.isnan(fl).all(axis=0), ~np.isnan(fl).all(axis=1)
    mean = np.nanmean(fl, axis=0)
    std = np.nanstd(fl, axis=0)
    comp = (fl - mean)[np.ix_(vs, vf)]
    isolated = base.copy()
    isolated[5, 2, 1] = np.nan      # a single missing value: the reference raises (sanitizer.py:115-122)
    np.savez_compressed(os.path.join(OUT, "g2_nan_patterns.npz"), full=full, valid_feature=vf, valid_sample=vs,
                        mean=mean, std=std, compact_centered=comp, total_variance=np.var(comp, axis=0, ddof=1).sum(),
                        s_exact=np.linalg.svd(comp, compute_uv=False), isolated=isolated, **VERS)
    # G3: config-1 shape 2920 x (25 x 53), decaying spectrum, k = 10, seeds {0, 5, 42}: scikit-learn outputs in
    # float64 and float32 (only s and a few rows/columns of U / Vt are stored to keep the fixture small)
    X3 = lowrank(2920, 25 * 53, 20, 9, np.float64)
    g3 = dict(n=2920, p=1325, rank=20, data_seed=9, k=10)
    for seed in (0, 5, 42):
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            U, s, Vt = randomized_svd(X3.astype(dt), n_components=10, random_state=seed)
            g3[f"s_{tag}_{seed}"] = s
            g3[f"U_head_{tag}_{seed}"] = U[:8]
            g3[f"Vt_head_{tag}_{seed}"] = Vt[:, :16]
    g3["s_exact"] = np.linalg.svd(X3, compute_uv=False)[:10]
    np.savez_compressed(os.path.join(OUT, "g3_config1_shape.npz"), **g3, **VERS)
    # G5b: the cross models' default route (use_pca=True, n_pca_modes=0.999, init_rank_reduction=0.3):
    # `_SVD` = scikit-learn randomized_svd with int(0.3 * rank) components, then the variance truncation
    # (linalg/_numpy/_svd.py:89-106, 215-241), here with the real solver on the G5 fields
    g5 = {}
    for nm, Zc in (("x", Xc), ("y", Yc)):
        n_pre = int(min(Zc.shape) * 0.3)
        U, s, Vt = randomized_svd(Zc, n_components=n_pre, random_state=3)
        cum = np.cumsum(s ** 2 / (Zc.shape[0] - 1) / np.var(Zc, axis=0, ddof=1).sum())
        g5[f"n_pre_{nm}"], g5[f"s_{nm}"], g5[f"cum_{nm}"] = n_pre, s, cum
        g5[f"n_keep_{nm}"] = min(n_pre, n_pre - int((cum >= 0.999).sum()) + 1)
        g5[f"absVt_{nm}"] = np.abs(Vt)
    np.savez_compressed(os.path.join(OUT, "g5_pca_route.npz"), pca_seed=3, **g5, **VERS)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote", sorted(os.listdir(OUT)), f"{tot/1024:.0f} kB")


if __name__ == "__main__":
    sys.exit(main())
