"""CPU oracle: numpy/scipy restatement of the xeofs hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product (`xeofs_amd/`).

Every function cites the reference file:line (relative to /root/reference) it
follows.  The reference (xeofs 3.0.4) is pure Python and its arithmetic lives
in third-party solvers that are NOT under /root/reference:

  * scikit-learn ``sklearn.utils.extmath.randomized_svd`` (pin >=1.0.2 in
    pyproject.toml:17; 1.7.2 installed here) -- restated in `randomized_svd`
    below from its published algorithm (Halko et al. 2009, as implemented in
    sklearn/utils/extmath.py `_randomized_range_finder` / `_randomized_svd`).
  * ``scipy.sparse.linalg.svds(solver="lobpcg")`` (scipy 1.15.3 here) -- called,
    not restated (`complex_svds`), it is the solver itself.
  * ``scipy.signal.hilbert`` -- restated with numpy FFT in `analytic_signal`.

Pinning status.  xeofs itself cannot be imported in the build container (no
xarray / dask), and the reference's own tests hold no golden numeric vectors
(only invariants: shapes, determinism, round trips).  The
restatement is therefore pinned (tests/test_oracle.py, tests/golden/*.npz made
by oracle/make_golden.py) against
  (a) the real third-party solvers run in the build container: bit-for-bit
      against sklearn's `randomized_svd`, to rounding against scipy `hilbert`,
      `svds` and exact LAPACK SVDs;
  (b) the output of the real `dask.array.linalg.svd_compressed` (tests/golden/g7, generated with the image's
      conda interpreter by oracle/make_golden_dask.py) for the dask branch;
  (c) every invariant the reference's tests state for this path.
The thin xeofs wrapper semantics (policy, sign rule, scaling, NaN handling) are
restated from source and are "parity unpinned" by reference-run outputs.
"""

from __future__ import annotations

import warnings

import numpy as np
import scipy.linalg as sla

FLOAT32_EPS = float(np.finfo(np.float32).eps)


# --------------------------------------------------------------------------- #
# R1/R2  Scaler                                                                #
# --------------------------------------------------------------------------- #
def sqrt_cos_lat_weights(lat_deg):
    """xeofs/utils/xarray_utils.py:256-270 `_np_sqrt_cos_lat_weights`."""
    return np.sqrt(np.cos(np.deg2rad(np.asarray(lat_deg, dtype=float))).clip(0, 1))


def scaler_fit(X, with_center=True, with_std=False):
    """xeofs/preprocessing/scaler.py:69-126 on the stacked (sample, feature) view.

    xarray float reductions default to skipna=True -> nanmean / nanstd(ddof=0);
    std is clipped below at float32 eps (:106-108).  Statistics keep the input
    dtype (float32 in, float32 out), as numpy does.
    """
    X = np.asarray(X)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        mean = np.nanmean(X, axis=0) if with_center else None
        std = None
        if with_std:
            std = np.clip(np.nanstd(X, axis=0), FLOAT32_EPS, None)
    return mean, std


def scaler_transform(X, mean=None, std=None, feature_weights=None):
    """xeofs/preprocessing/scaler.py:128-154.

    `weights_` is float64 ones when the user gives none
    (xeofs/utils/xarray_utils.py:78-100), so the result is always float64 (H2).
    `feature_weights` is the product coslat*weights broadcast to the feature axis.
    """
    X = np.asarray(X)
    if mean is not None:
        X = X - mean
    if std is not None:
        X = X / std
    w = np.ones(X.shape[1], dtype=np.float64) if feature_weights is None else np.asarray(feature_weights, dtype=np.float64)
    return X * w


# --------------------------------------------------------------------------- #
# R4  Sanitizer                                                                #
# --------------------------------------------------------------------------- #
def sanitizer_fit_transform(Xs, check_nans=True):
    """xeofs/preprocessing/sanitizer.py:46-126 (fit + transform on the same data).

    Returns (X_compact, valid_feature[P], valid_sample[n]).
    Raises ValueError("Input data contains partial NaN entries ...") like :115-122.
    """
    notnull = ~np.isnan(Xs)
    valid_feature = notnull.any(axis=0)
    valid_sample = notnull.any(axis=1)
    if not check_nans:
        return Xs, valid_feature, valid_sample
    per_sample = notnull.sum(axis=1)
    n_valid = int(valid_feature.sum())
    if (~np.isin(per_sample, [0, n_valid])).any():
        raise ValueError(
            "Input data contains partial NaN entries, which will cause the the SVD to fail."
        )
    return Xs[np.ix_(valid_sample, valid_feature)], valid_feature, valid_sample


def sanitizer_transform(Xs, valid_feature_fit, check_nans=True):
    """xeofs/preprocessing/sanitizer.py:80-126 on new data with a fitted mask."""
    notnull = ~np.isnan(Xs)
    valid_feature = notnull.any(axis=0)
    valid_sample = notnull.any(axis=1)
    if not check_nans:
        return Xs, valid_sample
    if not np.array_equal(valid_feature, valid_feature_fit):
        raise ValueError(
            "Input data had NaN features in different locations than the original data."
        )
    per_sample = notnull.sum(axis=1)
    if (~np.isin(per_sample, [0, int(valid_feature.sum())])).any():
        raise ValueError(
            "Input data contains partial NaN entries, which will cause the the SVD to fail."
        )
    return Xs[np.ix_(valid_sample, valid_feature)], valid_sample


def preprocess(X, center=True, standardize=False, feature_weights=None, check_nans=True):
    """Scaler -> Stacker(reshape, done by caller) -> Sanitizer, R1-R5.

    X: (n, P) stacked raw field (may hold NaN).  Returns dict with the float64
    compacted matrix and every fitted quantity.
    """
    mean, std = scaler_fit(X, center, standardize)
    Xs = scaler_transform(X, mean, std, feature_weights)
    Xc, vf, vs = sanitizer_fit_transform(Xs, check_nans)
    return dict(X=Xc, mean=mean, std=std, valid_feature=vf, valid_sample=vs)


# --------------------------------------------------------------------------- #
# R6  total variance                                                           #
# --------------------------------------------------------------------------- #
def total_variance(X):
    """xeofs/utils/xarray_utils.py:236-253: `data.var(dim, ddof=1).sum()`."""
    return np.var(X, axis=0, ddof=1).sum()


# --------------------------------------------------------------------------- #
# R8  randomized SVD (scikit-learn algorithm)                                  #
# --------------------------------------------------------------------------- #
def svd_flip(u, v, u_based_decision=True):
    """sklearn/utils/extmath.py `svd_flip` (sign so that largest-|.| entry is +)."""
    if u_based_decision:
        max_abs = np.argmax(np.abs(u.T), axis=1)
        signs = np.sign(u[max_abs, np.arange(u.shape[1])])
    else:
        max_abs = np.argmax(np.abs(v), axis=1)
        signs = np.sign(v[np.arange(v.shape[0]), max_abs])
    u = u * signs[np.newaxis, :]
    v = v * signs[:, np.newaxis]
    return u, v


def sketch_matrix(n_rows, size, random_state, dtype):
    """The Gaussian test matrix exactly as sklearn draws it
    (extmath.py `_randomized_range_finder`: `random_state.normal(size=(A.shape[1], size))`
    from a legacy `np.random.RandomState`, cast to the data dtype)."""
    if isinstance(random_state, np.random.RandomState):
        rs = random_state
    else:
        rs = np.random.RandomState(random_state)
    Q = rs.normal(size=(n_rows, size))
    if np.issubdtype(dtype, np.floating):
        Q = Q.astype(dtype, copy=False)
    return Q


def rsvd_n_iter(n_components, shape):
    """extmath.py `_randomized_svd`: n_iter = 7 if k < 0.1*min(shape) else 4."""
    return 7 if n_components < 0.1 * min(shape) else 4


def randomized_svd(M, n_components, n_oversamples=10, n_iter="auto",
                   power_iteration_normalizer="auto", transpose="auto",
                   flip_sign=True, random_state=None):
    """Restatement of sklearn.utils.extmath.randomized_svd (the callable the
    reference hands to apply_ufunc at xeofs/linalg/decomposer.py:141-146).

    Returns (U[n,k], s[k], Vt[k,p]) in M's dtype.  Checked bit-for-bit against
    scikit-learn 1.7.2 in tests/test_oracle.py.
    """
    M = np.asarray(M)
    n_random = n_components + n_oversamples
    n_samples, n_features = M.shape
    if n_iter == "auto":
        n_iter = rsvd_n_iter(n_components, M.shape)
    if transpose == "auto":
        transpose = n_samples < n_features
    if transpose:
        M = M.T
    Q = sketch_matrix(M.shape[1], n_random, random_state, M.dtype)
    if power_iteration_normalizer == "auto":
        power_iteration_normalizer = "none" if n_iter <= 2 else "LU"
    if power_iteration_normalizer == "QR":
        normalizer = lambda x: sla.qr(x, mode="economic", check_finite=False)
    elif power_iteration_normalizer == "LU":
        normalizer = lambda x: sla.lu(x, permute_l=True, check_finite=False)
    else:
        normalizer = lambda x: (x, None)
    for _ in range(n_iter):
        Q, _ = normalizer(M @ Q)
        Q, _ = normalizer(M.T @ Q)
    Q, _ = sla.qr(M @ Q, mode="economic", check_finite=False)
    B = Q.T @ M
    Uhat, s, Vt = sla.svd(B, full_matrices=False, lapack_driver="gesdd")
    del B
    U = Q @ Uhat
    if flip_sign:
        U, Vt = svd_flip(U, Vt, u_based_decision=not transpose)
    if transpose:
        return Vt[:n_components, :].T, s[:n_components], U[:, :n_components].T
    return U[:, :n_components], s[:n_components], Vt[:n_components, :]


# --------------------------------------------------------------------------- #
# R9  complex branch                                                           #
# --------------------------------------------------------------------------- #
def complex_svds(X, k, random_state=None):
    """xeofs/linalg/decomposer.py:149-160: scipy svds(lobpcg) + descending sort."""
    from scipy.sparse.linalg import svds

    U, s, VT = svds(X, k=k, solver="lobpcg", random_state=random_state)
    idx = np.argsort(s)[::-1]
    return U[:, idx], s[idx], VT[idx, :]


# --------------------------------------------------------------------------- #
# R11  sign rule                                                               #
# --------------------------------------------------------------------------- #
def deterministic_sign_multiplier(VT):
    """xeofs/utils/xarray_utils.py:273-301.

    Per mode: m = max_f VT, mi = min_f VT (numpy complex max/min are
    lexicographic); +1 if |m| >= |mi| (idxmax over coords [1, -1] returns the
    first maximum on ties) else -1.
    """
    m = VT.max(axis=1)
    mi = VT.min(axis=1)
    return np.where(np.abs(m) >= np.abs(mi), 1, -1)


# --------------------------------------------------------------------------- #
# R7/R11  Decomposer                                                           #
# --------------------------------------------------------------------------- #
def decomposer_fit(X, n_modes, init_rank_reduction=0.3, flip_signs=True,
                   solver="auto", random_state=None, solver_kwargs=None):
    """xeofs/linalg/decomposer.py:76-226 on a plain (sample, feature) ndarray.

    Returns (U[n,k], s[k], V[p,k]) with V = conj(VT).T (:226).
    """
    solver_kwargs = dict(solver_kwargs or {})
    X = np.asarray(X)
    is_based_on_variance = not isinstance(n_modes, (int, np.integer))
    if is_based_on_variance and not (0 < init_rank_reduction <= 1.0):
        raise ValueError("init_rank_reduction must be in the half open interval (0, 1].")
    rank = min(X.shape)
    n_pre = n_modes
    if is_based_on_variance:
        n_pre = int(rank * init_rank_reduction)
        if n_pre < 1:
            warnings.warn(
                f"`init_rank_reduction={init_rank_reduction}` is too low resulting in zero components. One component will be computed instead."
            )
            n_pre = 1
    if n_pre > rank:
        raise ValueError(
            f"n_modes must be less than or equal to the rank of the dataset (rank = {rank})."
        )
    use_complex = np.iscomplexobj(X)
    is_small = max(X.shape) < 500
    if solver == "auto":
        use_exact = bool(is_small and n_pre > int(0.8 * rank))
    elif solver == "full":
        use_exact = True
    elif solver == "randomized":
        use_exact = False
    else:
        raise ValueError(
            f"Unrecognized solver '{solver}'. Valid options are 'auto', 'full', and 'randomized'."
        )
    if use_exact:
        U, s, VT = np.linalg.svd(X, **solver_kwargs)  # full_matrices=True default, then sliced
        U, s, VT = U[:, :n_pre], s[:n_pre], VT[:n_pre, :]
    elif not use_complex:
        U, s, VT = randomized_svd(X, n_components=n_pre, random_state=random_state, **solver_kwargs)
    else:
        U, s, VT = complex_svds(X, n_pre, random_state=random_state)

    if is_based_on_variance:
        N = X.shape[0] - 1
        tv = np.var(X, axis=0, ddof=1).sum()
        cum = np.cumsum(s ** 2 / N / tv)
        n_req = n_pre - int((cum >= n_modes).sum()) + 1
        if n_req > n_pre:
            warnings.warn(
                f"Dataset has {n_pre} components, explaining {cum[-1]:.2%} of the variance. However, {n_modes:.2%} explained variance was requested. Please consider increasing `init_rank_reduction`."
            )
            n_req = n_pre
        U, s, VT = U[:, :n_req], s[:n_req], VT[:n_req, :]
    if flip_signs:
        sgn = deterministic_sign_multiplier(VT)
        VT = VT * sgn[:, None]
        U = U * sgn[None, :]
    return U, s, VT.conj().T


# --------------------------------------------------------------------------- #
# R12/R13  EOF model                                                           #
# --------------------------------------------------------------------------- #
def eof_fit(X, n_modes, center=True, standardize=False, feature_weights=None,
            random_state=None, solver="auto", solver_kwargs=None, check_nans=True):
    """xeofs/single/base_model_single_set.py:123-161 + xeofs/single/eof.py:85-118.

    X: raw stacked (n, P).  Returns dict keyed like the reference DataContainer
    (eof.py:110-115) plus the fitted preprocessing state.
    """
    pre = preprocess(X, center, standardize, feature_weights, check_nans)
    Xc = pre["X"]
    tv = total_variance(Xc)
    U, s, V = decomposer_fit(Xc, n_modes, random_state=random_state, solver=solver,
                             solver_kwargs=solver_kwargs)
    n = Xc.shape[0]
    return dict(
        input_data=Xc, components=V, scores=U * s, norms=s,
        explained_variance=s ** 2 / (n - 1), total_variance=tv,
        explained_variance_ratio=s ** 2 / (n - 1) / tv,
        U=U, **{k: pre[k] for k in ("mean", "std", "valid_feature", "valid_sample")},
    )


def eof_transform(Xc_new, components, norms=None, normalized=False):
    """xeofs/single/eof.py:123-132 (+ base_model_single_set.py:180-203 /norms)."""
    proj = Xc_new @ components
    if normalized:
        proj = proj / norms
    return proj


def eof_inverse_transform(scores, components):
    """xeofs/single/eof.py:134-156: xr.dot(comps.conj(), scores, dims='mode')."""
    return scores @ components.conj().T


# --------------------------------------------------------------------------- #
# R14/R15  cross-covariance + MCA                                              #
# --------------------------------------------------------------------------- #
def cross_covariance(X, Y):
    """xeofs/cross/cpcca.py:1007-1015 `_compute_cross_covariance_numpy`."""
    if X.shape[0] != Y.shape[0]:
        raise ValueError(
            f"Both data matrices must have the same number of samples but found {X.shape[0]} in the first and {Y.shape[0]} in the second."
        )
    return X.conj().T @ Y / (X.shape[0] - 1)


def pca_fit(Xc, n_modes=0.999, init_rank_reduction=0.3, random_state=None):
    """xeofs/preprocessing/pca.py:94-123: `SVD(n_modes, init_rank_reduction, random_state).fit_transform`
    (xeofs/linalg/_numpy/_svd.py:108-241 -- the same solver policy, sign rule and variance truncation as
    the Decomposer, restated in `decomposer_fit`).  The cross models build their PCA without a
    random_state (cross/base_model_cross_set.py:165-179), i.e. the reference's own result is only
    reproducible to the convergence of the randomized solver.  Returns (scores = X V, V, s)."""
    if isinstance(n_modes, str):
        if n_modes != "all":
            raise ValueError("`n_modes` must be an integer, float or 'all'")
        n_modes = min(Xc.shape)
    U, s, V = decomposer_fit(Xc, n_modes, init_rank_reduction=init_rank_reduction, random_state=random_state)
    return Xc @ V, V, s


def mca_fit(X, Y, n_modes, standardize=False, feature_weights_x=None, feature_weights_y=None,
            random_state=None, solver="auto", solver_kwargs=None, check_nans=True, use_pca=False,
            n_pca_modes=0.999, pca_init_rank_reduction=0.3, pca_random_state=None):
    """xeofs/cross/base_model_cross_set.py:269-321 with alpha=1 (MCA, cross/mca.py:107) +
    xeofs/cross/cpcca.py:168-225.  CPCCA always centres (cpcca.py:145).  With `use_pca` the
    analysis runs on the PC scores and the singular vectors are projected back
    (preprocessing/pca.py:125-168)."""
    px = preprocess(X, True, standardize, feature_weights_x, check_nans)
    py = preprocess(Y, True, standardize, feature_weights_y, check_nans)
    Xc, Yc = px["X"], py["X"]
    V1 = V2 = None
    if use_pca:
        Xc, V1, _ = pca_fit(Xc, n_pca_modes, pca_init_rank_reduction, pca_random_state)
        Yc, V2, _ = pca_fit(Yc, n_pca_modes, pca_init_rank_reduction, pca_random_state)
    C = cross_covariance(Xc, Yc)
    Q1, s, Q2 = decomposer_fit(C, n_modes, random_state=random_state, solver=solver,
                               solver_kwargs=solver_kwargs)
    tsc = (np.abs(C) ** 2).sum()
    scores1 = Xc @ Q1
    scores2 = Yc @ Q2
    norm1 = np.sqrt((scores1.conj() * scores1).sum(axis=0)).real
    norm2 = np.sqrt((scores2.conj() * scores2).sum(axis=0)).real
    return dict(
        input_data1=Xc, input_data2=Yc, components1=Q1 if V1 is None else V1 @ Q1,
        components2=Q2 if V2 is None else V2 @ Q2,
        scores1=scores1, scores2=scores2, singular_values=s, squared_covariance=s ** 2,
        total_squared_covariance=tsc, idx_modes_sorted=np.argsort(s)[::-1],
        norm1=norm1, norm2=norm2, C=C, pre_x=px, pre_y=py, pca_modes=(Xc.shape[1], Yc.shape[1]),
    )


# --------------------------------------------------------------------------- #
# R16  Hilbert transform                                                       #
# --------------------------------------------------------------------------- #
def analytic_signal(y):
    """scipy.signal.hilbert(y, axis=0): FFT, zero negative freqs, double positive."""
    N = y.shape[0]
    Yf = np.fft.fft(y, axis=0)
    h = np.zeros(N, dtype=Yf.real.dtype)
    if N % 2 == 0:
        h[0] = h[N // 2] = 1
        h[1:N // 2] = 2
    else:
        h[0] = 1
        h[1:(N + 1) // 2] = 2
    return np.fft.ifft(Yf * h[:, None], axis=0)


def pad_exp(y, decay_factor=0.2):
    """xeofs/utils/hilbert_transform.py:75-114 `_pad_exp`."""
    n = y.shape[0]
    x = np.arange(n)
    x_ext = np.arange(-n, 2 * n)
    coefs = np.polynomial.polynomial.polyfit(x, y, deg=1)
    yfit = np.polynomial.polynomial.polyval(x, coefs).T
    yfit_ext = np.polynomial.polynomial.polyval(x_ext, coefs).T
    y_ano = y - yfit
    amp_pre = y_ano[0][:, None]
    amp_pos = y_ano[-1][:, None]
    exp_ext = np.exp(-x / n / decay_factor)
    pad_pre = amp_pre * exp_ext[::-1]
    pad_pos = amp_pos * exp_ext
    y_ext = np.concatenate([pad_pre.T, y_ano, pad_pos.T], axis=0)
    return y_ext + yfit_ext


def hilbert_transform(y, padding="exp", decay_factor=0.2):
    """xeofs/utils/hilbert_transform.py:40-72 `_hilbert_transform_with_padding`."""
    n = y.shape[0]
    if padding == "exp":
        y = pad_exp(y, decay_factor)
    y = analytic_signal(y)
    if padding == "exp":
        y = y[n:2 * n]
    return y - y.mean(axis=0)


# --------------------------------------------------------------------------- #
# synthetic field of SURVEY.md §8d (shared by tests and bench)                 #
# --------------------------------------------------------------------------- #
def synthetic_field(n, n_lat, n_lon, rank=100, seed=0, dtype=np.float32, nan_frac=0.0):
    """Low-rank + noise field with a decaying spectrum."""
    p = n_lat * n_lon
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p)).astype(dtype)
    lat = np.linspace(-89.75, 89.75, n_lat)
    for j in range(rank):
        rj = np.random.default_rng(1000 + j)
        e = rj.standard_normal(n)
        t = np.empty(n)
        t[0] = e[0]
        for i in range(1, n):
            t[i] = 0.8 * t[i - 1] + 0.6 * e[i]
        t = (t - t.mean()) / t.std()
        gj = np.random.default_rng(2000 + j)
        # smooth spatial pattern: sum of a few random low-wavenumber harmonics
        yy, xx = np.meshgrid(np.linspace(0, np.pi, n_lat), np.linspace(0, 2 * np.pi, n_lon, endpoint=False), indexing="ij")
        g = np.zeros((n_lat, n_lon))
        for _ in range(4):
            ky, kx = gj.integers(1, 6, size=2)
            g += gj.standard_normal() * np.sin(ky * yy + gj.uniform(0, 2 * np.pi)) * np.cos(kx * xx + gj.uniform(0, 2 * np.pi))
        g = (g / np.linalg.norm(g)).ravel()
        X += (10.0 * 0.93 ** j) * np.outer(t, g).astype(dtype)
    X += (15.0 + 10.0 * np.cos(np.deg2rad(lat)))[None, :, None].repeat(n_lon, axis=2).reshape(1, p).astype(dtype)
    if nan_frac > 0:
        mask = np.random.default_rng(seed + 77).random(p) < nan_frac
        X[:, mask] = np.nan
    return X, lat


# --------------------------------------------------------------------------- #
# N2  Varimax / Promax rotation + EOFRotator                                    #
# --------------------------------------------------------------------------- #
def varimax(X, gamma=1.0, max_iter=1000, rtol=1e-8):
    """xeofs/linalg/_numpy/_rotation.py:95-187 `_varimax` (eager, numpy branch)."""
    X = np.array(X, dtype=np.result_type(X, np.float64), copy=True)
    n_samples, n_modes = X.shape
    if n_modes < 2:
        raise ValueError("Cannot rotate {:} modes (columns), but must be 2 or more.".format(n_modes))
    R = np.eye(n_modes)
    h = np.sqrt(np.sum(X * X.conj(), axis=1))
    eps = np.finfo(X.dtype).eps
    X = (1.0 / (h + eps))[:, np.newaxis] * X
    delta = 0.0
    XH = X.conj().T
    alpha = gamma / n_samples
    for _ in range(max_iter):
        delta_old = delta
        basis = X @ R
        basis2 = basis * basis.conj()
        W = np.sum(basis2, axis=0)
        transformed = XH @ (basis * (basis2 - (alpha * W)))
        U, svals, VT = np.linalg.svd(transformed)
        R = U @ VT
        delta = np.sum(svals)
        if (abs(delta - delta_old) / delta) < rtol:
            break
    if (abs(delta - delta_old) / delta) > rtol:
        raise RuntimeError("Rotation process did not converge.")
    X = h[:, np.newaxis] * X
    return X @ R, R


def promax(X, power=1, max_iter=1000, rtol=1e-8):
    """xeofs/linalg/_numpy/_rotation.py:6-92 `_promax`."""
    X, rot_mat = varimax(X, max_iter=max_iter, rtol=rtol)
    h = np.sqrt(np.sum(X * X.conj(), axis=1))
    eps = np.finfo(X.dtype).eps
    X = (1.0 / (h + eps))[:, np.newaxis] * X
    Xnorm = X / np.max(abs(X), axis=0)
    P = Xnorm * np.abs(Xnorm) ** (power - 1)
    L = np.linalg.inv(X.conj().T @ X) @ X.conj().T @ P
    try:
        sigma_inv = np.diag(np.diag(np.linalg.inv(L.conj().T @ L)))
    except np.linalg.LinAlgError:
        sigma_inv = np.diag(np.diag(np.linalg.pinv(L.conj().T @ L)))
    L = L @ np.sqrt(sigma_inv)
    Xrot = h[:, np.newaxis] * (X @ L)
    rot_mat = rot_mat @ L
    L_inv = np.linalg.inv(L)
    return Xrot, rot_mat, L_inv @ L_inv.conj().T


def eof_rotator_fit(eof, n_modes, power=1, max_iter=1000, rtol=1e-8):
    """xeofs/single/eof_rotator.py:119-205 on the dict returned by `eof_fit`
    (sorted by explained variance as `_sort_by_variance` does after compute, :207-218)."""
    comps = eof["components"][:, :n_modes]
    expvar = eof["explained_variance"][:n_modes]
    loadings = comps * np.sqrt(expvar)
    rot_loadings, rot_matrix, phi = promax(loadings, power=power, max_iter=max_iter, rtol=rtol)
    expvar_r = (np.abs(rot_loadings) ** 2).sum(axis=0)
    idx = np.argsort(expvar_r)[::-1]
    rot_components = rot_loadings / np.sqrt(expvar_r)
    n_samples = eof["input_data"].shape[0]
    norms = (expvar_r * (n_samples - 1)) ** 0.5
    scores = eof["scores"][:, :n_modes] / eof["norms"][:n_modes]
    RinvT = rot_matrix if power == 1 else np.linalg.inv(rot_matrix).conj().T
    scores = scores @ RinvT * norms
    sign = deterministic_sign_multiplier(rot_components.T)
    rot_components = rot_components * sign
    scores = scores * sign
    return dict(components=rot_components[:, idx], scores=scores[:, idx], norms=norms[idx],
                explained_variance=expvar_r[idx], total_variance=eof["total_variance"], idx_modes_sorted=idx,
                rotation_matrix=rot_matrix, phi_matrix=phi, modes_sign=sign[idx])


# --------------------------------------------------------------------------- #
# N3  EOFBootstrapper                                                           #
# --------------------------------------------------------------------------- #
def eof_bootstrap(eof, n_modes, n_bootstraps=20, seed=None, random_state=None):
    """xeofs/validation/bootstrapper.py:54-135 on the dict returned by `eof_fit` (+ "input_data" = the
    preprocessed matrix).  Each member: rows drawn with `default_rng(seed).choice(n, n, replace=True)`
    (:79), `EOF(n_modes, standardize=False, use_coslat=False).fit` (centre again, :88-89), scores =
    member.transform(input_data) (:95); afterwards the per-mode sign from the correlation with the
    model's scores (:112-121).  `random_state` seeds the members' solvers (unseeded in the reference)."""
    X = eof["input_data"]
    n = X.shape[0]
    rng = np.random.default_rng(seed)
    expvar, totvar, comps, scores = [], [], [], []
    for b in range(n_bootstraps):
        idx = rng.choice(n, n, replace=True)
        member = eof_fit(X[idx], n_modes, random_state=None if random_state is None else random_state + b)
        expvar.append(member["explained_variance"])
        totvar.append(member["total_variance"])
        comps.append(member["components"])
        scores.append(eof_transform(X - member["mean"], member["components"]))
    expvar, totvar, comps, scores = map(np.asarray, (expvar, totvar, comps, scores))
    ms = eof["scores"][:, :n_modes]
    corr = (scores * ms).mean(axis=1) / scores.std(axis=1) / ms.std(axis=0)
    signs = np.sign(corr)
    return dict(components=comps * signs[:, None, :], scores=scores * signs[:, None, :], norms=eof["norms"],
                explained_variance=expvar, total_variance=totvar)


# --------------------------------------------------------------------------- #
# N4  Whitener + CPCCA family (CCA alpha=0, RDA alpha=[0,1], MCA alpha=1)       #
# --------------------------------------------------------------------------- #
def fractional_matrix_power(C, power):
    """xeofs/linalg/_numpy/_utils.py:6-33: SVD of the symmetric matrix (solver 'full' here,
    whitener.py:125), singular values <= eps dropped, V s^power V^H."""
    U, s, VT = np.linalg.svd(C)
    sgn = deterministic_sign_multiplier(VT)
    V = (VT * sgn[:, None]).conj().T
    keep = s > np.finfo(s.dtype).eps
    V, s = V[:, keep], s[keep]
    out = V @ np.diag(s ** power) @ V.conj().T
    return out if np.iscomplexobj(C) else out.real


def whitener_fit(X, alpha):
    """xeofs/preprocessing/whitener.py:86-133: T = (X^H X / n)^((alpha-1)/2), Tinv = inv(T) (pinv fallback);
    identity when alpha == 1 (:46-60)."""
    if np.isclose(alpha, 1.0):
        return None, None
    C = X.conj().T @ X / X.shape[0]
    T = fractional_matrix_power(C, (alpha - 1) / 2)
    try:
        Tinv = np.linalg.inv(T)
    except np.linalg.LinAlgError:
        Tinv = np.linalg.pinv(T)
    return T, Tinv


def cpcca_fit(X, Y, n_modes, alpha=(0.2, 0.2), standardize=False, random_state=None, solver="auto", use_pca=True,
              n_pca_modes=0.999, pca_init_rank_reduction=0.3, pca_random_state=None, pca_solver="auto", hilbert=None):
    """xeofs/cross/base_model_cross_set.py:269-321 + xeofs/cross/cpcca.py:168-225 for general alpha.
    Returns the DataContainer entries plus the fitted transforms (V_i, T_i, Tinv_i).
    Complex X, Y: ComplexCPCCA / ComplexMCA (cpcca.py:1023-1173; the Decomposer takes its complex branch).
    hilbert = (padding, decay_factor): HilbertCPCCA / HilbertMCA (cpcca.py:1328-1500): `_augment_data` between the PCA
    and the whitener (base_model_cross_set.py:307-313)."""
    alpha = [alpha, alpha] if np.isscalar(alpha) else list(alpha)
    px = preprocess(X, True, standardize)
    py = preprocess(Y, True, standardize)
    data, Vs, Ts, Tinvs = [], [], [], []
    for Z, a in ((px["X"], alpha[0]), (py["X"], alpha[1])):
        V = None
        if use_pca:
            nm = min(Z.shape) if n_pca_modes == "all" else n_pca_modes
            _, _, V = decomposer_fit(Z, nm, init_rank_reduction=pca_init_rank_reduction,
                                     random_state=pca_random_state, solver=pca_solver)
            Z = Z @ V
        if hilbert is not None:
            Z = hilbert_transform(Z, hilbert[0], hilbert[1])
        T, Tinv = whitener_fit(Z, a)
        if T is not None:
            Z = Z @ T
        data.append(Z); Vs.append(V); Ts.append(T); Tinvs.append(Tinv)
    Xw, Yw = data
    C = cross_covariance(Xw, Yw)
    Q1, s, Q2 = decomposer_fit(C, n_modes, random_state=random_state, solver=solver)
    Cu = C if Tinvs[1] is None else C @ Tinvs[1]                     # cpcca.py:991-1000
    Cu = Cu.conj().T if Tinvs[0] is None else Cu.conj().T @ Tinvs[0]
    tsc = (np.abs(Cu) ** 2).sum()
    scores1, scores2 = Xw @ Q1, Yw @ Q2
    norm1 = np.sqrt((scores1.conj() * scores1).sum(axis=0)).real
    norm2 = np.sqrt((scores2.conj() * scores2).sum(axis=0)).real

    def back(Q, i):      # whitener.inverse_transform_components, pca.inverse_transform_components
        Q = Q if Tinvs[i] is None else Tinvs[i].conj().T @ Q
        return Q if Vs[i] is None else Vs[i] @ Q
    return dict(input_data1=Xw, input_data2=Yw, Q1=Q1, Q2=Q2, components1=back(Q1, 0), components2=back(Q2, 1),
                scores1=scores1, scores2=scores2, singular_values=s, squared_covariance=s ** 2,
                total_squared_covariance=tsc, norm1=norm1, norm2=norm2, V=Vs, T=Ts, Tinv=Tinvs,
                pre_x=px, pre_y=py)


def _unwhiten(Z, Tinv):
    return Z if Tinv is None else Z @ Tinv


def cpcca_transform(m, Xc=None, Yc=None, normalized=False):
    """base_model_cross_set.py:323-374 + cpcca.py:227-252 on preprocessed (centred) new data."""
    out = []
    for i, Z in enumerate((Xc, Yc)):
        if Z is None:
            continue
        if m["V"][i] is not None:
            Z = Z @ m["V"][i]
        if m["T"][i] is not None:
            Z = Z @ m["T"][i]
        S = Z @ m[f"Q{i + 1}"]
        out.append(S / m[f"norm{i + 1}"] if normalized else S)
    return out[0] if len(out) == 1 else out


def cpcca_predict(m, Xc):
    """cpcca.py:273-302: pseudo scores of Y from new X."""
    Z = Xc if m["V"][0] is None else Xc @ m["V"][0]
    Z = Z if m["T"][0] is None else Z @ m["T"][0]
    Rx, Ry = m["scores1"], m["scores2"]
    G = Rx.conj().T @ Ry / np.linalg.norm(Rx, axis=0) ** 2
    return Z @ m["Q1"] @ G


def cpcca_inverse_transform(m, scores, which):
    """cpcca.py:254-271 + base_model_cross_set.py:376-425: back to the preprocessed feature space."""
    i = which - 1
    Z = scores @ m[f"Q{which}"][:, :scores.shape[1]].conj().T
    Z = _unwhiten(Z, m["Tinv"][i])
    return Z if m["V"][i] is None else Z @ m["V"][i].conj().T


def cpcca_diagnostics(m):
    """cpcca.py:330-640: cross / auto correlation coefficients of the scores, squared covariance
    fraction and the three fractions of variance explained (Swenson 2015, eq. 15)."""
    Rx, Ry = m["scores1"], m["scores2"]
    k = Rx.shape[1]
    nrm = lambda Z: Z / Z.std(axis=0)
    n = Rx.shape[0]
    out = dict(cross_correlation_coefficients=np.diag(nrm(Rx).conj().T @ nrm(Ry) / (n - 1)).real,
               correlation_coefficients_X=nrm(Rx).conj().T @ nrm(Rx) / (n - 1),
               correlation_coefficients_Y=nrm(Ry).conj().T @ nrm(Ry) / (n - 1))
    X1, X2 = _unwhiten(m["input_data1"], m["Tinv"][0]), _unwhiten(m["input_data2"], m["Tinv"][1])
    scf, fxx, fyy, fyx = [], [], [], []
    tvx, tvy = (np.abs(X1) ** 2).sum() / (n - 1), (np.abs(X2) ** 2).sum() / (n - 1)
    Cx = X1.conj().T @ X1 / (n - 1)
    Tm = fractional_matrix_power(Cx, -0.5)
    tv_yx = np.linalg.norm(Tm @ X1.conj().T @ X2 / (n - 1)) ** 2
    for j in range(k):
        X1r = _unwhiten(Rx[:, [j]] @ m["Q1"][:, [j]].conj().T, m["Tinv"][0])
        X2r = _unwhiten(Ry[:, [j]] @ m["Q2"][:, [j]].conj().T, m["Tinv"][1])
        dX, dY = X1 - X1r, X2 - X2r
        scf.append(1 - np.linalg.norm(dX.conj().T @ dY / (n - 1)) ** 2 / m["total_squared_covariance"])
        fxx.append(1 - ((np.abs(dX) ** 2).sum() / (n - 1)) / tvx)
        fyy.append(1 - ((np.abs(dY) ** 2).sum() / (n - 1)) / tvy)
        fyx.append(1 - np.linalg.norm(Tm @ dX.conj().T @ dY / (n - 1)) ** 2 / tv_yx)
    scf = np.asarray(scf)
    out.update(squared_covariance_fraction=np.where(scf < 0, 0, scf), fraction_variance_X_explained_by_X=np.asarray(fxx),
               fraction_variance_Y_explained_by_Y=np.asarray(fyy), fraction_variance_Y_explained_by_X=np.asarray(fyx))
    return out


def pearson_patterns(data, scores):
    """xeofs/utils/optional/statistics.py:9-104 without multiple-test correction: correlation of every
    feature with every score series (centred data assumed, population std) and two-sided p-values from the
    beta distribution (scipy.stats.pearsonr reference)."""
    import scipy.stats as st

    n = data.shape[0]
    corr = (data / data.std(0)).conj().T @ (scores / scores.std(0)) / n
    a = n / 2 - 1
    return corr, 2 * st.beta(a, a, loc=-1, scale=2).cdf(-np.abs(corr))


def holm_sidak(p):
    """What `correction=<any valid name>` does in the reference (utils/optional/statistics.py:108-157): every mode's
    p-values go to statsmodels.stats.multitest.multipletests with that function's DEFAULTS (the wrapper forwards neither
    `method` nor `alpha`), i.e. Holm-Sidak step-down.  statsmodels (0.14; absent from both images, so this function is
    a restatement of its published algorithm and its parity is unpinned): sort; 1 - (1 - p_(i))^(m - i); running
    maximum; clip at 1; unsort.  Plain loops on purpose -- the product code is vectorised."""
    p = np.asarray(p, dtype=np.float64)
    out = np.empty_like(p)
    m = p.shape[0]
    for j in range(p.shape[1]):
        order = np.argsort(p[:, j], kind="stable")
        run = 0.0
        for rank, i in enumerate(order):
            raw = 1.0 - (1.0 - p[i, j]) ** (m - rank)
            run = max(run, raw)
            out[i, j] = min(run, 1.0)
    return out


def cpcca_patterns(m, kind="homogeneous"):
    """cpcca.py:642-845: data are taken back through whitener and PCA (i.e. the PCA-truncated fields)."""
    fields = []
    for i in range(2):
        Z = _unwhiten(m[f"input_data{i + 1}"], m["Tinv"][i])
        fields.append(Z if m["V"][i] is None else Z @ m["V"][i].conj().T)
    s1, s2 = (m["scores1"], m["scores2"]) if kind == "homogeneous" else (m["scores2"], m["scores1"])
    return pearson_patterns(fields[0], s1), pearson_patterns(fields[1], s2)


def cpcca_rotator_fit(m, n_modes, power=1, max_iter=1000, rtol=1e-8):
    """xeofs/cross/cpcca_rotator.py:122-263 on the dict returned by `cpcca_fit`: Varimax/Promax of the
    stacked feature-space loadings [Qx; Qy] sqrt(s), taken back to the analysis space (pca / whitener
    transform_components), sorted by squared covariance (:265-280, after compute)."""
    k = n_modes
    s = m["singular_values"][:k]
    scaling = np.sqrt(s)
    Qx, Qy = m["components1"][:, :k], m["components2"][:, :k]            # already whitener^-1, pca^-1
    p1 = Qx.shape[0]
    loadings = np.concatenate([Qx, Qy], axis=0) * scaling
    rot_loadings, rot_matrix, phi = promax(loadings, power=power, max_iter=max_iter, rtol=rtol)
    out = []
    for i, blk in enumerate((rot_loadings[:p1], rot_loadings[p1:])):
        Q = blk if m["V"][i] is None else m["V"][i].conj().T @ blk       # pca.transform_components
        Q = Q if m["T"][i] is None else m["T"][i].conj().T @ Q           # whitener.transform_components
        out.append(Q)
    norm1, norm2 = np.linalg.norm(out[0], axis=0), np.linalg.norm(out[1], axis=0)
    Q1r, Q2r = out[0] / norm1, out[1] / norm2
    sqcov = (norm1 * norm2) ** 2
    idx = np.argsort(sqcov)[::-1]
    RinvT = rot_matrix if power == 1 else np.linalg.inv(rot_matrix).conj().T
    sc1 = (m["scores1"][:, :k] / scaling) @ RinvT * norm1
    sc2 = (m["scores2"][:, :k] / scaling) @ RinvT * norm2
    sign = deterministic_sign_multiplier(rot_loadings.T)

    def back(Q, i):
        Q = Q if m["Tinv"][i] is None else m["Tinv"][i].conj().T @ Q
        return Q if m["V"][i] is None else m["V"][i] @ Q
    return dict(Q1=(Q1r * sign)[:, idx], Q2=(Q2r * sign)[:, idx], components1=back(Q1r * sign, 0)[:, idx],
                components2=back(Q2r * sign, 1)[:, idx], scores1=(sc1 * sign)[:, idx], scores2=(sc2 * sign)[:, idx],
                squared_covariance=sqcov[idx], norm1=norm1[idx], norm2=norm2[idx], idx_modes_sorted=idx,
                rotation_matrix=rot_matrix, phi_matrix=phi, modes_sign=sign[idx],
                total_squared_covariance=m["total_squared_covariance"])


def cpcca_rotator_transform(m, rr, Zc, which, n_modes, power=1, normalized=False):
    """cpcca_rotator.py:282-372 on preprocessed new data `Zc`: projection on the *back-projected* model
    components (whitener^-1, pca^-1), / sqrt(s), rotation, sort, sign, norm.  (For alpha != 1 this is not
    the map that produced the fitted scores -- the reference projects on Tinv^H Q, not T Q.)"""
    k = n_modes
    scaling = np.sqrt(m["singular_values"][:k])
    R = rr["rotation_matrix"]
    RinvT = R if power == 1 else np.linalg.inv(R).conj().T
    proj = (Zc @ m[f"components{which}"][:, :k] / scaling) @ RinvT
    proj = proj[:, rr["idx_modes_sorted"]] * rr["modes_sign"]
    return proj if normalized else proj * rr[f"norm{which}"]


# --------------------------------------------------------------------------- #
# R10  dask branch: dask.array.linalg.svd_compressed                            #
# --------------------------------------------------------------------------- #
def svd_compressed(X, k, seed=None, n_power_iter=4, n_oversamples=10, min_subspace_size=20):
    """The algorithm of `dask.array.linalg.svd_compressed` (dask is an un-vendored dependency, pinned
    `dask>=2023.0.1` in pyproject.toml; called at xeofs/linalg/decomposer.py:163-171 with
    `n_power_iter=4`), restated from its published implementation (Halko et al. 2009, alg. 4.3/5.1):
    comp_level = min(max(min_subspace_size, k + n_oversamples), min(m, n)); Omega ~ N(0,1) (n x comp_level);
    Y = X Omega; n_power_iter times Y = X (X^T Y) (iterator="power": no re-normalisation); q = tsqr(Y);
    B = q^T X; SVD of B; u = q u_B; truncate to k; svd_flip.  Only the random stream differs from dask's
    chunked generator (any Gaussian sketch spans the same subspace to the solver's tolerance)."""
    X = np.asarray(X, dtype=np.float64)
    m, n = X.shape
    comp = min(max(min_subspace_size, k + n_oversamples), min(m, n))
    Y = X @ np.random.RandomState(seed).standard_normal(size=(n, comp))
    for _ in range(n_power_iter):
        Y = X @ (X.T @ Y)
    q, _ = np.linalg.qr(Y)
    u, s, vt = np.linalg.svd(q.T @ X, full_matrices=False)
    u = q @ u
    u, s, vt = u[:, :k], s[:k], vt[:k]
    u, vt = svd_flip(u, vt)
    return u, s, vt
