"""xeofs_amd.cross.CPCCA / MCA / CCA / RDA -- drop-ins for xeofs.cross.CPCCA (xeofs/cross/cpcca.py:23-1020),
MCA (cross/mca.py:20-123, alpha = 1), CCA (cross/cca.py, alpha = 0) and RDA (cross/rda.py, alpha = [0, 1]);
fit / transform / inverse_transform / predict as in xeofs/cross/base_model_cross_set.py:269-460.

Pipeline per field: fused HIP preprocess -> resident matrix -> PCA pre-reduction (`xeofs_amd.pca.ResidentPCA`,
default on) -> fractional whitening T = (Z^T Z / n)^((alpha-1)/2) in PC space (m x m, host algebra as in
preprocessing/whitener.py:115-133) -> matrix-free randomized SVD of the cross-covariance of the two analysis
matrices (`eofx_crosscov_rsvd_f32`; C is never formed) -> singular vectors taken back through whitener and PCA
(V (Tinv^H Q), a panel product on the resident V).

With alpha = 1 and use_pca = False the analysis runs directly on the resident feature-space matrices
(the SURVEY.md §8 hot path, row R14/R15).  The diagnostics of Swenson (2015) are evaluated through rank-one
algebra on panel products (A B, A^T R) instead of per-mode reconstructions of the full fields.
"""

from __future__ import annotations

import datetime
import warnings

import numpy as np

from .. import __version__, engine, labelled
from .._deferred import Deferred
from ..linalg.decomposer import MAX_SKETCH, sanity_check_n_modes
from ..pca import ResidentPCA
from ..preprocessing import Preprocessor

MAX_DENSE_WHITEN = 8192     # whitening without PCA forms the p x p covariance (as the reference does)


def _pair(v):
    return list(v) if isinstance(v, (list, tuple)) else [v, v]


def fractional_matrix_power(C, power):
    """linalg/_numpy/_utils.py:6-33 for a real symmetric PSD matrix: V s^power V^T, s <= eps dropped."""
    w, V = np.linalg.eigh(0.5 * (C + C.T))
    keep = w > np.finfo(w.dtype).eps
    return (V[:, keep] * w[keep] ** power) @ V[:, keep].T


NEAR_DIAGONAL_TOL = 1e-8    # bound on the second-order term of `near_diagonal_powers` (relative to the matrix function)


def near_diagonal_powers(C, powers):
    """C^power for each of `powers` when C is symmetric positive definite and NEARLY DIAGONAL -- the covariance of PC
    scores (whitener.py:106-123 after pca.py:120-131: the columns are orthogonal up to rounding).  No eigen-decomposition:
    with C = D + E the matrix function is f(D) + F o E + O(|E|^2), F_ij = (f(d_i) - f(d_j)) / (d_i - d_j) the first divided
    differences (Daleckii-Krein) -- a bounded factor, so that close diagonal entries are harmless.  The dropped second-order
    term is at most ~ m rho^2 relative, rho = max |E_ij| / sqrt(d_i d_j); the route is taken when that bound is below
    NEAR_DIAGONAL_TOL and no diagonal entry is near the reference's cut-off (`s > eps`, linalg/_numpy/_utils.py:20-23).
    C: float64 torch tensor (any device).  Returns the list of matrices, or None (the caller takes the eigen-decomposition)."""
    import torch

    m = C.shape[0]
    d = C.diagonal().clone()
    if m == 0 or not bool(torch.isfinite(C).all()) or not bool((d > 0).all()):
        return None
    dmax, dmin = float(d.max()), float(d.min())
    if dmin <= 1e-12 * dmax or dmin <= 1e4 * float(torch.finfo(C.dtype).eps):
        return None
    r = torch.rsqrt(d)
    E = C - torch.diag(d)
    rho = float((E * r[:, None] * r[None, :]).abs().max()) if m > 1 else 0.0
    if m * rho * rho > NEAR_DIAGONAL_TOL:
        return None
    a, b = d[:, None], d[None, :]
    diff = a - b
    close = diff.abs() <= 1e-5 * torch.maximum(a, b)
    safe = torch.where(close, torch.ones_like(diff), diff)
    mid = 0.5 * (a + b)
    out = []
    for power in powers:
        fd = d ** power
        F = torch.where(close, power * mid ** (power - 1.0), (fd[:, None] - fd[None, :]) / safe)
        out.append(torch.diag(fd) + F * E)
    return out


def _whitener_is_identity(alpha) -> bool:
    """preprocessing/whitener.py:54-60: `(1.0 - alpha) < eps` -- every alpha >= 1 (and those within one ulp below it) is the
    identity transform; alpha = 1 - 1e-9 is NOT (np.isclose would call it one)."""
    return (1.0 - float(alpha)) < np.finfo(np.float64).eps


def _run_two(gens):
    """Drive two generators that each yield (at most once) a symmetric device matrix and take `torch.linalg.eigh` of it: the
    first one's eigen-problem runs on a second stream (started from a helper thread) while the second generator does its
    work -- and its own eigen-problem -- on the main stream.  Same inputs, same routine, same results as one after the
    other; the library call (order 1500: 36 ms of small launch-bound kernels) just no longer leaves the GPU idle."""
    torch = engine._torch()

    def first(g):
        if g is None:
            return None
        try:
            return next(g)
        except StopIteration:
            return None

    def finish(g, eig):
        try:
            g.send(eig)
        except StopIteration:
            pass

    M0 = first(gens[0])
    pending = None
    if M0 is not None and gens[1] is not None and M0.is_cuda:
        import threading

        main = torch.cuda.current_stream(M0.device)
        side = torch.cuda.Stream(M0.device)
        ready = torch.cuda.Event()
        ready.record(main)
        box = {}

        def run():
            try:
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    th, W = torch.linalg.eigh(M0)
                    done = torch.cuda.Event()
                    done.record(side)
                box["out"] = (th, W, done)
            except BaseException as e:      # re-raised in the caller's thread
                box["err"] = e

        M0.record_stream(side)
        pending = threading.Thread(target=run)
        pending.start()
    try:
        M1 = first(gens[1])
        eig1 = torch.linalg.eigh(M1) if M1 is not None else None
    except BaseException:
        if pending is not None:
            pending.join()
        raise
    if M0 is not None:
        if pending is not None:
            pending.join()
            if "err" in box:
                raise box["err"]
            th, W, done = box["out"]
            main.wait_event(done)
            th.record_stream(main)
            W.record_stream(main)
            eig0 = (th, W)
        else:
            eig0 = torch.linalg.eigh(M0)
        finish(gens[0], eig0)
    if M1 is not None:
        finish(gens[1], eig1)


def _fit_two_pcas(pcas, mats, total_variances):
    """The two PCA pre-reductions of a cross model (base_model_cross_set.py:307-313): each randomized fit has ONE library
    call, the order-ell `eigh` of its Rayleigh-Ritz matrix (36 of 85 ms at config 3); `_run_two` hides the first field's
    under the second field's Gram matrix, range finder and wide product."""
    _run_two([p.fit_steps(m, tv) if p is not None else None for p, m, tv in zip(pcas, mats, total_variances)])


class _Side:
    """One field of the cross model: resident matrix, optional PCA, optional whitener, analysis matrix."""

    _near_diagonal_ok = True      # the whitener of PC scores without an eigen-decomposition (near_diagonal_powers)
    whitener_route = None         # "near-diagonal" | "eigh" once a device whitener ran

    def __init__(self, ctx, mat, pca, alpha):
        self.ctx, self.mat, self.pca, self.alpha = ctx, mat, pca, alpha
        self.T = self.Tinv = None
        # the analysis matrix (n x m): PC scores, whitened -- on the DEVICE (float32 tensor) where a PCA produced it, and on
        # the host (float64, downloaded on first use) for the m x m / n x k diagnostics algebra
        self._Zdev = pca.scores_device() if (pca is not None and hasattr(pca, "scores_device")) else None
        self._Zhost = None if (pca is None or self._Zdev is not None) else pca.scores()
        self._whiten = None           # the device whitener as a generator (yields the covariance, takes its eigh): see steps()
        if not _whitener_is_identity(alpha):                     # whitener.py:54-60: identity when (1 - alpha) < eps
            if self._Zdev is not None:
                self._whiten = self._whiten_on_device(alpha)
            else:
                if self._Zhost is None:
                    if mat.p > MAX_DENSE_WHITEN:
                        raise NotImplementedError(
                            f"alpha < 1 without PCA needs the {mat.p} x {mat.p} feature covariance; use use_pca=True")
                    self._Zhost = mat.download().astype(np.float64)
                n, m = self._Zhost.shape
                self._warn_ill_conditioned(n, m)
                Cm = self._Zhost.T @ self._Zhost / n
                self.T = fractional_matrix_power(Cm, (alpha - 1) / 2)
                try:
                    self.Tinv = np.linalg.inv(self.T)
                except np.linalg.LinAlgError:
                    self.Tinv = np.linalg.pinv(self.T)
                self._Zhost = self._Zhost @ self.T
        self.n, self.m = mat.n, (mat.p if not self.has_Z else
                                 (self._Zdev.shape[1] if self._Zdev is not None else self._Zhost.shape[1]))

    def steps(self):
        """Finish the construction as a generator: a device whitener yields its covariance matrix once and takes
        `torch.linalg.eigh` of it (so that the two sides' eigen-problems can run side by side, `_run_two`); then the analysis
        matrix becomes resident."""
        if self._whiten is not None:
            yield from self._whiten
            self._whiten = None
        ctx, mat = self.ctx, self.mat
        if self._Zdev is not None:
            self.work = engine.from_dense(ctx, self._Zdev)       # device tensor in: nothing crosses PCIe
        else:
            self.work = mat if self._Zhost is None else engine.from_dense(ctx, self._Zhost.astype(np.float32))

    @staticmethod
    def _warn_ill_conditioned(n, m):
        if n < m:                                               # whitener.py:101-104
            warnings.warn(f"The number of samples ({n}) is smaller than the number of features ({m}), leading to "
                          "an ill-conditioned problem. This may cause unstable results. Consider using PCA to "
                          "reduce dimensionality and stabilize the problem by setting `use_pca=True`.")

    def _whiten_on_device(self, alpha):
        """Whitener.fit / transform (preprocessing/whitener.py:86-133) on the resident PC scores: C = Z^T Z / n through the
        float64 Gram kernel, T = C^((alpha - 1) / 2) from its eigen-decomposition (linalg/_numpy/_utils.py:6-33: eigenvalues
        <= eps dropped; order m <= int(0.3 rank), the one library call), Z T through panel_matmul.  T and its inverse go
        to the host (m x m) for the back-transforms; the n x m matrix never leaves HBM."""
        torch = engine._torch()
        ctx = self.ctx
        Zd = self._Zdev
        n, m = Zd.shape
        self._warn_ill_conditioned(n, m)
        Lm = (m + 31) // 32 * 32
        Zp = torch.zeros((self.mat.n_pad, Lm), dtype=torch.float32, device=Zd.device)
        Zp[:n, :m] = Zd
        Cm = engine.panel_gram(ctx, Zp)[:m, :m] / n
        Cs = (0.5 * (Cm + Cm.T)).contiguous()
        # PC scores are orthogonal up to rounding: the covariance is diagonal to first order and its matrix powers follow
        # without the order-m eigen-decomposition (36 ms of rocSOLVER launches at m = 1500); `_near_diagonal_ok = False`
        # or a covariance that is not nearly diagonal takes the eigen-decomposition
        short = near_diagonal_powers(Cs, [(alpha - 1) / 2, (1 - alpha) / 2]) if self._near_diagonal_ok else None
        self.whitener_route = "near-diagonal" if short is not None else "eigh"
        if short is not None:
            T, Tinv = short
        else:
            w, V = yield Cs
            keep = w > torch.finfo(w.dtype).eps
            Vk, wk = V[:, keep], w[keep]
            T = (Vk * wk ** ((alpha - 1) / 2)) @ Vk.T
            # np.linalg.inv(T) where T is regular (all eigenvalues kept), its pseudo-inverse otherwise (whitener.py:117-123)
            Tinv = (Vk * wk ** ((1 - alpha) / 2)) @ Vk.T
        Tp = torch.zeros((Lm, Lm), dtype=torch.float64, device=Zd.device)
        Tp[:m, :m] = T
        self._Zdev = engine.panel_matmul(ctx, Zp, Tp)[:n, :m].contiguous()
        self.T, self.Tinv = T.cpu().numpy(), Tinv.cpu().numpy()

    @property
    def has_Z(self):
        return self._Zdev is not None or self._Zhost is not None

    @property
    def Z(self):
        """the analysis matrix on the host (float64), or None when the field itself is the analysis matrix"""
        if self._Zhost is None and self._Zdev is not None:
            self._Zhost = self._Zdev.double().cpu().numpy()
        return self._Zhost

    # --- analysis-space <-> feature-space maps ------------------------------------------------
    def to_analysis(self, mat_new):
        """preprocessed new data (resident) -> analysis space: pca.transform, whitener.transform"""
        if not self.has_Z:
            return None                                          # stays resident
        Z = self.pca.transform(mat_new) if self.pca is not None else mat_new.download().astype(np.float64)
        return Z if self.T is None else Z @ self.T

    def components_back(self, Q):
        """whitener.inverse_transform_components, pca.inverse_transform_components: V (Tinv^H Q)"""
        Q = np.asarray(Q, dtype=np.float64)
        if self.Tinv is not None:
            Q = self.Tinv.T @ Q
        if self.pca is not None:
            return self.pca.back_project(Q)
        return Q.astype(np.float32)

    def data_back(self, Z):
        """analysis-space rows -> preprocessed feature space: whitener / pca inverse_transform_data"""
        if self.Tinv is not None:
            Z = Z @ self.Tinv
        if self.pca is not None:
            return engine.reconstruct(self.ctx, Z.astype(np.float32), self.pca.components())
        return Z.astype(np.float32)

    # --- unwhitened analysis matrix A (= input_data after whitener.inverse_transform_data) ------
    def A_host(self):
        if not self.has_Z:
            return None
        return self.Z if self.Tinv is None else self.Z @ self.Tinv

    def A_mul(self, B):
        """A B for B (m x k)"""
        A = self.A_host()
        if A is not None:
            return A @ B
        return engine.project(self.ctx, self.mat, np.ascontiguousarray(B, dtype=np.float32)).astype(np.float64)

    def A_tmul(self, R):
        """A^T R for R (n x k)"""
        A = self.A_host()
        if A is not None:
            return A.T @ R
        k = R.shape[1]
        L = engine.panel_width(k)
        Rp = engine.panel_import(self.ctx, np.ascontiguousarray(R, dtype=np.float32), self.mat.n_pad, L)
        out = engine.panel_tmul(self.ctx, self.mat, Rp, prec=self.ctx.precision[1])
        return self.mat.compact_rows(engine.panel_export(self.ctx, out, self.mat.p_phys, k)).astype(np.float64)

    def A_sumsq(self):
        A = self.A_host()
        return float((A * A).sum()) if A is not None else self.mat.sumsq()

    def feature_std_and_cov(self, R):
        """for the correlation patterns: per-feature population std of the (PCA-truncated) field and
        field^T R in feature space (p x k)"""
        n = self.n
        if self.pca is not None:
            A = self.A_host()
            num = self.pca.back_project(A.T @ R).astype(np.float64)
            S = A.T @ A
            w, E = np.linalg.eigh(0.5 * (S + S.T))
            half = (E * np.sqrt(np.clip(w, 0, None))) @ E.T
            std = self.pca.row_norms(half) / np.sqrt(n)
        elif self.has_Z:
            A = self.A_host()
            num, std = A.T @ R, np.sqrt((A * A).sum(axis=0) / n)
        else:
            num, std = self.A_tmul(R), engine.feature_norms(self.ctx, self.mat) / np.sqrt(n)
        return num, std

    def free(self):
        if self.has_Z and self.work is not None:
            self.work.free()
        self.work = None


class CPCCA(Deferred):
    _model_name = "Continuum Power CCA"

    def __init__(self, n_modes: int = 2, alpha=0.2, standardize=False, use_coslat=False, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, check_nans=True, compute: bool = True,
                 sample_name: str = "sample", feature_name="feature", solver: str = "auto", random_state=None,
                 solver_kwargs: dict = {}, **kwargs):
        sanity_check_n_modes(n_modes)
        if solver not in ("auto", "full", "randomized"):
            raise ValueError(f"Unrecognized solver '{solver}'. Valid options are 'auto', 'full', and 'randomized'.")
        self.n_modes = n_modes
        std, cos, chk = _pair(standardize), _pair(use_coslat), _pair(check_nans)
        self._params = dict(n_modes=n_modes, alpha=[float(a) for a in _pair(alpha)], standardize=std, use_coslat=cos,
                            check_nans=chk, use_pca=_pair(use_pca), n_pca_modes=_pair(n_pca_modes),
                            pca_init_rank_reduction=_pair(pca_init_rank_reduction), sample_name=sample_name,
                            feature_name=_pair(feature_name), random_state=random_state, compute=compute, solver=solver)
        self.alpha = self._params["alpha"]
        self.solver, self.random_state, self.solver_kwargs = solver, random_state, dict(solver_kwargs)
        self.sample_name = sample_name
        # CPCCA always centres (cpcca.py:145)
        # In-place layout: the fields are read where they lie (staged once when they come from the host), nothing is written;
        # whatever needs a layout (PCA, dense whitening, download) builds it on demand.  Without PCA and whitening the
        # resident matrices go straight to engine.crosscov_rsvd, which also takes land / sea masks in place (zero columns).
        direct = [not self._params["use_pca"][i] and _whitener_is_identity(self.alpha[i]) for i in range(2)]
        self.preprocessor1 = Preprocessor(True, std[0], cos[0], chk[0], in_place=True, masked_ok=direct[0])
        self.preprocessor2 = Preprocessor(True, std[1], cos[1], chk[1], in_place=True, masked_ok=direct[1])
        self.attrs = {"model": self._model_name, "software": "xeofs_amd", "version": __version__,
                      "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
        self.ctx = None
        self.data = {}
        self.pca = [None, None]
        self.side = [None, None]

    def get_params(self):
        return dict(self._params)

    # ------------------------------------------------------------------ fit
    def fit(self, X, Y, dim, weights_X=None, weights_Y=None):
        if (labelled.is_lazy(X) or labelled.is_lazy(Y)) and not self._params["compute"]:   # base_model_cross_set.py: defer
            return self._defer(lambda: self._fit_now(X, Y, dim, weights_X, weights_Y))
        return self._fit_now(X, Y, dim, weights_X, weights_Y)

    def _fit_now(self, X, Y, dim, weights_X=None, weights_Y=None):
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor1.ctx = self.preprocessor2.ctx = self.ctx
        ahead = self._sketch_ahead(X, Y, dim)       # drawn on a worker thread while the two fields are preprocessed
        mx = self.preprocessor1.fit_transform(X, dim, weights_X)
        my = self.preprocessor2.fit_transform(Y, dim, weights_Y)
        self.sample_dims = self.preprocessor1.sample_dims
        if mx.n != my.n:
            raise ValueError("Both data matrices must have the same number of samples but found "
                             f"{mx.n} in the first and {my.n} in the second.")
        k = int(self.n_modes)
        kw = dict(self.solver_kwargs)
        n_over, n_iter = int(kw.pop("n_oversamples", 10)), kw.pop("n_iter", "auto")
        # PCA pre-reduction and whitening (base_model_cross_set.py:307-313)
        mats, pres = (mx, my), (self.preprocessor1, self.preprocessor2)
        # (the reference's PCA is unseeded; here a model built with an integer random_state seeds the two pre-reductions with
        # random_state + i -- distinct starts for the two fields, the same fit for the same seed -- and an unseeded model takes
        # the process-wide counter of xeofs_amd.pca)
        rs = self.random_state
        seeds = [int(rs) + 1000003 * (i + 1) if isinstance(rs, (int, np.integer)) else None for i in range(2)]
        # (alpha = 1: no whitener reads the PC spectrum and the model's outputs are invariant to a rotation inside the PC space --
        # the pre-reduction may skip its order-ell eigen-decomposition when it keeps every mode anyway, pca.py `basis_only`)
        pcas = [ResidentPCA(self.ctx, self._params["n_pca_modes"][i], self._params["pca_init_rank_reduction"][i], random_state=seeds[i],
                            basis_only=_whitener_is_identity(self.alpha[i]))
                if self._params["use_pca"][i] else None for i in range(2)]
        _fit_two_pcas(pcas, mats, [pre.total_variance for pre in pres])
        for i in range(2):
            self.pca[i] = pcas[i]
            self.side[i] = _Side(self.ctx, mats[i], pcas[i], self.alpha[i])
        _run_two([sd.steps() for sd in self.side])       # the two whiteners' eigen-problems side by side
        sx, sy = self.side
        rank = min(sx.m, sy.m)
        if k > rank:
            raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {rank}).")
        small = max(sx.m, sy.m) < 500
        if self.solver == "full" or (self.solver == "auto" and small and k > int(0.8 * rank)):
            if rank > MAX_SKETCH:
                raise NotImplementedError(f"solver='full' on the cross path needs rank <= {MAX_SKETCH}; use 'randomized'")
            n_over, n_iter = rank - k, 0
        n_over = min(n_over, rank - k)         # a sketch as wide as the rank is already exact
        identity = sx.Tinv is None and sy.Tinv is None
        omega = None
        if ahead is not None:
            fut, l_ahead = ahead
            small_ = min(sx.work.p, sy.work.p)
            if l_ahead == k + n_over and fut.rows >= small_:      # numpy fills row by row: the leading rows ARE the draw
                omega = engine.SketchSlice(fut, small_)           # joined by the engine when it first needs the sketch
            else:
                fut.result()
        out = self._crosscov(sx, sy, k, n_over, n_iter, identity, omega)
        s = out["s"].astype(np.float64)
        self._q = [out["Q1"].astype(np.float64), out["Q2"].astype(np.float64)]     # in the analysis space
        tsc = out["total_squared_covariance"] if identity else self._unwhitened_tsc()
        comps = [self.side[i].components_back(self._q[i]) for i in range(2)]
        for sd in self.side:
            sd.free()
        self.data = dict(
            input_data1=sx.mat, input_data2=sy.mat, components1=comps[0], components2=comps[1],
            scores1=out["scores1"], scores2=out["scores2"], singular_values=s, squared_covariance=s ** 2,
            total_squared_covariance=tsc, idx_modes_sorted=np.argsort(s)[::-1],
            norm1=out["norm1"].astype(np.float64), norm2=out["norm2"].astype(np.float64),
        )
        return self

    def _crosscov(self, sx, sy, k, n_over, n_iter, identity, omega):
        try:
            return engine.crosscov_rsvd(self.ctx, sx.work, sy.work, k, n_over, n_iter, random_state=self.random_state,
                                        want_tsc=identity, omega=omega)
        except NotImplementedError:
            # a masked in-place pair the engine cannot take (a sketch as wide as the rank on a masked field): compact the
            # fields and go again
            if not (sx.work.masked or sy.work.masked):
                raise
            import logging

            logging.getLogger("xeofs_amd").info("cross-covariance: masked in-place matrices not usable here, compacting the fields")
            for sd, pre in ((sx, self.preprocessor1), (sy, self.preprocessor2)):
                if sd.work.masked:
                    sd.work = sd.mat = pre.recompact(sd.mat)
            return engine.crosscov_rsvd(self.ctx, sx.work, sy.work, k, n_over, n_iter, random_state=self.random_state,
                                        want_tsc=identity, omega=omega)

    _SKETCH_AHEAD_MIN = 20000

    def _sketch_ahead(self, X, Y, dim):
        """The matrix-free path (no PCA pre-reduction) sketches on the smaller FEATURE side: min(p1, p2) x (k + oversamples)
        normals of sklearn's RandomState stream, 6 ms at 129 600 x 30.  Its height depends on the NaN mask, but numpy
        fills the array row by row, so a draw as tall as the raw feature count contains the one that is needed as its
        leading rows: start it before the preprocessing.  -> (SketchFuture, width) or None"""
        try:
            if any(self._params["use_pca"]) or self.solver == "full" or not isinstance(self.random_state, (int, np.integer)):
                return None
            sd = (dim,) if isinstance(dim, str) else tuple(dim)
            ps = []
            for Z in (X, Y):
                vals, dims, _, _, _ = labelled.unpack(Z)
                n = int(np.prod([vals.shape[dims.index(d)] for d in sd]))
                ps.append(int(np.prod(tuple(vals.shape), dtype=np.int64)) // max(n, 1))
            kw = dict(self.solver_kwargs)
            l = int(self.n_modes) + int(kw.get("n_oversamples", 10))
            if min(ps) < self._SKETCH_AHEAD_MIN or l > min(ps):       # small draws are not worth a thread
                return None
            return engine.SketchFuture(min(ps), l, int(self.random_state)), l
        except Exception:
            return None

    def _unwhitened_tsc(self):
        """cpcca.py:991-1000: sum |Tinv1^T C Tinv2|^2 = || A1^T A2 ||_F^2 / (n-1)^2 on the unwhitened matrices"""
        sx, sy = self.side
        n = sx.n
        if sx._Zdev is not None and sy._Zdev is not None:
            # A1^T A2 = Tinv1^T (Z1^T Z2) Tinv2 with Z1^T Z2 the off-diagonal block of the float64 Gram matrix of the panel
            # [Z1 | Z2] (gram kernel on the resident analysis matrices); the m x m products on the host
            torch = engine._torch()
            Z1, Z2 = sx._Zdev, sy._Zdev
            m1, m2 = Z1.shape[1], Z2.shape[1]
            L1, L2 = (m1 + 31) // 32 * 32, (m2 + 31) // 32 * 32
            P = torch.zeros((sx.mat.n_pad, L1 + L2), dtype=torch.float32, device=Z1.device)
            P[:n, :m1] = Z1
            P[:n, L1:L1 + m2] = Z2
            C12 = engine.panel_gram(self.ctx, P)[:m1, L1:L1 + m2].cpu().numpy()
            if sx.Tinv is not None:
                C12 = sx.Tinv.T @ C12
            if sy.Tinv is not None:
                C12 = C12 @ sy.Tinv
            return float((C12 ** 2).sum()) / (n - 1) ** 2
        A1, A2 = sx.A_host(), sy.A_host()
        if A1 is not None and A2 is not None:
            return float(((A1.T @ A2) ** 2).sum()) / (n - 1) ** 2
        mats, tmp = [], []
        for sd, A in ((sx, A1), (sy, A2)):
            if A is None:
                mats.append(sd.mat)
            else:
                mats.append(engine.from_dense(self.ctx, A.astype(np.float32)))
                tmp.append(mats[-1])
        t = engine.vec_dot(self.ctx, mats[0].gram(0), mats[1].gram(0)) / (n - 1) ** 2
        for m in tmp:
            m.free()
        return t

    # ------------------------------------------------------------------ accessors (base_model_cross_set.py:465-523)
    def components(self, normalized: bool = True):
        q1, q2 = self.data["components1"], self.data["components2"]
        if not normalized:
            q1, q2 = q1 * self.data["norm1"].astype(q1.dtype), q2 * self.data["norm2"].astype(q2.dtype)
        return (self.preprocessor1.inverse_transform_components(q1, "components1", self.attrs),
                self.preprocessor2.inverse_transform_components(q2, "components2", self.attrs))

    def scores(self, normalized: bool = False):
        s1, s2 = self.data["scores1"], self.data["scores2"]
        if normalized:
            s1, s2 = s1 / self.data["norm1"].astype(s1.dtype), s2 / self.data["norm2"].astype(s2.dtype)
        return (self.preprocessor1.inverse_transform_scores(s1, "scores1", self.attrs),
                self.preprocessor2.inverse_transform_scores(s2, "scores2", self.attrs))

    def _project(self, which, Z):
        """preprocess -> pca -> whitener -> singular vectors; returns (scores n' x k, fields, valid samples)"""
        pre = self.preprocessor1 if which == 1 else self.preprocessor2
        sd = self.side[which - 1]
        mat, fields, vs = pre.transform(Z)
        Za = sd.to_analysis(mat)
        if Za is None:
            proj = engine.project(self.ctx, mat, self._q[which - 1].astype(np.float32)).astype(np.float64)
        else:
            proj = Za @ self._q[which - 1]
        mat.free()
        return proj, fields, vs

    def transform(self, X=None, Y=None, normalized: bool = False):
        """base_model_cross_set.py:323-374 + cpcca.py:227-252."""
        self.compute()          # a deferred fit (compute=False on a lazy input) runs now: the fitted state is needed
        if X is None and Y is None:
            raise ValueError("Either X or Y must be provided.")
        outs = []
        for which, Z in ((1, X), (2, Y)):
            if Z is None:
                continue
            pre = self.preprocessor1 if which == 1 else self.preprocessor2
            proj, fields, vs = self._project(which, Z)
            if normalized:
                proj = proj / self.data[f"norm{which}"]
            outs.append(pre.inverse_transform_scores(proj.astype(np.float32), f"scores{which}", self.attrs, fields, vs))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def predict(self, X):
        """base_model_cross_set.py:427-449 + cpcca.py:273-302: pseudo scores of Y from new X."""
        self.compute()          # a deferred fit (compute=False on a lazy input) runs now: the fitted state is needed
        proj, fields, vs = self._project(1, X)
        Rx, Ry = self.data["scores1"].astype(np.float64), self.data["scores2"].astype(np.float64)
        G = Rx.T @ Ry / np.linalg.norm(Rx, axis=0) ** 2
        return self.preprocessor2.inverse_transform_scores((proj @ G).astype(np.float32), "pseudo_scores_Y", self.attrs,
                                                           fields, vs)

    def inverse_transform(self, X=None, Y=None):
        """base_model_cross_set.py:376-425 + cpcca.py:254-271: scores (with a 'mode' dimension) back to the fields."""
        self.compute()          # a deferred fit (compute=False on a lazy input) runs now: the fitted state is needed
        if X is None and Y is None:
            raise ValueError("Either X or Y must be provided.")
        outs = []
        for which, S in ((1, X), (2, Y)):
            if S is None:
                continue
            pre = self.preprocessor1 if which == 1 else self.preprocessor2
            vals, dims, coords, _, _ = labelled.unpack(S)
            if "mode" not in dims:
                vals, dims = vals[None], ("mode",) + tuple(dims)
                coords = dict(coords, mode=np.array([1]))
            modes = np.asarray(coords["mode"]).astype(int)
            order = [dims.index("mode")] + [i for i, d in enumerate(dims) if d != "mode"]
            Sm = np.transpose(vals, order).reshape(len(modes), -1).T.astype(np.float64)     # (n', k')
            vs = ~np.isnan(Sm).all(axis=1)
            Za = Sm[vs] @ self._q[which - 1][:, modes - 1].T                                  # analysis space
            rec = self.side[which - 1].data_back(Za)
            f0 = pre.fields[0]
            sample_shape = tuple(vals.shape[dims.index(d)] for d in f0.sample_dims)
            fields = []
            for f in pre.fields:
                g = object.__new__(type(f))
                g.__dict__.update(f.__dict__)
                g.sample_shape = sample_shape
                g.coords = dict(f.coords, **{d: coords[d] for d in f.sample_dims if d in coords})
                fields.append(g)
            outs.append(pre.inverse_transform_data(rec, "reconstructed_data", fields, vs))
        return outs[0] if len(outs) == 1 else outs

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.preprocessor1.fields[0].like)

    def singular_values(self):
        return self._mode_array(self.data["singular_values"], "singular_values")

    def squared_covariance(self):
        return self._mode_array(self.data["squared_covariance"], "squared_covariance")

    def total_squared_covariance(self):
        return self.data["total_squared_covariance"]

    # ------------------------------------------------------------------ diagnostics (cpcca.py:330-640)
    @staticmethod
    def _corr(A, B):
        """cpcca.py:910-985 method='correlation': columns scaled by their population std, cross-products / (n-1)"""
        A, B = A.astype(np.float64), B.astype(np.float64)
        return (A / A.std(axis=0)).T @ (B / B.std(axis=0)) / (A.shape[0] - 1)

    def cross_correlation_coefficients(self):
        return self._mode_array(np.diag(self._corr(self.data["scores1"], self.data["scores2"])),
                                "cross_correlation_coefficients")

    def _mode_matrix(self, M, name):
        k = M.shape[0]
        return labelled.pack(M, ("mode_x", "mode_y"), {"mode_x": np.arange(1, k + 1), "mode_y": np.arange(1, k + 1)}, name,
                             dict(self.attrs), self.preprocessor1.fields[0].like)

    def correlation_coefficients_X(self):
        return self._mode_matrix(self._corr(self.data["scores1"], self.data["scores1"]), "correlation_coefficients_X")

    def correlation_coefficients_Y(self):
        return self._mode_matrix(self._corr(self.data["scores2"], self.data["scores2"]), "correlation_coefficients_Y")

    def _rank_one_terms(self):
        """per-mode pieces of the residuals d_i = A_i - r_i b_i^T (b_i = Tinv_i^T q_i), all modes batched into
        panel products: g1 = A1^T R2, g2 = A2^T R1, a_i = A_i b_i, ..."""
        R1, R2 = self.data["scores1"].astype(np.float64), self.data["scores2"].astype(np.float64)
        B = []
        for i in range(2):
            Q = self._q[i]
            B.append(Q if self.side[i].Tinv is None else self.side[i].Tinv.T @ Q)
        return R1, R2, B[0], B[1]

    def squared_covariance_fraction(self):
        """cpcca.py:418-497: 1 - ||d_X^T d_Y||_F^2 / ||X^T Y||_F^2 per mode (clipped at 0).

        D = M - b1 g2^T - g1 b2^T + c b1 b2^T with M = A1^T A2; ||D||^2 is expanded into inner products of
        n-vectors A_i x so that M (and any per-mode reconstruction of the fields) is never formed."""
        sx, sy = self.side
        n = sx.n
        R1, R2, B1, B2 = self._rank_one_terms()
        G1, G2 = sx.A_tmul(R2), sy.A_tmul(R1)                        # (m1 x k), (m2 x k)
        a1, a2b = sx.A_mul(B1), sy.A_mul(B2)                         # A1 b1, A2 b2   (n x k)
        a2g, a1g = sy.A_mul(G2), sx.A_mul(G1)                        # A2 g2, A1 g1   (n x k)
        c = (R1 * R2).sum(axis=0)
        nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)
        ng1, ng2 = (G1 * G1).sum(0), (G2 * G2).sum(0)
        M2 = self.data["total_squared_covariance"] * (n - 1) ** 2
        b1g1, g2b2 = (B1 * G1).sum(0), (G2 * B2).sum(0)
        D2 = (M2 + nb1 * ng2 + ng1 * nb2 + c ** 2 * nb1 * nb2 - 2 * (a1 * a2g).sum(0) - 2 * (a1g * a2b).sum(0)
              + 2 * c * (a1 * a2b).sum(0) + 2 * b1g1 * g2b2 - 2 * c * nb1 * g2b2 - 2 * c * b1g1 * nb2)
        scf = 1 - D2 / M2
        return self._mode_array(np.where(scf < 0, 0, scf), "squared_covariance_fraction")

    def _fve_self(self, i):
        sd = self.side[i]
        R = self.data[f"scores{i + 1}"].astype(np.float64)
        B = self._rank_one_terms()[2 + i]
        tot = sd.A_sumsq()
        res = tot - 2 * (R * sd.A_mul(B)).sum(0) + (R * R).sum(0) * (B * B).sum(0)
        return 1 - res / tot

    def fraction_variance_X_explained_by_X(self):
        return self._mode_array(self._fve_self(0), "fraction_variance_X_explained_by_X")

    def fraction_variance_Y_explained_by_Y(self):
        return self._mode_array(self._fve_self(1), "fraction_variance_Y_explained_by_Y")

    def fraction_variance_Y_explained_by_X(self):
        """cpcca.py:563-640: like the SCF but with (X^T X)^(-1/2) in front (needs X in a reduced space)."""
        sx, sy = self.side
        A1, A2 = sx.A_host(), sy.A_host()
        if A1 is None or A2 is None:
            raise NotImplementedError("fraction_variance_Y_explained_by_X needs the whitening of the X covariance: "
                                      "fit with use_pca=True")
        n = sx.n
        R1, R2, B1, B2 = self._rank_one_terms()
        Tm = fractional_matrix_power(A1.T @ A1 / (n - 1), -0.5)
        M = A1.T @ A2
        TM = Tm @ M
        tot = (TM ** 2).sum()
        G1, G2 = A1.T @ R2, A2.T @ R1
        c = (R1 * R2).sum(axis=0)
        out = np.empty(R1.shape[1])
        for j in range(R1.shape[1]):
            D = (TM - np.outer(Tm @ B1[:, j], G2[:, j]) - np.outer(Tm @ G1[:, j], B2[:, j])
                 + c[j] * np.outer(Tm @ B1[:, j], B2[:, j]))
            out[j] = 1 - (D ** 2).sum() / tot
        return self._mode_array(out, "fraction_variance_Y_explained_by_X")

    # ------------------------------------------------------------------ correlation patterns (cpcca.py:642-845)
    def _patterns(self, kind, correction, alpha):
        if correction is not None and correction not in MULTIPLE_TESTS:      # statistics.py:145-149
            raise ValueError(f"Your method '{correction}' is not in the accepted methods: {MULTIPLE_TESTS}")
        from scipy.special import betainc

        n = self.side[0].n
        S1, S2 = self.data["scores1"].astype(np.float64), self.data["scores2"].astype(np.float64)
        pairs = ((0, S1), (1, S2)) if kind == "homogeneous" else ((0, S2), (1, S1))
        pats, pvals = [], []
        for i, S in pairs:
            Sn = (S - S.mean(0)) / S.std(0)                       # statistics.py:50-54 (population std)
            num, std = self.side[i].feature_std_and_cov(Sn)
            with np.errstate(invalid="ignore", divide="ignore"):
                corr = num / std[:, None] / n
            a = n / 2 - 1                                         # statistics.py:85-101: beta(a, a) on [-1, 1]
            pv = 2 * betainc(a, a, np.clip((1 - np.abs(corr)) / 2, 0, 1))
            if correction is not None:
                pv = holm_sidak(pv)
            pre = self.preprocessor1 if i == 0 else self.preprocessor2
            side = "left" if i == 0 else "right"
            pats.append(pre.inverse_transform_components(corr.astype(np.float32), f"{side}_{kind}_patterns", self.attrs))
            pvals.append(pre.inverse_transform_components(pv.astype(np.float32), f"pvalues_of_{side}_{kind}_patterns",
                                                          self.attrs))
        return tuple(pats), tuple(pvals)

    def homogeneous_patterns(self, correction=None, alpha=0.05):
        return self._patterns("homogeneous", correction, alpha)

    def heterogeneous_patterns(self, correction=None, alpha=0.05):
        return self._patterns("heterogeneous", correction, alpha)


MULTIPLE_TESTS = ["bonferroni", "sidak", "holm-sidak", "holm", "simes-hochberg", "hommel", "fdr_bh", "fdr_by", "fdr_tsbh",
                  "fdr_tsbky"]          # utils/constants.py:21-32


def holm_sidak(p):
    """Adjusted p-values per column (mode) over the rows (features) -- what the reference's `correction=` delivers.

    utils/optional/statistics.py:150-157 hands every mode's p-values to statsmodels' `multipletests` WITHOUT forwarding
    `method` or `alpha`, so whatever valid name the caller passes, the adjustment is that function's default: Holm-Sidak
    step-down.  statsmodels (0.14, an optional dependency absent from both images) computes it as: sort ascending;
    raw_i = 1 - (1 - p_(i))^(m - i), i = 0..m-1 (via expm1 / log1p); running maximum; clip at 1; undo the sort.
    The rejection flags the reference also computes are discarded there (statistics.py:83-86)."""
    p = np.asarray(p, dtype=np.float64)
    m = p.shape[0]
    order = np.argsort(p, axis=0, kind="stable")                  # NaN (constant features) sort last and stay NaN
    ps = np.take_along_axis(p, order, axis=0)
    expo = np.arange(m, 0, -1, dtype=np.float64).reshape((m,) + (1,) * (p.ndim - 1))
    with np.errstate(invalid="ignore", divide="ignore"):
        raw = -np.expm1(expo * np.log1p(-ps))
    with np.errstate(invalid="ignore"):
        adj = np.minimum(np.where(np.isnan(raw), np.nan, np.fmax.accumulate(raw, axis=0)), 1.0)
    out = np.empty_like(adj)
    np.put_along_axis(out, order, adj, axis=0)
    return out


class MCA(CPCCA):
    """cross/mca.py:20-123: CPCCA with alpha = [1, 1]."""
    _model_name = "Maximum Covariance Analysis"

    def __init__(self, n_modes: int = 2, standardize=False, use_coslat=False, check_nans=True, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, compute: bool = True, sample_name: str = "sample",
                 feature_name="feature", solver: str = "auto", random_state=None, solver_kwargs: dict = {}):
        super().__init__(n_modes=n_modes, alpha=[1.0, 1.0], standardize=standardize, use_coslat=use_coslat,
                         use_pca=use_pca, n_pca_modes=n_pca_modes, pca_init_rank_reduction=pca_init_rank_reduction,
                         check_nans=check_nans, compute=compute, sample_name=sample_name, feature_name=feature_name,
                         solver=solver, random_state=random_state, solver_kwargs=solver_kwargs)
        self._params.pop("alpha")

    def covariance_fraction_CD95(self):
        """mca.py:127-189 (Cheng & Dunkerton 1995): CF_i = sigma_i / sum_j sigma_j over the retained modes, with the
        reference's warning when the estimate still moves by more than 1e-3 with the last mode."""
        s = np.asarray(self.data["singular_values"], dtype=np.float64)
        cf = s[0] / np.cumsum(s)
        if len(s) > 1 and (cf[-2] - cf[-1]) > 0.001:
            warnings.warn("The curent estimate of CF is sensitive to the number of modes retained. Please increase "
                          "`n_modes` for a better estimate.")
        return self._mode_array(s / s.sum(), "covariance_fraction")


class CCA(CPCCA):
    """cross/cca.py: CPCCA with alpha = [0, 0]."""
    _model_name = "Canonical Correlation Analysis"

    def __init__(self, n_modes: int = 2, standardize=False, use_coslat=False, check_nans=True, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, compute: bool = True, sample_name: str = "sample",
                 feature_name="feature", solver: str = "auto", random_state=None, solver_kwargs: dict = {}):
        super().__init__(n_modes=n_modes, alpha=[0.0, 0.0], standardize=standardize, use_coslat=use_coslat,
                         use_pca=use_pca, n_pca_modes=n_pca_modes, pca_init_rank_reduction=pca_init_rank_reduction,
                         check_nans=check_nans, compute=compute, sample_name=sample_name, feature_name=feature_name,
                         solver=solver, random_state=random_state, solver_kwargs=solver_kwargs)
        self._params.pop("alpha")


class RDA(CPCCA):
    """cross/rda.py: CPCCA with alpha = [0, 1]."""
    _model_name = "Redundancy Analysis"

    def __init__(self, n_modes: int = 2, standardize=False, use_coslat=False, check_nans=True, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, compute: bool = True, sample_name: str = "sample",
                 feature_name="feature", solver: str = "auto", random_state=None, solver_kwargs: dict = {}):
        super().__init__(n_modes=n_modes, alpha=[0.0, 1.0], standardize=standardize, use_coslat=use_coslat,
                         use_pca=use_pca, n_pca_modes=n_pca_modes, pca_init_rank_reduction=pca_init_rank_reduction,
                         check_nans=check_nans, compute=compute, sample_name=sample_name, feature_name=feature_name,
                         solver=solver, random_state=random_state, solver_kwargs=solver_kwargs)
        self._params.pop("alpha")
