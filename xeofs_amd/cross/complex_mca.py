"""The complex and Hilbert cross models: ComplexCPCCA / HilbertCPCCA (xeofs/cross/cpcca.py:1023-1500) and their fixed-alpha
children ComplexMCA / HilbertMCA (cross/mca.py:224-489, alpha = 1), ComplexCCA / HilbertCCA (cross/cca.py:123-355,
alpha = 0), ComplexRDA / HilbertRDA (cross/rda.py:123-355, alpha = [0, 1]), plus their rotators.

The `Complex*` classes take complex fields Z_x = U_x + i V_x, Z_y = U_y + i V_y; the `Hilbert*` classes take real fields and
augment them with their Hilbert transform.  Order of the stages as in base_model_cross_set.py:300-315:

    preprocess -> PCA pre-reduction (default) -> [Hilbert transform of the PC scores: `_augment_data`] -> fractional
    whitening T = (S^H S / n)^((alpha - 1) / 2) (identity for alpha = 1) -> cross-covariance C = S_x^H S_y / (n - 1) ->
    SVD of C -> scores, norms, total squared covariance of the UNwhitened matrices.

What runs on the GPU: the preprocessor, the PCA pre-reductions -- `ResidentPCA` on the real fields of the Hilbert model
(its analytic signal is taken of the n x m PC scores, through the same `eofx_hilbert_f32`), `ComplexResidentPCA`
(Hermitian Gram route) on complex fields -- and every product with a resident field (back-projection of the singular
vectors, projection of new data).  After the pre-reduction the analysis matrices are n x m with m of the order of tens to
hundreds, and the m_x x m_y complex cross-covariance matrix is decomposed exactly on the host (the reference hands it to
scipy's svds(lobpcg): same singular values / subspaces to its tolerance, vectors up to a unit phase per mode).
Without PCA (`use_pca=False`) the fields themselves are the analysis matrices: supported while they are small enough to
decompose densely (n * p <= MAX_DENSE); beyond that use the default PCA route.
"""

from __future__ import annotations

import datetime
import os
import warnings

import numpy as np

from .. import __version__, engine, labelled
from .._deferred import Deferred
from ..cpca import ComplexResidentPCA
from ..linalg.decomposer import sanity_check_n_modes
from ..pca import ResidentPCA
from ..preprocessing import Preprocessor

MAX_DENSE = 50_000_000      # elements of a field decomposed without PCA pre-reduction


def _pair(v):
    return list(v) if isinstance(v, (list, tuple)) else [v, v]


def _hermitian_power(C, power):
    """linalg/_numpy/_utils.py:6-33 for a Hermitian PSD matrix: V s^power V^H, s <= eps dropped"""
    w, V = np.linalg.eigh(0.5 * (C + C.conj().T))
    keep = w > np.finfo(w.dtype).eps
    return (V[:, keep] * w[keep] ** power) @ V[:, keep].conj().T


def _part32(vals, imag):
    """real / imaginary part of the input as a contiguous float32 array in ONE pass over it (the engine's working
    precision; a float64 copy of a 10 GB complex128 field first would double the host time of a fit), row blocks on a
    few threads (numpy's casting copy releases the GIL; one thread moves about 8 GB/s of a strided complex source)"""
    if not np.iscomplexobj(vals):
        return np.zeros(vals.shape, np.float32) if imag else np.ascontiguousarray(vals, dtype=np.float32)
    src = vals.imag if imag else vals.real
    if vals.ndim == 0 or vals.size < (1 << 24) or vals.shape[0] < 16:
        return np.ascontiguousarray(src, dtype=np.float32)
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(vals.shape, np.float32)
    nt = min(16, os.cpu_count() or 1, vals.shape[0])
    edges = np.linspace(0, vals.shape[0], nt + 1).astype(int)

    def block(j):
        out[edges[j]:edges[j + 1]] = src[edges[j]:edges[j + 1]]

    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(block, range(nt)))
    return out


def _leading_svd_device(C, k, device, block=24, max_iter=80, tol=1e-10):
    """k leading singular triplets of a dense complex matrix (hundreds x hundreds, float64) by block power iteration on
    the device: each step is two complex GEMMs and a QR of a (k + block)-column panel, iterated until the residuals
    ||C v - s u|| of the k wanted triplets are below tol * s_1 -- to the accuracy of the dense SVD it replaces
    (0.6 s through rocSOLVER at 1500 x 1500), which remains the fallback if the iteration has not converged.
    (The OTHER residual, ||C^H u - s v||, vanishes by construction -- V and s come from the factorisation of C^H Y
    itself -- and says nothing about convergence: ADVICE r02.)"""
    torch = engine._torch()
    dev = f"cuda:{device}"
    Cd = torch.as_tensor(C, device=dev)
    m1, m2 = Cd.shape
    l = min(k + block, m1, m2)
    g = torch.Generator().manual_seed(12345)
    Q = torch.randn((m2, l, 2), generator=g, dtype=torch.float64)
    Q = torch.view_as_complex(Q).to(dev)
    Q, _ = torch.linalg.qr(Q)
    CH = Cd.conj().T.contiguous()
    for it in range(max_iter):
        Y, _ = torch.linalg.qr(Cd @ Q)                 # m1 x l
        B = CH @ Y                                     # m2 x l   (= C^H Y)
        Q, R = torch.linalg.qr(B)
        if it >= 2 and it % 2 == 0:
            Ub, sb, Vbh = torch.linalg.svd(R.conj().T, full_matrices=False)    # Y^H C = R^H Q^H
            U = Y @ Ub[:, :k]
            V = Q @ Vbh.conj().T[:, :k]
            res = torch.linalg.norm(Cd @ V - U * sb[:k], dim=0).max()
            if float(res) <= tol * float(sb[0]):
                return U.cpu().numpy(), sb[:k].cpu().numpy(), V.conj().T.cpu().numpy()
    out = torch.linalg.svd(Cd, full_matrices=False)
    return tuple(a.cpu().numpy() for a in out)


def _sign_rule(VT):
    """utils/xarray_utils.py:273-301: +1 where |max| >= |min| per mode, numpy's lexicographic complex max / min"""
    mx, mn = VT.max(axis=1), VT.min(axis=1)
    return np.where(np.abs(mx) >= np.abs(mn), 1.0, -1.0)


class _Field:
    """one side after preprocessing and pre-reduction: analysis matrix S (n x m complex, host), maps back to features"""

    def __init__(self, S, back, project, parts, pca):
        self.S, self.back, self.project, self.parts, self.pca = S, back, project, parts, pca

    def free(self):
        for m in self.parts:
            if m is not None:
                m.free()
        self.parts = ()


class ComplexCPCCA(Deferred):
    """Drop-in for xeofs.cross.ComplexCPCCA (cross/cpcca.py:1023-1326)."""

    _model_name = "Complex CPCCA"
    _hilbert = False

    def __init__(self, n_modes: int = 2, alpha=0.2, standardize=False, use_coslat=False, check_nans=True, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, compute: bool = True, sample_name: str = "sample",
                 feature_name="feature", solver: str = "auto", random_state=None, solver_kwargs: dict = {}, **kwargs):
        sanity_check_n_modes(n_modes)
        self.alpha = [float(a) for a in _pair(alpha)]
        if any(a < 0 for a in self.alpha):       # preprocessing/whitener.py:63-66
            raise ValueError("`alpha` must be greater than or equal to 0")
        if solver not in ("auto", "full", "randomized"):
            raise ValueError(f"Unrecognized solver '{solver}'. Valid options are 'auto', 'full', and 'randomized'.")
        self.n_modes = n_modes
        std, cos, chk = _pair(standardize), _pair(use_coslat), _pair(check_nans)
        self._params = dict(n_modes=n_modes, standardize=std, use_coslat=cos, check_nans=chk, use_pca=_pair(use_pca),
                            n_pca_modes=_pair(n_pca_modes), pca_init_rank_reduction=_pair(pca_init_rank_reduction),
                            sample_name=sample_name, feature_name=_pair(feature_name), random_state=random_state,
                            compute=compute, solver=solver, alpha=list(self.alpha))
        self.sample_name = sample_name
        self.solver_kwargs = dict(solver_kwargs)
        # CPCCA always centres (cpcca.py:145); real and imaginary parts share centring, scaling and weights
        self.pre_re = [Preprocessor(True, std[i], cos[i], chk[i]) for i in range(2)]
        self.pre_im = [Preprocessor(True, False, cos[i], chk[i]) for i in range(2)]
        self.attrs = {"model": self._model_name, "software": "xeofs_amd", "version": __version__,
                      "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
        self.ctx = None
        self.data = {}
        self.field = [None, None]

    def get_params(self):
        return dict(self._params)

    # ------------------------------------------------------------------ fit
    def fit(self, X, Y, dim, weights_X=None, weights_Y=None):
        if labelled.is_lazy(X) or labelled.is_lazy(Y):      # linalg/decomposer.py:172-177
            raise NotImplementedError("Complex data together with dask is currently not implemented. See dask issue 7639 "
                                      "https://github.com/dask/dask/issues/7639")
        return self._fit_now(X, Y, dim, weights_X, weights_Y)

    def _complex_parts(self, i, Z, dim, weights):
        """preprocess Re and Im of a complex input with the same centring / weights (as `ComplexEOF`)"""
        vals, dims, coords, name, attrs = labelled.unpack(Z)
        if not np.iscomplexobj(vals):
            warnings.warn("Expected complex-valued data but found real-valued data. For Hilbert model, use corresponding "
                          "`Hilbert` class.")
        vals = np.asarray(vals)
        re = labelled.pack(_part32(vals, False), dims, coords, name, attrs, Z)
        im = labelled.pack(_part32(vals, True),
                           dims, coords, name, attrs, Z)
        pr, pi = self.pre_re[i], self.pre_im[i]
        std_c = None
        if self._params["standardize"][i]:
            # scaler.py:105-108 on complex data: numpy's std of a complex array is the real sqrt(var Re + var Im)
            pr.standardize = False
            eps = np.finfo(np.float32).eps       # peek_std clips each part at eps: undo that, combine, clip ONCE
            s_re, s_im = (np.where(v <= eps, 0.0, v) for v in (pr.peek_std(re, dim), pi.peek_std(im, dim)))
            std_c = np.maximum(np.sqrt(s_re ** 2 + s_im ** 2), eps)
        A = pr.fit_transform(re, dim, weights, std_override=std_c)
        B = pi.fit_transform(im, dim, weights, std_override=std_c)
        return A, B, pr.total_variance + pi.total_variance

    def _dense(self, A, B):
        if A.n * A.p > MAX_DENSE:
            raise NotImplementedError(f"use_pca=False on a {A.n} x {A.p} field: the complex cross models decompose without "
                                      "PCA pre-reduction only up to n * p = %d elements; use use_pca=True" % MAX_DENSE)
        S = A.download().astype(np.float64) + 1j * B.download().astype(np.float64)

        def back(Q):
            return np.asarray(Q, dtype=np.complex64)

        def project(An, Bn):
            return An.download().astype(np.float64) + 1j * Bn.download().astype(np.float64)

        return _Field(S, back, project, (A, B), None)

    def _make_field(self, i, Z, dim, weights):
        ctx = self.ctx
        A, B, tv = self._complex_parts(i, Z, dim, weights)
        if not self._params["use_pca"][i]:
            return self._dense(A, B)
        pca = ComplexResidentPCA(ctx, self._params["n_pca_modes"][i], self._params["pca_init_rank_reduction"][i])
        pca.fit(A, B, tv)
        return _Field(pca.scores(), pca.back_project, pca.transform, (A, B), pca)

    def _fit_now(self, X, Y, dim, weights_X=None, weights_Y=None):
        self.ctx = self.ctx or engine.default_context()
        for p in self.pre_re + self.pre_im:
            p.ctx = self.ctx
        fx = self._make_field(0, X, dim, weights_X)
        fy = self._make_field(1, Y, dim, weights_Y)
        self.field = [fx, fy]
        self.sample_dims = self.pre_re[0].sample_dims
        if fx.S.shape[0] != fy.S.shape[0]:
            raise ValueError("Both data matrices must have the same number of samples but found "
                             f"{fx.S.shape[0]} in the first and {fy.S.shape[0]} in the second.")
        # fractional whitening in the analysis space (preprocessing/whitener.py:86-141; identity when alpha == 1, :46-60)
        self.T, self.Tinv, Sw = [None, None], [None, None], []
        for i, fld in enumerate((fx, fy)):
            S = fld.S
            if not (1.0 - self.alpha[i]) < np.finfo(np.float64).eps:     # whitener.py:46-60: alpha >= 1 is the identity
                n_, m_ = S.shape
                if n_ < m_:                                          # whitener.py:101-104
                    warnings.warn(f"The number of samples ({n_}) is smaller than the number of features ({m_}), leading to "
                                  "an ill-conditioned problem. This may cause unstable results. Consider using PCA to "
                                  "reduce dimensionality and stabilize the problem by setting `use_pca=True`.")
                T = _hermitian_power(np.ascontiguousarray(S.conj().T) @ S / n_, (self.alpha[i] - 1) / 2)
                try:
                    Tinv = np.linalg.inv(T)
                except np.linalg.LinAlgError:
                    Tinv = np.linalg.pinv(T)
                self.T[i], self.Tinv[i] = T, Tinv
                S = S @ T
            Sw.append(S)
        Sx, Sy = Sw
        n = Sx.shape[0]
        k = int(self.n_modes)
        C = np.ascontiguousarray(Sx.conj().T) @ Sy / (n - 1)     # cpcca.py:1008-1016
        rank = min(C.shape)
        if k > rank:
            raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {rank}).")
        if rank >= 512:      # PC-space matrices of hundreds of modes: the leading triplets on the device
            U, s, VT = _leading_svd_device(C, k, self.ctx.device)
        else:
            U, s, VT = np.linalg.svd(C, full_matrices=False)
        U, s, VT = np.ascontiguousarray(U[:, :k]), s[:k], np.ascontiguousarray(VT[:k])
        sgn = _sign_rule(VT)
        Q1, Q2 = U * sgn[None, :], (VT * sgn[:, None]).conj().T   # decomposer.py:226: V = conj(VT).T
        scores1, scores2 = Sx @ Q1, Sy @ Q2
        norm1 = np.sqrt((scores1.conj() * scores1).sum(axis=0).real)
        norm2 = np.sqrt((scores2.conj() * scores2).sum(axis=0).real)
        Cu = C if self.Tinv[1] is None else C @ self.Tinv[1]        # cpcca.py:991-1000: of the unwhitened matrices
        Cu = Cu.conj().T if self.Tinv[0] is None else Cu.conj().T @ self.Tinv[0]

        def back(fld, Q, i):      # whitener.inverse_transform_components, then pca.inverse_transform_components
            return fld.back(Q if self.Tinv[i] is None else self.Tinv[i].conj().T @ Q)

        for fld in (fx, fy):             # the resident parts are not read again (transform preprocesses new data and
            fld.free()                   # projects through the PCA's own panel): give their HBM back
        self._Su = (fx.S, fy.S)          # unwhitened analysis matrices (host): the diagnostics below work on them
        self._back = lambda i, Q: back(self.field[i], Q, i)
        self.data = dict(Q1=Q1, Q2=Q2, components1=back(fx, Q1, 0), components2=back(fy, Q2, 1), scores1=scores1,
                         scores2=scores2, singular_values=s, squared_covariance=s ** 2,
                         total_squared_covariance=float((np.abs(Cu) ** 2).sum()), norm1=norm1, norm2=norm2)
        return self

    # ------------------------------------------------------------------ accessors
    def _pre(self, which):
        return self.pre_re[which - 1]

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.pre_re[0].fields[0].like)

    def _components(self, normalized):
        c1, c2 = self.data["components1"], self.data["components2"]
        if not normalized:                                       # cpcca.py:308-316
            c1, c2 = c1 * self.data["norm1"].astype(np.float32), c2 * self.data["norm2"].astype(np.float32)
        return c1, c2

    def _scores(self, normalized):
        s1, s2 = self.data["scores1"], self.data["scores2"]
        if normalized:                                           # cpcca.py:318-329
            s1, s2 = s1 / self.data["norm1"], s2 / self.data["norm2"]
        return s1, s2

    def _wrap_components(self, c1, c2, name):
        return (self.pre_re[0].inverse_transform_components(c1, name + "_X", self.attrs),
                self.pre_re[1].inverse_transform_components(c2, name + "_Y", self.attrs))

    def _wrap_scores(self, s1, s2, name):
        return (self.pre_re[0].inverse_transform_scores(s1, name + "_X", self.attrs),
                self.pre_re[1].inverse_transform_scores(s2, name + "_Y", self.attrs))

    def components(self, normalized: bool = True):
        return self._wrap_components(*self._components(normalized), "components")

    def scores(self, normalized: bool = False):
        return self._wrap_scores(*self._scores(normalized), "scores")

    def components_amplitude(self, normalized: bool = True):
        c1, c2 = self._components(normalized)
        return self._wrap_components(np.abs(c1), np.abs(c2), "components_amplitude")

    def components_phase(self, normalized: bool = True):
        c1, c2 = self._components(normalized)
        return self._wrap_components(np.angle(c1), np.angle(c2), "components_phase")

    def scores_amplitude(self, normalized: bool = False):
        s1, s2 = self._scores(normalized)
        return self._wrap_scores(np.abs(s1), np.abs(s2), "scores_amplitude")

    def scores_phase(self, normalized: bool = False):
        s1, s2 = self._scores(normalized)
        return self._wrap_scores(np.angle(s1), np.angle(s2), "scores_phase")

    def singular_values(self):
        return self._mode_array(self.data["singular_values"], "singular_values")

    def squared_covariance(self):
        return self._mode_array(self.data["squared_covariance"], "squared_covariance")

    def total_squared_covariance(self):
        return self.data["total_squared_covariance"]

    # ---- diagnostics of cpcca.py:342-575 in complex algebra, in the n x m analysis space (a PCA basis has orthonormal
    # columns, so Frobenius norms of feature-space residuals equal those of their PC-space coordinates)
    def _rank_one_terms(self):
        """mode j: the whitened reconstruction r_j q_j^H un-whitened (whitener.inverse_transform_data: . @ T^-1) is
        r_j b_j^H with b_j = T^-H q_j"""
        B = [self.data[f"Q{i + 1}"] if self.Tinv[i] is None else self.Tinv[i].conj().T @ self.data[f"Q{i + 1}"] for i in (0, 1)]
        return self.data["scores1"], self.data["scores2"], B[0], B[1]

    @staticmethod
    def _deflated_norms(Sx, Sy, R1, R2, B1, B2, M2):
        """||(Sx - r1 b1^H)^H (Sy - r2 b2^H)||_F^2 per mode without forming Sx^H Sy: with M = Sx^H Sy the product is
        D = M - g1 b2^H - b1 g2^H + c b1 b2^H (g1 = Sx^H r2, g2 = Sy^H r1, c = r1^H r2) and ||D||^2 expands into inner
        products of n-vectors (M2 = ||M||_F^2)."""
        dot = lambda a, b: (a.conj() * b).sum(axis=0)            # column-wise <a, b>
        G1, G2 = Sx.conj().T @ R2, Sy.conj().T @ R1              # (m1 x k), (m2 x k)
        a1, a2b = Sx @ B1, Sy @ B2
        a1g, a2g = Sx @ G1, Sy @ G2
        c = dot(R1, R2)
        nb1, nb2, ng1, ng2 = dot(B1, B1).real, dot(B2, B2).real, dot(G1, G1).real, dot(G2, G2).real
        return (M2 + nb1 * ng2 + ng1 * nb2 + np.abs(c) ** 2 * nb1 * nb2
                - 2 * dot(a2g, a1).real - 2 * dot(a2b, a1g).real + 2 * (c * dot(a2b, a1)).real
                + 2 * (dot(B1, G1) * dot(B2, G2)).real - 2 * (c * nb1 * dot(B2, G2)).real - 2 * (c * dot(G1, B1) * nb2).real)

    def squared_covariance_fraction(self):
        """cpcca.py:418-512: SCF_i = 1 - ||d_X,i^H d_Y,i||_F^2 / ||X^H Y||_F^2 with d the residual of the un-whitened data
        after its reconstruction by mode i (clipped at 0) -- for every alpha; with alpha = 1 it equals sigma_i^2 / TSC."""
        Sx, Sy = self._Su
        n = Sx.shape[0]
        R1, R2, B1, B2 = self._rank_one_terms()
        M2 = self.data["total_squared_covariance"] * (n - 1) ** 2
        scf = 1 - self._deflated_norms(Sx, Sy, R1, R2, B1, B2, M2) / M2
        return self._mode_array(np.where(scf < 0, 0, scf), "squared_covariance_fraction")

    def _fve_self(self, i):
        """cpcca.py:514-639: 1 - ||S - r b^H||_F^2 / ||S||_F^2 per mode"""
        S = self._Su[i]
        R, B = self._rank_one_terms()[i], self._rank_one_terms()[2 + i]
        tot = (np.abs(S) ** 2).sum()
        res = tot - 2 * ((S @ B).conj() * R).sum(axis=0).real + (np.abs(R) ** 2).sum(0) * (np.abs(B) ** 2).sum(0)
        return 1 - res / tot

    def fraction_variance_X_explained_by_X(self):
        return self._mode_array(self._fve_self(0), "fraction_variance_X_explained_by_X")

    def fraction_variance_Y_explained_by_Y(self):
        return self._mode_array(self._fve_self(1), "fraction_variance_Y_explained_by_Y")

    @staticmethod
    def _corr(A, B):
        """cpcca.py:910-1022 method='correlation': columns divided by numpy's (real, population) std of a complex array,
        then A^H B / (n - 1)"""
        return (A / A.std(axis=0)).conj().T @ (B / B.std(axis=0)) / (A.shape[0] - 1)

    def cross_correlation_coefficients(self):
        return self._mode_array(np.diag(self._corr(self.data["scores1"], self.data["scores2"])), "cross_correlation_coefficients")

    def _mode_matrix(self, M, name):
        k = M.shape[0]
        return labelled.pack(M, ("mode_x", "mode_y"), {"mode_x": np.arange(1, k + 1), "mode_y": np.arange(1, k + 1)}, name,
                             dict(self.attrs), self.pre_re[0].fields[0].like)

    def correlation_coefficients_X(self):
        return self._mode_matrix(self._corr(self.data["scores1"], self.data["scores1"]), "correlation_coefficients_X")

    def correlation_coefficients_Y(self):
        return self._mode_matrix(self._corr(self.data["scores2"], self.data["scores2"]), "correlation_coefficients_Y")

    def covariance_fraction_CD95(self):
        """mca.py:127-189"""
        s = self.data["singular_values"]
        cf = s[0] / np.cumsum(s)
        if len(s) > 1 and (cf[-2] - cf[-1]) > 0.001:
            warnings.warn("The curent estimate of CF is sensitive to the number of modes retained. Please increase "
                          "`n_modes` for a better estimate.")
        return self._mode_array(s / s.sum(), "covariance_fraction")

    # ------------------------------------------------------------------ transform / inverse
    def transform(self, X=None, Y=None, normalized: bool = False):
        """base_model_cross_set.py:323-374 + cpcca.py:227-252"""
        if X is None and Y is None:
            raise ValueError("Either X or Y must be given.")
        out = []
        for i, Z in enumerate((X, Y)):
            if Z is None:
                continue
            vals, dims, coords, name, attrs = labelled.unpack(Z)
            vals = np.asarray(vals)
            re = labelled.pack(_part32(vals, False), dims, coords, name, attrs, Z)
            im = labelled.pack(_part32(vals, True),
                               dims, coords, name, attrs, Z)
            An, fields, vs = self.pre_re[i].transform(re)
            Bn, _, _ = self.pre_im[i].transform(im)
            S = self.field[i].project(An, Bn)
            if self.T[i] is not None:
                S = S @ self.T[i]
            S = S @ self.data[f"Q{i + 1}"]
            An.free()
            Bn.free()
            if normalized:
                S = S / self.data[f"norm{i + 1}"]
            out.append(self.pre_re[i].inverse_transform_scores(S, "scores_" + "XY"[i], self.attrs, fields, vs))
        return out[0] if len(out) == 1 else out


    def inverse_transform(self, X=None, Y=None):
        """base_model_cross_set.py:376-425 + cpcca.py:254-271: scores (with a 'mode' dimension) back to the fields --
        S conj(Q)^T in the analysis space, un-whitened, out of the PCA basis, then every part un-scaled by its own
        Scaler (the imaginary part of a Hilbert model carries no mean)."""
        if X is None and Y is None:
            raise ValueError("Either X or Y must be provided.")
        outs = []
        for i, S in enumerate((X, Y)):
            if S is None:
                continue
            pre = self.pre_re[i]
            vals, dims, coords, _, _ = labelled.unpack(S)
            if "mode" not in dims:
                vals, dims = vals[None], ("mode",) + tuple(dims)
                coords = dict(coords, mode=np.array([1]))
            modes = np.asarray(coords["mode"]).astype(int)
            order = [dims.index("mode")] + [j for j, d in enumerate(dims) if d != "mode"]
            Sm = np.transpose(vals, order).reshape(len(modes), -1).T.astype(np.complex128)      # (n', k')
            vs = ~np.isnan(Sm).all(axis=1)
            Za = Sm[vs] @ self.data[f"Q{i + 1}"][:, modes - 1].conj().T                          # analysis space (n' x m)
            rec = np.asarray(self._back(i, np.ascontiguousarray(Za.conj().T))).conj().T         # (V T^-H Za^H)^H = Za T^-1 V^H
            f0 = pre.fields[0]
            sample_shape = tuple(vals.shape[dims.index(d)] for d in f0.sample_dims)
            fields = []
            for f in pre.fields:
                g = object.__new__(type(f))
                g.__dict__.update(f.__dict__)
                g.sample_shape = sample_shape
                g.coords = dict(f.coords, **{d: coords[d] for d in f.sample_dims if d in coords})
                fields.append(g)
            re = pre.inverse_transform_data(rec.real, "reconstructed_data", fields, vs)
            if self._hilbert:      # un-scale only: inverse(Im) - inverse(0)
                im = pre.inverse_transform_data(rec.imag, "reconstructed_data", fields, vs)
                zero = pre.inverse_transform_data(np.zeros_like(rec.imag), "reconstructed_data", fields, vs)
            else:
                im = self.pre_im[i].inverse_transform_data(rec.imag, "reconstructed_data", fields, vs)
                zero = None

            def join(a, b, z):
                va, d_, c_, nm, at = labelled.unpack(a)
                vb = labelled.unpack(b)[0] - (labelled.unpack(z)[0] if z is not None else 0.0)
                return labelled.pack(va + 1j * vb, d_, c_, nm, at, a)

            if isinstance(re, list):
                outs.append([join(a, b, z) for a, b, z in zip(re, im, zero or [None] * len(re))])
            else:
                outs.append(join(re, im, zero))
        return outs[0] if len(outs) == 1 else outs


class HilbertCPCCA(ComplexCPCCA):
    """Drop-in for xeofs.cross.HilbertCPCCA (cpcca.py:1328-1500): real fields, analytic signal of the (PCA-reduced) data
    along the sample axis, optional exponential padding against spectral leakage."""

    _model_name = "Hilbert CPCCA"
    _hilbert = True

    def __init__(self, n_modes: int = 2, alpha=0.2, padding="exp", decay_factor=0.2, **kwargs):
        super().__init__(n_modes=n_modes, alpha=alpha, **kwargs)
        self._params["padding"] = _pair(padding)
        self._params["decay_factor"] = _pair(decay_factor)

    def _analytic(self, S, i):
        """utils/hilbert_transform.py:40-72 of an n x m real matrix along the samples, through the engine"""
        mat = engine.from_dense(self.ctx, np.ascontiguousarray(S, dtype=np.float32))
        im, _ = engine.hilbert(self.ctx, mat, self._params["padding"][i], float(self._params["decay_factor"][i]))
        out = S.astype(np.float64) + 1j * im.download().astype(np.float64)
        im.free()
        mat.free()
        return out

    def _make_field(self, i, Z, dim, weights):
        ctx = self.ctx
        pre = self.pre_re[i]
        pre.standardize = self._params["standardize"][i]
        mat = pre.fit_transform(Z, dim, weights)
        if not self._params["use_pca"][i]:
            im, _ = engine.hilbert(ctx, mat, self._params["padding"][i], float(self._params["decay_factor"][i]))
            return self._dense(mat, im)
        pca = ResidentPCA(ctx, self._params["n_pca_modes"][i], self._params["pca_init_rank_reduction"][i])
        pca.fit(mat, pre.total_variance)

        def back(Q):          # V Q with a real V: the two parts separately
            Q = np.asarray(Q)
            return pca.back_project(Q.real).astype(np.complex64) + 1j * pca.back_project(Q.imag)

        return _Field(self._analytic(pca.scores(), i), back, None, (mat,), pca)

    def transform(self, X=None, Y=None, normalized: bool = False):
        raise NotImplementedError("Hilbert models do not support the transform method.")


class ComplexCPCCARotator:
    """Drop-in for xeofs.cross.ComplexCPCCARotator (cross/cpcca_rotator.py:472-534 over :20-420): Varimax
    (power = 1) / Promax rotation of a fitted complex cross model.  The stacked complex feature-space loadings [Qx; Qy] sqrt(s)
    are rotated on the device as one [Re | Im] panel (`rotation.cpromax_panel`); their images in the
    analysis space are Q sqrt(s) rotation_matrix -- k x k algebra (as in `CPCCARotator`)."""

    _model_name = "Rotated Complex CPCCA"
    _hilbert = False

    def __init__(self, n_modes: int = 10, power: int = 1, max_iter: int | None = None, rtol: float = 1e-8,
                 compute: bool = True):
        if max_iter is None:
            max_iter = 1000 if compute else 100
        self._params = dict(n_modes=n_modes, power=power, max_iter=max_iter, rtol=rtol, compute=compute)
        self.attrs = {"model": self._model_name}
        self.attrs.update(self._params)
        self.attrs.update({"software": "xeofs_amd", "version": __version__,
                           "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")})
        self.data, self.model_data = {}, {}
        self.sorted = False

    def get_params(self):
        return dict(self._params)

    def _rot_mat_inv_trans(self, R):
        return np.linalg.inv(R).conj().T if self._params["power"] > 1 else R

    def fit(self, model):
        """cpcca_rotator.py:122-263 (+ the post-compute sort by squared covariance) with complex loadings"""
        getattr(model, "compute", lambda: None)()      # a deferred fit runs now: ctx / preprocessor / data are read below
        from .. import rotation

        torch = engine._torch()
        self.model = model
        self.ctx = model.ctx
        self.pre = model.pre_re
        k = int(self._params["n_modes"])
        s = np.asarray(model.data["singular_values"], dtype=np.float64)[:k]
        k = s.size
        scaling = np.sqrt(s)
        C1 = np.asarray(model.data["components1"])[:, :k]
        C2 = np.asarray(model.data["components2"])[:, :k]
        p1 = C1.shape[0]
        Xrot, ptot, k, rot_matrix, phi = rotation.cpromax_panel(self.ctx, np.concatenate([C1, C2], axis=0),
                                                                power=self._params["power"], max_iter=self._params["max_iter"],
                                                                rtol=self._params["rtol"], col_scale=scaling)
        Qr = [model.data[f"Q{i + 1}"][:, :k] * scaling @ rot_matrix for i in range(2)]
        norm1, norm2 = np.linalg.norm(Qr[0], axis=0), np.linalg.norm(Qr[1], axis=0)
        sqcov = (norm1 * norm2) ** 2
        idx = np.argsort(sqcov)[::-1]
        RinvT = self._rot_mat_inv_trans(rot_matrix)
        sc1 = (np.asarray(model.data["scores1"])[:, :k] / scaling) @ RinvT * norm1
        sc2 = (np.asarray(model.data["scores2"])[:, :k] / scaling) @ RinvT * norm2
        # sign rule on the stacked rotated loadings (xarray_utils.py:273-301; numpy's lexicographic complex max / min)
        CH = Xrot.shape[1] // 2
        amax, amin = engine.panel_colargminmax(self.ctx, Xrot, ptot)
        cols = torch.arange(k, device=Xrot.device)
        pick = lambda ix: (Xrot[ix[:k], cols].double().cpu().numpy(), Xrot[ix[:k], cols + CH].double().cpu().numpy())
        (mr, mi), (nr, ni) = pick(amax), pick(amin)
        sign = np.where(np.hypot(mr, mi) >= np.hypot(nr, ni), 1.0, -1.0)
        F = []
        for norm, lo, hi in ((norm1, 0, p1), (norm2, p1, ptot)):
            M = np.zeros((k, k), dtype=complex)
            M[idx, np.arange(k)] = sign[idx] / norm[idx]
            blk = engine.panel_matmul(self.ctx, Xrot[lo:], rotation._dev(rotation._cembed(M, CH), Xrot))[:hi - lo].cpu().numpy()
            c = np.empty((hi - lo, k), np.complex64)
            c.real, c.imag = blk[:, :k], blk[:, CH:CH + k]
            F.append(c)
        del Xrot
        self.model_data = dict(singular_values=np.asarray(model.data["singular_values"]), components1=C1, components2=C2)
        self.data = dict(
            components1=F[0], components2=F[1],
            scores1=(sc1 * sign)[:, idx].astype(np.complex64), scores2=(sc2 * sign)[:, idx].astype(np.complex64),
            squared_covariance=sqcov[idx], total_squared_covariance=model.data["total_squared_covariance"],
            idx_modes_sorted=idx, norm1=norm1[idx], norm2=norm2[idx], rotation_matrix=rot_matrix, phi_matrix=phi,
            modes_sign=sign[idx],
        )
        self.sorted = True
        return self

    # ------------------------------------------------------------------ transform (cpcca_rotator.py:282-372)
    def transform(self, X=None, Y=None, normalized: bool = False):
        if self._hilbert:
            raise NotImplementedError("Hilbert models do not support the transform method.")
        if X is None and Y is None:
            raise ValueError("No data provided. Please provide X and/or Y.")
        k = self.data["norm1"].size
        RinvT = self._rot_mat_inv_trans(self.data["rotation_matrix"])
        scaling = np.sqrt(np.asarray(self.model_data["singular_values"], dtype=np.float64)[:k])
        outs = []
        for which, Z in ((1, X), (2, Y)):
            if Z is None:
                continue
            un = self.model.transform(**{"XY"[which - 1]: Z})          # unrotated scores: data . back-projected components
            vals, dims, coords, _, _ = labelled.unpack(un)
            kk = vals.shape[0]
            S = np.asarray(vals).reshape(kk, -1).T[:, :k]
            ok = ~np.isnan(S).all(axis=1)
            proj = np.full(S.shape, np.nan, dtype=complex)
            proj[ok] = (S[ok] / scaling) @ RinvT
            proj = proj[:, self.data["idx_modes_sorted"]] * self.data["modes_sign"]
            if not normalized:
                proj = proj * self.data[f"norm{which}"]
            coords = dict(coords, mode=np.arange(1, k + 1))
            outs.append(labelled.pack(proj.T.reshape((k,) + vals.shape[1:]), dims, coords, f"scores{which}", dict(self.attrs), un))
        return outs[0] if len(outs) == 1 else outs

    # ------------------------------------------------------------------ accessors
    def _components(self, normalized):
        q1, q2 = self.data["components1"], self.data["components2"]
        if not normalized:
            q1, q2 = q1 * self.data["norm1"].astype(np.float32), q2 * self.data["norm2"].astype(np.float32)
        return q1, q2

    def _scores(self, normalized):
        s1, s2 = self.data["scores1"], self.data["scores2"]
        if normalized:
            s1, s2 = s1 / self.data["norm1"].astype(np.float32), s2 / self.data["norm2"].astype(np.float32)
        return s1, s2

    def _wc(self, c1, c2, name):
        return (self.pre[0].inverse_transform_components(c1, name + "1", self.attrs),
                self.pre[1].inverse_transform_components(c2, name + "2", self.attrs))

    def _ws(self, s1, s2, name):
        return (self.pre[0].inverse_transform_scores(s1, name + "1", self.attrs),
                self.pre[1].inverse_transform_scores(s2, name + "2", self.attrs))

    def components(self, normalized: bool = True):
        return self._wc(*self._components(normalized), "components")

    def scores(self, normalized: bool = False):
        return self._ws(*self._scores(normalized), "scores")

    def components_amplitude(self, normalized: bool = True):
        c1, c2 = self._components(normalized)
        return self._wc(np.abs(c1), np.abs(c2), "components_amplitude")

    def components_phase(self, normalized: bool = True):
        c1, c2 = self._components(normalized)
        return self._wc(np.angle(c1), np.angle(c2), "components_phase")

    def scores_amplitude(self, normalized: bool = False):
        s1, s2 = self._scores(normalized)
        return self._ws(np.abs(s1), np.abs(s2), "scores_amplitude")

    def scores_phase(self, normalized: bool = False):
        s1, s2 = self._scores(normalized)
        return self._ws(np.angle(s1), np.angle(s2), "scores_phase")

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.pre[0].fields[0].like)

    def squared_covariance(self):
        return self._mode_array(self.data["squared_covariance"], "squared_covariance")

    def squared_covariance_fraction(self):
        return self._mode_array(self.data["squared_covariance"] / self.data["total_squared_covariance"],
                                "squared_covariance_fraction")

    def rotation_matrix(self):
        return self.data["rotation_matrix"]

    def phi_matrix(self):
        return self.data["phi_matrix"]

    def fit_transform(self, *a, **k):
        raise NotImplementedError("The fit_transform method is not implemented for the rotator classes.")


class HilbertCPCCARotator(ComplexCPCCARotator):
    """Drop-in for xeofs.cross.HilbertCPCCARotator (cpcca_rotator.py:536-600); `transform` is not implemented there either."""

    _model_name = "Rotated Hilbert CPCCA"
    _hilbert = True


class ComplexMCARotator(ComplexCPCCARotator):
    """Drop-in for xeofs.cross.ComplexMCARotator (cross/mca_rotator.py:77-150)."""

    _model_name = "Rotated Complex MCA"


class HilbertMCARotator(HilbertCPCCARotator):
    """Drop-in for xeofs.cross.HilbertMCARotator (cross/mca_rotator.py:153-230)."""

    _model_name = "Rotated Hilbert MCA"


# --------------------------------------------------------------------------------------------------------------------
# the fixed-alpha children (cross/mca.py, cca.py, rda.py) and the rotator names of cross/cpcca_rotator.py, mca_rotator.py
# --------------------------------------------------------------------------------------------------------------------
def _fixed_alpha(base, alpha, name, doc):
    def __init__(self, n_modes: int = 2, **kwargs):
        kwargs.pop("alpha", None)
        base.__init__(self, n_modes=n_modes, alpha=alpha, **kwargs)
        self._params.pop("alpha", None)          # hard-coded for these classes (mca.py:120-123)
        self.attrs["model"] = name

    return type(name.replace(" ", ""), (base,), {"__init__": __init__, "__doc__": doc, "_model_name": name})


ComplexMCA = _fixed_alpha(ComplexCPCCA, [1.0, 1.0], "Complex MCA", "Drop-in for xeofs.cross.ComplexMCA (cross/mca.py:224-338).")
ComplexCCA = _fixed_alpha(ComplexCPCCA, [0.0, 0.0], "Complex CCA", "Drop-in for xeofs.cross.ComplexCCA (cross/cca.py:123-237).")
ComplexRDA = _fixed_alpha(ComplexCPCCA, [0.0, 1.0], "Complex RDA", "Drop-in for xeofs.cross.ComplexRDA (cross/rda.py:123-237).")
HilbertMCA = _fixed_alpha(HilbertCPCCA, [1.0, 1.0], "Hilbert MCA", "Drop-in for xeofs.cross.HilbertMCA (cross/mca.py:340-489).")
HilbertCCA = _fixed_alpha(HilbertCPCCA, [0.0, 0.0], "Hilbert CCA", "Drop-in for xeofs.cross.HilbertCCA (cross/cca.py:239-355).")
HilbertRDA = _fixed_alpha(HilbertCPCCA, [0.0, 1.0], "Hilbert RDA", "Drop-in for xeofs.cross.HilbertRDA (cross/rda.py:239-355).")
