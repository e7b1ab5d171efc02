"""xeofs_amd.cross.MCA -- drop-in for xeofs.cross.MCA (xeofs/cross/mca.py:88-123 =
CPCCA with alpha=[1,1]; fit xeofs/cross/base_model_cross_set.py:269-321, algorithm
xeofs/cross/cpcca.py:168-225).

The cross-covariance matrix C = X^T Y/(n-1) is never materialised: the engine applies it as a
matrix-free operator X^T (Y .) inside the randomized SVD (`eofx_crosscov_rsvd_f32`).

With `use_pca=True` (the reference default, base_model_cross_set.py:165-179, 307-308) each field is
first reduced to the principal components that explain `n_pca_modes` (99.9 %) of its variance
(`xeofs_amd.pca.ResidentPCA`: exact, two wide passes over the resident matrix); the cross-covariance
analysis then runs on the n x m score matrices and the singular vectors are projected back (V Q).
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled
from ..pca import ResidentPCA
from ..linalg.decomposer import MAX_SKETCH, sanity_check_n_modes
from ..preprocessing import Preprocessor


def _pair(v):
    return list(v) if isinstance(v, (list, tuple)) else [v, v]


class MCA:
    def __init__(self, n_modes: int = 2, standardize=False, use_coslat=False, check_nans=True, use_pca=True,
                 n_pca_modes=0.999, pca_init_rank_reduction=0.3, compute: bool = True, sample_name: str = "sample",
                 feature_name="feature", solver: str = "auto", random_state=None, solver_kwargs: dict = {}):
        sanity_check_n_modes(n_modes)
        if solver not in ("auto", "full", "randomized"):
            raise ValueError(f"Unrecognized solver '{solver}'. Valid options are 'auto', 'full', and 'randomized'.")
        self.n_modes = n_modes
        std, cos, chk = _pair(standardize), _pair(use_coslat), _pair(check_nans)
        self._params = dict(n_modes=n_modes, standardize=std, use_coslat=cos, check_nans=chk, use_pca=_pair(use_pca),
                            n_pca_modes=_pair(n_pca_modes), pca_init_rank_reduction=_pair(pca_init_rank_reduction),
                            sample_name=sample_name, feature_name=_pair(feature_name), random_state=random_state,
                            compute=compute, solver=solver)
        self.solver, self.random_state, self.solver_kwargs = solver, random_state, dict(solver_kwargs)
        # CPCCA always centres (cpcca.py:145)
        self.preprocessor1 = Preprocessor(True, std[0], cos[0], chk[0])
        self.preprocessor2 = Preprocessor(True, std[1], cos[1], chk[1])
        self.attrs = {"model": "Maximum Covariance Analysis", "software": "xeofs_amd", "version": __version__,
                      "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
        self.ctx = None
        self.data = {}

    def fit(self, X, Y, dim, weights_X=None, weights_Y=None):
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor1.ctx = self.preprocessor2.ctx = self.ctx
        mx = self.preprocessor1.fit_transform(X, dim, weights_X)
        my = self.preprocessor2.fit_transform(Y, dim, weights_Y)
        self.sample_dims = self.preprocessor1.sample_dims
        k = int(self.n_modes)
        kw = dict(self.solver_kwargs)
        n_over, n_iter = int(kw.pop("n_oversamples", 10)), kw.pop("n_iter", "auto")
        # PCA pre-reduction (base_model_cross_set.py:307-308): the analysis runs on the PC scores
        self.pca = [None, None]
        work = [mx, my]
        for i, (mat, pre) in enumerate(((mx, self.preprocessor1), (my, self.preprocessor2))):
            if self._params["use_pca"][i]:
                pca = ResidentPCA(self.ctx, self._params["n_pca_modes"][i], self._params["pca_init_rank_reduction"][i])
                pca.fit(mat, pre.total_variance)
                self.pca[i] = pca
                work[i] = engine.from_dense(self.ctx, pca.scores().astype(np.float32))
        wx, wy = work
        rank = min(wx.p, wy.p)
        if k > rank:
            raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {rank}).")
        small = max(wx.p, wy.p) < 500
        if self.solver == "full" or (self.solver == "auto" and small and k > int(0.8 * rank)):
            if rank > MAX_SKETCH:
                raise NotImplementedError(f"solver='full' on the cross path needs rank <= {MAX_SKETCH}; use 'randomized'")
            n_over, n_iter = rank - k, 0
        n_over = min(n_over, rank - k)         # a sketch as wide as the rank is already exact
        out = engine.crosscov_rsvd(self.ctx, wx, wy, k, n_over, n_iter, random_state=self.random_state)
        s = out["s"].astype(np.float64)
        self._q = [out["Q1"].astype(np.float64), out["Q2"].astype(np.float64)]     # singular vectors in PC space
        comps = [self.pca[i].back_project(self._q[i]) if self.pca[i] is not None else out[f"Q{i + 1}"]
                 for i in range(2)]
        for i in range(2):
            if self.pca[i] is not None:
                work[i].free()
        self.data = dict(
            input_data1=mx, input_data2=my, components1=comps[0], components2=comps[1],
            scores1=out["scores1"], scores2=out["scores2"], singular_values=s, squared_covariance=s ** 2,
            total_squared_covariance=out["total_squared_covariance"], idx_modes_sorted=np.argsort(s)[::-1],
            norm1=out["norm1"].astype(np.float64), norm2=out["norm2"].astype(np.float64),
        )
        return self

    # accessors (base_model_cross_set.py:465-523): 2-tuples
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

    def transform(self, X=None, Y=None, normalized: bool = False):
        """base_model_cross_set.py:323-374: project new data onto the fitted singular vectors."""
        if X is None and Y is None:
            raise ValueError("Either X or Y must be provided.")
        outs = []
        for which, Z in ((1, X), (2, Y)):
            if Z is None:
                continue
            pre = self.preprocessor1 if which == 1 else self.preprocessor2
            mat, fields, vs = pre.transform(Z)
            pca = self.pca[which - 1]
            if pca is not None:     # pca.transform -> X V (pca.py:125-134), then the PC-space singular vectors
                proj = (pca.transform(mat) @ self._q[which - 1]).astype(np.float32)
            else:
                proj = engine.project(self.ctx, mat, self.data[f"components{which}"])
            mat.free()
            if normalized:
                proj = proj / self.data[f"norm{which}"].astype(proj.dtype)
            outs.append(pre.inverse_transform_scores(proj, f"scores{which}", self.attrs, fields, vs))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.preprocessor1.fields[0].like)

    def singular_values(self):
        return self._mode_array(self.data["singular_values"], "singular_values")

    def squared_covariance(self):
        return self._mode_array(self.data["squared_covariance"], "squared_covariance")

    def squared_covariance_fraction(self):
        return self._mode_array(self.data["squared_covariance"] / self.data["total_squared_covariance"],
                                "squared_covariance_fraction")

    def total_squared_covariance(self):
        return self.data["total_squared_covariance"]
