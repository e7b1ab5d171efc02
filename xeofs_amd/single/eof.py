"""xeofs_amd.single.EOF -- drop-in for xeofs.single.EOF (xeofs/single/eof.py:17-240,
xeofs/single/base_model_single_set.py:30-336): same constructor, fit / transform /
inverse_transform / components / scores / singular_values / explained_variance(_ratio).

Everything numeric runs in the HIP engine: fused preprocess -> resident matrix -> randomized SVD.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled
from ..linalg.decomposer import Decomposer
from ..preprocessing import Preprocessor


class EOF:
    def __init__(self, n_modes: int = 2, center: bool = True, standardize: bool = False, use_coslat: bool = False,
                 check_nans=True, sample_name: str = "sample", feature_name: str = "feature", compute: bool = True,
                 random_state: int | None = None, solver: str = "auto", solver_kwargs: dict = {}, **kwargs):
        self.n_modes = n_modes
        self.sample_name, self.feature_name = sample_name, feature_name
        self._params = dict(n_modes=n_modes, center=center, standardize=standardize, use_coslat=use_coslat,
                            check_nans=check_nans, sample_name=sample_name, feature_name=feature_name,
                            random_state=random_state, compute=compute, solver=solver)
        self._solver_kwargs = dict(solver_kwargs)
        self._decomposer_kwargs = dict(n_modes=n_modes, solver=solver, random_state=random_state, compute=compute,
                                       component_dim_name="mode", solver_kwargs=solver_kwargs, **kwargs)
        self.ctx = None
        self.preprocessor = Preprocessor(center, standardize, use_coslat, check_nans)
        # attrs as the reference stores them (base_model.py:38-46): bools/None stringified
        self.attrs = {"model": "EOF analysis", "software": "xeofs_amd", "version": __version__,
                      "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
        self.attrs.update({k: (str(v) if isinstance(v, bool) or v is None else v) for k, v in self._params.items()})
        self.data = {}

    def get_params(self):
        return dict(self._params)

    # ------------------------------------------------------------------ fit
    def fit(self, X, dim, weights=None):
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor.ctx = self.ctx
        mat = self.preprocessor.fit_transform(X, dim, weights)      # fused HIP preprocess
        self.sample_dims = self.preprocessor.sample_dims
        return self._fit_algorithm(mat)

    def _fit_algorithm(self, mat):
        """xeofs/single/eof.py:85-118."""
        total_variance = self.preprocessor.total_variance                     # eof.py:93
        dec = Decomposer(ctx=self.ctx, **self._decomposer_kwargs)
        dec.fit(mat, dims=(self.sample_name, self.feature_name), total_variance=total_variance)
        s = dec.s_.astype(np.float64)
        n_samples = mat.n
        self.data = dict(
            input_data=mat,                                                   # stays resident in HBM
            components=dec.V_, scores=dec.U_ * dec.s_, norms=s,
            explained_variance=s ** 2 / (n_samples - 1), total_variance=total_variance,
        )
        return self

    # ------------------------------------------------------------------ transform / inverse
    def transform(self, X, normalized: bool = False):
        """base_model_single_set.py:180-203 + eof.py:123-132: preprocess with the fitted state, X V."""
        mat, fields, vs = self.preprocessor.transform(X)
        proj = engine.project(self.ctx, mat, self.data["components"])
        mat.free()
        if normalized:
            proj = proj / self.data["norms"].astype(proj.dtype)
        return self.preprocessor.inverse_transform_scores(proj, "scores", self.attrs, fields, vs)

    def fit_transform(self, X, dim, weights=None, **kwargs):
        return self.fit(X, dim, weights).transform(X, **kwargs)

    def inverse_transform(self, scores, normalized: bool = False):
        """base_model_single_set.py:205-286 + eof.py:134-156: Xhat = scores . conj(V)^T, then un-scale."""
        vals, dims, coords, _, _ = labelled.unpack(scores)
        if "mode" not in dims:
            raise ValueError("scores must have a 'mode' dimension")
        modes = np.asarray(coords["mode"]).astype(int)
        order = [dims.index("mode")] + [i for i, d in enumerate(dims) if d != "mode"]
        S = np.transpose(vals, order).reshape(len(modes), -1).T            # (n_samples, k')
        vs = ~np.isnan(S).all(axis=1)
        S = np.ascontiguousarray(S[vs], dtype=np.float32)
        if normalized:
            S = S * self.data["norms"][modes - 1].astype(np.float32)
        V = np.ascontiguousarray(self.data["components"][:, modes - 1])
        rec = engine.reconstruct(self.ctx, S, V)
        f0 = self.preprocessor.fields[0]
        sample_shape = tuple(vals.shape[dims.index(d)] for d in f0.sample_dims)
        fields = []
        for f in self.preprocessor.fields:
            g = object.__new__(type(f))
            g.__dict__.update(f.__dict__)
            g.sample_shape = sample_shape
            g.coords = dict(f.coords, **{d: coords[d] for d in f.sample_dims if d in coords})
            fields.append(g)
        return self.preprocessor.inverse_transform_data(rec, "reconstructed_data", fields, vs)

    # ------------------------------------------------------------------ accessors
    def components(self, normalized: bool = True):
        V = self.data["components"]
        if not normalized:
            V = V * self.data["norms"].astype(V.dtype)
        return self.preprocessor.inverse_transform_components(V, "components", self.attrs)

    def scores(self, normalized: bool = False):
        S = self.data["scores"]
        if normalized:
            S = S / self.data["norms"].astype(S.dtype)
        return self.preprocessor.inverse_transform_scores(S, "scores", self.attrs)

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.preprocessor.fields[0].like)

    def singular_values(self):
        return self._mode_array(self.data["norms"], "norms")

    def explained_variance(self):
        return self._mode_array(self.data["explained_variance"], "explained_variance")

    def explained_variance_ratio(self):
        return self._mode_array(self.data["explained_variance"] / self.data["total_variance"],
                                "explained_variance_ratio")
