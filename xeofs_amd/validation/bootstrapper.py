"""xeofs_amd.validation.EOFBootstrapper -- drop-in for xeofs.validation.EOFBootstrapper
(xeofs/validation/bootstrapper.py:40-135): refit the EOF model `n_bootstraps` times on rows of the
preprocessed data drawn with replacement.

The fitted model's matrix is already resident in HBM, so a bootstrap member is a row gather inside the
statistics / apply kernels (`eofx_resample_f32`: re-centre + both layouts, no host round trip), the usual
randomized SVD, and one projection of the *original* resident matrix on the member's components
(`bst_model.transform(input_data)` = (X - 1 mean_b^T) V_b = X V_b - mean_b^T V_b).
The resampling indices come from `np.random.default_rng(seed).choice(n, n, replace=True)` exactly as in the
reference, so a seed selects the same bootstrap members.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled
from ..single.eof import EOF


class EOFBootstrapper(EOF):
    def __init__(self, n_bootstraps: int = 20, seed=None):
        self._params = {"n_bootstraps": n_bootstraps, "seed": seed}
        self.attrs = {"model": "Bootstrapped EOF analysis"}
        self.attrs.update(self._params)
        self.attrs.update({"software": "xeofs_amd", "version": __version__,
                           "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")})
        self.ctx = None
        self.data = {}

    def get_params(self):
        return dict(self._params)

    def fit(self, model: EOF, random_state=None):
        """`random_state` seeds the members' randomized SVDs (the reference leaves them unseeded)."""
        self.model = model
        self.ctx = ctx = model.ctx
        self.preprocessor = model.preprocessor
        self.sample_name, self.feature_name = model.sample_name, model.feature_name
        self.sample_dims = getattr(model, "sample_dims", None)
        mat = model.data["input_data"]
        n, p = mat.n, mat.p
        k = int(np.asarray(model.data["components"]).shape[1])     # the fitted number of modes (n_modes may be a variance target)
        n_boot = int(self._params["n_bootstraps"])
        rng = np.random.default_rng(self._params["seed"])
        expvar = np.empty((n_boot, k))
        totvar = np.empty(n_boot)
        comps = np.empty((n_boot, p, k), np.float32)
        scores = np.empty((n_boot, n, k), np.float32)
        for b in range(n_boot):
            idx = rng.choice(n, n, replace=True)                       # bootstrapper.py:79
            bmat, mean_b, tv = engine.resample(ctx, mat, idx, center=True)
            U, s, V = engine.rsvd(ctx, bmat, k, random_state=None if random_state is None else random_state + b)
            bmat.free()
            s64 = s.astype(np.float64)
            expvar[b] = s64 ** 2 / (n - 1)                             # eof.py:104
            totvar[b] = tv
            comps[b] = V
            # bst_model.transform(input_data): centre with the member's mean, project (eof.py:123-132)
            proj = engine.project(ctx, mat, V).astype(np.float64) - mean_b @ V.astype(np.float64)
            scores[b] = proj
        # sign of each member's modes from the correlation with the model's scores (bootstrapper.py:112-121)
        ms = np.asarray(model.data["scores"], dtype=np.float64)[:, :k]
        sc = scores.astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            corr = (sc * ms).mean(axis=1) / sc.std(axis=1) / ms.std(axis=0)
        signs = np.sign(corr)                                          # (n_boot, k)
        comps *= signs[:, None, :].astype(np.float32)
        scores *= signs[:, None, :].astype(np.float32)
        self.data = dict(input_data=mat, components=comps, scores=scores, norms=np.asarray(model.data["norms"]),
                         explained_variance=expvar, total_variance=totvar)
        return self

    # accessors: the bootstrap axis "n" leads (xr.concat(..., dim="n"), bootstrapper.py:100-110)
    def components(self, normalized: bool = True):
        out = []
        for b in range(self.data["components"].shape[0]):
            V = self.data["components"][b]
            if not normalized:
                V = V * self.data["norms"].astype(V.dtype)
            out.append(self.preprocessor.inverse_transform_components(V, "components", self.attrs))
        return labelled.concat(out, "n", np.arange(1, len(out) + 1))

    def scores(self, normalized: bool = False):
        out = []
        for b in range(self.data["scores"].shape[0]):
            S = self.data["scores"][b]
            if normalized:
                S = S / self.data["norms"].astype(S.dtype)
            out.append(self.preprocessor.inverse_transform_scores(S, "scores", self.attrs))
        return labelled.concat(out, "n", np.arange(1, len(out) + 1))

    def explained_variance(self):
        ev = self.data["explained_variance"]
        return labelled.pack(ev, ("n", "mode"), {"n": np.arange(1, ev.shape[0] + 1), "mode": np.arange(1, ev.shape[1] + 1)},
                             "explained_variance", dict(self.attrs), self.preprocessor.fields[0].like)

    def total_variance(self):
        tv = self.data["total_variance"]
        return labelled.pack(tv, ("n",), {"n": np.arange(1, tv.shape[0] + 1)}, "total_variance", dict(self.attrs),
                             self.preprocessor.fields[0].like)

    def explained_variance_ratio(self):
        ev, tv = self.data["explained_variance"], self.data["total_variance"]
        return labelled.pack(ev / tv[:, None], ("n", "mode"),
                             {"n": np.arange(1, ev.shape[0] + 1), "mode": np.arange(1, ev.shape[1] + 1)},
                             "explained_variance_ratio", dict(self.attrs), self.preprocessor.fields[0].like)
