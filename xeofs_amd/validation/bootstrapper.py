"""xeofs_amd.validation.EOFBootstrapper -- drop-in for xeofs.validation.EOFBootstrapper
(xeofs/validation/bootstrapper.py:40-135): refit the EOF model `n_bootstraps` times on rows of the
preprocessed data drawn with replacement.

The fitted model's matrix X is already resident in HBM and a member never copies it: rows drawn with replacement and
re-centred are  X_b = H X  with the n x n matrix  H = G - 1 c^T / n  (G the row selector, c the draw counts), so both
products of the member's randomized SVD run on the ORIGINAL matrix with the small sample-side panel transformed,
    X_b^T Z = X^T (H^T Z),      X_b Y = H (X Y)
(`BootstrapOps`: a deterministic segment sum and a row gather on n x 64 panels), the member's total variance is
(c . |x_r|^2 - n |m_b|^2) / (n - 1) from the row norms of X (once) and one extra product X^T c, and
`bst_model.transform(input_data)` = (X - 1 m_b^T) V_b = X V_b - c^T (X V_b) / n.  A member therefore costs its 16
passes over the field plus one, and bootstrapping an in-place model keeps HBM at 1x the field when it has to (where
one more copy fits, the sample-contiguous layout is built once for all members: `ensure_sample_layout`; the earlier
`eofx_resample_f32` route -- gather inside the statistics / apply kernels into a second two-layout matrix -- remains
for callers that want the resampled matrix itself).
The resampling indices come from `np.random.default_rng(seed).choice(n, n, replace=True)` exactly as in the
reference, so a seed selects the same bootstrap members.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled
from ..sharded import HipPanelOps, sharded_rsvd
from ..single.eof import EOF


class _Solo:
    """the panel-level driver's communicator for one rank (a member never communicates, whatever the process group)"""
    active, rank, world = False, 0, 1

    def sum_(self, t):
        return t

    max_ = min_ = sum_


class BootstrapOps(HipPanelOps):
    """Panel products of the bootstrap member X_b = H X on the resident X (H = G - 1 c^T / n, see the module docstring).
    Sample-side panels are indexed by draw; H and H^T are applied by two small HIP kernels (`eofx_panel_bootstrap_f32`:
    row gather / segment sums over the sorted draw + a rank-one term, float64 sums in draw order, no atomics), so a member
    is reproducible bit for bit ON A GIVEN LAYOUT of the matrix and its trace holds no library GEMM.  (The in-place route
    maps the field with a fused multiply-add, the sample-contiguous copy was written with a multiply: one ulp apart when the
    scale is not a power of two.  The bootstrapper builds that copy only where HBM has room, and releases it afterwards.)"""

    def __init__(self, ctx, mat, idx):
        super().__init__(ctx, mat)
        torch = engine._torch()
        dev = f"cuda:{ctx.device}"
        n = mat.n
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        order = np.argsort(idx, kind="stable").astype(np.int64)       # the draws of every source row, in draw order
        counts = np.bincount(idx, minlength=n)
        rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.idx = torch.as_tensor(idx, device=dev)
        self.order = torch.as_tensor(order, device=dev)
        self.rowptr = torch.as_tensor(rowptr, device=dev)
        self.counts = torch.as_tensor(counts, device=dev).double()

    def _ht(self, Zn):            # H^T Z = G^T Z - c (1^T Z) / n : segment sums over the sorted draw + a rank-one term
        return engine.panel_bootstrap(self.ctx, Zn, self.n, self.idx, self.order, self.rowptr, True)

    def _h(self, Wn):             # H W = W[idx] - 1 (c^T W) / n : a row gather + a rank-one term
        return engine.panel_bootstrap(self.ctx, Wn, self.n, self.idx, self.order, self.rowptr, False)

    def tmul(self, Zn, final=False):
        return super().tmul(self._ht(Zn), final)

    def mul(self, Yp, final=False):
        return self._h(super().mul(Yp, final))

    def mean_sumsq(self):
        """|m_b|^2 = |X^T c|^2 / n^2 (one product with c in the first column of a 32-wide panel)"""
        torch = engine._torch()
        Z = torch.zeros((self.n_pad, 32), dtype=torch.float32, device=self.idx.device)
        Z[:self.n, 0] = self.counts.float()
        Y = super().tmul(Z, True)
        return float(self.gram(Y)[0, 0]) / float(self.n) ** 2


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

    def fit(self, model: EOF, random_state=None, sample_layout="auto"):
        """`random_state` seeds the members' randomized SVDs (the reference leaves them unseeded).
        `sample_layout`: "auto" builds the sample-contiguous copy of the model's matrix for the members' X Y passes where HBM
        has room (13 % faster) and releases it afterwards; True / False force the choice.  With a seed the members are
        reproducible bit for bit FOR A GIVEN CHOICE; between the two layouts the Scaler map is a multiply or a fused
        multiply-add, i.e. results agree to one float32 ulp of the field (pass True / False where runs on machines
        with different free memory must agree bit for bit)."""
        getattr(model, "compute", lambda: None)()      # a deferred fit runs now: ctx / preprocessor / data are read below
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
        r2 = engine.sample_norms(ctx, mat) ** 2                        # |x_r|^2, once
        built_here = False
        if n_boot >= 2:      # every member is a full decomposition of the same matrix: where HBM has room for the
            had = mat.has_sample_layout()                  # sample-contiguous layout, its X Y passes run 13 % faster over it
            if sample_layout == "auto":
                built_here = mat.ensure_sample_layout(only_if_room=True) and not had
            elif sample_layout:
                built_here = mat.ensure_sample_layout(only_if_room=False) and not had
        comm = _Solo()
        try:
            for b in range(n_boot):
                idx = rng.choice(n, n, replace=True)                       # bootstrapper.py:79
                ops = BootstrapOps(ctx, mat, idx)
                U, s, V = sharded_rsvd(ops, comm, k, p, 0, random_state=None if random_state is None else random_state + b)
                s64 = s.astype(np.float64)
                expvar[b] = s64 ** 2 / (n - 1)                             # eof.py:104
                c = ops.counts.cpu().numpy()
                totvar[b] = (float(c @ r2) - n * ops.mean_sumsq()) / (n - 1)
                comps[b] = V
                # bst_model.transform(input_data): centre with the member's mean, project (eof.py:123-132)
                proj = engine.project(ctx, mat, V).astype(np.float64)
                scores[b] = proj - (c @ proj) / n
        finally:
            if built_here:   # the model's matrix goes back to the footprint it had; the layout is rebuilt on demand
                mat.release_sample_layout()
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
