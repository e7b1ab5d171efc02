"""xeofs_amd.cross.CPCCARotator / MCARotator -- drop-ins for xeofs.cross.CPCCARotator
(xeofs/cross/cpcca_rotator.py:20-420) and MCARotator (cross/mca_rotator.py:5-75): Varimax (power=1) /
Promax rotation of a fitted cross model.

The rotation acts on the stacked feature-space loadings [Qx; Qy] sqrt(s), a ((p1 + p2) x k) panel that is
rotated on the GPU (`xeofs_amd.rotation.promax`, fused step kernel).  Because the rotated loadings are
`loadings @ rotation_matrix`, their images in the analysis space (pca / whitener `transform_components`,
cpcca_rotator.py:186-189) are Q sqrt(s) rotation_matrix -- k x k algebra, no second trip through V.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled, rotation


class CPCCARotator:
    _model_name = "Rotated CPCCA"

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
        """cpcca_rotator.py:122-263 (+ the post-compute sort by squared covariance)."""
        getattr(model, "compute", lambda: None)()      # a deferred fit runs now: ctx / preprocessor / data are read below
        self.model = model
        self.ctx = model.ctx
        self.preprocessor1, self.preprocessor2 = model.preprocessor1, model.preprocessor2
        self.sample_name = model.sample_name
        k = int(self._params["n_modes"])
        s = np.asarray(model.data["singular_values"], dtype=np.float64)[:k]
        k = s.size
        scaling = np.sqrt(s)
        C1 = np.asarray(model.data["components1"])[:, :k]
        C2 = np.asarray(model.data["components2"])[:, :k]
        p1 = C1.shape[0]
        # stacked loadings [Qx; Qy] sqrt(s): scaled, rotated, normalised, signed and ordered on the resident panel
        Xrot, ptot, k, rot_matrix, phi = rotation.promax_panel(self.ctx, np.concatenate([C1, C2], axis=0),
                                                               power=self._params["power"], max_iter=self._params["max_iter"],
                                                               rtol=self._params["rtol"], col_scale=scaling)
        # analysis-space images of the rotated loadings: Q sqrt(s) rotation_matrix
        Qr = [model._q[i][:, :k] * scaling @ rot_matrix for i in range(2)]
        norm1, norm2 = np.linalg.norm(Qr[0], axis=0), np.linalg.norm(Qr[1], axis=0)
        sqcov = (norm1 * norm2) ** 2
        idx = np.argsort(sqcov)[::-1]
        RinvT = self._rot_mat_inv_trans(rot_matrix)
        sc1 = (np.asarray(model.data["scores1"], dtype=np.float64)[:, :k] / scaling) @ RinvT * norm1
        sc2 = (np.asarray(model.data["scores2"], dtype=np.float64)[:, :k] / scaling) @ RinvT * norm2
        mx, mn = engine.panel_colminmax(self.ctx, Xrot, ptot)                 # xarray_utils.py:273-301
        mx, mn = mx.cpu().numpy()[:k].astype(np.float64), mn.cpu().numpy()[:k].astype(np.float64)
        sign = np.where(np.abs(mx) >= np.abs(mn), 1.0, -1.0)
        # feature-space components (what `components()` back-projects to): rotated loadings / norm, signed, sorted
        L = Xrot.shape[1]
        F = []
        for norm, lo, hi in ((norm1, 0, p1), (norm2, p1, ptot)):
            M = np.zeros((L, L))
            M[idx, np.arange(k)] = sign[idx] / norm[idx]
            blk = engine.panel_matmul(self.ctx, Xrot[lo:], rotation._dev(M, Xrot))
            F.append(engine.panel_export(self.ctx, blk, hi - lo, k))
        del Xrot
        self.model_data = dict(singular_values=np.asarray(model.data["singular_values"]), components1=C1, components2=C2)
        self.data = dict(
            input_data1=model.data["input_data1"], input_data2=model.data["input_data2"],
            components1=F[0], components2=F[1],
            scores1=(sc1 * sign)[:, idx].astype(np.float32), scores2=(sc2 * sign)[:, idx].astype(np.float32),
            squared_covariance=sqcov[idx], total_squared_covariance=model.data["total_squared_covariance"],
            idx_modes_sorted=idx, norm1=norm1[idx], norm2=norm2[idx], rotation_matrix=rot_matrix, phi_matrix=phi,
            modes_sign=sign[idx],
        )
        self.sorted = True
        return self

    # ------------------------------------------------------------------ transform (cpcca_rotator.py:282-372)
    def transform(self, X=None, Y=None, normalized: bool = False):
        if X is None and Y is None:
            raise ValueError("No data provided. Please provide X and/or Y.")
        k = self.data["norm1"].size
        RinvT = self._rot_mat_inv_trans(self.data["rotation_matrix"])
        scaling = np.sqrt(np.asarray(self.model_data["singular_values"], dtype=np.float64)[:k])
        outs = []
        for which, Z in ((1, X), (2, Y)):
            if Z is None:
                continue
            pre = self.preprocessor1 if which == 1 else self.preprocessor2
            mat, fields, vs = pre.transform(Z)
            proj = engine.project(self.ctx, mat, np.ascontiguousarray(self.model_data[f"components{which}"])).astype(np.float64)
            mat.free()
            proj = (proj / scaling) @ RinvT
            proj = proj[:, self.data["idx_modes_sorted"]] * self.data["modes_sign"]
            if not normalized:
                proj = proj * self.data[f"norm{which}"]
            outs.append(pre.inverse_transform_scores(proj.astype(np.float32), f"scores{which}", self.attrs, fields, vs))
        return outs[0] if len(outs) == 1 else outs

    # ------------------------------------------------------------------ accessors
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

    def _mode_array(self, values, name):
        k = len(values)
        return labelled.pack(np.asarray(values), ("mode",), {"mode": np.arange(1, k + 1)}, name, dict(self.attrs),
                             self.preprocessor1.fields[0].like)

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


class MCARotator(CPCCARotator):
    """cross/mca_rotator.py:5-75."""
    _model_name = "Rotated MCA"
