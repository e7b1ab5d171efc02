"""xeofs_amd.single.EOFRotator -- drop-in for xeofs.single.EOFRotator
(xeofs/single/eof_rotator.py:17-305): Varimax (power=1) / Promax (power>1) rotation of a fitted EOF
model.  The rotation loop runs on the resident loadings panel (xeofs_amd/rotation.py,
`eofx_panel_rot_step_f64`); the m x m algebra (SVD, inverse, sort, sign) is host work as in the reference.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, rotation
from .eof import EOF, ComplexEOF


class EOFRotator(EOF):
    _complex = False

    def __init__(self, n_modes: int = 2, power: int = 1, max_iter: int | None = None, rtol: float = 1e-8,
                 compute: bool = True):
        if max_iter is None:
            max_iter = 1000 if compute else 100
        self.n_modes = n_modes
        self._params = dict(n_modes=n_modes, power=power, max_iter=max_iter, rtol=rtol, compute=compute)
        self.attrs = {"model": "Rotated EOF analysis"}
        self.attrs.update(self._params)
        self.attrs.update({"software": "xeofs_amd", "version": __version__,
                           "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")})
        self.ctx = None
        self.preprocessor = None
        self.data = {}
        self.model_data = {}
        self.sorted = False

    # ------------------------------------------------------------------ fit
    def fit(self, model):
        """eof_rotator.py:103-205 (+ `_sort_by_variance`, :207-218; the engine is eager)."""
        getattr(model, "compute", lambda: None)()      # a deferred fit runs now: ctx / preprocessor / data are read below
        self.ctx = model.ctx
        self.preprocessor = model.preprocessor
        self.sample_name, self.feature_name = model.sample_name, model.feature_name
        self.sample_dims = getattr(model, "sample_dims", None)
        for a in ("preprocessor_imag", "_real_reconstruction"):      # the complex models' inverse_transform needs them
            if hasattr(model, a):
                setattr(self, a, getattr(model, a))
        m = int(self._params["n_modes"])
        power = self._params["power"]
        comps = np.asarray(model.data["components"])[:, :m]
        is_complex = np.iscomplexobj(comps)     # ComplexEOF / HilbertEOF models (eof_rotator.py:294-400)
        if is_complex and not self._complex:
            raise TypeError("EOFRotator rotates real models; use ComplexEOFRotator / HilbertEOFRotator for this model")
        m = comps.shape[1]
        expvar = np.asarray(model.data["explained_variance"], dtype=np.float64)[:m]
        # loadings = components * sqrt(expvar); rotation, explained variance, normalisation, sign and ordering all
        # happen on the resident panel -- the host sees the finished components once
        panel, finish = (rotation.cpromax_panel, rotation.cfinish_on_device) if is_complex else \
            (rotation.promax_panel, rotation.finish_on_device)
        Xrot, p, m, rot_matrix, phi = panel(self.ctx, comps, power=power, max_iter=self._params["max_iter"],
                                            rtol=self._params["rtol"], col_scale=np.sqrt(expvar))
        rot_sorted, expvar_r, idx, sign = finish(self.ctx, Xrot, p, m)
        del Xrot
        inp = model.data["input_data"]
        n_samples = (inp[0] if isinstance(inp, tuple) else inp).n
        norms = (expvar_r * (n_samples - 1)) ** 0.5
        svals = np.asarray(model.data["norms"], dtype=np.float64)[:m]
        scores = np.asarray(model.data["scores"])[:, :m].astype(np.complex128 if is_complex else np.float64) / svals
        RinvT = self._rot_mat_inv_trans(rot_matrix)
        scores = scores @ RinvT * norms
        scores = scores * sign
        self.model_data = dict(singular_values=np.asarray(model.data["norms"]), components=np.asarray(model.data["components"]))
        self.data = dict(
            input_data=model.data["input_data"],
            components=rot_sorted,
            scores=np.ascontiguousarray(scores[:, idx].astype(np.complex64 if is_complex else np.float32)),
            norms=norms[idx], explained_variance=expvar_r[idx], total_variance=model.data["total_variance"],
            idx_modes_sorted=idx, rotation_matrix=rot_matrix, phi_matrix=phi, modes_sign=sign[idx],
        )
        self.sorted = True
        return self

    def _rot_mat_inv_trans(self, R):
        """eof_rotator.py:264-288: R itself for the orthogonal (Varimax) case, inv(R)^H for Promax."""
        if self._params["power"] > 1:
            return np.linalg.inv(R).conj().T
        return R

    # ------------------------------------------------------------------ transform
    def transform(self, X, normalized: bool = False):
        """eof_rotator.py:220-257: project on the unrotated components, rotate, sort, scale, sign."""
        m = self.data["components"].shape[1]
        mat, fields, vs = self.preprocessor.transform(X)
        V = np.ascontiguousarray(self.model_data["components"][:, :m])
        proj = engine.project(self.ctx, mat, V).astype(np.float64)
        mat.free()
        proj = proj / np.asarray(self.model_data["singular_values"], dtype=np.float64)[:m]
        proj = proj @ self._rot_mat_inv_trans(self.data["rotation_matrix"])
        proj = proj[:, self.data["idx_modes_sorted"]]
        proj = proj * self.data["norms"] * self.data["modes_sign"]
        if normalized:
            proj = proj / self.data["norms"]
        return self.preprocessor.inverse_transform_scores(proj.astype(np.float32), "scores", self.attrs, fields, vs)

    def fit_transform(self, model):
        raise NotImplementedError("The fit_transform method is not implemented for the EOFRotator class.")

    def rotation_matrix(self):
        return self.data["rotation_matrix"]

    def phi_matrix(self):
        return self.data["phi_matrix"]


class ComplexEOFRotator(EOFRotator, ComplexEOF):
    """Drop-in for xeofs.single.ComplexEOFRotator (eof_rotator.py:294-337): Varimax / Promax rotation of a fitted
    `ComplexEOF` model; the complex loadings are rotated on the device as a [Re | Im] panel (`rotation.cpromax_panel`),
    up to 32 modes."""

    _complex = True

    def __init__(self, n_modes: int = 2, power: int = 1, max_iter: int | None = None, rtol: float = 1e-8,
                 compute: bool = True):
        EOFRotator.__init__(self, n_modes=n_modes, power=power, max_iter=max_iter, rtol=rtol, compute=compute)
        self.attrs.update({"model": "Rotated Complex EOF analysis"})

    def transform(self, X, normalized: bool = False):
        raise NotImplementedError("ComplexEOFRotator/HilbertEOFRotator does not support transform()")


class HilbertEOFRotator(ComplexEOFRotator):
    """Drop-in for xeofs.single.HilbertEOFRotator (eof_rotator.py:339-400; `transform` is not implemented there either)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.attrs.update({"model": "Rotated Hilbert EOF analysis"})
