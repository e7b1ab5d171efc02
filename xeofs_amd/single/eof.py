"""xeofs_amd.single.EOF -- drop-in for xeofs.single.EOF (xeofs/single/eof.py:17-240,
xeofs/single/base_model_single_set.py:30-336): same constructor, fit / transform /
inverse_transform / components / scores / singular_values / explained_variance(_ratio).

Everything numeric runs in the HIP engine: fused preprocess -> resident matrix -> randomized SVD.
"""

from __future__ import annotations

import datetime

import numpy as np

from .. import __version__, engine, labelled
from .._deferred import Deferred
from ..linalg.decomposer import Decomposer
from ..preprocessing import Preprocessor


# Iteration rule of the complex models when `solver_kwargs` names none.  The reference's complex branch is a CONVERGED solver
# (scipy svds(solver="lobpcg"), xeofs/linalg/decomposer.py:149-160): on the bench's config-5 field its 20 modes are good to 1e-5
# (profiles/r06_r9_evidence.txt) while scikit-learn's fixed count ("auto": 7 products) leaves the modes that sit a per cent above
# the noise bulk 4e-5 .. 3e-3 off.  "converge" continues the block-Krylov recurrence until every wanted value is good to 2e-6
# (at most 20 products, lobpcg's own limit); `solver_kwargs={"n_iter": "auto"}` or an integer selects a fixed count.
COMPLEX_N_ITER_DEFAULT = "converge"


class EOF(Deferred):
    """Drop-in for xeofs.single.EOF (xeofs/single/eof.py:15-240).

    Aliasing contract of the default in-place layout: when `fit` is handed a DEVICE tensor (a torch tensor in HBM), the
    engine writes no copy of it -- the fitted model keeps a reference to that tensor and every later pass over the data
    (`inverse_transform`, the rotators, the bootstrapper, `ResidentMatrix.download`) reads it again through the Scaler
    map.  The tensor must therefore stay unmodified for as long as the model (or anything fitted on it) is in use;
    `model.data["input_data"].release_raw()` builds the engine's own layouts and drops the reference.  Host inputs
    (numpy / labelled arrays) are staged into HBM by the engine and owned by the model: nothing to observe there."""

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
        # rSVD streams the field where it lies; for the plain EOF model (every later use of the matrix goes through
        # engine.project / the panel ops) a land / sea mask stays in place as well (zero columns, layout mode 3)
        self.preprocessor = Preprocessor(center, standardize, use_coslat, check_nans, in_place=True,
                                         masked_ok=type(self).__name__ in ("EOF", "EOFBootstrapper"))
        # attrs as the reference stores them (base_model.py:38-46): bools/None stringified
        self.attrs = {"model": "EOF analysis", "software": "xeofs_amd", "version": __version__,
                      "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")}
        self.attrs.update({k: (str(v) if isinstance(v, bool) or v is None else v) for k, v in self._params.items()})
        self.data = {}

    def get_params(self):
        return dict(self._params)

    # ------------------------------------------------------------------ fit
    def fit(self, X, dim, weights=None):
        if labelled.is_lazy(X) and not self._params["compute"]:     # base_model_single_set.py:157-159: defer
            return self._defer(lambda: self._fit_now(X, dim, weights))
        return self._fit_now(X, dim, weights)

    def _fit_now(self, X, dim, weights=None):
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor.ctx = self.ctx
        self._decomposer_kwargs["lazy_input"] = labelled.is_lazy(X)
        omega = None if self._decomposer_kwargs["lazy_input"] else self._sketch_ahead(X, dim)
        # Preprocessor.fit_transform + Decomposer.fit (eof.py:85-97) as ONE engine call where the shape allows: the
        # column statistics ride on the first pass of the randomized SVD (eofx_fit_f32)
        dec = Decomposer(ctx=self.ctx, **self._decomposer_kwargs)
        mat = self.preprocessor.fit_transform_decompose(X, dim, weights, dec, omega)
        self.sample_dims = self.preprocessor.sample_dims
        return self._fit_algorithm(mat, dec=dec)

    def _sketch_ahead(self, X, dim):
        """Start drawing the sketch (host, sklearn's RandomState stream) on a worker thread so it
        overlaps the upload + preprocess kernels.  Only when the shape is knowable up front
        (integer n_modes, integer seed, single array without NaN compaction changing min(n, p))."""
        try:
            rs, k = self._params["random_state"], self.n_modes
            if not isinstance(k, (int, np.integer)) or not isinstance(rs, (int, np.integer)):
                return None
            vals, dims, _, _, _ = labelled.unpack(X)
            sd = (dim,) if isinstance(dim, str) else tuple(dim)
            n = int(np.prod([vals.shape[dims.index(d)] for d in sd]))
            p = int(np.prod(tuple(vals.shape), dtype=np.int64)) // max(n, 1)
            if n >= p:      # the sketch lives on the feature side: its size depends on the NaN mask
                return None
            n_over = int(self._solver_kwargs.get("n_oversamples", 10))
            return engine.SketchFuture(n, int(k) + n_over, int(rs))
        except Exception:
            return None

    def _fit_algorithm(self, mat, omega=None, dec=None):
        """xeofs/single/eof.py:85-118."""
        total_variance = self.preprocessor.total_variance                     # eof.py:93
        if dec is None:
            dec = Decomposer(ctx=self.ctx, **self._decomposer_kwargs)
            dec.fit(mat, dims=(self.sample_name, self.feature_name), total_variance=total_variance, omega=omega)
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
        self.compute()          # a deferred fit (compute=False on a lazy input) runs now: the fitted state is needed
        mat, fields, vs = self.preprocessor.transform(X)
        proj = engine.project(self.ctx, mat, self.data["components"])
        mat.free()
        if normalized:
            proj = proj / self.data["norms"].astype(proj.dtype)
        return self.preprocessor.inverse_transform_scores(proj, "scores", self.attrs, fields, vs)

    def fit_transform(self, X, dim, weights=None, **kwargs):
        return self.fit(X, dim, weights).transform(X, **kwargs)

    def _parse_scores(self, scores, normalized, dtype):
        """base_model_single_set.py:205-286: scores with a (possibly scalar) 'mode' coordinate -> (S [n', k'] of the valid
        samples, mode numbers, valid-sample mask, the fields relabelled with the scores' sample coordinates)."""
        vals, dims, coords, _, _ = labelled.unpack(scores)
        dims = tuple(dims)
        if "mode" not in dims:          # a single selected mode: "Handle scalar mode in xr.dot" (line 276)
            m = np.asarray(coords.get("mode", 1)).reshape(-1)[:1]
            vals, dims = np.asarray(vals)[None], ("mode",) + dims
            coords = dict(coords, mode=m)
        modes = np.asarray(coords["mode"]).astype(int).reshape(-1)
        order = [dims.index("mode")] + [i for i, d in enumerate(dims) if d != "mode"]
        S = np.transpose(vals, order).reshape(len(modes), -1).T            # (n_samples, k')
        vs = ~np.isnan(S).all(axis=1)
        S = np.ascontiguousarray(S[vs], dtype=dtype)
        if normalized:
            S = S * self.data["norms"][modes - 1].astype(S.real.dtype)
        f0 = self.preprocessor.fields[0]
        sample_shape = tuple(vals.shape[dims.index(d)] for d in f0.sample_dims)
        fields = []
        for f in self.preprocessor.fields:
            g = object.__new__(type(f))
            g.__dict__.update(f.__dict__)
            g.sample_shape = sample_shape
            g.coords = dict(f.coords, **{d: coords[d] for d in f.sample_dims if d in coords})
            fields.append(g)
        return S, modes, vs, fields

    def inverse_transform(self, scores, normalized: bool = False):
        """base_model_single_set.py:205-286 + eof.py:134-156: Xhat = scores . conj(V)^T, then un-scale."""
        self.compute()          # a deferred fit (compute=False on a lazy input) runs now: the fitted state is needed
        S, modes, vs, fields = self._parse_scores(scores, normalized, np.float32)
        V = np.ascontiguousarray(self.data["components"][:, modes - 1])
        rec = engine.reconstruct(self.ctx, S, V)
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


class ComplexEOF(EOF):
    """Drop-in for xeofs.single.ComplexEOF (xeofs/single/eof.py:243-446): EOF analysis of complex
    data.  The complex matrix is held as two resident real matrices; the decomposition is the
    complex randomized SVD of `xeofs_amd.complex_svd` (the reference's complex branch,
    linalg/decomposer.py:149-160)."""

    def __init__(self, n_modes: int = 2, padding: str = "exp", decay_factor: float = 0.2, center: bool = True,
                 standardize: bool = False, use_coslat: bool = False, check_nans: bool = True,
                 sample_name: str = "sample", feature_name: str = "feature", compute: bool = True,
                 random_state: int | None = None, solver: str = "auto", solver_kwargs: dict = {}, **kwargs):
        super().__init__(n_modes=n_modes, center=center, standardize=standardize, use_coslat=use_coslat,
                         check_nans=check_nans, sample_name=sample_name, feature_name=feature_name,
                         compute=compute, random_state=random_state, solver=solver, solver_kwargs=solver_kwargs,
                         **kwargs)
        self.attrs.update({"model": "Complex EOF analysis", "padding": padding, "decay_factor": decay_factor})
        self._params.update({"padding": padding, "decay_factor": decay_factor})
        self.padding, self.decay_factor = padding, decay_factor
        # both parts stay in place where the engine streams them as they lie (eofx_rsvd_c64's lean layout: sketches of up
        # to 32 complex columns); wider sketches and the Hermitian-Gram route want the written layouts, built in one pass
        lean = self._lean_ok()
        self.preprocessor.in_place = lean
        self.preprocessor_imag = Preprocessor(center, False, use_coslat, check_nans, in_place=lean)

    def _lean_ok(self):
        n_over = int(dict(self._solver_kwargs).get("n_oversamples", 10))
        return isinstance(self.n_modes, (int, np.integer)) and int(self.n_modes) + n_over <= 32

    def _complex_parts(self, X, dim, weights):
        """preprocess Re and Im of a complex input with the same centring / weights"""
        vals, dims, coords, name, attrs = labelled.unpack(X)
        re = labelled.pack(np.ascontiguousarray(vals.real), dims, coords, name, attrs, X)
        im = labelled.pack(np.ascontiguousarray(vals.imag), dims, coords, name, attrs, X)
        self.preprocessor_imag.ctx = self.ctx
        std_c = None
        if self._params["standardize"]:
            # scaler.py:105-108 on complex data: numpy's std of a complex array is the real
            # sqrt(mean |z - mean|^2) = sqrt(var Re + var Im); both parts are divided by it
            self.preprocessor.standardize = False
            eps = np.finfo(np.float32).eps       # peek_std clips each part at eps: undo that, combine, clip ONCE
            s_re, s_im = (np.where(v <= eps, 0.0, v)
                          for v in (self.preprocessor.peek_std(re, dim), self.preprocessor_imag.peek_std(im, dim)))
            std_c = np.maximum(np.sqrt(s_re ** 2 + s_im ** 2), eps)
        A = self.preprocessor.fit_transform(re, dim, weights, std_override=std_c)
        B = self.preprocessor_imag.fit_transform(im, dim, weights, std_override=std_c)
        tv = self.preprocessor.total_variance + self.preprocessor_imag.total_variance
        return A, B, tv

    def _fit_now(self, X, dim, weights=None):
        self._reject_lazy(X)
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor.ctx = self.ctx
        A, B, tv = self._complex_parts(X, dim, weights)
        self.sample_dims = self.preprocessor.sample_dims
        return self._fit_complex(A, B, tv)

    @staticmethod
    def _reject_lazy(X):
        if labelled.is_lazy(X):     # linalg/decomposer.py:172-177
            raise NotImplementedError("Complex data together with dask is currently not implemented. See dask issue 7639 "
                                      "https://github.com/dask/dask/issues/7639")

    def _fit_complex(self, A, B, total_variance, omega=None):
        kw = dict(self._solver_kwargs)
        n_over = int(kw.get("n_oversamples", 10))
        if not isinstance(self.n_modes, (int, np.integer)) or int(self.n_modes) + n_over > 64:
            return self._fit_complex_wide(A, B, total_variance)
        om = None if omega is None else omega.result()
        if om is not None and om.shape[0] != min(A.n, A.p):     # samples or features were dropped: draw again
            om = None
        U, s, V = engine.rsvd_c64(self.ctx, A, B, int(self.n_modes), n_over,
                                  kw.get("n_iter", COMPLEX_N_ITER_DEFAULT), self._params["random_state"], omega=om)
        s64 = s.astype(np.float64)
        self.data = dict(input_data=(A, B), components=V, scores=U * s, norms=s64,
                         explained_variance=s64 ** 2 / (A.n - 1), total_variance=total_variance)
        return self

    def _fit_complex_wide(self, A, B, total_variance):
        """More modes than the 64-column complex sketch holds, or a variance-based (float) n_modes -- which asks for
        int(init_rank_reduction * rank) modes first and truncates by explained variance (decomposer.py:89-106,
        _svd.py:215-241): the exact Hermitian-Gram route of the complex cross models' PCA pre-reduction
        (xeofs_amd/cpca.py), followed by the decomposer's sign rule."""
        from ..cpca import ComplexResidentPCA

        pca = ComplexResidentPCA(self.ctx, self.n_modes, self._decomposer_kwargs.get("init_rank_reduction", 0.3))
        pca.fit(A, B, total_variance)
        V = pca.components()                                          # p x m complex64
        mx, mn = V.conj().max(axis=0), V.conj().min(axis=0)           # utils/xarray_utils.py:273-301 on VT = conj(V)^T
        sgn = np.where(np.abs(mx) >= np.abs(mn), 1.0, -1.0)
        V *= sgn.astype(np.float32)
        s64 = np.asarray(pca.s, dtype=np.float64)
        scores = (pca.U * sgn) * s64
        self.data = dict(input_data=(A, B), components=V, scores=scores.astype(np.complex64), norms=s64,
                         explained_variance=s64 ** 2 / (A.n - 1), total_variance=total_variance)
        return self

    def transform(self, X, normalized=False):
        raise NotImplementedError("ComplexEOF/HilbertEOF does not support transform() (as in the reference)")

    _real_reconstruction = False      # HilbertEOF: eof.py:564-567 keeps the real part only

    def inverse_transform(self, scores, normalized: bool = False):
        """eof.py:134-156 with complex scores and components: Xhat = S conj(V)^T, i.e.
        Re = Sr Vr^T + Si Vi^T and Im = Si Vr^T - Sr Vi^T -- two real products of width 2 k' on the GPU; each part is
        un-scaled with the centring of its own part (the reference's Scaler holds one complex mean)."""
        S, modes, vs, fields = self._parse_scores(scores, normalized, np.complex64)
        V = self.data["components"][:, modes - 1]
        Vri = np.ascontiguousarray(np.concatenate([V.real, V.imag], axis=1), dtype=np.float32)
        Sre = np.ascontiguousarray(np.concatenate([S.real, S.imag], axis=1), dtype=np.float32)
        re = self.preprocessor.inverse_transform_data(engine.reconstruct(self.ctx, Sre, Vri), "reconstructed_data", fields, vs)
        if self._real_reconstruction:
            return re
        Sim = np.ascontiguousarray(np.concatenate([S.imag, -S.real], axis=1), dtype=np.float32)
        im = self.preprocessor_imag.inverse_transform_data(engine.reconstruct(self.ctx, Sim, Vri), "reconstructed_data", fields, vs)

        def join(a, b):
            va, dims, coords, name, attrs = labelled.unpack(a)
            return labelled.pack(va + 1j * labelled.unpack(b)[0], dims, coords, name, attrs, a)

        return [join(a, b) for a, b in zip(re, im)] if isinstance(re, list) else join(re, im)

    def components_amplitude(self, normalized=True):
        c = self.components(normalized)
        return self._map(c, np.abs, "components_amplitude")

    def components_phase(self, normalized=True):
        c = self.components(normalized)
        return self._map(c, np.angle, "components_phase")

    def scores_amplitude(self, normalized=False):
        return self._map(self.scores(normalized), np.abs, "scores_amplitude")

    def scores_phase(self, normalized=False):
        return self._map(self.scores(normalized), np.angle, "scores_phase")

    @staticmethod
    def _map(obj, fn, name):
        def one(a):
            vals, dims, coords, _, attrs = labelled.unpack(a)
            return labelled.pack(fn(vals), dims, coords, name, attrs, a)

        return [one(a) for a in obj] if isinstance(obj, list) else one(obj)


class HilbertEOF(ComplexEOF):
    """Drop-in for xeofs.single.HilbertEOF (xeofs/single/eof.py:449-560): the real input is
    preprocessed, Hilbert-transformed along the sample axis on the GPU (`eofx_hilbert_f32`,
    utils/hilbert_transform.py) and decomposed as a complex matrix."""

    _real_reconstruction = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.attrs.update({"model": "Hilbert EOF analysis"})

    def _fit_now(self, X, dim, weights=None):
        self._reject_lazy(X)
        self.ctx = self.ctx or engine.default_context()
        self.preprocessor.ctx = self.ctx
        omega = self._sketch_ahead(X, dim)          # drawn on a worker thread while the Hilbert stage runs
        centred = bool(self._params["center"])
        # A centred field stays in place (the raw field through the Scaler map): the Hilbert stage then builds the
        # sample-contiguous layout it works on for the duration of its kernel only and leaves Im in that layout alone.
        self.preprocessor.in_place = centred and self._lean_ok()
        self.preprocessor.for_hilbert = True        # (the statistics pass writes the sample-contiguous raw field on its way)
        # a land / sea mask (all-NaN grid points) stays in place too: zero columns through the Hilbert stage and the passes
        # of the decomposition (round 5; `_hilbert_masked_ok = False` forces the compaction route, for comparisons)
        self.preprocessor.masked_ok = self.preprocessor.in_place and getattr(self, "_hilbert_masked_ok", True)
        A = self.preprocessor.fit_transform(X, dim, weights)
        self.sample_dims = self.preprocessor.sample_dims
        if centred and self._operator_route_ok(A):
            return self._fit_operator(A, omega)
        B, A2 = engine.hilbert(self.ctx, A, self.padding, self.decay_factor, want_real=not centred)
        if A2 is not None:          # eof.py:546-555: the analytic signal is re-centred per feature
            A.free()
            A = A2
            tv_re = A.sumsq() / (A.n - 1)
        elif A.layout()[1] and not A.layout()[0]:    # in place: the Scaler's statistics already hold it (no layout is built)
            tv_re = float(self.preprocessor.total_variance)
        else:
            tv_re = A.sumsq() / (A.n - 1)
        tv = tv_re + B.sumsq() / (A.n - 1)
        return self._fit_complex(A, B, tv, omega)

    def _operator_route_ok(self, A):
        """The imaginary part is never written (engine.rsvd_hilbert_c64): the Hilbert stage is one n x n matrix along the
        samples, applied to the sample-side panels of the decomposition, and every pass streams the real field once.
        Needs a centred field (the analytic signal is then centred as the reference re-centres it, eof.py:546-555), a sketch
        of at most 64 complex columns and a series the resident operator holds; `_hilbert_operator_ok = False` forces the
        two-part route (comparisons)."""
        n_over = int(dict(self._solver_kwargs).get("n_oversamples", 10))
        return (getattr(self, "_hilbert_operator_ok", True) and isinstance(self.n_modes, (int, np.integer))
                and int(self.n_modes) + n_over <= 64 and A.n <= engine.HILBERT_OPERATOR_MAX_SAMPLES
                and 2 * A.n <= A.p          # the n x n operator must be cheaper to hold and stream than the n x p imaginary part
                and not (A.masked and A.p < A.n))

    def _fit_operator(self, A, omega):
        kw = dict(self._solver_kwargs)
        n_over = int(kw.get("n_oversamples", 10))
        if A.layout()[1] and not A.layout()[0]:      # in place: the Scaler's statistics already hold it
            tv_re = float(self.preprocessor.total_variance)
        else:
            tv_re = A.sumsq() / (A.n - 1)
        # total variance of the imaginary part: the transform kernel with its stores switched off (consumes the
        # sample-contiguous raw field the statistics pass wrote)
        tv = tv_re + engine.hilbert_sumsq(self.ctx, A, self.padding, self.decay_factor) / (A.n - 1)
        om = None if omega is None else omega.result()
        if om is not None and om.shape[0] != min(A.n, A.p):     # samples or features were dropped: draw again
            om = None
        U, s, V = engine.rsvd_hilbert_c64(self.ctx, A, int(self.n_modes), self.padding, self.decay_factor, n_over,
                                          kw.get("n_iter", COMPLEX_N_ITER_DEFAULT), self._params["random_state"], omega=om)
        s64 = s.astype(np.float64)
        self.data = dict(input_data=(A, None), components=V, scores=U * s, norms=s64,
                         explained_variance=s64 ** 2 / (A.n - 1), total_variance=tv)
        return self
