"""Host-side mirror of the reference preprocessing pipeline's *bookkeeping*
(xeofs/preprocessing/preprocessor.py:119-366): dims -> (sample, feature) stacking order, cos-lat
weights, feature concatenation of lists, NaN re-expansion and unstacking on the way back.

All arithmetic on the data (NaN mask, statistics, centring/scaling/weighting, compaction,
total variance) is the fused HIP pass `eofx_preprocess_f32` / `eofx_apply_f32`.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np

from . import engine, labelled


def _as_tuple(d):
    return (d,) if isinstance(d, str) or not isinstance(d, Sequence) else tuple(d)


def sqrt_cos_lat_weights(lat):
    """xeofs/utils/xarray_utils.py:256-270."""
    return np.sqrt(np.cos(np.deg2rad(np.asarray(lat, dtype=float))).clip(0, 1))


class _Field:
    """Stacking bookkeeping of one input array (xeofs/preprocessing/stacker.py:157-256)."""

    def __init__(self, obj, sample_dims):
        vals, dims, coords, name, attrs = labelled.unpack(obj)
        missing = [d for d in sample_dims if d not in dims]
        if missing:
            raise ValueError(f"sample dimension(s) {missing} not found in data dims {dims}")
        self.like, self.dims, self.coords, self.name, self.attrs = obj, dims, coords, name, attrs
        self.sample_dims = tuple(sample_dims)
        self.feature_dims = tuple(d for d in dims if d not in sample_dims)  # order of X.dims (stacker.py:192-193)
        order = [dims.index(d) for d in self.sample_dims + self.feature_dims]
        v = vals.permute(order) if labelled._is_torch(vals) else np.transpose(vals, order)   # device tensors stay on the device
        self.sample_shape = tuple(v.shape[:len(self.sample_dims)])
        self.feature_shape = tuple(v.shape[len(self.sample_dims):])
        self.n = int(np.prod(self.sample_shape, dtype=np.int64))
        self.P = int(np.prod(self.feature_shape, dtype=np.int64))
        self.matrix = v.reshape(self.n, self.P)

    def feature_vector(self, w_obj):
        """broadcast a weights array defined on (a subset of) the feature dims to the stacked axis"""
        vals, dims, _, _, _ = labelled.unpack(w_obj)
        bad = [d for d in dims if d not in self.feature_dims]
        if bad:
            raise ValueError(f"weights have dimensions {bad} that are not feature dimensions")
        shape = [vals.shape[dims.index(d)] if d in dims else 1 for d in self.feature_dims]
        perm = [dims.index(d) for d in self.feature_dims if d in dims]
        return np.broadcast_to(np.transpose(vals, perm).reshape(shape), self.feature_shape).reshape(-1).astype(np.float64)

    def coslat_vector(self):
        if "lat" not in self.feature_dims:
            raise ValueError("use_coslat=True requires a feature dimension called 'lat'")
        w = sqrt_cos_lat_weights(self.coords["lat"])
        shape = [w.size if d == "lat" else 1 for d in self.feature_dims]
        return np.broadcast_to(w.reshape(shape), self.feature_shape).reshape(-1)


class Preprocessor:
    def __init__(self, center=True, standardize=False, use_coslat=False, check_nans=True, ctx=None, in_place=False,
                 masked_ok=False):
        self.center, self.standardize, self.use_coslat, self.check_nans = center, standardize, use_coslat, check_nans
        self.ctx = ctx
        # masked_ok (with in_place): a field with all-NaN grid points (land / sea mask) stays in place too -- the engine keeps
        # the masked features as zero columns (layout mode 3) and xeofs_amd.engine compacts the factors; for models whose
        # every use of the resident matrix goes through engine.rsvd / fit / project / the panel-level ops
        self.masked_ok = bool(masked_ok) and bool(in_place)
        # in_place: the engine writes no copy of the matrix, the passes of the decomposition stream the (staged) field
        # through the Scaler map (include/eofx.h, layout policy); a layout is built later only if something asks for it
        self.in_place = in_place

    # ------------------------------------------------------------------ forward
    def _fields(self, X, sample_dims):
        self.is_list = isinstance(X, (list, tuple))
        self.is_dataset = False
        if labelled.is_dataset(X):
            self.is_dataset, self._ds_like = True, X
            X = [X[v] for v in X.data_vars]
        xs = list(X) if isinstance(X, (list, tuple)) else [X]
        fields = [_Field(x, sample_dims) for x in xs]
        if len({f.n for f in fields}) != 1:
            raise ValueError("all input arrays must share the sample dimensions")
        return fields

    def _stack(self, fields, weights):
        ws = None
        if weights is not None or self.use_coslat:
            wl = list(weights) if isinstance(weights, (list, tuple)) else [weights] * len(fields)
            ws = []
            for f, w in zip(fields, wl):
                v = np.ones(f.P)
                if self.use_coslat:
                    v = v * f.coslat_vector()
                if w is not None:
                    v = v * f.feature_vector(w)
                ws.append(v)
            ws = np.concatenate(ws)
        if any(labelled._is_torch(f.matrix) for f in fields):      # resident input: concatenate on the device
            import torch

            dev = next(f.matrix.device for f in fields if labelled._is_torch(f.matrix))
            mats = [(f.matrix if labelled._is_torch(f.matrix) else torch.as_tensor(np.asarray(f.matrix))).to(dev, torch.float32)
                    for f in fields]
            M = mats[0] if len(mats) == 1 else torch.cat(mats, dim=1)
            return M.contiguous(), ws
        mats = [np.asarray(f.matrix, dtype=np.float32) for f in fields]
        M = mats[0] if len(mats) == 1 else np.concatenate(mats, axis=1)
        return np.ascontiguousarray(M), ws

    def peek_std(self, X, sample_dims):
        """per-feature standard deviation (ddof = 0, clipped at float32 eps: scaler.py:105-108) of the stacked field,
        without building the matrix -- the complex models combine the deviations of the two parts"""
        ctx = self.ctx or engine.default_context()
        M, _ = self._stack(self._fields(X, _as_tuple(sample_dims)), None)
        _, st = engine.preprocess(ctx, M, True, True, None, self.check_nans, build=False)
        return st["std"]

    def fit_transform(self, X, sample_dims, weights=None, std_override=None):
        """std_override: per-feature deviations to scale with instead of this field's own (complex input: one real
        deviation sqrt(var Re + var Im) for both parts, numpy's std of a complex array)."""
        self.sample_dims = _as_tuple(sample_dims)
        ctx = self.ctx or engine.default_context()
        self.fields = self._fields(X, self.sample_dims)
        M, self.feature_weights = self._stack(self.fields, weights)
        if std_override is not None:
            _, st = engine.preprocess(ctx, M, self.center, False, self.feature_weights, self.check_nans, build=False)
            self.mean_, self.std_ = (st["mean"] if self.center else None), np.asarray(std_override, dtype=np.float64)
            self.valid_feature, self.valid_sample = st["valid_feature"], st["valid_sample"]
            mat, _ = engine.apply(ctx, M, self.mean_, self.std_, self.feature_weights, self.valid_feature, self.check_nans)
            self.total_variance = mat.sumsq() / (mat.n - 1)
            return mat
        mat, st = engine.preprocess(ctx, M, self.center, self.standardize, self.feature_weights, self.check_nans,
                                    in_place=self.in_place, allow_masked=self.masked_ok,
                                    for_hilbert=bool(getattr(self, "for_hilbert", False)) and self.in_place)
        self.mean_, self.std_ = (st["mean"] if self.center else None), (st["std"] if self.standardize else None)
        self.valid_feature, self.valid_sample = st["valid_feature"], st["valid_sample"]
        self.total_variance = st["total_variance"]
        self._stacked = M if mat.masked else None      # (a reference, not a copy) for recompact()
        return mat

    def recompact(self, mat):
        """The compacted matrix of a field that fit_transform left masked in place (layout mode 3), for a consumer that
        cannot work with zero columns; frees `mat`.  Same statistics, same preprocessed values."""
        ctx = self.ctx or engine.default_context()
        M, self._stacked = self._stacked, None
        if M is None:
            raise RuntimeError("recompact: the stacked field is no longer held")
        mat.free()
        new, _ = engine.preprocess(ctx, M, self.center, self.standardize, self.feature_weights, self.check_nans,
                                   want_stats=False, in_place=False, allow_masked=False)
        return new

    def fit_transform_decompose(self, X, sample_dims, weights, decomposer, omega=None):
        """Preprocessor.fit_transform and Decomposer.fit in ONE engine call where the shape allows (engine.fit /
        eofx_fit_f32: the Scaler's statistics ride on the first pass of the randomized SVD); otherwise the two steps.
        -> the resident matrix; `decomposer` holds U_, s_, V_ afterwards."""
        self.sample_dims = _as_tuple(sample_dims)
        ctx = self.ctx or engine.default_context()
        self.fields = self._fields(X, self.sample_dims)
        M, self.feature_weights = self._stack(self.fields, weights)
        n, P = M.shape
        plan = decomposer.fused_plan(n, P) if self.in_place and ctx.precision[0] == "f16x3" else None
        if plan is None:
            mat, st = engine.preprocess(ctx, M, self.center, self.standardize, self.feature_weights, self.check_nans,
                                        in_place=self.in_place, allow_masked=self.masked_ok)
        else:
            k, n_over, n_iter = plan
            mat, st, U, s, V = engine.fit(ctx, M, k, self.center, self.standardize, self.feature_weights, self.check_nans,
                                          n_over, n_iter, random_state=decomposer.random_state,
                                          flip=bool(decomposer.flip_signs), omega=omega, allow_masked=self.masked_ok)
        self.mean_, self.std_ = (st["mean"] if self.center else None), (st["std"] if self.standardize else None)
        self.valid_feature, self.valid_sample = st["valid_feature"], st["valid_sample"]
        self.total_variance = st["total_variance"]
        if plan is None:
            decomposer.fit(mat, total_variance=self.total_variance, omega=omega)
        else:
            decomposer.adopt(mat, U, s, V, self.total_variance, plan)
        return mat

    def transform(self, X):
        ctx = self.ctx or engine.default_context()
        fields = self._fields_like(X)
        M, _ = self._stack(fields, None)
        mat, vs = engine.apply(ctx, M, self.mean_, self.std_, self.feature_weights, self.valid_feature, self.check_nans,
                               in_place=self.in_place, allow_masked=self.masked_ok)
        return mat, fields, vs

    def _fields_like(self, X):
        if labelled.is_dataset(X):
            X = [X[v] for v in X.data_vars]
        xs = list(X) if isinstance(X, (list, tuple)) else [X]
        fields = [_Field(x, self.sample_dims) for x in xs]
        if [f.P for f in fields] != [f.P for f in self.fields]:
            raise ValueError("Cannot transform data. Feature coordinates are different.")
        return fields

    # ------------------------------------------------------------------ backward
    def _wrap(self, outs):
        if self.is_dataset:
            return labelled.make_dataset(self._ds_like, {f.name: o for f, o in zip(self.fields, outs)})
        return outs if self.is_list else outs[0]

    def inverse_transform_components(self, V, name="components", attrs=None):
        """(p_valid, k) -> per-field arrays with dims (mode, *feature_dims); NaN where masked."""
        k = V.shape[1]
        if self.valid_feature.all():        # nothing masked: one plain copy (207 MB at config 4) instead of a NaN fill + a masked scatter
            full = np.array(V, copy=True)
        else:
            full = np.full((self.valid_feature.size, k), np.nan, dtype=V.dtype)
            full[self.valid_feature] = V
        outs, off = [], 0
        for f in self.fields:
            blk = full[off:off + f.P].T.reshape((k,) + f.feature_shape)
            off += f.P
            coords = {d: f.coords[d] for d in f.feature_dims}
            coords["mode"] = np.arange(1, k + 1)
            outs.append(labelled.pack(blk, ("mode",) + f.feature_dims, coords, name, dict(attrs or {}), f.like))
        return self._wrap(outs)

    def inverse_transform_scores(self, S, name="scores", attrs=None, fields=None, valid_sample=None):
        """(n_valid, k) -> array with dims (mode, *sample_dims); NaN rows for dropped samples."""
        f = (fields or self.fields)[0]
        vs = self.valid_sample if valid_sample is None else valid_sample
        k = S.shape[1]
        if np.all(vs):
            full = np.array(S, copy=True)
        else:
            full = np.full((vs.size, k), np.nan, dtype=S.dtype)
            full[vs] = S
        blk = full.T.reshape((k,) + f.sample_shape)
        coords = {d: f.coords[d] for d in f.sample_dims}
        coords["mode"] = np.arange(1, k + 1)
        return labelled.pack(blk, ("mode",) + f.sample_dims, coords, name, dict(attrs or {}), f.like)

    def inverse_transform_data(self, X2d, name="reconstructed_data", fields=None, valid_sample=None):
        """(n_valid, p_valid) preprocessed-space matrix -> original dims and units
        (scaler.py:165-190 un-scaling, sanitizer.py:128-153 NaN re-expansion, stacker unstack)."""
        fields = fields or self.fields
        vs = self.valid_sample if valid_sample is None else valid_sample
        vf = self.valid_feature
        X = np.asarray(X2d, dtype=np.float64)
        if self.feature_weights is not None:
            X = X / self.feature_weights[vf]
        if self.std_ is not None:
            X = X * self.std_[vf]
        if self.mean_ is not None:
            X = X + self.mean_[vf]
        full = np.full((vs.size, vf.size), np.nan)
        full[np.ix_(vs, vf)] = X
        outs, off = [], 0
        for f in fields:
            blk = full[:, off:off + f.P].reshape(f.sample_shape + f.feature_shape)
            off += f.P
            src = f.sample_dims + f.feature_dims
            blk = np.transpose(blk, [src.index(d) for d in f.dims])
            outs.append(labelled.pack(blk, f.dims, {d: f.coords[d] for d in f.dims}, name, {}, f.like))
        return self._wrap(outs)
