"""Labelled-array plumbing for the model classes.

The reference's public surface is xarray in / xarray out.  When xarray is importable the model
classes accept and return real `xarray.DataArray` / `Dataset` objects; in images without xarray
(this build image and the GPU box) a minimal stand-in `DataArray` carries dims / coords / attrs /
name so the same API can be exercised.  Only label bookkeeping lives here -- no numerics.
"""

from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the image
    import xarray as _xr
except Exception:  # xarray absent
    _xr = None


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch") and hasattr(a, "device")


class DataArray:
    """Minimal labelled n-d array (dims, coords, name, attrs) mirroring the xarray names used
    by the reference's accessors."""

    def __init__(self, data, dims=None, coords=None, name=None, attrs=None, chunks=None):
        # a torch tensor (e.g. a field that already lives in HBM) is kept as it is: the preprocessing kernels
        # read it in place, nothing crosses PCIe
        self.values = data if _is_torch(data) else np.asarray(data)
        self.chunks = chunks          # like xarray: None for an in-memory array, chunk sizes for a dask-backed one
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(self.values.ndim))
        self.dims = tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError("dims do not match the data rank")
        self.coords = {}
        for k, v in (coords or {}).items():
            self.coords[k] = np.asarray(v)
        for d, n in zip(self.dims, self.values.shape):
            self.coords.setdefault(d, np.arange(n))
        self.name = name
        self.attrs = dict(attrs or {})

    data = property(lambda self: self.values)
    shape = property(lambda self: tuple(self.values.shape))
    ndim = property(lambda self: self.values.ndim)
    dtype = property(lambda self: self.values.dtype)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.values.shape))

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def transpose(self, *dims):
        if not dims:
            dims = self.dims[::-1]
        perm = [self.dims.index(d) for d in dims]
        return DataArray(self.values.transpose(perm), dims, self.coords, self.name, self.attrs)

    def sel(self, **kw):
        out = self
        for d, key in kw.items():
            ax = out.dims.index(d)
            c = out.coords[d]
            if isinstance(key, slice):
                lo = c.min() if key.start is None else key.start
                hi = c.max() if key.stop is None else key.stop
                idx = np.nonzero((c >= lo) & (c <= hi))[0]
            else:
                keys = np.atleast_1d(key)
                idx = np.array([int(np.nonzero(c == kk)[0][0]) for kk in keys])
            coords = dict(out.coords)
            coords[d] = c[idx]
            out = DataArray(np.take(out.values, idx, axis=ax), out.dims, coords, out.name, out.attrs)
        return out

    def isel(self, **kw):
        out = self
        for d, idx in kw.items():
            ax = out.dims.index(d)
            idx = np.arange(out.shape[ax])[idx] if isinstance(idx, slice) else np.atleast_1d(idx)
            coords = dict(out.coords)
            coords[d] = out.coords[d][idx]
            out = DataArray(np.take(out.values, idx, axis=ax), out.dims, coords, out.name, out.attrs)
        return out

    def copy(self):
        return DataArray(self.values.copy(), self.dims, self.coords, self.name, self.attrs)

    def __repr__(self):
        return f"<xeofs_amd.DataArray {self.name!r} {dict(self.sizes)}>"


class Dataset:
    """Minimal stand-in for `xarray.Dataset`: named DataArrays sharing their dimensions.  The reference accepts
    Datasets (its README quickstart feeds one) and returns Datasets with the same data_vars
    (preprocessing/stacker.py:203-206, 271-275)."""

    def __init__(self, data_vars, attrs=None):
        self.data_vars = {}
        for k, v in dict(data_vars).items():
            if not isinstance(v, DataArray):
                raise TypeError("Dataset variables must be DataArrays")
            self.data_vars[k] = DataArray(v.values, v.dims, v.coords, k, v.attrs, v.chunks)
        self.attrs = dict(attrs or {})

    def __getitem__(self, k):
        return self.data_vars[k]

    def __iter__(self):
        return iter(self.data_vars)

    def keys(self):
        return self.data_vars.keys()

    def __repr__(self):
        return f"<xeofs_amd.Dataset {list(self.data_vars)}>"


def is_dataset(obj) -> bool:
    return isinstance(obj, Dataset) or (_xr is not None and isinstance(obj, _xr.Dataset))


def make_dataset(like, data_vars):
    """a Dataset of the same flavour as `like`"""
    if _xr is not None and isinstance(like, _xr.Dataset):
        return _xr.Dataset(data_vars)
    return Dataset(data_vars)


def is_xarray(obj) -> bool:
    return _xr is not None and isinstance(obj, (_xr.DataArray, _xr.Dataset))


def unpack(obj):
    """-> (values ndarray, dims tuple, coords dict, name, attrs) from either array flavour."""
    if _xr is not None and isinstance(obj, _xr.DataArray):
        coords = {k: np.asarray(v.values) for k, v in obj.coords.items() if k in obj.dims or v.ndim == 0}   # + scalar coords
        return np.asarray(obj.values), tuple(obj.dims), coords, obj.name, dict(obj.attrs)
    if isinstance(obj, DataArray):
        return obj.values, obj.dims, dict(obj.coords), obj.name, dict(obj.attrs)
    raise TypeError(f"Data must be a DataArray (xarray or xeofs_amd.labelled), got {type(obj)}")


def pack(values, dims, coords, name, attrs, like):
    """Build an output array of the same flavour as `like`."""
    coords = {k: v for k, v in coords.items() if k in dims}
    if _xr is not None and isinstance(like, _xr.DataArray):
        return _xr.DataArray(values, dims=dims, coords=coords, name=name, attrs=attrs)
    return DataArray(values, dims, coords, name, attrs)


def concat(objs, dim, coord):
    """Stack equally shaped arrays along a new leading dimension `dim` (xr.concat(..., dim=dim) +
    assign_coords); a list of lists (multi-field outputs) is stacked field by field."""
    if isinstance(objs[0], (list, tuple)):
        return [concat([o[i] for o in objs], dim, coord) for i in range(len(objs[0]))]
    if _xr is not None and isinstance(objs[0], _xr.DataArray):
        return _xr.concat(objs, dim=dim).assign_coords({dim: np.asarray(coord)})
    vals, dims, coords, name, attrs = unpack(objs[0])
    stacked = np.stack([unpack(o)[0] for o in objs], axis=0)
    coords = dict(coords)
    coords[dim] = np.asarray(coord)
    return DataArray(stacked, (dim,) + tuple(dims), coords, name, attrs)


def is_lazy(obj) -> bool:
    """True for a dask-backed (chunked) array: `xarray.DataArray.chunks is not None` -- the reference then
    takes its dask branch (linalg/decomposer.py:104, 163-171)."""
    if isinstance(obj, (list, tuple)):
        return any(is_lazy(o) for o in obj)
    if isinstance(obj, Dataset):
        return any(is_lazy(v) for v in obj.data_vars.values())
    return bool(getattr(obj, "chunks", None))
