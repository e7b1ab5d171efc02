"""Thin Python handles over the C ABI (include/eofx.h): context, resident matrix, and the
hot-path calls.  numpy arrays are host buffers, torch CUDA tensors are device buffers; the
library detects which.  No arithmetic happens in this file.
"""

from __future__ import annotations

import ctypes as C
import numbers

import numpy as np

from . import _lib
from ._lib import ptr, raise_for


def _f32c(a):
    """float32 C-contiguous host array or a device tensor, unchanged if already so."""
    if isinstance(a, np.ndarray):
        return np.ascontiguousarray(a, dtype=np.float32)
    if hasattr(a, "data_ptr"):  # torch tensor
        import torch

        if a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(torch.float32).contiguous()
        return a
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class Context:
    """One GPU + one HIP stream (eofx_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        """stream: a hipStream_t handle; None = torch's CURRENT stream on `device` at construction (the default stream
        unless the caller is inside `torch.cuda.stream(...)`), so torch operations issued between engine calls stay ordered
        with the engine's kernels.  `use_stream` re-binds later."""
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            try:
                import torch

                if torch.cuda.is_available():
                    stream = int(torch.cuda.current_stream(int(device)).cuda_stream)
            except Exception:
                stream = None
        rc = self.lib.eofx_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            raise _lib.EofxError(
                f"eofx_ctx_create(device={device}) failed ({rc}): no usable MI355X / HIP runtime. "
                "xeofs_amd has no CPU fallback.")
        self.handle = h
        self.device = int(device)
        self.precision = ("f16x3", "f16x3")

    def synchronize(self):
        raise_for(self.lib.eofx_ctx_synchronize(self.handle), self.handle)

    def use_stream(self, stream=None):
        """bind the context to a hipStream_t handle / a torch.cuda.Stream (None: torch's current stream on this device)"""
        if stream is None:
            stream = _torch().cuda.current_stream(self.device)
        handle = int(getattr(stream, "cuda_stream", stream) or 0)
        raise_for(self.lib.eofx_ctx_set_stream(self.handle, C.c_void_p(handle)), self.handle)

    def set_precision(self, power="f16x3", final="f16x3"):
        """Arithmetic of the matrix passes: "f16x3" (scaled split-fp16 MFMA, default), "f32" (exact-f32
        MFMA), "bf16x3", "bf16x6" (split-bf16 MFMA); see include/eofx.h."""
        raise_for(self.lib.eofx_ctx_set_precision(self.handle, _lib.PREC[power], _lib.PREC[final]), self.handle)
        self.precision = (power, final)

    def trim(self):
        """Return cached resident-matrix buffers to the device."""
        raise_for(self.lib.eofx_ctx_trim(self.handle), self.handle)

    def profile(self, enable: bool = True):
        raise_for(self.lib.eofx_ctx_profile(self.handle, int(enable)), self.handle)

    def profile_read(self):
        """-> dict(launches, ms, flops, bytes) of the atb_f32 launches since the last read."""
        n = C.c_int64()
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        raise_for(self.lib.eofx_ctx_profile_read(self.handle, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)),
                  self.handle)
        kl, km = (C.c_int64 * 3)(), (C.c_double * 3)()
        raise_for(self.lib.eofx_ctx_profile_by_kernel(self.handle, kl, km), self.handle)
        by_kernel = {name: dict(launches=int(kl[i]), ms=float(km[i])) for i, name in enumerate(("atb", "axb")) if kl[i]}
        return dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value, by_kernel=by_kernel)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.eofx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class ResidentMatrix:
    """A preprocessed (sample x feature) matrix resident in HBM (eofx_mat)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle
        n, p, npad, ppad = (C.c_int64() for _ in range(4))
        ctx.lib.eofx_mat_shape(handle, C.byref(n), C.byref(p), C.byref(npad), C.byref(ppad))
        self.n, self.p, self.n_pad, self.p_pad = n.value, p.value, npad.value, ppad.value
        self._keepalive = None     # the device field a raw-mode matrix reads (must outlive it)
        # Masked in-place matrix (layout mode 3, include/eofx.h): all-NaN grid points stay in the matrix as zero columns.
        # `p` is the number of VALID features -- what every caller means by it --, `p_phys` the column count of the
        # engine's matrix; `valid_index` (int64, ascending) maps one onto the other.  The functions of this module
        # compact / scatter the feature axis of the factors, so callers never see the zero columns.
        self.p_phys = self.p
        self.valid_index = None
        mk, pv = C.c_int(), C.c_int64()
        ctx.lib.eofx_mat_masked(handle, C.byref(mk), C.byref(pv))
        self.masked = bool(mk.value)
        if self.masked:
            self.p = pv.value

    def set_valid(self, valid_feature):
        """the boolean mask of valid features of a masked matrix (from the preprocessing statistics)"""
        idx = np.flatnonzero(np.asarray(valid_feature, dtype=bool)).astype(np.int64)
        if idx.size != self.p or (idx.size and idx[-1] >= self.p_phys):
            raise ValueError("valid-feature mask does not match the masked matrix")
        self.valid_index = idx
        self._valid_dev = None

    def _vidx(self, device=None):
        """valid_index as a torch tensor on `device` (cached)"""
        torch = _torch()
        dev = device if device is not None else f"cuda:{self.ctx.device}"
        if getattr(self, "_valid_dev", None) is None or str(self._valid_dev.device) != str(torch.device(dev)):
            self._valid_dev = torch.as_tensor(self.valid_index, device=dev)
        return self._valid_dev

    def compact_rows(self, V):
        """rows of the valid features of a factor with p_phys rows (numpy array or torch tensor)"""
        if not self.masked:
            return V
        if hasattr(V, "data_ptr"):
            return V.index_select(0, self._vidx(V.device)).contiguous()
        return np.ascontiguousarray(V[self.valid_index])

    def scatter_rows(self, V):
        """a factor with p rows -> p_phys rows, zeros at the masked features"""
        if not self.masked:
            return V
        if hasattr(V, "data_ptr"):
            torch = _torch()
            out = torch.zeros((self.p_phys,) + tuple(V.shape[1:]), dtype=V.dtype, device=V.device)
            out.index_copy_(0, self._vidx(V.device), V)
            return out
        out = np.zeros((self.p_phys,) + V.shape[1:], dtype=V.dtype)
        out[self.valid_index] = V
        return out

    @property
    def shape(self):
        return (self.n, self.p)

    def download(self) -> np.ndarray:
        out = np.empty((self.n, self.p_phys), dtype=np.float32)
        raise_for(self.ctx.lib.eofx_mat_download_f32(self.ctx.handle, self.handle, ptr(out)), self.ctx.handle)
        return np.ascontiguousarray(out[:, self.valid_index]) if self.masked else out

    def sumsq(self) -> float:
        out = C.c_double()
        raise_for(self.ctx.lib.eofx_mat_sumsq_f64(self.ctx.handle, self.handle, C.byref(out)), self.ctx.handle)
        return out.value

    def gram(self, side: int = 0):
        """side 0: X X^T as an [n_pad, n_pad] float32 device tensor; side 1: X^T X [p_pad, p_pad]
        (rows/columns beyond n / p are zero)."""
        torch = _torch()
        if side and self.masked:
            raise NotImplementedError("feature-space Gram matrix of a masked in-place matrix")
        d = self.p_pad if side else self.n_pad
        G = torch.empty((d, d), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        raise_for(self.ctx.lib.eofx_mat_gram_f32(self.ctx.handle, self.handle, int(side), ptr(G)), self.ctx.handle)
        return G

    def sample_gram(self):
        return self.gram(0)

    def cross_gram(self, other: "ResidentMatrix", side: int = 0):
        """side 0: A_self A_other^T [n_pad, n_pad]; side 1: A_self^T A_other [p_pad, p_pad] (float32 device tensor)"""
        torch = _torch()
        d = self.p_pad if side else self.n_pad
        G = torch.empty((d, d), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        raise_for(self.ctx.lib.eofx_mat_cross_gram_f32(self.ctx.handle, self.handle, other.handle, int(side), ptr(G)),
                  self.ctx.handle)
        return G

    def layout(self):
        """-> (has the feature-contiguous layout, reads the raw field instead): see eofx_ctx_set_layout"""
        hx, hr = C.c_int(), C.c_int()
        self.ctx.lib.eofx_mat_layout(self.handle, C.byref(hx), C.byref(hr))
        return bool(hx.value & 1), bool(hr.value)

    def has_sample_layout(self):
        """False for an in-place matrix until an entry point that needs the sample-contiguous layout has built it"""
        hx = C.c_int()
        self.ctx.lib.eofx_mat_layout(self.handle, C.byref(hx), None)
        return bool(hx.value & 2)

    def ensure_sample_layout(self, only_if_room: bool = True) -> bool:
        """Build the sample-contiguous layout ahead of repeated decompositions on this matrix (their X Y passes run
        faster over it than over the raw field).  only_if_room: only when HBM holds one more copy of the field with 8 GB
        to spare.  -> whether the layout exists afterwards (never for a masked in-place matrix)."""
        built = C.c_int()
        raise_for(self.ctx.lib.eofx_mat_ensure_sample_layout(self.ctx.handle, self.handle, int(only_if_room), C.byref(built)),
                  self.ctx.handle)
        return bool(built.value)

    def release_sample_layout(self):
        """Drop the sample-contiguous layout again (a matrix that still holds the raw field and its map, or the other
        layout, rebuilds it on demand); the memory returns to the context's pool."""
        raise_for(self.ctx.lib.eofx_mat_release_sample_layout(self.ctx.handle, self.handle), self.ctx.handle)

    def release_raw(self):
        """Drop the reference to the raw field (raw / in-place mode); missing layouts are built first / on demand."""
        raise_for(self.ctx.lib.eofx_mat_release_raw(self.ctx.handle, self.handle), self.ctx.handle)
        self._keepalive = None

    def free(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.eofx_mat_destroy(self.ctx.handle, self.handle)
        self.handle = None
        self._keepalive = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _host_staging(shape):
    """float32 host array the engine will upload: page-locked when a GPU is present (the upload of the
    sketch then is a plain DMA instead of a staged pageable copy; torch caches the pinned blocks)"""
    try:
        torch = _torch()
        if torch.cuda.is_available():
            return torch.empty(shape, dtype=torch.float32, pin_memory=True).numpy()
    except Exception:
        pass
    return np.empty(shape, dtype=np.float32)


def _host_out(shape, dtype=np.float32):
    """host array the engine will fill: page-locked above a few MB, so that the download of a 1M-row factor (207 MB of
    components at config 4) is one DMA at PCIe speed instead of a staged pageable copy at a fifth of it"""
    dtype = np.dtype(dtype)
    if int(np.prod(shape, dtype=np.int64)) * dtype.itemsize >= (8 << 20):
        try:
            torch = _torch()
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.complex64): torch.complex64}.get(dtype)
            if tdt is not None and torch.cuda.is_available():
                return torch.empty(tuple(shape), dtype=tdt, pin_memory=True).numpy()
        except Exception:
            pass
    return np.empty(shape, dtype=dtype)


def sketch_matrix(rows: int, size: int, random_state=None) -> np.ndarray:
    """The Gaussian test matrix exactly as scikit-learn draws it for
    randomized_svd (sklearn/utils/extmath.py `_randomized_range_finder`):
    ``check_random_state(seed).normal(size=(A.shape[1], size))`` cast to float32,
    so a given `random_state` means the same thing as in the reference
    (xeofs/linalg/decomposer.py:141-146)."""
    if random_state is None or random_state is np.random:
        rs = np.random.mtrand._rand
    elif isinstance(random_state, numbers.Integral):
        seed = int(random_state)
        if not 0 <= seed < 2 ** 32:
            raise ValueError("Seed must be between 0 and 2**32 - 1")
        # native generator: the same legacy MT19937 / polar-method stream as RandomState(seed).normal,
        # bit for bit (tests/test_abi.py), ~3x faster than numpy for the 600k draws of a 10000 x 60 sketch
        out = _host_staging((rows, size))
        raise_for(_lib.load().eofx_sketch_gaussian_f32(seed, rows, size, ptr(out)))
        return out
    elif isinstance(random_state, np.random.RandomState):
        rs = random_state
    else:
        raise ValueError(f"{random_state!r} cannot be used to seed a numpy.random.RandomState instance")
    return np.ascontiguousarray(rs.normal(size=(rows, size)).astype(np.float32))


class SketchFuture:
    """Draws the sketch matrix on a worker thread so the ~10 ms of legacy-RandomState sampling
    (600k normals at n=10000, l=60) overlap the preprocess kernels of the same fit (the ctypes
    call into the engine releases the GIL).  `.result()` joins."""

    def __init__(self, rows, size, random_state=None):
        import threading

        self.rows, self.size = int(rows), int(size)
        self._out = None
        self._err = None

        def run():
            try:
                self._out = sketch_matrix(rows, size, random_state)
            except Exception as e:  # re-raised in the caller's thread
                self._err = e

        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def result(self):
        self._t.join()
        if self._err is not None:
            raise self._err
        return self._out


class SketchSlice:
    """The leading `rows` rows of a SketchFuture (numpy fills the draw row by row, so they ARE the smaller draw); joined on
    `.result()`, which engine.crosscov_rsvd calls as late as the engine allows."""

    def __init__(self, future, rows):
        self.future, self.rows = future, int(rows)

    def result(self):
        return np.ascontiguousarray(self.future.result()[:self.rows])


def from_dense(ctx: Context, X) -> ResidentMatrix:
    X = _f32c(X)
    n, p = X.shape
    h = C.c_void_p()
    raise_for(ctx.lib.eofx_mat_from_dense_f32(ctx.handle, ptr(X), n, p, p, C.byref(h)), ctx.handle)
    return ResidentMatrix(ctx, h)


def _layout_mode(keep_raw, in_place, allow_masked):
    return (3 if allow_masked else 2) if in_place else int(bool(keep_raw))


def preprocess(ctx: Context, X, center=True, standardize=False, feature_weights=None,
               check_nans=True, want_stats=True, build=True, keep_raw=False, in_place=False, allow_masked=False,
               for_hilbert=False):
    """Scaler + Sanitizer + total variance on the stacked raw (n, P) field.
    Returns (ResidentMatrix | None, stats dict).  keep_raw: raw mode (include/eofx.h, eofx_ctx_set_layout) -- the
    feature-contiguous layout is not written, the products read the raw field through the Scaler map; in_place:
    NO layout is written, both products stream the field where it lies (the sample-contiguous layout is built on
    demand by the entry points that need it).  Either way a device field must stay unmodified until
    `release_raw()` / `free()` (the matrix holds a reference to it); a host field is staged and owned.
    allow_masked (with in_place): a field whose NaN pattern is a mask of all-NaN grid points stays in place as well -- the
    masked features become zero columns of the engine's matrix (layout mode 3, include/eofx.h) and this module
    compacts / scatters the feature axis of every factor, so `mat.p` and the factors have the valid features only.
    for_hilbert (with in_place): `hilbert(ctx, mat, ...)` is the next call on this matrix -- the statistics pass also writes the
    raw field in the sample-contiguous layout (eofx_ctx_set_sample_raw) and the Hilbert stage saves its transposing copy."""
    X = _f32c(X)
    n, P = X.shape
    w = None if feature_weights is None else np.ascontiguousarray(feature_weights, dtype=np.float64)
    if w is not None and w.shape != (P,):
        raise ValueError("feature_weights must have one entry per stacked feature")
    mean = np.empty(P, np.float64) if want_stats else None
    std = np.empty(P, np.float64) if want_stats else None
    vf = np.empty(P, np.uint8)
    vs = np.empty(n, np.uint8)
    n_out, p_out = C.c_int64(), C.c_int64()
    tv = C.c_double()
    h = C.c_void_p()
    ctx.lib.eofx_ctx_set_layout(ctx.handle, _layout_mode(keep_raw, in_place, allow_masked))
    ctx.lib.eofx_ctx_set_sample_raw(ctx.handle, int(bool(for_hilbert and in_place and build)))
    try:
        rc = ctx.lib.eofx_preprocess_f32(ctx.handle, ptr(X), n, P, int(center), int(standardize), ptr(w),
                                         int(check_nans), C.byref(h) if build else None, ptr(mean), ptr(std),
                                         ptr(vf), ptr(vs), C.byref(n_out), C.byref(p_out), C.byref(tv))
    finally:
        ctx.lib.eofx_ctx_set_layout(ctx.handle, 0)
        ctx.lib.eofx_ctx_set_sample_raw(ctx.handle, 0)
    raise_for(rc, ctx.handle)
    stats = dict(mean=mean, std=std, valid_feature=vf.astype(bool), valid_sample=vs.astype(bool),
                 n=n_out.value, p=p_out.value, total_variance=tv.value)
    mat = ResidentMatrix(ctx, h) if build else None
    if mat is not None and (keep_raw or in_place) and hasattr(X, "data_ptr"):
        mat._keepalive = X
    if mat is not None and mat.masked:
        mat.set_valid(stats["valid_feature"])
    return mat, stats


def fit(ctx: Context, X, k: int, center=True, standardize=False, feature_weights=None, check_nans=True,
        n_oversamples: int = 10, n_iter: int | str = "auto", random_state=None, flip: bool = True, omega=None,
        want_stats=True, device_out: bool = False, in_place: bool = True, allow_masked: bool = False):
    """Scaler + Sanitizer + randomized SVD in one engine call (eofx_fit_f32): with the in-place layout the column
    statistics ride on the first pass of the decomposition, so the field is read 2 n_iter + 2 times, not 2 n_iter + 3.
    Same results as `preprocess(..., in_place=True)` followed by `rsvd(...)` (which is what the engine falls back to by
    itself for NaN fields, other precisions, wide sketches, n >= P).
    -> (ResidentMatrix, stats dict (as `preprocess`, plus `fused`), U[n', k], s[k], V[p', k])"""
    X = _f32c(X)
    n, P = X.shape
    k = int(k)
    w = None if feature_weights is None else np.ascontiguousarray(feature_weights, dtype=np.float64)
    if w is not None and w.shape != (P,):
        raise ValueError("feature_weights must have one entry per stacked feature")
    small = min(n, P)
    if omega is None:
        omega = sketch_matrix(small, k + n_oversamples, random_state)
    elif isinstance(omega, SketchFuture):
        omega = omega.result()
    omega = np.ascontiguousarray(omega, dtype=np.float32)
    if omega.ndim != 2 or omega.shape[1] != k + n_oversamples or omega.shape[0] < small:
        raise ValueError(f"omega must have shape {(small, k + n_oversamples)}")
    mean = np.empty(P, np.float64) if want_stats else None
    std = np.empty(P, np.float64) if want_stats else None
    vf = np.empty(P, np.uint8)
    vs = np.empty(n, np.uint8)
    n_out, p_out = C.c_int64(), C.c_int64()
    tv = C.c_double()
    fused = C.c_int()
    h = C.c_void_p()
    if device_out:
        torch = _torch()
        U = torch.empty((n, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
        V = torch.empty((P, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
    else:
        U = _host_out((n, k))
        V = _host_out((P, k))
    s = np.empty(k, np.float32)
    it = -1 if n_iter == "auto" else int(n_iter)
    ctx.lib.eofx_ctx_set_layout(ctx.handle, _layout_mode(False, in_place, allow_masked))
    try:
        rc = ctx.lib.eofx_fit_f32(ctx.handle, ptr(X), n, P, int(center), int(standardize), ptr(w), int(check_nans), k,
                                  int(n_oversamples), it, ptr(omega), omega.shape[0], int(flip), C.byref(h), ptr(mean),
                                  ptr(std), ptr(vf), ptr(vs), C.byref(n_out), C.byref(p_out), C.byref(tv), ptr(U), ptr(s),
                                  ptr(V), C.byref(fused))
    finally:
        ctx.lib.eofx_ctx_set_layout(ctx.handle, 0)
    raise_for(rc, ctx.handle)
    mat = ResidentMatrix(ctx, h)
    if hasattr(X, "data_ptr"):
        mat._keepalive = X
    stats = dict(mean=mean, std=std, valid_feature=vf.astype(bool), valid_sample=vs.astype(bool),
                 n=n_out.value, p=p_out.value, total_variance=tv.value, fused=bool(fused.value))
    if mat.masked:
        mat.set_valid(stats["valid_feature"])
    if mat.n != n or mat.p_phys != P:     # the factors were written densely with the compacted shape
        U = U.reshape(-1)[: mat.n * k].reshape(mat.n, k)
        V = V.reshape(-1)[: mat.p_phys * k].reshape(mat.p_phys, k)
    return mat, stats, U, s, mat.compact_rows(V)


# ---- communicator of the native feature-sharded fit (include/eofx.h: eofx_fit_sharded_f32) -------------------------------
def comm_unique_id() -> bytes:
    """128 bytes from ncclGetUniqueId (rank 0 draws them and hands them to every rank)"""
    buf = C.create_string_buffer(128)
    raise_for(_lib.load().eofx_comm_unique_id(buf))
    return buf.raw


def comm_init_rccl(ctx: Context, unique_id: bytes, world: int, rank: int):
    """attach an RCCL communicator to the context: its collectives are enqueued on the context's own stream"""
    raise_for(ctx.lib.eofx_ctx_comm_init_rccl(ctx.handle, C.c_char_p(bytes(unique_id)), int(world), int(rank)), ctx.handle)
    ctx._comm_cb = None
    ctx._comm_attached = (int(world), int(rank))


def comm_set_callback(ctx: Context, fn, world: int, rank: int):
    """attach a host callback as the collective: fn(ptr, count, dtype, op, stream) -> 0, all-reducing `count` elements of the
    device buffer in place (dtype 0 f32 / 1 f64 / 2 i32, op 0 sum / 1 max / 2 min), in stream order"""
    def tramp(_user, buf, count, dtype, op, stream):
        try:
            return int(fn(buf, int(count), int(dtype), int(op), stream) or 0)
        except BaseException as e:        # must not propagate through the C frames
            ctx._comm_err = e
            return 1

    cb = _lib.ALLREDUCE_FN(tramp)
    raise_for(ctx.lib.eofx_ctx_comm_set_callback(ctx.handle, cb, None, int(world), int(rank)), ctx.handle)
    ctx._comm_cb = cb      # keep the trampoline alive
    ctx._comm_attached = (int(world), int(rank))


def comm_clear(ctx: Context):
    raise_for(ctx.lib.eofx_ctx_comm_clear(ctx.handle), ctx.handle)
    ctx._comm_cb = None
    ctx._comm_attached = None


def comm_attached(ctx: Context):
    """(world, rank) of the communicator attached to the context, or None"""
    return getattr(ctx, "_comm_attached", None)


def comm_selftest(ctx: Context) -> bool:
    """One round of every collective the sharded fit uses on known values (collective: every rank calls it)."""
    ok = C.c_int()
    raise_for(ctx.lib.eofx_ctx_comm_selftest(ctx.handle, C.byref(ok)), ctx.handle)
    return bool(ok.value)


def comm_probe(ctx: Context, cases, reps: int = 20):
    """eofx_ctx_comm_probe (collective): cases = [(count, "f32" | "f64" | "i32"), ...] -> (ranks the communicator reduces over,
    [mean microseconds of an all-reduce(sum) per case])"""
    codes = {"f32": 0, "f64": 1, "i32": 2}
    counts = np.ascontiguousarray([int(c) for c, _ in cases], dtype=np.int64)
    dtypes = np.ascontiguousarray([codes[d] for _, d in cases], dtype=np.int32)
    seen = np.zeros(1, np.float64)
    us = np.zeros(len(cases), np.float64)
    raise_for(ctx.lib.eofx_ctx_comm_probe(ctx.handle, len(cases), ptr(counts), ptr(dtypes), int(reps), ptr(seen), ptr(us)), ctx.handle)
    return float(seen[0]), [float(x) for x in us]


def comm_stats(ctx: Context):
    """-> dict(calls, bytes, ms) of the collectives of the native sharded fit since the last call (ms: while profiling)"""
    calls, nbytes, ms = C.c_int64(), C.c_int64(), C.c_double()
    raise_for(ctx.lib.eofx_ctx_comm_stats(ctx.handle, C.byref(calls), C.byref(nbytes), C.byref(ms)), ctx.handle)
    return dict(calls=calls.value, bytes=nbytes.value, ms=ms.value)


def fit_sharded(ctx: Context, X, k: int, p_total: int, center=True, standardize=False, feature_weights=None,
                n_oversamples: int = 10, n_iter: int | str = "auto", random_state=None, flip: bool = True, omega=None,
                want_stats=True, device_out: bool = False, allow_masked: bool = False):
    """`fit` on this rank's slice X [n, p_local] of a field with `p_total` features, the ranks joined by the communicator
    attached to the context (eofx_fit_sharded_f32: every collective is issued by the engine on its own stream).
    -> (ResidentMatrix, stats, U[n, k] replicated, s[k], V[p_local, k]) or None when some rank cannot take the fused path
    (the ranks agree on that; nothing has been built -- use the panel-level driver, xeofs_amd.sharded)."""
    X = _f32c(X)
    n, P = X.shape
    k = int(k)
    w = None if feature_weights is None else np.ascontiguousarray(feature_weights, dtype=np.float64)
    if w is not None and w.shape != (P,):
        raise ValueError("feature_weights must have one entry per stacked feature of the slice")
    if omega is None:
        omega = sketch_matrix(n, k + n_oversamples, random_state)
    elif hasattr(omega, "result"):
        omega = omega.result()
    omega = np.ascontiguousarray(omega, dtype=np.float32)
    if omega.ndim != 2 or omega.shape[1] != k + n_oversamples or omega.shape[0] < n:
        raise ValueError(f"omega must have shape {(n, k + n_oversamples)}")
    mean = np.empty(P, np.float64) if want_stats else None
    std = np.empty(P, np.float64) if want_stats else None
    vf = np.empty(P, np.uint8)
    tv = C.c_double()
    h = C.c_void_p()
    if device_out:
        torch = _torch()
        U = torch.empty((n, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
        V = torch.empty((P, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
    else:
        U = _host_out((n, k))
        V = _host_out((P, k))
    s = np.empty(k, np.float32)
    it = -1 if n_iter == "auto" else int(n_iter)
    ctx._comm_err = None
    ctx.lib.eofx_ctx_set_layout(ctx.handle, _layout_mode(False, True, allow_masked))
    try:
        rc = ctx.lib.eofx_fit_sharded_f32(ctx.handle, ptr(X), n, P, int(p_total), int(center), int(standardize), ptr(w), k,
                                          int(n_oversamples), it, ptr(omega), omega.shape[0], int(flip), C.byref(h),
                                          ptr(mean), ptr(std), ptr(vf), C.byref(tv), ptr(U), ptr(s), ptr(V))
    finally:
        ctx.lib.eofx_ctx_set_layout(ctx.handle, 0)
    if getattr(ctx, "_comm_err", None) is not None:
        raise ctx._comm_err
    if rc == 1:
        return None
    raise_for(rc, ctx.handle)
    mat = ResidentMatrix(ctx, h)
    if hasattr(X, "data_ptr"):
        mat._keepalive = X
    stats = dict(mean=mean, std=std, valid_feature=vf.astype(bool), valid_sample=np.ones(n, bool), n=n,
                 p=mat.p, total_variance=tv.value, fused=True)
    if mat.masked:
        mat.set_valid(stats["valid_feature"])
    return mat, stats, U, s, mat.compact_rows(V)


def fit_first(ctx: Context, X, Zn, l: int, center=True, standardize=False, feature_weights=None, check_nans=True,
              want_stats=True):
    """The statistics-carrying first pass on its own (eofx_fit_first_f32): Yp = X'^T Zn for the device panel Zn
    [n_pad, L] (first l < L columns in use) together with the preprocessing of X (in-place layout).
    -> (ResidentMatrix, stats dict as `preprocess` plus `fused`, Yp [p_pad, L] device tensor or None when samples were
    dropped and the caller has to redo the product with the surviving rows of Z)"""
    torch = _torch()
    X = _f32c(X)
    n, P = X.shape
    w = None if feature_weights is None else np.ascontiguousarray(feature_weights, dtype=np.float64)
    if w is not None and w.shape != (P,):
        raise ValueError("feature_weights must have one entry per stacked feature")
    L = Zn.shape[1]
    Yp = torch.empty(((P + 511) // 512 * 512, L), dtype=torch.float32, device=Zn.device)
    mean = np.empty(P, np.float64) if want_stats else None
    std = np.empty(P, np.float64) if want_stats else None
    vf = np.empty(P, np.uint8)
    vs = np.empty(n, np.uint8)
    n_out, p_out = C.c_int64(), C.c_int64()
    tv = C.c_double()
    fused = C.c_int()
    h = C.c_void_p()
    ctx.lib.eofx_ctx_set_layout(ctx.handle, 2)
    try:
        rc = ctx.lib.eofx_fit_first_f32(ctx.handle, ptr(X), n, P, int(center), int(standardize), ptr(w), int(check_nans),
                                        ptr(Zn), L, int(l), ptr(Yp), C.byref(h), ptr(mean), ptr(std), ptr(vf), ptr(vs),
                                        C.byref(n_out), C.byref(p_out), C.byref(tv), C.byref(fused))
    finally:
        ctx.lib.eofx_ctx_set_layout(ctx.handle, 0)
    raise_for(rc, ctx.handle)
    mat = ResidentMatrix(ctx, h)
    if hasattr(X, "data_ptr"):
        mat._keepalive = X
    stats = dict(mean=mean, std=std, valid_feature=vf.astype(bool), valid_sample=vs.astype(bool),
                 n=n_out.value, p=p_out.value, total_variance=tv.value, fused=bool(fused.value))
    if n_out.value != n:
        Yp = None
    elif mat.p_pad != Yp.shape[0]:
        Yp = Yp[: mat.p_pad].contiguous()
    return mat, stats, Yp


def fit_info(ctx: Context):
    """-> dict(fused, preprocess_ms, reason) of the last `fit` on this context (include/eofx.h, eofx_ctx_fit_info)"""
    info = (C.c_double * 3)()
    raise_for(ctx.lib.eofx_ctx_fit_info(ctx.handle, info), ctx.handle)
    return dict(fused=bool(info[0]), preprocess_ms=float(info[1]), reason=int(info[2]))


def last_iterations(ctx: Context) -> int:
    """power iterations of the last `rsvd_c64` on this context (its n_iter="auto" iterates until the Ritz values stand still)"""
    out = C.c_int()
    raise_for(ctx.lib.eofx_ctx_last_iterations(ctx.handle, C.byref(out)), ctx.handle)
    return out.value


def apply(ctx: Context, X, mean, std, feature_weights, valid_feature, check_nans=True, in_place=False, allow_masked=False):
    """Preprocessor.transform on new data with fitted state.  in_place: as in `preprocess` -- nothing is written, the
    projection that follows streams the (staged) field through the fitted map."""
    X = _f32c(X)
    n, P = X.shape
    f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    mean, std, w = f64(mean), f64(std), f64(feature_weights)
    vf = np.ascontiguousarray(valid_feature, dtype=np.uint8)
    vs = np.empty(n, np.uint8)
    n_out = C.c_int64()
    h = C.c_void_p()
    ctx.lib.eofx_ctx_set_layout(ctx.handle, _layout_mode(False, in_place, allow_masked))
    try:
        rc = ctx.lib.eofx_apply_f32(ctx.handle, ptr(X), n, P, ptr(mean), ptr(std), ptr(w), ptr(vf),
                                    int(check_nans), C.byref(h), ptr(vs), C.byref(n_out))
    finally:
        ctx.lib.eofx_ctx_set_layout(ctx.handle, 0)
    raise_for(rc, ctx.handle)
    mat = ResidentMatrix(ctx, h)
    if in_place and hasattr(X, "data_ptr"):
        mat._keepalive = X
    if mat.masked:
        mat.set_valid(vf)
    return mat, vs.astype(bool)


def rsvd(ctx: Context, mat: ResidentMatrix, k: int, n_oversamples: int = 10, n_iter: int | str = "auto",
         random_state=None, flip: bool = True, omega=None, device_out: bool = False):
    """randomized SVD of the resident matrix -> (U[n,k], s[k], V[p,k]) float32 host arrays
    (or torch device tensors for U and V with `device_out=True`: nothing crosses PCIe)."""
    k = int(k)
    small = min(mat.n, mat.p)
    if omega is None:
        omega = sketch_matrix(small, k + n_oversamples, random_state)
    omega = np.ascontiguousarray(omega, dtype=np.float32)
    if omega.shape != (small, k + n_oversamples):
        raise ValueError(f"omega must have shape {(small, k + n_oversamples)}")
    if mat.masked and mat.p < mat.n:
        raise NotImplementedError("masked in-place matrix with fewer valid features than samples")   # the engine never builds one
    if device_out:
        torch = _torch()
        U = torch.empty((mat.n, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
        V = torch.empty((mat.p_phys, k), dtype=torch.float32, device=f"cuda:{ctx.device}")
    else:
        U = _host_out((mat.n, k))
        V = _host_out((mat.p_phys, k))
    s = np.empty(k, np.float32)
    it = -1 if n_iter == "auto" else int(n_iter)
    rc = ctx.lib.eofx_rsvd_f32(ctx.handle, mat.handle, k, int(n_oversamples), it, ptr(omega), int(flip),
                               ptr(U), ptr(s), ptr(V))
    raise_for(rc, ctx.handle)
    return U, s, mat.compact_rows(V)


def project(ctx: Context, mat: ResidentMatrix, V) -> np.ndarray:
    V = _f32c(V)
    if V.ndim != 2 or V.shape[0] != mat.p:
        raise ValueError(f"components have {V.shape[0] if V.ndim == 2 else V.shape} features, the matrix has {mat.p} "
                         "(a different NaN pattern in the new data?)")
    k = V.shape[1]
    out = np.empty((mat.n, k), np.float32)
    V = mat.scatter_rows(V)
    raise_for(ctx.lib.eofx_project_f32(ctx.handle, mat.handle, ptr(V), k, ptr(out)), ctx.handle)
    return out


def reconstruct(ctx: Context, S, V) -> np.ndarray:
    S, V = _f32c(S), _f32c(V)
    if S.ndim != 2 or V.ndim != 2 or S.shape[1] != V.shape[1]:
        raise ValueError(f"scores {S.shape} and components {V.shape} do not have the same number of modes")
    n, k = S.shape
    p = V.shape[0]
    out = np.empty((n, p), np.float32)
    raise_for(ctx.lib.eofx_reconstruct_f32(ctx.handle, ptr(S), ptr(V), n, p, k, ptr(out)), ctx.handle)
    return out


def crosscov_rsvd(ctx: Context, x: ResidentMatrix, y: ResidentMatrix, k: int, n_oversamples: int = 10,
                  n_iter: int | str = "auto", random_state=None, flip: bool = True, omega=None,
                  want_tsc: bool = True):
    """Matrix-free rSVD of C = X^T Y/(n-1) -> dict (cpcca.py:168-225 quantities)."""
    k = int(k)
    small = min(x.p, y.p)
    if (x.masked or y.masked) and k > small:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {small}).")
    # the engine orients C by the VALID feature counts, as the reference does (sklearn transposes when rows < cols; round 6 --
    # before, it used the physical widths and pairs whose two orders disagree had to be compacted)
    small_mat = x if x.p < y.p else y
    if small_mat.masked and k + n_oversamples >= small:
        raise NotImplementedError("a sketch as wide as the rank on a masked in-place matrix (compact the field)")
    if x.n != y.n:
        raise ValueError(f"Both data matrices must have the same number of samples but found {x.n} in the first and "
                         f"{y.n} in the second.")

    def sketch():
        om = omega.result() if hasattr(omega, "result") else omega     # a SketchFuture is joined as late as possible
        if om is None:
            om = sketch_matrix(small, k + n_oversamples, random_state)
        om = np.ascontiguousarray(om, dtype=np.float32)
        if om.shape != (small, k + n_oversamples):
            raise ValueError(f"omega must have shape {(small, k + n_oversamples)}")
        # masked in-place matrices keep their all-NaN grid points as zero columns: the sketch gets zero rows there, the
        # singular vectors come back with zero rows there (compacted below)
        return np.ascontiguousarray(small_mat.scatter_rows(om))

    n = x.n
    Q1 = np.empty((x.p_phys, k), np.float32)
    Q2 = np.empty((y.p_phys, k), np.float32)
    s = np.empty(k, np.float32)
    s1 = np.empty((n, k), np.float32)
    s2 = np.empty((n, k), np.float32)
    n1 = np.empty(k, np.float32)
    n2 = np.empty(k, np.float32)
    tsc = C.c_double(float("nan"))
    it = -1 if n_iter == "auto" else int(n_iter)
    # the engine asks for the sketch when it first needs it (eofx_crosscov_rsvd_lazy_f32): with the total squared
    # covariance wanted, the Gram matrices are queued before that and a SketchFuture finishes beside them
    held = {}

    def provide(_user):
        try:
            held["om"] = sketch()
            return held["om"].ctypes.data
        except BaseException as e:      # re-raised below, in the caller's frame
            held["err"] = e
            return None

    cb = _lib.SKETCH_FN(provide)
    rc = ctx.lib.eofx_crosscov_rsvd_lazy_f32(ctx.handle, x.handle, y.handle, k, int(n_oversamples), it, cb, None,
                                             int(flip), ptr(Q1), ptr(s), ptr(Q2), ptr(s1), ptr(s2), ptr(n1), ptr(n2),
                                             C.byref(tsc) if want_tsc else None)
    if "err" in held:
        raise held["err"]
    raise_for(rc, ctx.handle)
    return dict(Q1=x.compact_rows(Q1), Q2=y.compact_rows(Q2), s=s, scores1=s1, scores2=s2, norm1=n1, norm2=n2,
                total_squared_covariance=tsc.value)


def comm_allreduce_host(ctx: Context, values, op: str = "sum") -> np.ndarray:
    """all-reduce of a small host vector of doubles over the communicator attached to the context
    (eofx_ctx_comm_allreduce_f64): the global facts a sharded model needs between engine calls"""
    buf = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
    ctx._comm_err = None
    rc = ctx.lib.eofx_ctx_comm_allreduce_f64(ctx.handle, ptr(buf), buf.size, {"sum": 0, "max": 1, "min": 2}[op])
    if getattr(ctx, "_comm_err", None) is not None:
        raise ctx._comm_err
    raise_for(rc, ctx.handle)
    return buf


def crosscov_rsvd_sharded(ctx: Context, x: ResidentMatrix, y: ResidentMatrix, k: int, p1_total: int, p1_offset: int,
                          p2_total: int, p2_offset: int, n_oversamples: int = 10, n_iter: int | str = "auto",
                          random_state=None, flip: bool = True, omega=None, want_tsc: bool = True):
    """`crosscov_rsvd` with both fields sharded along their feature axes (eofx_crosscov_rsvd_sharded_f32; the communicator is
    the one attached to the context).  x / y: this rank's slices; p*_total / p*_offset: feature counts over all ranks and this
    slice's position on the global VALID-feature axis.  omega: the GLOBAL sketch [min(p1_total, p2_total), k + n_oversamples]
    (one draw, identical on every rank; drawn here from random_state when None).  Q1 / Q2 come back as this rank's rows."""
    k = int(k)
    if x.n != y.n:
        raise ValueError(f"Both data matrices must have the same number of samples but found {x.n} in the first and "
                         f"{y.n} in the second.")
    small = min(int(p1_total), int(p2_total))
    if k > small:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {small}).")
    l_req = k + int(n_oversamples)
    if l_req > small:
        raise ValueError("sketch wider than rank not supported on the cross path")
    on_x = int(p1_total) < int(p2_total)            # the sketch lives on the narrower field's feature axis
    sm, off = (x, int(p1_offset)) if on_x else (y, int(p2_offset))
    om = omega.result() if hasattr(omega, "result") else omega
    if l_req == small:      # full-width sketch: the identity (see eofx_rsvd_f32)
        om = np.eye(small, dtype=np.float32)
    elif om is None:
        om = sketch_matrix(small, l_req, random_state)
    om = np.asarray(om, dtype=np.float32)
    if om.shape != (small, l_req):
        raise ValueError(f"omega must have shape {(small, l_req)}")
    rows = np.ascontiguousarray(sm.scatter_rows(np.ascontiguousarray(om[off:off + sm.p])), dtype=np.float32)
    if rows.shape[0] == 0:
        rows = np.zeros((1, l_req), np.float32)
    n = x.n
    Q1 = np.empty((max(x.p_phys, 1), k), np.float32)
    Q2 = np.empty((max(y.p_phys, 1), k), np.float32)
    s = np.empty(k, np.float32)
    s1 = np.empty((n, k), np.float32)
    s2 = np.empty((n, k), np.float32)
    n1 = np.empty(k, np.float32)
    n2 = np.empty(k, np.float32)
    tsc = C.c_double(float("nan"))
    it = -1 if n_iter == "auto" else int(n_iter)
    ctx._comm_err = None
    rc = ctx.lib.eofx_crosscov_rsvd_sharded_f32(ctx.handle, x.handle, y.handle, int(p1_total), int(p1_offset), int(p2_total),
                                                int(p2_offset), k, int(n_oversamples), it, ptr(rows), int(flip), ptr(Q1),
                                                ptr(s), ptr(Q2), ptr(s1), ptr(s2), ptr(n1), ptr(n2),
                                                C.byref(tsc) if want_tsc else None)
    if getattr(ctx, "_comm_err", None) is not None:
        raise ctx._comm_err
    raise_for(rc, ctx.handle)
    return dict(Q1=x.compact_rows(Q1[:x.p_phys]), Q2=y.compact_rows(Q2[:y.p_phys]), s=s, scores1=s1, scores2=s2, norm1=n1,
                norm2=n2, total_squared_covariance=tsc.value)


def host_eigh(A: np.ndarray):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    w = np.empty(n)
    V = np.empty((n, n))
    raise_for(_lib.load().eofx_host_eigh_f64(ptr(A), n, ptr(w), ptr(V)))
    return w, V


# --------------------------------------------------------------------------- #
# panel-level steps on torch device tensors (used by the feature-sharded path)  #
# --------------------------------------------------------------------------- #
def _torch():
    import torch

    return torch


def panel_width(l: int) -> int:
    return (int(l) + 31) // 32 * 32


def panel_import(ctx: Context, src, rows_pad: int, L: int):
    torch = _torch()
    src = _f32c(src)
    rows, l = src.shape
    P = torch.empty((rows_pad, L), dtype=torch.float32, device=f"cuda:{ctx.device}")
    raise_for(ctx.lib.eofx_panel_import_f32(ctx.handle, ptr(src), rows, l, ptr(P), rows_pad, L), ctx.handle)
    return P


def panel_export(ctx: Context, P, rows: int, k: int, sign=None, device_out: bool = False):
    if device_out:
        torch = _torch()
        out = torch.empty((rows, k), dtype=torch.float32, device=P.device)
    else:
        out = np.empty((rows, k), np.float32)
    sg = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
    raise_for(ctx.lib.eofx_panel_export_f32(ctx.handle, ptr(P), rows, P.shape[1], k, ptr(sg), ptr(out)), ctx.handle)
    return out


def panel_tmul(ctx: Context, mat: ResidentMatrix, Zn, out=None, prec="f32"):
    """Yp[p_pad, L] = X^T Zn[n_pad, L]"""
    torch = _torch()
    L = Zn.shape[1]
    if out is None:
        out = torch.empty((mat.p_pad, L), dtype=torch.float32, device=Zn.device)
    raise_for(ctx.lib.eofx_panel_tmul_f32(ctx.handle, mat.handle, ptr(Zn), ptr(out), L, _lib.PREC[prec]), ctx.handle)
    return out


def panel_mul(ctx: Context, mat: ResidentMatrix, Yp, out=None, prec="f32"):
    """Wn[n_pad, L] = X Yp[p_pad, L]"""
    torch = _torch()
    L = Yp.shape[1]
    if out is None:
        out = torch.empty((mat.n_pad, L), dtype=torch.float32, device=Yp.device)
    raise_for(ctx.lib.eofx_panel_mul_f32(ctx.handle, mat.handle, ptr(Yp), ptr(out), L, _lib.PREC[prec]), ctx.handle)
    return out


def panel_gram(ctx: Context, P, out=None):
    torch = _torch()
    L = P.shape[1]
    if out is None:
        out = torch.empty((L, L), dtype=torch.float64, device=P.device)
    raise_for(ctx.lib.eofx_panel_gram_f64(ctx.handle, ptr(P), P.shape[0], L, ptr(out)), ctx.handle)
    return out


def panel_cholqr(ctx: Context, P, l: int, G, out=None):
    torch = _torch()
    if out is None:
        out = torch.empty_like(P)
    raise_for(ctx.lib.eofx_panel_cholqr_f32(ctx.handle, ptr(P), P.shape[0], P.shape[1], int(l), ptr(G), ptr(out)),
              ctx.handle)
    return out


def panel_rinv(ctx: Context, G, l: int):
    """R^-1 (L x L float64 device tensor) of the Cholesky factor of the leading l x l block of G"""
    torch = _torch()
    out = torch.empty_like(G)
    raise_for(ctx.lib.eofx_panel_rinv_f64(ctx.handle, ptr(G), G.shape[0], int(l), ptr(out)), ctx.handle)
    return out


def panel_matmul(ctx: Context, P, M, out=None):
    torch = _torch()
    Lo = M.shape[1]
    if out is None:
        out = torch.empty((P.shape[0], Lo), dtype=torch.float32, device=P.device)
    raise_for(ctx.lib.eofx_panel_matmul_f32(ctx.handle, ptr(P), P.shape[0], P.shape[1], ptr(M), Lo, ptr(out)),
              ctx.handle)
    return out


def panel_colminmax(ctx: Context, P, rows: int):
    torch = _torch()
    L = P.shape[1]
    mx = torch.empty(L, dtype=torch.float32, device=P.device)
    mn = torch.empty(L, dtype=torch.float32, device=P.device)
    raise_for(ctx.lib.eofx_panel_colminmax_f32(ctx.handle, ptr(P), int(rows), L, ptr(mx), ptr(mn)), ctx.handle)
    return mx, mn


# --------------------------------------------------------------------------- #
# complex / Hilbert path                                                        #
# --------------------------------------------------------------------------- #
def hilbert(ctx: Context, mat: ResidentMatrix, padding="exp", decay_factor: float = 0.2, want_real: bool = False):
    """Analytic signal along the sample axis (xeofs/utils/hilbert_transform.py:40-72).
    Returns (imag ResidentMatrix, real ResidentMatrix | None).  As in the reference only
    padding == "exp" pads; any other value means no padding."""
    if mat.masked and want_real:      # (checked BEFORE the call: its two result matrices would leak)
        raise NotImplementedError("re-centred real part of a masked in-place matrix (preprocess with center=True)")
    hi, hr = C.c_void_p(), C.c_void_p()
    rc = ctx.lib.eofx_hilbert_f32(ctx.handle, mat.handle, int(padding == "exp"), float(decay_factor),
                                  C.byref(hi), C.byref(hr) if want_real else None)
    raise_for(rc, ctx.handle)
    B = ResidentMatrix(ctx, hi)
    # a masked in-place input (all-NaN grid points kept as zero columns): Im is a plain written matrix over the SAME physical
    # columns, zeros at the masked ones; `rsvd_c64(ctx, mat, B, ...)` compacts the feature axis of its factors through `mat`
    return B, (ResidentMatrix(ctx, hr) if want_real else None)


def cmat_mul(ctx: Context, A: ResidentMatrix, B: ResidentMatrix, P, conj_left: bool, final: bool = False):
    """one pass of Z = A + iB over a [Re | Im] panel: Z^H P (conj_left, P on the sample side) or Z P -- one launch"""
    torch = _torch()
    out = torch.empty((A.p_pad if conj_left else A.n_pad, P.shape[1]), dtype=torch.float32, device=P.device)
    raise_for(ctx.lib.eofx_cmat_mul_f32(ctx.handle, A.handle, B.handle, int(conj_left), ptr(P), P.shape[1], int(final),
                                        ptr(out)), ctx.handle)
    return out


def cpanel_combine(ctx: Context, P1, P2, conj_left: bool, out=None):
    torch = _torch()
    if out is None:
        out = torch.empty_like(P1)
    raise_for(ctx.lib.eofx_cpanel_combine_f32(ctx.handle, ptr(P1), ptr(P2), int(conj_left), P1.shape[0],
                                              P1.shape[1], ptr(out)), ctx.handle)
    return out


def panel_colargminmax(ctx: Context, P, rows: int):
    torch = _torch()
    L = P.shape[1]
    amax = torch.empty(L, dtype=torch.int64, device=P.device)
    amin = torch.empty(L, dtype=torch.int64, device=P.device)
    raise_for(ctx.lib.eofx_panel_colargminmax_f32(ctx.handle, ptr(P), int(rows), L, ptr(amax), ptr(amin)), ctx.handle)
    return amax, amin


# --------------------------------------------------------------------------- #
# rotation steps                                                                #
# --------------------------------------------------------------------------- #
def panel_row_normalize(ctx: Context, P, out=None):
    torch = _torch()
    if out is None:
        out = torch.empty_like(P)
    raise_for(ctx.lib.eofx_panel_row_normalize_f32(ctx.handle, ptr(P), P.shape[0], P.shape[1], ptr(out)), ctx.handle)
    return out


def panel_rot_step(ctx: Context, X, R, aux, mode: int, power: float = 1.0):
    """one pass over the normalised loadings panel -> L x L float64 (device) ; see include/eofx.h"""
    torch = _torch()
    L = X.shape[1]
    G = torch.empty((L, L), dtype=torch.float64, device=X.device)
    raise_for(ctx.lib.eofx_panel_rot_step_f64(ctx.handle, ptr(X), X.shape[0], L, ptr(R), ptr(aux), int(mode),
                                              float(power), ptr(G)), ctx.handle)
    return G


def cpanel_colabsmax(ctx: Context, P, rows: int):
    """max over the rows of |column| for the L/2 complex columns of a [Re | Im] panel -> float32 device tensor [L/2]"""
    torch = _torch()
    L = P.shape[1]
    out = torch.empty(L // 2, dtype=torch.float32, device=P.device)
    raise_for(ctx.lib.eofx_cpanel_colabsmax_f32(ctx.handle, ptr(P), int(rows), L, ptr(out)), ctx.handle)
    return out


def vec_dot(ctx: Context, a, b) -> float:
    """float64 dot product of two equally sized float32 device tensors (fixed reduction tree)"""
    out = C.c_double()
    raise_for(ctx.lib.eofx_vec_dot_f64(ctx.handle, ptr(a), ptr(b), a.numel(), C.byref(out)), ctx.handle)
    return out.value


def resample(ctx: Context, mat: ResidentMatrix, rows, center: bool = True):
    """Bootstrap member of a resident matrix: rows drawn with replacement, re-centred.
    -> (ResidentMatrix, mean[p] float64, total_variance)"""
    if mat.masked:
        raise NotImplementedError("resampled copy of a masked in-place matrix (the bootstrapper works on the matrix in place)")
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    mean = np.empty(mat.p, np.float64)
    tv = C.c_double()
    h = C.c_void_p()
    raise_for(ctx.lib.eofx_resample_f32(ctx.handle, mat.handle, ptr(rows), rows.size, int(center), C.byref(h), ptr(mean),
                                        C.byref(tv)), ctx.handle)
    return ResidentMatrix(ctx, h), mean, tv.value


def panel_bootstrap(ctx: Context, P, n: int, idx, order, rowptr, transpose: bool):
    """H P (transpose False) or H^T P (True) of a bootstrap member on a sample-side panel (eofx_panel_bootstrap_f32);
    idx / order / rowptr: int64 device tensors describing the draw"""
    torch = _torch()
    out = torch.empty_like(P)
    raise_for(ctx.lib.eofx_panel_bootstrap_f32(ctx.handle, ptr(P), int(n), P.shape[0], P.shape[1], ptr(idx), ptr(order),
                                               ptr(rowptr), int(bool(transpose)), ptr(out)), ctx.handle)
    return out


def panel_rownorm(ctx: Context, P, rows: int) -> np.ndarray:
    out = np.empty(rows, np.float64)
    raise_for(ctx.lib.eofx_panel_rownorm_f64(ctx.handle, ptr(P), rows, P.shape[1], ptr(out)), ctx.handle)
    return out


def feature_norms(ctx: Context, mat: ResidentMatrix) -> np.ndarray:
    """sqrt(sum over samples of x^2) per feature of the resident matrix"""
    out = np.empty(mat.p_phys, np.float64)
    raise_for(ctx.lib.eofx_mat_feature_norms_f64(ctx.handle, mat.handle, ptr(out)), ctx.handle)
    return out[mat.valid_index] if mat.masked else out


def sample_norms(ctx: Context, mat: ResidentMatrix) -> np.ndarray:
    """sqrt(sum over features of x^2) per sample of the resident matrix (no layout is built for an in-place matrix)"""
    out = np.empty(mat.n, np.float64)
    raise_for(ctx.lib.eofx_mat_sample_norms_f64(ctx.handle, mat.handle, ptr(out)), ctx.handle)
    return out


def _c64_arguments(ctx: Context, A: ResidentMatrix, k: int, n_oversamples: int, n_iter, random_state, omega, device_out: bool):
    """the start matrix, the iteration rule and the output buffers shared by the two complex decompositions"""
    r = min(A.n, A.p)
    if k > r:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {r}).")
    l = min(k + n_oversamples, r)
    if l == r:      # full-width sketch spans everything: identity, not an ill-conditioned square Gaussian
        omega = np.zeros((r, k + n_oversamples), np.float32)
        omega[np.arange(l), np.arange(l)] = 1.0
    elif omega is None:
        omega = np.ascontiguousarray(sketch_matrix(r, k + n_oversamples, random_state), dtype=np.float32)
    else:
        omega = np.ascontiguousarray(omega, dtype=np.float32)
        if omega.shape != (r, k + n_oversamples):
            raise ValueError(f"omega must have shape {(r, k + n_oversamples)}")
    # "auto": scikit-learn's count, as the real branch; "converge": until the Ritz values stand still (at most 20 iterations)
    it = -1 if n_iter in ("auto", None) else -2 if n_iter == "converge" else int(n_iter)
    if device_out:
        torch = _torch()
        dev = f"cuda:{ctx.device}"
        U = torch.empty((A.n, k), dtype=torch.complex64, device=dev)
        V = torch.empty((A.p_phys, k), dtype=torch.complex64, device=dev)
    else:
        U = _host_out((A.n, k), np.complex64)
        V = _host_out((A.p_phys, k), np.complex64)
    return omega, it, U, np.empty(k, np.float32), V


def rsvd_c64(ctx: Context, A: ResidentMatrix, B: ResidentMatrix, k: int, n_oversamples: int = 10, n_iter="auto",
             random_state=None, flip: bool = True, omega=None, device_out: bool = False):
    """complex randomized SVD of Z = A + iB (eofx_rsvd_c64) -> (U[n,k] complex64, s[k] float32, V[p,k] complex64);
    device_out: U and V stay on the device as torch complex64 tensors (V is 8 p k bytes: 166 MB at config 5)"""
    k = int(k)
    if B.masked:
        raise NotImplementedError("complex rSVD with a masked imaginary part (only the real part may be a masked in-place matrix)")
    if A.masked and (B.p_phys != A.p_phys or A.p < A.n):
        raise NotImplementedError("masked in-place real part: the imaginary part must cover the same physical columns and n < p")
    omega, it, U, s, V = _c64_arguments(ctx, A, k, n_oversamples, n_iter, random_state, omega, device_out)
    raise_for(ctx.lib.eofx_rsvd_c64(ctx.handle, A.handle, B.handle, k, int(n_oversamples), it, ptr(omega), int(flip),
                                    ptr(U), ptr(s), ptr(V)), ctx.handle)
    return U, s, A.compact_rows(V)       # (masked real part: the rows of the valid features)


def _c64_sharded_arguments(ctx: Context, A: ResidentMatrix, k: int, p_total: int, n_oversamples: int, n_iter, random_state,
                           omega, device_out: bool):
    """as `_c64_arguments` for a slice of a field with `p_total` valid features over all ranks (n < p_total: the start matrix
    lives on the replicated sample side -- one draw, identical on every rank)"""
    n = A.n
    if not n < int(p_total):
        raise ValueError("the sharded complex decomposition needs more features over all ranks than samples")
    if k > n:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {n}).")
    l = min(k + n_oversamples, n)
    if l == n:
        omega = np.zeros((n, k + n_oversamples), np.float32)
        omega[np.arange(l), np.arange(l)] = 1.0
    elif omega is None:
        omega = np.ascontiguousarray(sketch_matrix(n, k + n_oversamples, random_state), dtype=np.float32)
    else:
        omega = np.ascontiguousarray(omega.result() if hasattr(omega, "result") else omega, dtype=np.float32)
        if omega.shape != (n, k + n_oversamples):
            raise ValueError(f"omega must have shape {(n, k + n_oversamples)}")
    it = -1 if n_iter in ("auto", None) else -2 if n_iter == "converge" else int(n_iter)
    rows_v = max(A.p_phys, 1)
    if device_out:
        torch = _torch()
        dev = f"cuda:{ctx.device}"
        U = torch.empty((n, k), dtype=torch.complex64, device=dev)
        V = torch.empty((rows_v, k), dtype=torch.complex64, device=dev)
    else:
        U = _host_out((n, k), np.complex64)
        V = _host_out((rows_v, k), np.complex64)
    return omega, it, U, np.empty(k, np.float32), V


def rsvd_sharded_c64(ctx: Context, A: ResidentMatrix, B: ResidentMatrix, k: int, p_total: int, n_oversamples: int = 10,
                     n_iter="auto", random_state=None, flip: bool = True, omega=None, device_out: bool = False):
    """`rsvd_c64` on this rank's slice of the feature axis (eofx_rsvd_sharded_c64; collectives through the communicator
    attached to the context) -> (U[n,k] replicated, s[k], V[p_local,k])"""
    k = int(k)
    if B.masked or (A.masked and B.p_phys != A.p_phys):
        raise NotImplementedError("masked in-place real part: the imaginary part must be a written matrix over the same physical columns")
    omega, it, U, s, V = _c64_sharded_arguments(ctx, A, k, p_total, n_oversamples, n_iter, random_state, omega, device_out)
    ctx._comm_err = None
    rc = ctx.lib.eofx_rsvd_sharded_c64(ctx.handle, A.handle, B.handle, int(p_total), k, int(n_oversamples), it, ptr(omega),
                                       int(flip), ptr(U), ptr(s), ptr(V))
    if getattr(ctx, "_comm_err", None) is not None:
        raise ctx._comm_err
    raise_for(rc, ctx.handle)
    return U, s, A.compact_rows(V[:A.p_phys])


def rsvd_hilbert_sharded_c64(ctx: Context, A: ResidentMatrix, k: int, p_total: int, padding="exp", decay_factor: float = 0.2,
                             n_oversamples: int = 10, n_iter="auto", random_state=None, flip: bool = True, omega=None,
                             device_out: bool = False):
    """`rsvd_hilbert_c64` on this rank's slice (eofx_rsvd_hilbert_sharded_c64): the operator route of the analytic signal,
    feature-sharded -- every pass streams the rank's REAL slice once, the Hilbert operator acts on the replicated sample-side
    panel.  Reference: single/eof.py:546-555 -> linalg/decomposer.py:149-160."""
    k = int(k)
    omega, it, U, s, V = _c64_sharded_arguments(ctx, A, k, p_total, n_oversamples, n_iter, random_state, omega, device_out)
    ctx._comm_err = None
    rc = ctx.lib.eofx_rsvd_hilbert_sharded_c64(ctx.handle, A.handle, int(p_total), int(padding == "exp"), float(decay_factor), k,
                                               int(n_oversamples), it, ptr(omega), int(flip), ptr(U), ptr(s), ptr(V))
    if getattr(ctx, "_comm_err", None) is not None:
        raise ctx._comm_err
    raise_for(rc, ctx.handle)
    return U, s, A.compact_rows(V[:A.p_phys])


HILBERT_OPERATOR_MAX_SAMPLES = 16384      # eofx_rsvd_hilbert_c64 holds the n x n operator resident (2 GB at the limit)


def rsvd_hilbert_c64(ctx: Context, A: ResidentMatrix, k: int, padding="exp", decay_factor: float = 0.2, n_oversamples: int = 10,
                     n_iter="auto", random_state=None, flip: bool = True, omega=None, device_out: bool = False):
    """complex randomized SVD of the analytic signal Z = A + i H(A) WITHOUT its imaginary part in memory
    (eofx_rsvd_hilbert_c64): the Hilbert stage is one n x n matrix along the samples, applied to the sample-side panels,
    and every product streams the real field once.  Same results as `rsvd_c64(ctx, A, hilbert(ctx, A)[0], ...)`
    (reference single/eof.py:433-447 + decomposer.py:149-160)."""
    k = int(k)
    if A.masked and A.p < A.n:
        raise NotImplementedError("masked in-place matrix with fewer valid features than samples")
    omega, it, U, s, V = _c64_arguments(ctx, A, k, n_oversamples, n_iter, random_state, omega, device_out)
    raise_for(ctx.lib.eofx_rsvd_hilbert_c64(ctx.handle, A.handle, int(padding == "exp"), float(decay_factor), k,
                                            int(n_oversamples), it, ptr(omega), int(flip), ptr(U), ptr(s), ptr(V)), ctx.handle)
    return U, s, A.compact_rows(V)


def hilbert_operator(ctx: Context, n: int, padding="exp", decay_factor: float = 0.2) -> np.ndarray:
    """Hc [n, n] float32 with Im = Hc A for the Hilbert stage along the samples (eofx_hilbert_operator_f32)"""
    out = np.empty((int(n), int(n)), np.float32)
    raise_for(ctx.lib.eofx_hilbert_operator_f32(ctx.handle, int(n), int(padding == "exp"), float(decay_factor), ptr(out)), ctx.handle)
    return out


def hilbert_sumsq(ctx: Context, A: ResidentMatrix, padding="exp", decay_factor: float = 0.2) -> float:
    """sum of squares of the imaginary part `hilbert(ctx, A)` would hold, without writing it (eofx_hilbert_sumsq_f64):
    total variance of the analytic signal = (A.sumsq() + this) / (n - 1)  (reference single/eof.py:93)"""
    out = C.c_double()
    raise_for(ctx.lib.eofx_hilbert_sumsq_f64(ctx.handle, A.handle, int(padding == "exp"), float(decay_factor), C.byref(out)),
              ctx.handle)
    return float(out.value)
