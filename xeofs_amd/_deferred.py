"""`compute=False` on a lazy (chunked) input: the reference leaves the fitted quantities as dask graphs and evaluates them
in `model.compute()` (xeofs/base_model.py:57-82, single/base_model_single_set.py:157-159,
linalg/decomposer.py:163-171, 272-298).  The engine is eager, so the WHOLE fit is the deferred unit here: `fit` returns at
once, `compute()` -- or the first read of a fitted quantity, which is what touching a lazy DataArray does in the
reference -- runs it."""

from __future__ import annotations


class Deferred:
    _pending = None
    _data = None

    @property
    def data(self):
        if self._pending is not None:
            self.compute()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def _defer(self, thunk):
        self._pending = thunk
        return self

    def compute(self, **kwargs):
        """Run a deferred fit (no-op otherwise); keyword arguments are accepted for signature parity with
        `dask.compute()` and ignored."""
        thunk, self._pending = self._pending, None
        if thunk is not None:
            thunk()
        return self

    @property
    def is_deferred(self) -> bool:
        return self._pending is not None
