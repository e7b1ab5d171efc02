"""Decomposer -- same constructor, policy, errors, warnings and outputs as the reference's
xeofs/linalg/decomposer.py:15-226, with the solver seam (decomposer.py:141-146, the callable
given to xr.apply_ufunc) replaced by the HIP engine call `eofx_rsvd_f32`.

`fit` takes the resident (sample x feature) matrix and leaves numpy arrays
    U_ (sample x mode), s_ (mode), V_ (feature x mode)   [V_ = conj(VT).T, decomposer.py:226]
"""

from __future__ import annotations

import warnings

import numpy as np

from .. import engine

MAX_SKETCH = 256  # EOFX_MAX_SKETCH: sketches up to 64 wide are factorised on the device, wider ones on the host;
                  # beyond it the Decomposer switches to the exact small-side Gram route (xeofs_amd/pca.py)


def sanity_check_n_modes(n_modes):
    """xeofs/utils/sanity_checks.py:105-119."""
    if isinstance(n_modes, bool) or not isinstance(n_modes, (int, float, np.integer, np.floating)):
        raise TypeError("n_modes must be an integer or float")
    if isinstance(n_modes, (int, np.integer)):
        if n_modes < 1:
            raise ValueError("If n_modes is an integer, it must be greater than 0.")
    elif not (0.0 < n_modes <= 1.0):
        raise ValueError("If n_modes is a float, it must be strictly between 0 and 1.")


class Decomposer:
    def __init__(self, n_modes, init_rank_reduction=0.3, flip_signs=True, compute=True, solver="auto",
                 random_state=None, component_dim_name="mode", solver_kwargs={}, ctx=None, lazy_input=False):
        sanity_check_n_modes(n_modes)
        self.is_based_on_variance = not isinstance(n_modes, (int, np.integer))
        if self.is_based_on_variance and not (0 < init_rank_reduction <= 1.0):
            raise ValueError("init_rank_reduction must be in the half open interval (0, 1].")
        self.n_modes = n_modes
        self.n_modes_precompute = n_modes
        self.init_rank_reduction = init_rank_reduction
        self.flip_signs = flip_signs
        self.compute = compute
        self.solver = solver
        self.random_state = random_state
        self.component_dim_name = component_dim_name
        self.solver_kwargs = dict(solver_kwargs)
        self.ctx = ctx
        # the input was a dask-backed array: the reference then calls dask.array.linalg.svd_compressed
        # (decomposer.py:104, 163-171).  The engine is eager and resident, so the data are materialised, but the
        # branch keeps its parameters: sketch width max(20, k + 10), n_power_iter (default 4) power passes.
        self.lazy_input = bool(lazy_input)

    def policy(self, n, p, quiet=False):
        """The solver ladder of decomposer.py:86-131 for an (n x p) matrix -> (k, n_oversamples, n_iter, wide):
        the sketch the engine is asked for, and whether it is wider than the sketch kernels hold.  Raises / warns as
        the reference does (quiet: no warning, for a provisional look before the NaN compaction is known)."""
        rank = min(n, p)
        k_pre = self.n_modes
        if self.is_based_on_variance:
            k_pre = int(rank * self.init_rank_reduction)
            if k_pre < 1:
                if not quiet:
                    warnings.warn(
                        f"`init_rank_reduction={self.init_rank_reduction}` is too low resulting in zero components. One component will be computed instead."
                    )
                k_pre = 1
        if k_pre > rank:
            raise ValueError(
                f"n_modes must be less than or equal to the rank of the dataset (rank = {rank})."
            )
        is_small_data = max(n, p) < 500
        if self.solver == "auto":
            use_exact = bool(is_small_data and k_pre > int(0.8 * rank) and not self.lazy_input)
        elif self.solver == "full":
            use_exact = True
        elif self.solver == "randomized":
            use_exact = False
        else:
            raise ValueError(
                f"Unrecognized solver '{self.solver}'. "
                "Valid options are 'auto', 'full', and 'randomized'."
            )
        k = int(k_pre)
        kw = dict(self.solver_kwargs)
        for name in ("power_iteration_normalizer", "transpose", "flip_sign", "svd_lapack_driver"):
            kw.pop(name, None)  # sklearn knobs without effect on the result here
        if use_exact:
            # A full-width sketch spans the whole row/column space: the same kernels then return the
            # exact truncated SVD (no power iterations needed).
            n_over, n_iter = rank - k, 0
            wide = rank > MAX_SKETCH
        elif self.lazy_input:
            # svd_compressed: comp_level = min(max(20, k + n_oversamples), rank), `n_power_iter` passes
            # (re-orthonormalised here; dask's default iterator="power" does not, which float32 could not afford)
            kw.pop("compute", None)
            n_over = min(max(20, k + int(kw.pop("n_oversamples", 10))), rank) - k
            n_iter = int(kw.pop("n_power_iter", 4))
            wide = k + n_over > MAX_SKETCH
        else:
            n_over = int(kw.pop("n_oversamples", 10))
            n_iter = kw.pop("n_iter", "auto")
            wide = min(k + n_over, rank) > MAX_SKETCH
        return k, n_over, n_iter, wide

    def fit(self, X, dims=("sample", "feature"), total_variance=None, omega=None):
        """`omega`: optional pre-drawn sketch (engine.SketchFuture / ndarray) so the host-side sampling can
        overlap the preprocessing kernels; it must be the matrix `sketch_matrix` would draw."""
        ctx = self.ctx or engine.default_context()
        mat = X if isinstance(X, engine.ResidentMatrix) else engine.from_dense(ctx, np.asarray(X))
        n, p = mat.shape
        rank = min(n, p)
        k, n_over, n_iter, wide = self.policy(n, p)
        self.n_modes_precompute = k
        if wide:
            # more modes than the sketch kernels hold (e.g. float n_modes -> int(0.3 * rank) modes): at that
            # width the exact small-side Gram route is cheaper than the randomized passes (xeofs_amd/pca.py)
            from ..pca import ResidentPCA

            # solver="full" means the exact decomposition; otherwise this is the reference's randomized solver at a width
            # the sketch kernels do not hold (float n_modes -> int(0.3 rank) modes): its algorithm on the resident Gram matrix
            pca = ResidentPCA(ctx, k, flip_signs=bool(self.flip_signs), solver="exact" if self.solver == "full" else "auto",
                              random_state=self.random_state if isinstance(self.random_state, (int, np.integer)) else None
                              ).fit(mat, total_variance)
            U, s, V = pca.U.astype(np.float32), pca.s.astype(np.float32), pca.components()
            return self._finish(U, s, V, n, k, total_variance)
        # the per-mode sign rule (xarray_utils.py:273-301) runs on the GPU; truncating modes afterwards
        # does not change the sign of the kept ones
        om = None
        if omega is not None:
            om = omega.result() if hasattr(omega, "result") else omega
            if om.shape != (rank, k + n_over):
                om = None      # drawn for another policy branch: fall back to drawing it now
        U, s, V = engine.rsvd(ctx, mat, k, n_over, n_iter, random_state=self.random_state, flip=bool(self.flip_signs),
                              omega=om)
        return self._finish(U, s, V, n, k, total_variance)

    def fused_plan(self, n, P):
        """(k, n_oversamples, n_iter) when an (n x P) raw field can go through the engine's fused fit (eofx_fit_f32:
        statistics during the first pass of the randomized SVD), else None.  Only where the policy cannot change under
        NaN compaction: randomized branch, sketch of at most 64 columns on the sample side, not 'small data'."""
        if self.lazy_input or self.is_based_on_variance or max(n, P) < 500 or n >= P:
            return None
        try:
            k, n_over, n_iter, wide = self.policy(n, P, quiet=True)
        except ValueError:
            return None
        if wide or n_iter == 0 or not (0 < k + n_over < 64) or (k + n_over) % 32 == 0 or k + n_over >= n:
            return None
        return k, n_over, n_iter

    def adopt(self, mat, U, s, V, total_variance, plan):
        """Take the factors the fused fit produced with `plan` = fused_plan(raw shape).  If NaN compaction changed the
        shape so much that the solver ladder would have chosen differently, decompose again the reference's way."""
        n, p = mat.shape
        k, n_over, n_iter = plan
        if self.policy(n, p) != (k, n_over, n_iter, False):
            return self.fit(mat, total_variance=total_variance)
        self.n_modes_precompute = k
        return self._finish(U, s, V, n, k, total_variance)

    def _finish(self, U, s, V, n, k, total_variance):
        """variance-fraction truncation, decomposer.py:179-212"""
        if self.is_based_on_variance:
            if self.lazy_input:      # decomposer.py:191-193
                raise ValueError("Estimating the number of modes to keep based on variance is not supported with dask "
                                 "arrays. Please explicitly specifiy the number of modes to keep by using an integer for "
                                 "the number of modes.")
            if total_variance is None:
                raise ValueError("variance-based truncation needs the total variance of the input")
            cum = np.cumsum(s.astype(np.float64) ** 2 / (n - 1) / total_variance)
            n_req = k - int((cum >= self.n_modes).sum()) + 1
            if n_req > k:
                warnings.warn(
                    f"Dataset has {k} components, explaining {cum[-1]:.2%} of the variance. However, {self.n_modes:.2%} explained variance was requested. Please consider increasing `init_rank_reduction`."
                )
                n_req = k
            U, s, V = U[:, :n_req], s[:n_req], V[:, :n_req]
        self.U_, self.s_, self.V_ = U, s, V
        return self
