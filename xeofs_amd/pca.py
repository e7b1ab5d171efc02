"""PCA pre-reduction of a resident matrix: xeofs/preprocessing/pca.py:94-171 on top of
xeofs/linalg/_numpy/_svd.py:89-106,108-241 (`n_modes` float = explained-variance target inside the first
int(rank * init_rank_reduction) modes).

The reference computes those int(0.3 * rank) modes with an *unseeded* randomized SVD (sketch width ~0.3 n,
10 passes over X).  At that width the MI355X-first route is the exact one, in two wide passes over X:

  1. G = X X^T (sample space, n x n; or X^T X when p < n) through the streaming `atb` kernel
     (`eofx_mat_gram_f32`: 128-column tiles, those below the diagonal skipped and mirrored; the matrix is read
     ~n_pad/256 times from HBM),
  2. symmetric eigendecomposition of G (float64, rocSOLVER through torch.linalg.eigh; 0.2 s at n = 5000) ->
     the whole spectrum, hence the reference's truncation rule evaluated exactly, and the basis U_m,
  3. B = X^T U_m (wide panel product, the same kernel) and a Rayleigh-Ritz step on B^T B (float64 Gram, m x m
     eigh): singular values / vectors accurate to float32 rounding independent of their size (the Gram matrix
     of step 1 only has to deliver the subspace).

V stays resident in HBM as a [p_pad, Lm] panel: `transform` (X_new V) and the back-projection of the cross
model's singular vectors (V Q) are panel products.
"""

from __future__ import annotations

import warnings

import numpy as np

from . import engine


def _round32(m):
    return (int(m) + 31) // 32 * 32


# Seeds of unseeded randomized fits: the reference's PCA draws from numpy's global, unseeded generator, so the two fields
# of a cross model start from INDEPENDENT Gaussian matrices.  A fixed seed shared by both would correlate the two sample-space
# starts -- and with them the noise ends of the two PC spaces (the total squared covariance of a default-argument MCA at
# config 3 came out 28 % high that way).  One process-wide counter: every unseeded fit gets the next stream; a process that
# makes the same calls in the same order gets the same results.
_unseeded_fits = [0]


def _next_seed():
    _unseeded_fits[0] += 1
    return 0x5EED0000 + _unseeded_fits[0]


class ResidentPCA:
    def __init__(self, ctx, n_modes=0.999, init_rank_reduction: float = 0.3, flip_signs: bool = True, solver: str = "auto",
                 random_state=None, n_iter: int = 4, n_oversamples: int = 10, basis_only: bool = False):
        """solver: "randomized" = the reference's own algorithm (scikit-learn's randomized_svd: n_oversamples = 10, 4 power
        iterations at this width, re-normalised every step) carried out on the RESIDENT sample-space Gram matrix -- every
        product n x n x l, no pass over the field inside the iteration, no order-n eigen-decomposition; "exact": the
        eigen-decomposition of the Gram matrix (rounds 1-3; rocSOLVER through torch, 0.2 s at n = 5000); "auto":
        randomized where the eigen-decomposition is the expensive step (n <= p, n >= 1024, sketch narrower than 0.6 n), else exact.  random_state: seed of the
        Gaussian start (the reference's is unseeded; None here takes the next stream of a process-wide counter: independent starts for
        the two fields of a cross model, the same results for the same sequence of calls)."""
        self.ctx = ctx
        self.solver = solver
        # basis_only (round 5): the caller needs the PC SUBSPACE, not its individual modes (a cross model
        # with alpha = 1: MCA's outputs are invariant to a rotation inside the PC space).  When the variance target is then
        # out of reach of all the computed directions -- decided from trace(B^T B) alone -- every mode is kept and the order-ell
        # eigen-decomposition of the randomized route (36 of 85 ms at config 3) is skipped: see `_basis_without_spectrum`.
        self.basis_only = bool(basis_only)
        self.random_state = random_state
        self.n_iter, self.n_oversamples = int(n_iter), int(n_oversamples)
        self.flip_signs = flip_signs
        self.n_modes = n_modes
        self.init_rank_reduction = init_rank_reduction
        self.is_based_on_variance = not isinstance(n_modes, (int, np.integer, str))
        if self.is_based_on_variance and not (0 < init_rank_reduction <= 1.0):
            raise ValueError("init_rank_reduction must be in the half open interval (0, 1].")

    # ------------------------------------------------------------------ policy (_svd.py:89-106)
    def _n_modes_precompute(self, rank: int) -> int:
        if self.is_based_on_variance:
            n_pre = int(rank * self.init_rank_reduction)
            if n_pre < 1:
                warnings.warn(f"`init_rank_reduction={self.init_rank_reduction}` is too low resulting in zero "
                              "components. One component will be computed instead.")
                n_pre = 1
            return n_pre
        if isinstance(self.n_modes, str):
            if self.n_modes != "all":
                raise ValueError("`n_modes` must be an integer, float or 'all'")
            return rank
        if self.n_modes > rank:
            raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {rank}).")
        return int(self.n_modes)

    # ------------------------------------------------------------------ fit
    def fit(self, mat, total_variance: float | None = None, comm=None, p_total: int | None = None):
        """`comm` (xeofs_amd.sharded.Comm): `mat` is this rank's slice of the feature axis (p_total features
        in all); the n x n Gram matrix and the m x m Rayleigh-Ritz Gram matrix are all-reduced, V stays
        sharded by rows, scores / singular values are replicated.  `total_variance` must then be the global
        one."""
        steps = self.fit_steps(mat, total_variance, comm, p_total)
        try:
            M = next(steps)
        except StopIteration:
            return self
        try:
            steps.send(engine._torch().linalg.eigh(M))
        except StopIteration:
            pass
        return self

    def fit_steps(self, mat, total_variance: float | None = None, comm=None, p_total: int | None = None):
        """`fit` as a generator: the randomized route YIELDS its order-ell Rayleigh-Ritz matrix and expects
        `torch.linalg.eigh` of it to be sent back (the one library call of the route: 36 of 85 ms at config 3, launch-bound) --
        a caller with two fields lets one field's eigen-problem run on a second stream under the other field's kernels
        (xeofs_amd/cross/cpcca.py).  The exact route never yields."""
        torch = engine._torch()
        ctx = self.ctx
        n, p = mat.n, mat.p
        sharded = comm is not None and getattr(comm, "active", False)
        p_all = int(p_total) if (sharded and p_total is not None) else p
        rank = min(n, p_all)
        n_pre = self._n_modes_precompute(rank)
        side = 0 if n <= p_all else 1                   # Gram matrix on the small side
        if sharded and side == 1:
            raise NotImplementedError("feature-sharded PCA needs n <= p (the Gram matrix lives on the sample side)")
        r = n if side == 0 else p
        Gf = mat.gram(side)
        if sharded:
            Gf = comm.sum_(Gf)                          # X X^T = sum over the feature shards
        ell = min(n_pre + self.n_oversamples, r)
        # "auto": the order-r eigen-decomposition is a few milliseconds up to r ~ 1000 and 0.2 s at 5000 (cubic): the exact route
        # below that size, the reference's randomized algorithm on the resident Gram matrix above it
        randomized = self.solver == "randomized" or (self.solver == "auto" and side == 0 and not sharded and
                                                     ell <= 0.6 * r and r >= 1024)
        if randomized and (side != 0 or ell >= r or sharded):
            randomized = False          # (the randomized route has no collectives: a feature-sharded fit takes the exact one)
        if self.solver not in ("auto", "randomized", "exact"):
            raise ValueError(f"Unrecognized solver '{self.solver}'. Valid options are 'auto', 'randomized', and 'exact'.")
        self.solver_used = "randomized" if randomized else "exact"
        self.spectrum_known = True
        dev = Gf.device
        if randomized:
            # ---- the reference's randomized solver on the resident Gram matrix (see _range_randomized) -----------------------
            if total_variance is None:
                total_variance = float(Gf.diagonal()[:r].double().sum()) / (n - 1)
            Q, GQ, Gm = self._range_randomized(Gf, r, ell)
            del Gf
            try:
                Le = Q.shape[1]
                # B = X^T Q (p x ell: the one wide product over the field) and the Rayleigh-Ritz step on B^T B = Q^T G Q,
                # accumulated in float64 from exact products of the field -- as in the exact route below
                B = engine.panel_tmul(ctx, mat, Q, prec=ctx.precision[1])
                Mfull = engine.panel_gram(ctx, B)
                M = Mfull[:ell, :ell]
                M = 0.5 * (M + M.T)
                fast = None
                if self.basis_only and self.is_based_on_variance and n_pre <= ell and \
                        float(M.diagonal().sum()) / (n - 1) / total_variance < self.n_modes:
                    fast = self._basis_without_spectrum(M, Mfull, ell, n_pre, n, total_variance)
                if fast is not None:
                    Wm, lam_h = fast                        # ell x n_pre: B Wm is an orthonormal basis of the kept subspace
                    m = n_pre
                    Lm = _round32(m)
                    Wt = torch.zeros((Le, Lm), dtype=torch.float64, device=dev)
                    Wt[:ell, :m] = Wm
                    self.Vp = engine.panel_matmul(ctx, B, Wt)
                    del B
                    Sc = engine.panel_matmul(ctx, GQ, Wt)[:n, :m].double()     # X V = (G Q) Wm
                    s = torch.sqrt((Sc * Sc).sum(dim=0)).clamp_min(1e-300)     # column norms: scores = U s stays X V
                    U = Sc / s
                    self.spectrum_known = False
                else:
                    th, W = yield M                         # order ell: the one library call of this route (see fit)
                    th = torch.flip(th, (0,)).clamp_min(0.0)
                    W = torch.flip(W, (1,))
                    lam_h = th.cpu().numpy()
                    m = self._truncate(lam_h, n_pre, n, total_variance)
                    Lm = _round32(m)
                    th, W = th[:m], W[:, :m]
                    s = torch.sqrt(th)
                    tiny = float(th[0]) * 1e-14 if m else 0.0
                    inv = torch.where(th > tiny, 1.0 / torch.sqrt(th.clamp_min(1e-300)), torch.zeros_like(th))
                    Wt = torch.zeros((Le, Lm), dtype=torch.float64, device=dev)
                    Wt[:ell, :m] = W * inv
                    self.Vp = engine.panel_matmul(ctx, B, Wt)   # V = B W theta^-1/2
                    del B
                    # The subspace of a randomized solver is not invariant, so U s and X V differ for the unconverged modes; the
                    # reference keeps V and defines the scores as X V (preprocessing/pca.py:120-131).  X V = X X^T Q W theta^-1/2
                    # = (G Q) W theta^-1/2, and G Q is at hand from the range finder's last product: no further pass.
                    U = engine.panel_matmul(ctx, GQ, Wt)[:n, :m].double() * inv
            finally:
                Gm.free()
        else:
            G = Gf[:r, :r].double()
            del Gf
            G = 0.5 * (G + G.T)
            if not bool(torch.isfinite(G).all()):
                raise np.linalg.LinAlgError("SVD failed. This may be due to isolated NaN values in the data.")
            lam, E = torch.linalg.eigh(G)                   # ascending
            lam = torch.flip(lam, (0,)).clamp_min(0.0)
            E = torch.flip(E, (1,))
            lam_h = lam.cpu().numpy()
            if total_variance is None:
                total_variance = float(lam_h.sum()) / (n - 1)
            m = self._truncate(lam_h, n_pre, n, total_variance)
            Lm = _round32(m)
            rows_small = mat.n_pad if side == 0 else mat.p_pad
            Es = torch.zeros((rows_small, Lm), dtype=torch.float32, device=dev)
            Es[:r, :m] = E[:, :m].float()
            # tall-side panel B = A^T E (A = X for side 0): columns ~ s_j v_j
            B = engine.panel_tmul(ctx, mat, Es, prec=ctx.precision[1]) if side == 0 else \
                engine.panel_mul(ctx, mat, Es, prec=ctx.precision[1])
            M = engine.panel_gram(ctx, B)
            if sharded:
                M = comm.sum_(M)
            M = M[:m, :m]
            M = 0.5 * (M + M.T)
            th, W = torch.linalg.eigh(M)
            th = torch.flip(th, (0,)).clamp_min(0.0)
            W = torch.flip(W, (1,))
            s = torch.sqrt(th)
            tiny = float(th[0]) * 1e-14 if m else 0.0
            inv = torch.where(th > tiny, 1.0 / torch.sqrt(th.clamp_min(1e-300)), torch.zeros_like(th))
            Wt = torch.zeros((Lm, Lm), dtype=torch.float64, device=dev)
            Wt[:m, :m] = W * inv                            # B W theta^-1/2 : orthonormal tall-side vectors
            Tall = engine.panel_matmul(ctx, B, Wt)
            del B
            Small = (E[:, :m] @ W)                          # r x m float64, orthonormal small-side vectors
            if side == 0:
                self.Vp, U = Tall, Small                    # V: p_pad x Lm panel (device), U: n x m
            else:                                           # features are the small side: V = E W, U = tall side
                Vp = torch.zeros((mat.p_pad, Lm), dtype=torch.float32, device=dev)
                Vp[:p, :m] = Small.float()
                self.Vp, U = Vp, Tall[:n, :m].double()
        # deterministic sign (utils/xarray_utils.py:273-301 on V), applied before the truncation in _svd.py:208-213
        # (a masked in-place matrix -- layout mode 3 -- keeps its all-NaN grid points as zero columns: V has p_phys rows,
        # zero at the masked features; zeros never change the sign rule, and every export below compacts the rows)
        mx, mn = engine.panel_colminmax(ctx, self.Vp, mat.p_phys)
        if sharded:
            mx, mn = comm.max_(mx), comm.min_(mn)
        mx, mn = mx.double()[:m], mn.double()[:m]
        sign = torch.where(mx.abs() >= mn.abs(), torch.ones_like(mx), -torch.ones_like(mx))
        if not self.flip_signs:
            sign = torch.ones_like(sign)
        self.sign = sign.cpu().numpy()
        self.m, self.Lm, self.n, self.p, self.p_pad = m, Lm, n, p, mat.p_pad
        self.p_phys, self._masked_index = mat.p_phys, (mat.valid_index if mat.masked else None)
        # `_col_scale`: what the scores are scaled by (X V = U * _col_scale): the singular values, or -- on the basis-only path,
        # where no individual mode exists -- the norms of the columns of X V; `s` is the spectrum and says so when there is none
        self._col_scale = s.cpu().numpy()
        Ud = U * sign                                   # n x m float64 (device)
        self._U_dev, self._U_host = Ud, None            # downloaded on first use (60 MB at config 3)
        self._scores_dev = (Ud * s).float().contiguous()   # X V on the device: the analysis matrix of the cross models
        self.singular_values_all = np.sqrt(lam_h) if self.spectrum_known else None
        self.total_variance = total_variance

    @property
    def s(self):
        """singular values of the kept modes.  The basis-only path (a cross model with alpha = 1 that keeps every computed
        direction) forms no individual mode: there is no spectrum to return, and a placeholder would be silently wrong."""
        if not getattr(self, "spectrum_known", True):
            raise RuntimeError("this PCA was fitted on the basis-only path (basis_only=True, variance target out of reach): the "
                               "kept SUBSPACE is exact but no individual mode / singular value was computed; fit with "
                               "basis_only=False for the spectrum")
        return self._col_scale

    @s.setter
    def s(self, value):      # (the complex subclass assigns its spectrum directly)
        self._col_scale = value
        self.spectrum_known = True

    def _basis_without_spectrum(self, M, Mfull, ell, n_pre, n, total_variance):
        """The randomized route when only the kept SUBSPACE matters and every computed mode is kept (`basis_only`, variance
        target out of reach): the reference keeps the leading n_pre of the ell = n_pre + n_oversamples Ritz directions of
        M = B^T B (linalg/_numpy/_svd.py:215-241 after sklearn's truncation) -- i.e. it drops the span of the n_oversamples
        SMALLEST ones.  Those few are found without the order-ell eigen-decomposition: the blocked device Cholesky factor of M
        (engine.panel_rinv, 4.7 ms at 1510) gives M^-1 = R^-1 R^-T, a 32-column inverse subspace iteration converges on the bottom
        of the spectrum, and the kept subspace is the orthogonal complement of B w_bottom inside span(B):
        B R^-1 N with N an orthonormal basis of the complement of R w_bottom (a 10-column Householder QR).  No individual
        mode is formed; `s` / `singular_values_all` are then not the spectrum (spectrum_known = False).
        -> (ell x n_pre float64 device matrix Wm with (B Wm)^T (B Wm) = I, host array of per-mode variances as far as known)
        or None when the Cholesky factor is not usable (dependent columns): the caller takes the eigen-decomposition."""
        torch = engine._torch()
        ctx = self.ctx
        dev = M.device
        nb = ell - n_pre
        Ri = engine.panel_rinv(ctx, Mfull.contiguous(), ell)[:ell, :ell]      # upper triangular, M^-1 = Ri Ri^T
        d = Ri.diagonal()
        if not bool(torch.isfinite(Ri).all()) or bool((d <= 0).any()):
            return None
        if nb == 0:
            Wm, lam_bottom = Ri, np.zeros(0)
        else:
            blk = min(ell, nb + 22)
            gen = torch.Generator(device=dev)
            gen.manual_seed(0x5EED)
            Y = torch.randn((ell, blk), generator=gen, device=dev, dtype=torch.float64)
            RiT = Ri.T.contiguous()

            def cholqr(Z):      # (two small library calls; torch.linalg.qr of a 1510 x 32 panel costs milliseconds)
                Lc = torch.linalg.cholesky(Z.T @ Z)
                return torch.linalg.solve_triangular(Lc, Z.T, upper=False).T

            # The bottom of the sketch's spectrum decays smoothly, so the iteration is CHECKED, not trusted: after
            # 4 x 5 inverse steps and a Rayleigh-Ritz step, the residual |M w - theta w| / theta of each of the nb dropped Ritz
            # directions must be <= 5e-2 (a dropped direction then mixes at most that much of a kept one: its variance is within
            # 0.3 % of an exact bottom direction's, far inside what the reference's unseeded sketch varies by from run to run);
            # two more rounds are tried, then the caller takes the order-ell eigen-decomposition (return None).
            self.fast_path_residual = None
            for outer in range(6):
                for _ in range(5):                                            # five inverse steps between orthonormalisations
                    Y = Ri @ (RiT @ Y)
                    Y = Y / Y.norm(dim=0, keepdim=True)
                Y = cholqr(Y)
                if outer < 3:
                    continue
                Yq = cholqr(Y)
                MY = M @ Yq
                T = (Yq.T @ MY).cpu().numpy()                                 # 32 x 32: the host solves it
                tv_h, Ws_h = np.linalg.eigh(0.5 * (T + T.T))                  # ascending: the bottom of the spectrum first
                Wsb = torch.as_tensor(Ws_h[:, :nb], device=dev)
                Wb = Yq @ Wsb
                lam_bottom = np.maximum(tv_h[:nb], 0.0)
                res = (MY @ Wsb - Wb * torch.as_tensor(tv_h[:nb], device=dev)).norm(dim=0).cpu().numpy()
                self.fast_path_residual = float(np.max(res / np.maximum(np.abs(tv_h[:nb]), 1e-300)))
                if self.fast_path_residual <= 5e-2:
                    break
            else:
                return None
            D = torch.linalg.solve_triangular(Ri, Wb, upper=True)             # R w_bottom
            Qf, _ = torch.linalg.qr(D, mode="complete")
            Wm = Ri @ Qf[:, nb:]
        kept = float(M.diagonal().sum()) - float(lam_bottom.sum())
        warnings.warn(f"Dataset has {n_pre} components, explaining {kept / (n - 1) / total_variance:.2%} of the variance. However, "
                      f"{self.n_modes:.2%} explained variance was requested. Please consider increasing "
                      "`init_rank_reduction`.")
        lam_h = np.full(n_pre, np.nan)                     # (no spectrum on this path: `s` / `singular_values_all` say so)
        self._kept_variance = kept / (n - 1)
        return Wm, lam_h

    def _truncate(self, lam_h, n_pre, n, total_variance):
        """number of modes kept (linalg/_numpy/_svd.py:215-241): the first n_pre eigenvalue estimates against the target"""
        m = n_pre
        if self.is_based_on_variance:
            cum = np.cumsum(lam_h[:n_pre] / (n - 1) / total_variance)
            m = n_pre - int((cum >= self.n_modes).sum()) + 1
            if m > n_pre:
                warnings.warn(f"Dataset has {n_pre} components, explaining {cum[-1]:.2%} of the variance. However, "
                              f"{self.n_modes:.2%} explained variance was requested. Please consider increasing "
                              "`init_rank_reduction`.")
                m = n_pre
        return m

    def _range_randomized(self, Gf, r, ell):
        """Orthonormal basis Q (n x ell) of the reference's randomized range finder (linalg/_numpy/_svd.py:170-186 -> sklearn
        randomized_svd: Gaussian start, n_iter power iterations each followed by a re-normalisation) run IN SAMPLE SPACE on
        the resident Gram matrix G = X X^T: range((X X^T)^q X Omega) = range(G^q Y0).  The start Y0 = G Omega' (Omega' Gaussian
        n x ell) stands in for X Omega -- both are a Gaussian combination of the field's columns pushed once through X.  Per
        step: one product G Q (the split-fp16 streaming kernel over the 100 MB matrix), the float64 Gram matrix of the n x ell
        panel, its blocked Cholesky factor + triangular inverse on the device (eofx_abi.hip::launch_rinv_blocked) and the
        product with it -- engine kernels only, nothing of order n is factorised.
        -> (Q [n_pad, round32(ell)] float32, G Q (same shape), the Gram matrix as a ResidentMatrix: the caller frees it)"""
        torch = engine._torch()
        ctx = self.ctx
        dev = Gf.device
        if not bool(torch.isfinite(Gf[:r, :r]).all()):
            raise np.linalg.LinAlgError("SVD failed. This may be due to isolated NaN values in the data.")
        Gm = engine.from_dense(ctx, Gf[:r, :r].contiguous())          # the Gram matrix as a resident (r x r) matrix
        try:
            L = _round32(ell)
            gen = torch.Generator(device=dev)
            gen.manual_seed(_next_seed() if self.random_state is None else int(self.random_state))
            Q = torch.zeros((Gm.n_pad, L), dtype=torch.float32, device=dev)
            Q[:r, :ell] = torch.randn((r, ell), generator=gen, device=dev, dtype=torch.float32)
            prec = ctx.precision[1]
            for _ in range(self.n_iter + 1):
                Y = engine.panel_tmul(ctx, Gm, Q, prec=prec)           # G Q (G symmetric); [p_pad == n_pad, L]
                S = engine.panel_gram(ctx, Y)
                Q = engine.panel_cholqr(ctx, Y, ell, S)
            Y = engine.panel_tmul(ctx, Gm, Q, prec=prec)
        except BaseException:
            Gm.free()
            raise
        return Q, Y, Gm

    @property
    def U(self):
        """left singular vectors (n x m float64, host): X V / s"""
        if getattr(self, "_U_host", None) is None:
            self._U_host = self._U_dev.cpu().numpy()
        return self._U_host

    @U.setter
    def U(self, value):      # (the complex subclass assigns its host array directly)
        self._U_host = value

    def scores_device(self):
        """X V (n x m float32) as a device tensor: stays in HBM for the cross models' analysis"""
        return self._scores_dev

    # ------------------------------------------------------------------ PC-space views
    def scores(self):
        """X V = U s  (n x m), what `PCA.transform` returns for the training data (pca.py:125-134)."""
        return self.U * self._col_scale

    def _compact(self, V):
        """rows of the valid features of an exported factor with p_phys rows"""
        return V if self._masked_index is None else np.ascontiguousarray(V[self._masked_index])

    def _vp_for(self, mat_new):
        """V as a panel over the columns of `mat_new`: the fitted matrix may have been masked in place (V has p_phys rows)
        while the new one is compacted, or the other way round"""
        if self._masked_index is None and not mat_new.masked:
            return self.Vp
        if self._masked_index is not None and mat_new.masked and mat_new.p_phys == self.p_phys:
            return self.Vp
        torch = engine._torch()
        idx = torch.as_tensor(self._masked_index if self._masked_index is not None else mat_new.valid_index, device=self.Vp.device)
        out = torch.zeros((mat_new.p_pad, self.Lm), dtype=torch.float32, device=self.Vp.device)
        if self._masked_index is not None:          # fitted masked, new data compacted: gather the valid rows
            out[:self.p] = self.Vp.index_select(0, idx)
        else:                                       # fitted compacted, new data masked in place: scatter
            out.index_copy_(0, idx, self.Vp[:self.p])
        return out

    def transform(self, mat_new):
        """X_new V (n' x m) for a resident matrix preprocessed with the fitted state."""
        out = engine.panel_mul(self.ctx, mat_new, self._vp_for(mat_new), prec=self.ctx.precision[1])
        return out[:mat_new.n, :self.m].double().cpu().numpy() * self.sign

    def back_project(self, Q):
        """V Q (p x k): components from PC space back to feature space (pca.py:158-168)."""
        torch = engine._torch()
        Q = np.asarray(Q, dtype=np.float64)
        k = Q.shape[1]
        M = np.zeros((self.Lm, _round32(k)))
        M[:self.m, :k] = Q * self.sign[:, None]
        out = engine.panel_matmul(self.ctx, self.Vp, torch.as_tensor(M, device=self.Vp.device))
        return self._compact(engine.panel_export(self.ctx, out, self.p_phys, k))

    def components(self):
        """V (p x m) float32 on the host."""
        return self._compact(engine.panel_export(self.ctx, self.Vp, self.p_phys, self.m, self.sign))

    def row_norms(self, M):
        """Euclidean norms of the rows of V diag(sign) M (p,), M: m x m -- the per-feature standard deviations
        of the PCA-truncated field when M = (A^T A)^(1/2)."""
        torch = engine._torch()
        Mp = np.zeros((self.Lm, self.Lm))
        Mp[:self.m, :self.m] = np.asarray(M, dtype=np.float64) * self.sign[:, None]
        out = engine.panel_matmul(self.ctx, self.Vp, torch.as_tensor(Mp, device=self.Vp.device))
        return self._compact(engine.panel_rownorm(self.ctx, out, self.p_phys))
