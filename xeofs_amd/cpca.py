"""PCA pre-reduction of a COMPLEX resident field Z = A + iB (the complex cross models' default path:
xeofs/cross/cpcca.py:1023-1173 on top of xeofs/preprocessing/pca.py:94-171 and xeofs/linalg/_numpy/_svd.py:89-241).

Same exact route as the real `xeofs_amd.pca.ResidentPCA`, with Hermitian algebra:

  1. G = Z Z^H = (A A^T + B B^T) + i (B A^T - A B^T)   (n x n, four wide launches of the streaming kernel:
     `eofx_mat_gram_f32` twice, `eofx_mat_cross_gram_f32` once; the antisymmetric part comes from its transpose),
  2. Hermitian eigendecomposition of G in float64 -> the whole spectrum, hence the reference's truncation rule
     (`n_modes` float = explained-variance target inside the first int(rank * init_rank_reduction) modes) evaluated
     exactly, and the basis E_m,
  3. P = Z^H E_m (two wide products on the parts + recombination) and a Rayleigh-Ritz step on P^H P (float64 Gram of
     the real [Re | Im] panel, m x m Hermitian eigh): singular values / vectors accurate to float32 rounding.

A complex panel of m columns is a real [rows_pad, 2 Lh] panel, Lh = round_up(m, 32): columns [0, m) real parts,
[Lh, Lh + m) imaginary parts.  V stays resident; scores U s live on the host (n x m complex128).
With more samples than features the Gram matrix is Z^H Z (features x features) and the roles of the two sides swap.
"""

from __future__ import annotations

import warnings

import numpy as np

from . import engine
from .pca import ResidentPCA, _round32


def embed_right(M, Lh_in, Lh_out):
    """real (2 Lh_in x 2 Lh_out) matrix E with [Pr | Pi] @ E = [Re(P M) | Im(P M)] for a complex M (l x m)"""
    l, m = M.shape
    E = np.zeros((2 * Lh_in, 2 * Lh_out))
    E[:l, :m] = M.real
    E[Lh_in:Lh_in + l, :m] = -M.imag
    E[:l, Lh_out:Lh_out + m] = M.imag
    E[Lh_in:Lh_in + l, Lh_out:Lh_out + m] = M.real
    return E


def hermitian_from_real_gram(G, Lh, m):
    """complex m x m Gram P^H P from the real (2 Lh x 2 Lh) Gram of [Pr | Pi]"""
    rr, ri = G[:m, :m], G[:m, Lh:Lh + m]
    ir, ii = G[Lh:Lh + m, :m], G[Lh:Lh + m, Lh:Lh + m]
    H = (rr + ii) + 1j * (ri - ir)
    return 0.5 * (H + H.conj().T)


def _eigh_desc(H):
    """Hermitian eigendecomposition, descending: on the device where torch offers it, else on the host"""
    torch = engine._torch()
    if torch.is_tensor(H):
        try:
            w, V = torch.linalg.eigh(H)
            return torch.flip(w, (0,)).cpu().numpy(), torch.flip(V, (1,)).cpu().numpy()
        except Exception:
            H = H.cpu().numpy()
    w, V = np.linalg.eigh(H)
    return w[::-1].copy(), V[:, ::-1].copy()


class ComplexResidentPCA(ResidentPCA):
    def fit(self, A, B, total_variance: float | None = None):
        torch = engine._torch()
        ctx = self.ctx
        n, p = A.n, A.p
        if (B.n, B.p) != (n, p):
            raise ValueError("real and imaginary parts must have the same shape")
        side = 0 if n <= p else 1                              # Hermitian Gram matrix on the small side
        r = n if side == 0 else p
        rank = min(n, p)
        n_pre = self._n_modes_precompute(rank)
        if side == 0:      # Z Z^H = (A A^T + B B^T) + i (B A^T - A B^T)
            Gr = (A.gram(0)[:r, :r].double() + B.gram(0)[:r, :r].double())
            X = B.cross_gram(A, 0)[:r, :r].double()            # B A^T
        else:              # Z^H Z = (A^T A + B^T B) + i (A^T B - B^T A)
            Gr = (A.gram(1)[:r, :r].double() + B.gram(1)[:r, :r].double())
            X = A.cross_gram(B, 1)[:r, :r].double()            # A^T B
        G = torch.complex(0.5 * (Gr + Gr.T), X - X.T)
        del Gr, X
        if not bool(torch.isfinite(G.real).all() and torch.isfinite(G.imag).all()):
            raise np.linalg.LinAlgError("SVD failed. This may be due to isolated NaN values in the data.")
        lam, E = _eigh_desc(G)
        del G
        lam = np.clip(lam, 0.0, None)
        if total_variance is None:
            total_variance = float(lam.sum()) / (n - 1)
        m = n_pre
        if self.is_based_on_variance:                          # _svd.py:215-241
            cum = np.cumsum(lam[:n_pre] / (n - 1) / total_variance)
            m = n_pre - int((cum >= self.n_modes).sum()) + 1
            if m > n_pre:
                warnings.warn(f"Dataset has {n_pre} components, explaining {cum[-1]:.2%} of the variance. However, "
                              f"{self.n_modes:.2%} explained variance was requested. Please consider increasing "
                              "`init_rank_reduction`.")
                m = n_pre
        Lh = _round32(m)
        dev = f"cuda:{ctx.device}"
        Es = torch.zeros((A.n_pad if side == 0 else A.p_pad, 2 * Lh), dtype=torch.float32, device=dev)
        Em = np.ascontiguousarray(E[:, :m])      # (BLAS needs plain strides: a sliced / reversed view multiplies 100x slower)
        Es[:r, :m] = torch.as_tensor(Em.real, dtype=torch.float32)
        Es[:r, Lh:Lh + m] = torch.as_tensor(Em.imag, dtype=torch.float32)
        pr = ctx.precision[1]
        if side == 0:      # tall side = features: P = Z^H E_m
            P = engine.cpanel_combine(ctx, engine.panel_tmul(ctx, A, Es, prec=pr), engine.panel_tmul(ctx, B, Es, prec=pr), True)
        else:              # tall side = samples:  P = Z E_m
            P = engine.cpanel_combine(ctx, engine.panel_mul(ctx, A, Es, prec=pr), engine.panel_mul(ctx, B, Es, prec=pr), False)
        H = hermitian_from_real_gram(engine.panel_gram(ctx, P).cpu().numpy(), Lh, m)
        th, W = _eigh_desc(torch.as_tensor(H, device=dev) if m >= 512 else H)
        th, W = np.clip(th, 0.0, None), np.ascontiguousarray(W)
        s = np.sqrt(th)
        tiny = th[0] * 1e-14 if m else 0.0
        inv = np.where(th > tiny, 1.0 / np.sqrt(np.maximum(th, 1e-300)), 0.0)
        Tall = engine.panel_matmul(ctx, P, torch.as_tensor(embed_right(W * inv, Lh, Lh), device=P.device))   # rows_pad x 2 Lh
        del P
        Small = Em @ W                                          # r x m complex128, orthonormal
        if side == 0:
            self.Vp, self.U = Tall, Small                       # V: p_pad x 2 Lh panel, U: n x m
        else:                                                   # features are the small side: V = E_m W, U = the tall side
            Vp = torch.zeros((A.p_pad, 2 * Lh), dtype=torch.float32, device=dev)
            Vp[:p, :m] = torch.as_tensor(Small.real, dtype=torch.float32)
            Vp[:p, Lh:Lh + m] = torch.as_tensor(Small.imag, dtype=torch.float32)
            t = Tall[:n].double().cpu().numpy()
            self.Vp, self.U = Vp, t[:, :m] + 1j * t[:, Lh:Lh + m]
        self.s = s
        self.m, self.Lh, self.n, self.p, self.p_pad = m, Lh, n, p, A.p_pad
        self.singular_values_all = np.sqrt(lam)
        self.total_variance = total_variance
        return self

    # ------------------------------------------------------------------ PC-space views
    def scores(self):
        """Z V = U s (n x m complex128)"""
        return self.U * self.s

    def transform(self, A_new, B_new):
        """Z_new V (n' x m complex) for the parts of new data preprocessed with the fitted state"""
        pr = self.ctx.precision[1]
        out = engine.cpanel_combine(self.ctx, engine.panel_mul(self.ctx, A_new, self.Vp, prec=pr),
                                    engine.panel_mul(self.ctx, B_new, self.Vp, prec=pr), False)
        o = out[:A_new.n].double().cpu().numpy()
        return o[:, :self.m] + 1j * o[:, self.Lh:self.Lh + self.m]

    def back_project(self, Q):
        """V Q (p x k complex64): components from PC space back to feature space (pca.py:158-168)"""
        torch = engine._torch()
        Q = np.asarray(Q, dtype=np.complex128)
        k = Q.shape[1]
        Lk = _round32(k)
        out = engine.panel_matmul(self.ctx, self.Vp, torch.as_tensor(embed_right(Q, self.Lh, Lk), device=self.Vp.device))
        o = out[:self.p].cpu().numpy()
        res = np.empty((self.p, k), np.complex64)
        res.real, res.imag = o[:, :k], o[:, Lk:Lk + k]
        return res

    def components(self):
        o = self.Vp[:self.p].cpu().numpy()
        res = np.empty((self.p, self.m), np.complex64)
        res.real, res.imag = o[:, :self.m], o[:, self.Lh:self.Lh + self.m]
        return res
