"""Randomized SVD of a complex (sample x feature) matrix Z = A + iB held as two resident real
matrices -- the complex branch of the decomposer (xeofs/linalg/decomposer.py:149-160, which the
reference hands to scipy's svds(lobpcg)).

Every pass over the data is two launches of the same real kernel that serves the real path
(one over A, one over B) on a 64-wide real panel holding [Re | Im] of a complex panel of up to
32 columns, followed by a tiny complex recombination kernel.  Orthonormalisation is a complex
Cholesky-free QR from the Hermitian Gram matrix (one 64x64 float64 Gram of the real panel gives
all four blocks); the l x l Hermitian eigen-problems are solved on the host in float64.

Singular values / subspaces match the reference's LOBPCG result to tolerance; singular vectors
are defined up to a unit complex phase per mode (LOBPCG's phase is arbitrary too), after which
the reference's +-1 sign rule (xarray_utils.py:273-301, numpy's lexicographic complex max/min)
is applied.
"""

from __future__ import annotations

import numpy as np

from . import engine

HALF = 32          # complex panels are [Re(32 cols) | Im(32 cols)]
LP = 2 * HALF


def _embed_right(M):
    """real 64x64 matrix E with [Pr|Pi] @ E = [Re(P M) | Im(P M)] for complex M (l x m, padded)."""
    E = np.zeros((LP, LP))
    l, m = M.shape
    E[:l, :m] = M.real
    E[HALF:HALF + l, :m] = -M.imag
    E[:l, HALF:HALF + m] = M.imag
    E[HALF:HALF + l, HALF:HALF + m] = M.real
    return E


def _hermitian_gram(G, l):
    """complex l x l Gram P^H P from the real 64x64 Gram of [Pr|Pi]."""
    rr, ri = G[:l, :l], G[:l, HALF:HALF + l]
    ir, ii = G[HALF:HALF + l, :l], G[HALF:HALF + l, HALF:HALF + l]
    H = (rr + ii) + 1j * (ri - ir)
    return 0.5 * (H + H.conj().T)


class ComplexOps:
    """Complex panel steps on (A, B) through the C ABI."""

    def __init__(self, ctx, A, B):
        if A.shape != B.shape:
            raise ValueError("real and imaginary parts must have the same shape")
        self.ctx, self.A, self.B = ctx, A, B
        self.n, self.p, self.n_pad, self.p_pad = A.n, A.p, A.n_pad, A.p_pad

    def zh_mul(self, Wn, final=False):      # feature-side panel = Z^H W
        pr = self.ctx.precision[1 if final else 0]
        P1 = engine.panel_tmul(self.ctx, self.A, Wn, prec=pr)
        P2 = engine.panel_tmul(self.ctx, self.B, Wn, prec=pr)
        return engine.cpanel_combine(self.ctx, P1, P2, True, out=P1)

    def z_mul(self, Yp, final=False):       # sample-side panel = Z Y
        pr = self.ctx.precision[1 if final else 0]
        P1 = engine.panel_mul(self.ctx, self.A, Yp, prec=pr)
        P2 = engine.panel_mul(self.ctx, self.B, Yp, prec=pr)
        return engine.cpanel_combine(self.ctx, P1, P2, False, out=P1)

    def gram(self, P, l):
        return _hermitian_gram(engine.panel_gram(self.ctx, P).cpu().numpy(), l)

    def right_mul(self, P, M):
        torch = engine._torch()
        return engine.panel_matmul(self.ctx, P, torch.as_tensor(_embed_right(M), device=P.device))

    def orth(self, P, l):
        """Q = P (V diag(w^-1/2)) with P^H P = V diag(w) V^H: orthonormal columns spanning range(P);
        numerically dependent directions are dropped (zero columns)."""
        w, V = np.linalg.eigh(self.gram(P, l))
        good = w > 1e-13 * max(w.max(), 0.0)
        T = np.zeros((l, l), dtype=complex)
        T[:, good] = V[:, good] / np.sqrt(w[good])
        return self.right_mul(P, T)


def complex_rsvd(ctx, A, B, k: int, n_oversamples: int = 10, n_iter="auto", random_state=None, flip=True):
    """-> (U[n,k] complex64, s[k] float32, V[p,k] complex64) with Z ~ U diag(s) V^H and V = conj(VT).T."""
    torch = engine._torch()
    ops = ComplexOps(ctx, A, B)
    n, p = ops.n, ops.p
    r = min(n, p)
    if k > r:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {r}).")
    l = min(k + n_oversamples, r)
    if l > HALF:
        raise NotImplementedError(f"complex sketch width {l} > {HALF} is not supported by this build "
                                  f"(n_modes + n_oversamples <= {HALF})")
    if n_iter == "auto" or n_iter is None:
        n_iter = 7 if k < 0.1 * r else 4
    omega = engine.sketch_matrix(r, k + n_oversamples, random_state)[:, :l]   # real Gaussian start
    transposed = n < p
    small_rows, small_pad = (n, ops.n_pad) if transposed else (p, ops.p_pad)
    host = np.zeros((small_rows, LP), np.float32)
    host[:, :l] = omega
    Z = engine.panel_import(ctx, host, small_pad, LP)
    fwd = (lambda P, f=False: ops.zh_mul(P, f)) if transposed else (lambda P, f=False: ops.z_mul(P, f))
    bwd = (lambda P, f=False: ops.z_mul(P, f)) if transposed else (lambda P, f=False: ops.zh_mul(P, f))
    for _ in range(int(n_iter)):
        Z = ops.orth(bwd(fwd(Z)), l)
    Q = ops.orth(ops.orth(fwd(Z, True), l), l)
    Bt = bwd(Q, True)                                   # = B^H with B = Q^H A_op
    w, Uh = np.linalg.eigh(ops.gram(Bt, l))             # B B^H = Uh diag(w) Uh^H
    order = np.argsort(w)[::-1][:k]
    s = np.sqrt(np.maximum(w[order], 0.0))
    Uh = Uh[:, order]
    with np.errstate(divide="ignore"):
        inv = np.where(s > 0, 1.0 / s, 0.0)
    Tall = ops.right_mul(Q, Uh)                         # A_op = Tall diag(s) Small^H
    Small = ops.right_mul(Bt, Uh * inv)
    if transposed:      # A_op = Z^H  ->  Z = Small diag(s) Tall^H : U = Small, V = Tall
        Up, Vp = Small, Tall
    else:               # A_op = Z    ->  Z = Tall diag(s) Small^H
        Up, Vp = Tall, Small
    sign = np.ones(k)
    if flip:
        # VT = conj(V)^T; numpy max/min of complex arrays are lexicographic (real part, then imag)
        amax, amin = engine.panel_colargminmax(ctx, Vp, p)
        amax, amin = amax[:k].cpu(), amin[:k].cpu()
        cols = torch.arange(k)
        vmax = (Vp[amax, cols].cpu().numpy() - 1j * Vp[amax, cols + HALF].cpu().numpy())
        # conj flips the imaginary part; lexicographic ties on the real part are measure-zero
        vmin = (Vp[amin, cols].cpu().numpy() - 1j * Vp[amin, cols + HALF].cpu().numpy())
        sign = np.where(np.abs(vmax) >= np.abs(vmin), 1.0, -1.0)
    sg = np.concatenate([sign, np.ones(HALF - k), sign, np.ones(HALF - k)])

    def export(P, rows):
        full = engine.panel_export(ctx, P, rows, LP, sg)
        return (full[:, :k] + 1j * full[:, HALF:HALF + k]).astype(np.complex64)

    return export(Up, n), s.astype(np.float32), export(Vp, p)
