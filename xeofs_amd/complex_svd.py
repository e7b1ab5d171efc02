"""Randomized SVD of a complex (sample x feature) matrix Z = A + iB held as two resident real
matrices -- the complex branch of the decomposer (xeofs/linalg/decomposer.py:149-160, which the
reference hands to scipy's svds(lobpcg)).

Every pass over the data is one launch of the streaming kernel in its two-matrix form
(`eofx_cmat_mul_f32`; other precisions than the default: one launch per part + a recombination
kernel) on a 64-wide real panel holding [Re | Im] of a complex panel of up to 32 columns.  Orthonormalisation is a complex
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

HALF = 32          # complex panels are [Re(32 cols) | Im(32 cols)]; sketches of 33 .. 64 columns: [Re(64) | Im(64)]
LP = 2 * HALF
MAX_HALF = 64


def _embed_right(M, half=HALF):
    """real (2 half) x (2 half) matrix E with [Pr|Pi] @ E = [Re(P M) | Im(P M)] for complex M (l x m, padded)."""
    E = np.zeros((2 * half, 2 * half))
    l, m = M.shape
    E[:l, :m] = M.real
    E[half:half + l, :m] = -M.imag
    E[:l, half:half + m] = M.imag
    E[half:half + l, half:half + m] = M.real
    return E


def _hermitian_gram(G, l, half=HALF):
    """complex l x l Gram P^H P from the real Gram of [Pr|Pi]."""
    rr, ri = G[:l, :l], G[:l, half:half + l]
    ir, ii = G[half:half + l, :l], G[half:half + l, half:half + l]
    H = (rr + ii) + 1j * (ri - ir)
    return 0.5 * (H + H.conj().T)


class ComplexOps:
    """Complex panel steps on this rank's (A, B) through the C ABI.  With a `comm`
    (xeofs_amd.sharded.Comm) the feature axis is sharded over ranks exactly as in the real path:
    sample-side panels are all-reduced after every Z Y, Gram matrices of feature-side panels are
    all-reduced before the host factorisation."""

    def __init__(self, ctx, A, B):
        if A.shape != B.shape:
            raise ValueError("real and imaginary parts must have the same shape")
        self.ctx, self.A, self.B = ctx, A, B
        self.n, self.p, self.n_pad, self.p_pad = A.n, A.p, A.n_pad, A.p_pad
        self.half = HALF

    def set_half(self, half):               # 32 or 64 complex columns per panel
        self.half = int(half)

    def import_panel(self, host, side):
        return engine.panel_import(self.ctx, host, self.n_pad if side == "n" else self.p_pad, 2 * self.half)

    def zh_mul(self, Wn, final=False):      # feature-side panel = Z^H W   (local)
        pr = self.ctx.precision[1 if final else 0]
        if pr == "f16x3":                   # one launch over both parts
            return engine.cmat_mul(self.ctx, self.A, self.B, Wn, True, final)
        P1 = engine.panel_tmul(self.ctx, self.A, Wn, prec=pr)
        P2 = engine.panel_tmul(self.ctx, self.B, Wn, prec=pr)
        return engine.cpanel_combine(self.ctx, P1, P2, True, out=P1)

    def z_mul(self, Yp, final=False):       # sample-side panel = Z Y     (partial sum over features)
        pr = self.ctx.precision[1 if final else 0]
        if pr == "f16x3":
            return engine.cmat_mul(self.ctx, self.A, self.B, Yp, False, final)
        P1 = engine.panel_mul(self.ctx, self.A, Yp, prec=pr)
        P2 = engine.panel_mul(self.ctx, self.B, Yp, prec=pr)
        return engine.cpanel_combine(self.ctx, P1, P2, False, out=P1)

    def gram_real(self, P):                 # 64 x 64 float64 Gram of the real [Re|Im] panel (device)
        return engine.panel_gram(self.ctx, P)

    def right_mul(self, P, M):
        torch = engine._torch()
        return engine.panel_matmul(self.ctx, P, torch.as_tensor(_embed_right(M, self.half), device=P.device))

    def matmul_real(self, P, E):             # P [rows x L] (any multiple of 64 columns) times the real matrix E [L x Lo]
        torch = engine._torch()
        return engine.panel_matmul(self.ctx, P, torch.as_tensor(np.ascontiguousarray(E, dtype=np.float64), device=P.device))

    def argminmax(self, P, rows):
        return engine.panel_colargminmax(self.ctx, P, rows)

    def export(self, P, rows, sign):
        return engine.panel_export(self.ctx, P, rows, 2 * self.half, sign)


class HilbertOperatorOps(ComplexOps):
    """The analytic signal Z = (I + i Hc) A of this rank's REAL slice A without its imaginary part (the panel-level form of
    the engine's operator route, eofx_rsvd_hilbert_c64): Hc is the n x n matrix of the Hilbert stage along the samples
    (`engine.hilbert_operator`), resident as a matrix of its own; Z^H W = A^T (W - i Hc^T W) and Z Y = (I + i Hc)(A Y), so every
    product streams the real slice once and applies Hc to an n x LP sample-side panel.  Feature-sharded: A Y is a partial sum
    over the rank's features and (I + i Hc) is linear, so the driver's all-reduce of the finished panel gives the same sum.
    Reference: single/eof.py:546-555 (the analytic signal) -> linalg/decomposer.py:149-160."""

    def __init__(self, ctx, A, Hop):
        if A.masked:
            raise NotImplementedError("panel-level operator route on a masked in-place slice (preprocess without allow_masked)")
        if Hop.n != A.n or Hop.p != A.n:
            raise ValueError("the Hilbert operator must be n x n")
        self.ctx, self.A, self.B, self.Hop = ctx, A, None, Hop
        self.n, self.p, self.n_pad, self.p_pad = A.n, A.p, A.n_pad, A.p_pad
        self.half = HALF

    def zh_mul(self, Wn, final=False):
        pr = self.ctx.precision[1 if final else 0]
        T = engine.panel_tmul(self.ctx, self.Hop, Wn, prec=pr)                  # Hc^T [Wr | Wi]
        R = engine.cpanel_combine(self.ctx, Wn, T, True, out=T)                 # W - i Hc^T W
        return engine.panel_tmul(self.ctx, self.A, R, prec=pr)

    def z_mul(self, Yp, final=False):
        pr = self.ctx.precision[1 if final else 0]
        T = engine.panel_mul(self.ctx, self.A, Yp, prec=pr)                     # A [Yr | Yi]  (partial sum over this slice)
        HT = engine.panel_mul(self.ctx, self.Hop, T, prec=pr)                   # Hc T
        return engine.cpanel_combine(self.ctx, T, HT, False, out=T)             # T + i Hc T


# numpy solves the Rayleigh-Ritz problem here: every sketch width up to 64 at 7 products fits WITHOUT a restart.  The engine entry
# (rsvd_c64_impl, csrc/eofx_abi.hip) keeps its host solver below order 384 with thick restarts instead, so for sketches of
# 49 .. 64 columns the two drivers run different recurrences (same Krylov space up to the restart, both converge to the same
# modes; results agree to the convergence level, not mode for mode in the unconverged tail) -- a deliberate divergence.
KRYLOV_MAX_ORDER = 8 * 64


def _block_krylov(ops, comm, Z0, q, l, half, small, tall, fwd, bwd, gram, orth, orth_tall, adaptive=False, k=None, q_first_check=4,
                  products=None):
    """The q products of the decomposition as a block Lanczos recurrence on the small side with a Rayleigh-Ritz step over the
    whole Krylov space (the engine entry `eofx_rsvd_c64` does the same on one GPU, csrc/eofx_abi.hip; round 5: the reference's
    complex branch is a block Krylov-class solver, scipy svds(lobpcg), decomposer.py:149-160).  Returns the tall Ritz panel
    A_op K y -- a linear combination of the tall panels of the products, no extra pass.  Sharded: the small-side Gram matrices
    are all-reduced when the small side is the feature side; everything else is local or replicated.
    adaptive (n_iter="converge"): q is the LIMIT; every third product from `q_first_check` on the Ritz values of the blocks multiplied
    so far are compared with those of the previous check and the recurrence stops when every wanted value is good to 2e-6 by its
    own history -- the rule of the engine entry (rsvd_c64_impl, csrc/eofx_abi.hip: rise D_c between two checks, ratio rho of two
    successive rises, distance to go D_c rho / (1 - rho); vectors of modes separated by 4 % in sigma^2: error / gap <= 1e-5).
    `products` (a list) receives the number of products made."""
    torch = engine._torch()
    lp = 2 * half
    k = l if k is None else int(k)

    def real_gram(P, side):
        G = ops.gram_real(P)
        if side == "p":
            G = comm.sum_(G)
        return G.detach().cpu().numpy() if torch.is_tensor(G) else np.asarray(G)

    def cblock(G, r0, c0):                       # complex l x l block of a real Gram matrix at (r0, c0)
        rr, ri = G[r0:r0 + l, c0:c0 + l], G[r0:r0 + l, c0 + half:c0 + half + l]
        ir, ii = G[r0 + half:r0 + half + l, c0:c0 + l], G[r0 + half:r0 + half + l, c0 + half:c0 + half + l]
        return (rr + ii) + 1j * (ri - ir)

    def cholqr(P, side, dref=None, tolref=0.0):
        """Cholesky-QR with the dependency rules of eofx_rsvd_c64 -> (Q, R, live columns)"""
        Hm = cblock(real_gram(P, side), 0, 0)
        Hm = 0.5 * (Hm + Hm.conj().T)
        d0 = Hm.diagonal().real.copy()
        A = Hm.copy()
        dead = np.zeros(l, bool)
        for j in range(l):
            d = A[j, j].real
            dj = not (d > 1e-13 * d0[j]) or not (d0[j] > 0.0) or (dref is not None and not (d > tolref * dref[j]))
            dead[j] = dj
            rjj = 1.0 if dj else np.sqrt(d)
            A[j, j] = rjj
            A[j, j + 1:] *= 0.0 if dj else 1.0 / rjj
            if not dj and j + 1 < l:
                A[j + 1:, j + 1:] -= np.outer(A[j, j + 1:].conj(), A[j, j + 1:])
        R = np.triu(A)
        R[dead, dead] = 0.0
        T = np.zeros((l, l), complex)
        live = ~dead
        if live.any():
            T[np.ix_(live, live)] = np.linalg.inv(np.triu(A)[np.ix_(live, live)])
        return ops.right_mul(P, T), R, int(live.sum()), d0

    def embed_stack(blocks):                     # real (len(blocks) lp) x lp matrix applying complex l x l blocks on the right
        return np.concatenate([_embed_right(c, half) for c in blocks], axis=0)

    def assemble(nbr, nWr, last=None):
        """H = K^H M K over the first nbr blocks from the products of the first nWr ones (+ the last diagonal block)"""
        G = real_gram(torch.cat(K[:nbr] + W[:nWr], dim=1), small)
        m = nbr * l
        raw = {}
        for i in range(nWr):
            for j in range(nbr):
                cb = cblock(G, j * lp, (nbr + i) * lp)
                raw[(j, i)] = cb if Rf[i] is None else cb @ Rf[i]
        if last is not None:
            raw[(nbr - 1, nbr - 1)] = last
        Hm = np.zeros((m, m), complex)
        for a in range(nbr):
            for b in range(a, nbr):
                u, v = raw.get((a, b)), raw.get((b, a))
                if u is None and v is None:
                    continue
                val = u if v is None else (v.conj().T if u is None else 0.5 * (u + v.conj().T))
                Hm[a * l:(a + 1) * l, b * l:(b + 1) * l] = val
                if a != b:
                    Hm[b * l:(b + 1) * l, a * l:(a + 1) * l] = val.conj().T
        return 0.5 * (Hm + Hm.conj().T)

    def ritz_values(nbr):
        return np.linalg.eigvalsh(assemble(nbr, nbr))[::-1][:l]

    next_check, th_prev, rise_prev = int(q_first_check), None, None
    K, W, slots, Rf = [], [], [], []
    Zb, _, _, _ = cholqr(Z0, small)
    K.append(Zb)
    orth_rest = orth_tall
    exhausted = False
    for it in range(q):
        Y = fwd(K[-1])
        if it == 0 or orth_rest:
            Y, R, _, _ = cholqr(Y, tall)
            Rf.append(R)
        else:
            Rf.append(None)
        slots.append(Y)
        Wb = bwd(Y)
        W.append(Wb)
        nbk = len(K)
        Kc = torch.cat(K, dim=1)
        G = real_gram(torch.cat([Kc, Wb], dim=1), small)
        c = [cblock(G, b * lp, nbk * lp) for b in range(nbk)]
        if it == 0 and not orth_tall and q > 1:
            H00 = c[0] if Rf[0] is None else c[0] @ Rf[0]
            wv = np.linalg.eigvalsh(0.5 * (H00 + H00.conj().T))
            orth_rest = not (wv[0] > 0.0) or np.sqrt(wv[-1] / wv[0]) > 30.0
        V1 = ops.matmul_real(torch.cat([Kc, Wb], dim=1), np.concatenate([-embed_stack(c), np.eye(lp)], axis=0))
        Gv = cblock(real_gram(V1, small), 0, 0)
        dref = Gv.diagonal().real + sum((np.abs(cb) ** 2).sum(axis=0) for cb in c)
        Zn, _, live, _ = cholqr(V1, small, dref, 1e-10)
        if live > 0:
            G2 = real_gram(torch.cat([Kc, Zn], dim=1), small)
            c2 = [cblock(G2, b * lp, nbk * lp) for b in range(nbk)]
            V2 = ops.matmul_real(torch.cat([Kc, Zn], dim=1), np.concatenate([-embed_stack(c2), np.eye(lp)], axis=0))
            Zn, _, live, _ = cholqr(V2, small, np.ones(l), 0.25)
        if live == 0:
            exhausted = True
            break
        K.append(Zn)
        if adaptive and it + 1 >= next_check and it + 1 < q:
            th = ritz_values(len(W))
            worst = 1e300 if th_prev is None else 0.0
            rise = np.zeros(k)
            if th_prev is not None:
                for j in range(k):
                    t = max(th[j], 1e-300)
                    rise[j] = abs(th[j] - th_prev[j]) / t
                    rho = 0.5 if rise_prev is None or not rise_prev[j] > 0.0 else min(0.7, max(0.02, rise[j] / rise_prev[j]))
                    est = rise[j] * rho / (1.0 - rho)
                    score = est / 4e-6
                    gaps = ([(th[j - 1] - th[j]) / t] if j > 0 else []) + ([(th[j] - th[j + 1]) / t] if j + 1 < len(th) else [])
                    gap = min(gaps) if gaps else 1e300
                    if 0.04 <= gap < 1e300:
                        score = max(score, est / gap / 1e-5)
                    worst = max(worst, score)
            if worst <= 1.0:
                break
            if th_prev is not None:
                rise_prev = rise
            th_prev = th[:k].copy()
            next_check = it + 1 + 3
    if products is not None:
        products.append(len(W))
    nb, nW = len(K), len(W)
    Hqq = None
    if not exhausted:
        Y = fwd(K[-1])
        Gq = cblock(real_gram(Y, tall), 0, 0)
        Hqq = 0.5 * (Gq + Gq.conj().T)
        if orth_rest:
            Y, R, _, _ = cholqr(Y, tall)
            Rf.append(R)
        else:
            Rf.append(None)
        slots.append(Y)
    Hm = assemble(nb, nW, Hqq)
    wv, y = np.linalg.eigh(0.5 * (Hm + Hm.conj().T))
    y = y[:, ::-1][:, :l]
    coeff = []
    for b in range(nb):
        yb = y[b * l:(b + 1) * l]
        coeff.append(yb if Rf[b] is None else Rf[b] @ yb)
    return ops.matmul_real(torch.cat(slots, dim=1), embed_stack(coeff))


class _NoComm:
    rank, world = 0, 1

    def sum_(self, t):
        return t


def _fix_null_small(P, rows, k, first, half):
    """Numerically null modes (more modes asked for than the matrix has rank): the small-side vectors B^H u / s are rounding noise
    there, where the reference's solver (scipy svds ends in a dense SVD of A V) returns orthonormal columns whatever the values.
    Columns [first, k) of the REPLICATED small-side panel [Re | Im] are re-orthonormalised on the host against all the columns
    before them (two rounds of Gram-Schmidt in complex128; a column inside their span is replaced by the first unit vector that
    is not) -- the host algorithm of the engine entry (csrc/eofx_abi.hip, rsvd_c64_impl); values and the tall side are untouched."""
    torch = engine._torch()
    h = P.detach().cpu().numpy() if torch.is_tensor(P) else np.asarray(P)
    Z = h[:rows, :k].astype(np.complex128) + 1j * h[:rows, half:half + k].astype(np.complex128)

    def project_out(v, upto):
        for _ in range(2):
            for c in range(upto):
                v = v - np.vdot(Z[:, c], v) * Z[:, c]
        return v, float(np.linalg.norm(v))

    next_unit = 0
    for j in range(first, k):
        v = Z[:, j].copy()
        nn0 = float(np.linalg.norm(v))
        if np.isfinite(nn0) and nn0 > 0.0:
            v, nn = project_out(v, j)
            nn /= nn0
        else:
            nn = 0.0
        while not nn > 1e-3 and next_unit < rows:
            v = np.zeros(rows, np.complex128)
            v[next_unit] = 1.0
            next_unit += 1
            v, nn = project_out(v, j)
            if nn > 0.1:
                break
            nn = 0.0
        nrm = float(np.linalg.norm(v))
        Z[:, j] = v / nrm if nrm > 0.0 else 0.0
    out = h.copy()
    out[:rows, first:k] = Z[:, first:k].real.astype(np.float32)
    out[:rows, half + first:half + k] = Z[:, first:k].imag.astype(np.float32)
    if torch.is_tensor(P):
        return torch.as_tensor(out, device=P.device)
    return out


def complex_rsvd(ctx, A, B, k: int, n_oversamples: int = 10, n_iter="auto", random_state=None, flip=True,
                 ops=None, comm=None, p_total=None, p_offset=0, omega=None):
    """-> (U[n,k] complex64, s[k] float32, V[p_local,k] complex64) with Z ~ U diag(s) V^H, V = conj(VT).T.

    Single GPU: `complex_rsvd(ctx, A, B, k, ...)`.  Feature-sharded: every rank passes its slice
    (A, B) with `comm`, the global feature count `p_total` and its `p_offset`; U and s come back
    replicated, V holds this rank's features."""
    ops = ops or ComplexOps(ctx, A, B)
    comm = comm or _NoComm()
    n, p_loc = ops.n, ops.p
    p = p_loc if p_total is None else int(p_total)
    r = min(n, p)
    if k > r:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {r}).")
    l = min(k + n_oversamples, r)
    if l > MAX_HALF:
        raise NotImplementedError(f"complex sketch width {l} > {MAX_HALF} is not supported by this build "
                                  f"(n_modes + n_oversamples <= {MAX_HALF})")
    half = HALF if l <= HALF else MAX_HALF      # wider sketches: 128-column real panels, two column blocks per pass
    if hasattr(ops, "set_half"):
        ops.set_half(half)
    elif half != HALF:
        raise NotImplementedError(f"these panel operations hold {HALF} complex columns (sketch width {l})")
    lp = 2 * half
    adaptive = n_iter == "converge"
    auto_count = 7 if k < 0.1 * r else 4
    if n_iter == "auto" or n_iter is None:
        n_iter = auto_count
    elif adaptive:      # until every wanted value is good to 2e-6 (at most 20 products, and what the Rayleigh-Ritz order allows)
        n_iter = max(2, min(20, KRYLOV_MAX_ORDER // max(l, 1) - 1))
    if l == r:      # full-width sketch spans everything: identity, not an ill-conditioned square Gaussian
        omega = np.eye(r, dtype=np.float32)
    elif omega is not None:                                                       # the caller's draw (one for all ranks)
        omega = np.asarray(omega.result() if hasattr(omega, "result") else omega, dtype=np.float32)
        if omega.shape != (r, k + n_oversamples):
            raise ValueError(f"omega must have shape {(r, k + n_oversamples)}")
        omega = omega[:, :l]
    else:
        omega = engine.sketch_matrix(r, k + n_oversamples, random_state)[:, :l]   # real Gaussian start
    transposed = n < p      # A_op = Z^H: tall side = features (sharded), small side = samples

    def to_feature(P, f=False):
        return ops.zh_mul(P, f)

    def to_sample(P, f=False):
        return comm.sum_(ops.z_mul(P, f))

    def gram(P, side):
        G = ops.gram_real(P)
        if side == "p":
            G = comm.sum_(G)
        return _hermitian_gram(G.detach().cpu().numpy(), l, half)

    def orth(P, side):
        """Q = P (V diag(w^-1/2)), P^H P = V diag(w) V^H; dependent directions -> zero columns"""
        w, V = np.linalg.eigh(gram(P, side))
        good = w > 1e-13 * max(w.max(), 0.0)
        T = np.zeros((l, l), dtype=complex)
        T[:, good] = V[:, good] / np.sqrt(w[good])
        return ops.right_mul(P, T)

    if transposed:
        small, tall, fwd, bwd = "n", "p", to_feature, to_sample
        rows0 = omega
    else:
        small, tall, fwd, bwd = "p", "n", to_sample, to_feature
        rows0 = omega[p_offset:p_offset + p_loc]
    host = np.zeros((rows0.shape[0], lp), np.float32)
    host[:, :l] = rows0
    Z = ops.import_panel(host, small)
    # like the real driver: re-normalise the tall panel inside the iteration while it is small (sharded.py)
    tall_total = n if tall == "n" else p
    from .sharded import _orth_tall

    orth_tall = _orth_tall(tall_total, lp, getattr(ctx, "precision", ("f16x3",))[0])
    orth_rest = orth_tall
    q = int(n_iter)
    made = []
    if q >= 1 and (q + 1) * l <= KRYLOV_MAX_ORDER and hasattr(ops, "matmul_real"):
        Yx = _block_krylov(ops, comm, Z, q, l, half, small, tall, fwd, bwd, gram, orth, orth_tall, adaptive=adaptive, k=k,
                           q_first_check=max(2, auto_count - 3), products=made)
        Q = orth(orth(Yx, tall), tall)
    else:
        for it in range(q):
            Yt = fwd(Z)
            if it == 0 or orth_rest:           # the first iteration always re-normalises the tall panel (rsvd_core)
                Yt = orth(Yt, tall)
            Wp = bwd(Yt)
            if it == 0 and not orth_tall and q > 1:      # peaked spectrum: keep the step (eofx_peaked_spectrum's rule)
                wv = np.linalg.eigvalsh(gram(Wp, small))
                orth_rest = not (wv[0] > 0.0) or np.sqrt(wv[-1] / wv[0]) > 30.0
            Z = orth(Wp, small)
        Q = orth(orth(fwd(Z), tall), tall)                  # range basis: a subspace only, power-pass precision
    Bt = bwd(Q, True)                                   # = B^H with B = Q^H A_op
    w, Uh = np.linalg.eigh(gram(Bt, small))             # B B^H = Uh diag(w) Uh^H
    order = np.argsort(w)[::-1][:k]
    s = np.sqrt(np.maximum(w[order], 0.0))
    Uh = Uh[:, order]
    with np.errstate(divide="ignore"):
        inv = np.where(s > 0, 1.0 / s, 0.0)
    Tall = ops.right_mul(Q, Uh)                         # A_op = Tall diag(s) Small^H
    Small = ops.right_mul(Bt, Uh * inv)
    # numerically null modes: the small-side factor stays orthonormal (the replicated sample side of the sharded case, or any
    # single-rank call; a SHARDED small side would need its Gram matrix reduced -- n >= p_total, not a sharded use case)
    first_null = k
    while first_null > 0 and not (s[first_null - 1] > 3e-6 * s[0]):
        first_null -= 1
    if first_null < k and s[0] > 0.0 and (small == "n" or getattr(comm, "world", 1) == 1):
        Small = _fix_null_small(Small, n if small == "n" else p_loc, k, first_null, half)
    Up, Vp = (Small, Tall) if transposed else (Tall, Small)
    sign = np.ones(k)
    if flip:
        # VT = conj(V)^T; numpy's max/min of complex arrays are lexicographic (real part, then imag),
        # ties on the real part are measure-zero.  Global over the feature shards.
        torch = engine._torch()
        amax, amin = ops.argminmax(Vp, p_loc)
        cols = torch.arange(k)
        cand = []
        for idx in (amax[:k].cpu(), amin[:k].cpu()):
            if p_loc > 0:
                vr = Vp[idx, cols].detach().cpu().numpy().astype(np.float64)
                vi = Vp[idx, cols + half].detach().cpu().numpy().astype(np.float64)
            else:
                vr, vi = np.full(k, np.nan), np.zeros(k)
            cand.append((vr, -vi))          # conj flips the imaginary part
        (mr, mi), (nr, ni) = cand
        if comm.world > 1:
            mr, mi, nr, ni = _global_lex_extrema(comm, mr, mi, nr, ni)
        sign = np.where(np.hypot(mr, mi) >= np.hypot(nr, ni), 1.0, -1.0)
    sg = np.concatenate([sign, np.ones(half - k), sign, np.ones(half - k)])

    def export(P, rows):
        """signed [Re | Im] columns -> interleaved complex64 on the device -> one copy into a page-locked host array
        (the 64-wide panel of a 1M-row factor is 265 MB; its k complex columns are 166 MB at k = 20)"""
        torch = engine._torch()
        if not torch.is_tensor(P) or rows == 0:
            full = ops.export(P, rows, sg)
            out = np.empty((rows, k), dtype=np.complex64)
            out.real = full[:, :k]
            out.imag = full[:, half:half + k]
            return out
        sgt = torch.as_tensor(sign, dtype=torch.float32, device=P.device)
        z = torch.stack((P[:rows, :k] * sgt, P[:rows, half:half + k] * sgt), dim=-1)     # [rows, k, 2]
        out = engine._host_out((rows, k), np.complex64)
        torch.from_numpy(out.view(np.float32).reshape(rows, k, 2)).copy_(z)
        return out

    return export(Up, n), s.astype(np.float32), export(Vp, p_loc)


def _global_lex_extrema(comm, mr, mi, nr, ni):
    """combine per-rank lexicographic (real, imag) max / min candidates over the ranks"""
    import torch
    import torch.distributed as dist

    k = mr.size
    mine = torch.tensor(np.stack([mr, mi, nr, ni]), dtype=torch.float64)
    bufs = [torch.empty_like(mine) for _ in range(comm.world)]
    dev = None
    if dist.get_backend(comm.group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = mine.to(dev)
        bufs = [b.to(dev) for b in bufs]
    dist.all_gather(bufs, mine, group=comm.group)
    allc = torch.stack(bufs).cpu().numpy()            # (world, 4, k)
    out = [np.empty(k) for _ in range(4)]
    for j in range(k):
        mx = max(((allc[r, 0, j], allc[r, 1, j]) for r in range(comm.world) if not np.isnan(allc[r, 0, j])))
        mn = min(((allc[r, 2, j], allc[r, 3, j]) for r in range(comm.world) if not np.isnan(allc[r, 2, j])))
        out[0][j], out[1][j], out[2][j], out[3][j] = mx[0], mx[1], mn[0], mn[1]
    return out
