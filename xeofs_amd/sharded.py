"""Feature-sharded randomized SVD: one process per GPU, space axis split across ranks,
RCCL (torch.distributed backend "nccl") all-reduces of the small panels between passes.

Rank g holds X_g = X[:, p_g] (all samples, its slice of the stacked feature axis) as a
ResidentMatrix.  Per pass the only traffic is an all-reduce(sum) of the (n x L) float32
panel (2.4 MB at n=10000, L=64) or of an (L x L) float64 Gram matrix (32 kB); the panels
on the sample side are replicated bit-identically on every rank, the feature-side panels
stay sharded.  The step sequence is the same as the single-GPU driver
`eofx_rsvd_f32` (csrc/eofx_abi.hip `rsvd_core`); at world size 1 the two agree bitwise.

The arithmetic lives behind `PanelOps`; the product implementation is `HipPanelOps`
(C ABI calls).  The orchestration is backend-agnostic so that its collectives and
sharding logic are exercised by world_size-2 gloo tests without a GPU.
"""

from __future__ import annotations

import numpy as np


class Comm:
    """Collectives over torch.distributed (nccl == RCCL on ROCm; gloo on CPU)."""

    def __init__(self, group=None, force=False):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        init = dist.is_available() and dist.is_initialized()
        # `force` issues the collectives even at world size 1 (exercises the RCCL calls on one GPU)
        self.active = init and (dist.get_world_size(group) > 1 or force)
        self.rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def sum_(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def max_(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def min_(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t


class HipPanelOps:
    """Panel steps on this rank's ResidentMatrix through the C ABI (include/eofx.h)."""

    def __init__(self, ctx, mat):
        from . import engine

        self.e = engine
        self.ctx = ctx
        self.mat = mat
        self.n, self.p = mat.n, mat.p          # p = local feature count
        self.n_pad, self.p_pad = mat.n_pad, mat.p_pad

    def import_panel(self, src, side):
        rows_pad = self.n_pad if side == "n" else self.p_pad
        L = self.e.panel_width(src.shape[1])
        return self.e.panel_import(self.ctx, src, rows_pad, L)

    def tmul(self, Zn, final=False):   # feature panel = X_g^T Zn
        return self.e.panel_tmul(self.ctx, self.mat, Zn, prec=self.ctx.precision[1 if final else 0])

    def mul(self, Yp, final=False):    # sample panel (partial sum over this rank's features) = X_g Yp
        return self.e.panel_mul(self.ctx, self.mat, Yp, prec=self.ctx.precision[1 if final else 0])

    def gram(self, P):
        return self.e.panel_gram(self.ctx, P)

    def cholqr(self, P, l, G):
        return self.e.panel_cholqr(self.ctx, P, l, G)

    def matmul(self, P, M):
        import torch

        Md = torch.as_tensor(np.ascontiguousarray(M, dtype=np.float64), device=P.device)
        return self.e.panel_matmul(self.ctx, P, Md)

    def colminmax(self, P, rows):
        return self.e.panel_colminmax(self.ctx, P, rows)

    def export(self, P, rows, k, sign=None, device_out=False):
        return self.e.panel_export(self.ctx, P, rows, k, sign, device_out)

    def eigh(self, G):
        return self.e.host_eigh(G)


def rsvd_auto_iters(k, n, p):
    """sklearn extmath._randomized_svd: 7 if n_components < 0.1 * min(M.shape) else 4."""
    return 7 if k < 0.1 * min(n, p) else 4


def sharded_rsvd(ops, comm: Comm, k: int, p_total: int, p_offset: int, n_oversamples: int = 10,
                 n_iter="auto", random_state=None, flip: bool = True, omega=None, device_out: bool = False):
    """Randomized SVD of X = [X_0 | X_1 | ...] with the feature axis sharded over ranks.

    Returns (U[n, k] replicated, s[k] replicated, V_local[p_g, k]) as float32 numpy arrays.
    `omega` is the global sketch matrix (min(n, p_total) x (k + n_oversamples)), identical
    on every rank (same seed), drawn as scikit-learn does.
    """
    from .engine import sketch_matrix

    n, p_loc = ops.n, ops.p
    r = min(n, p_total)
    if k > r:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {r}).")
    l_req = k + n_oversamples
    l = min(l_req, r)
    if n_iter == "auto" or n_iter is None or (isinstance(n_iter, int) and n_iter < 0):
        n_iter = rsvd_auto_iters(k, n, p_total)
    if omega is None:
        omega = sketch_matrix(r, l_req, random_state)
    omega = np.ascontiguousarray(omega[:, :l], dtype=np.float32)
    if l == r:      # full-width sketch spans everything: identity instead of an (ill-conditioned) square Gaussian
        omega = np.eye(r, dtype=np.float32)
    transposed = n < p_total   # A = X^T: tall side = features (sharded), small side = samples

    # side bookkeeping: "n" panels are replicated, "p" panels are sharded by rows
    if transposed:
        small, tall = "n", "p"
        Z = ops.import_panel(omega, "n")
    else:
        small, tall = "p", "n"
        Z = ops.import_panel(omega[p_offset:p_offset + p_loc], "p")

    def to_side(P, side, final=False):
        """product that lands on `side` from a panel on the other side; `final` selects the
        precision of the last two passes (eofx_ctx_set_precision)"""
        if side == "p":
            return ops.tmul(P, final)               # local, no communication
        return comm.sum_(ops.mul(P, final))         # partial sums over the feature shards

    def gram(P, side):
        G = ops.gram(P)
        return comm.sum_(G) if side == "p" else G

    for _ in range(int(n_iter)):
        Yt = to_side(Z, tall)
        W = to_side(Yt, small)
        Z = ops.cholqr(W, l, gram(W, small))
    Yt = to_side(Z, tall)                        # range basis: a subspace only, power-pass precision
    Q = ops.cholqr(Yt, l, gram(Yt, tall))
    Q = ops.cholqr(Q, l, gram(Q, tall))          # CholeskyQR2
    Bt = to_side(Q, small, True)
    G = gram(Bt, small)
    Gh = G.detach().cpu().numpy()[:l, :l]
    Gh = 0.5 * (Gh + Gh.T)
    if not np.isfinite(Gh).all():
        raise np.linalg.LinAlgError("SVD failed. This may be due to isolated NaN values in the data.")
    w, Uh = ops.eigh(Gh)
    s = np.sqrt(np.maximum(w[:k], 0.0))
    L = Z.shape[1]
    Lo = (k + 31) // 32 * 32
    M1 = np.zeros((L, Lo))
    M2 = np.zeros((L, Lo))
    M1[:l, :k] = Uh[:, :k]
    with np.errstate(divide="ignore"):
        inv = np.where(s > 0, 1.0 / s, 0.0)
    M2[:l, :k] = Uh[:, :k] * inv
    Tv = ops.matmul(Q, M1)      # singular vectors on the tall side
    Sv = ops.matmul(Bt, M2)     # singular vectors on the small side
    Vp, Up = (Tv, Sv) if transposed else (Sv, Tv)
    sign = None
    if flip:  # xeofs sign rule on VT: global per-mode max / min over all features
        mx, mn = ops.colminmax(Vp, p_loc)
        mx, mn = comm.max_(mx), comm.min_(mn)
        mxh, mnh = mx.detach().cpu().numpy()[:k], mn.detach().cpu().numpy()[:k]
        sign = np.where(np.abs(mxh) >= np.abs(mnh), 1.0, -1.0)
    if device_out:   # results stay in HBM (torch tensors); nothing crosses PCIe
        U = ops.export(Up, n, k, sign, True)
        V = ops.export(Vp, p_loc, k, sign, True)
    else:
        U = ops.export(Up, n, k, sign)
        V = ops.export(Vp, p_loc, k, sign)
    return U, s.astype(np.float32), V


def shard_bounds(p_total: int, world: int, rank: int):
    """Contiguous, balanced split of the stacked feature axis."""
    base, rem = divmod(p_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
