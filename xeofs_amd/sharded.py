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
        self._prof = None

    # Measurement aid (bench.py `comm`): with profiling on, every collective is bracketed by events on the current
    # stream (device tensors: the stream waits for RCCL's, so the interval covers the collective) or by wall time
    # (host tensors); profile_read() -> dict(calls, bytes, ms) since profile(True) and resets.
    def profile(self, enable=True):
        self._prof = dict(calls=0, bytes=0, ms=0.0, events=[]) if enable else None

    def profile_read(self):
        pr = self._prof or dict(calls=0, bytes=0, ms=0.0, events=[])
        ms = pr["ms"]
        if pr["events"]:
            import torch

            torch.cuda.synchronize()
            ms += sum(a.elapsed_time(b) for a, b in pr["events"])
        out = dict(calls=pr["calls"], bytes=pr["bytes"], ms=ms)
        if self._prof is not None:
            self.profile(True)
        return out

    def _reduce(self, t, op):
        if not self.active:
            return t
        pr = self._prof
        if pr is None:
            self.dist.all_reduce(t, op=op, group=self.group)
            return t
        pr["calls"] += 1
        pr["bytes"] += t.numel() * t.element_size()
        if t.is_cuda:
            import torch

            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.dist.all_reduce(t, op=op, group=self.group)
            b.record()
            pr["events"].append((a, b))
        else:
            import time

            t0 = time.perf_counter()
            self.dist.all_reduce(t, op=op, group=self.group)
            pr["ms"] += 1e3 * (time.perf_counter() - t0)
        return t

    def sum_(self, t):
        return self._reduce(t, self.dist.ReduceOp.SUM)

    def max_(self, t):
        return self._reduce(t, self.dist.ReduceOp.MAX)

    def min_(self, t):
        return self._reduce(t, self.dist.ReduceOp.MIN)


class NativeComm:
    """rank / world of the communicator attached to an engine context (`attach_native`, or `engine.comm_init_rccl` by a host
    that brings its own rendezvous): what the model-level drivers below need when EVERY collective goes through the engine
    (`native=True`) and no torch.distributed process group exists."""

    active = False
    group = None

    def __init__(self, ctx):
        from . import engine

        att = engine.comm_attached(ctx)
        if att is None:
            raise ValueError("no communicator is attached to the engine context")
        self.world, self.rank = att


class _DeviceView:
    """a raw device pointer as something torch.as_tensor accepts (zero copy)"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(count),), "typestr": typestr, "version": 2}


def attach_native(ctx, comm: Comm) -> bool:
    """Give the engine context its own communicator over the ranks of `comm`, so that `engine.fit_sharded`
    (eofx_fit_sharded_f32) issues every collective itself, on its own stream, between its kernels.
    nccl backend: an RCCL communicator of the engine's own -- rank 0 draws the unique id, torch.distributed carries the 128
    bytes (the one thing it is used for), every rank calls ncclCommInitRank.  Other backends (gloo: CPU tests, two processes
    on one GPU): a host callback that synchronises, reduces a host copy through torch.distributed and copies back.
    -> whether a communicator was attached (False without an initialised process group)."""
    import torch
    from . import engine

    dist = comm.dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    world, rank = comm.world, comm.rank

    def agreed(ok_local: bool) -> bool:      # every rank attaches, or none does: the fit's control flow is collective
        flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32,
                            device=f"cuda:{ctx.device}" if dist.get_backend(comm.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=comm.group)
        return bool(int(flag.item()))

    if dist.get_backend(comm.group) == "nccl":
        try:
            uid = engine.comm_unique_id() if rank == 0 else None
        except Exception as e:              # librccl not loadable: the same on every rank of a node, the vote covers the rest
            uid, err = None, e
        box = [uid]
        dist.broadcast_object_list(box, src=0, group=comm.group)
        ok = False
        if box[0] is not None:
            try:
                engine.comm_init_rccl(ctx, box[0], world, rank)
                ok = engine.comm_selftest(ctx)
            except Exception as e:
                import warnings

                warnings.warn(f"xeofs_amd: the engine's RCCL communicator is not usable ({e}); "
                              "the sharded fit takes the panel-level driver over torch.distributed")
        if agreed(ok):
            return True
        try:
            engine.comm_clear(ctx)
        except Exception:
            pass
        return False
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}
    types = {0: "<f4", 1: "<f8", 2: "<i4"}

    def allreduce(buf, count, dtype, op, _stream):
        torch.cuda.synchronize()
        t = torch.as_tensor(_DeviceView(buf, count, types[dtype]), device=f"cuda:{ctx.device}")
        h = t.cpu()
        dist.all_reduce(h, op=ops[op], group=comm.group)
        t.copy_(h)
        torch.cuda.synchronize()
        return 0

    engine.comm_set_callback(ctx, allreduce, world, rank)
    if agreed(engine.comm_selftest(ctx)):
        return True
    engine.comm_clear(ctx)
    return False


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
        if side == "p":
            src = self.mat.scatter_rows(src)      # masked in-place matrix: zero rows at the masked features
        return self.e.panel_import(self.ctx, src, rows_pad, L)

    def tmul(self, Zn, final=False):   # feature panel = X_g^T Zn
        return self.e.panel_tmul(self.ctx, self.mat, Zn, prec=self.ctx.precision[1 if final else 0])

    def mul(self, Yp, final=False):    # sample panel (partial sum over this rank's features) = X_g Yp
        return self.e.panel_mul(self.ctx, self.mat, Yp, prec=self.ctx.precision[1 if final else 0])

    def gram(self, P):
        return self.e.panel_gram(self.ctx, P)

    def cholqr(self, P, l, G):
        return self.e.panel_cholqr(self.ctx, P, l, G)

    def rinv(self, G, l):
        return self.e.panel_rinv(self.ctx, G, l)

    def matmul(self, P, M):
        import torch

        Md = M if torch.is_tensor(M) else torch.as_tensor(np.ascontiguousarray(M, dtype=np.float64), device=P.device)
        return self.e.panel_matmul(self.ctx, P, Md)

    def _feature_side(self, P, rows):
        return self.mat.masked and P.shape[0] == self.p_pad and rows == self.p

    def colminmax(self, P, rows):
        if self._feature_side(P, rows):
            rows = self.mat.p_phys               # zero rows cannot change which of |max|, |min| is larger
        return self.e.panel_colminmax(self.ctx, P, rows)

    def export(self, P, rows, k, sign=None, device_out=False):
        if self._feature_side(P, rows):          # masked in-place matrix: drop the rows of the masked features
            return self.mat.compact_rows(self.e.panel_export(self.ctx, P, self.mat.p_phys, k, sign, device_out))
        return self.e.panel_export(self.ctx, P, rows, k, sign, device_out)

    def eigh(self, G):
        return self.e.host_eigh(G)

    def sample_gram(self):             # X_g X_g^T, [n_pad, n_pad] float32 on the device
        return self.mat.sample_gram()

    def dot(self, a, b):
        return self.e.vec_dot(self.ctx, a, b)


def rsvd_auto_iters(k, n, p):
    """sklearn extmath._randomized_svd: 7 if n_components < 0.1 * min(M.shape) else 4."""
    return 7 if k < 0.1 * min(n, p) else 4


def _resolve_sketch(k, r, n_oversamples, omega, random_state):
    """sketch width policy shared by the drivers: l = min(k + n_oversamples, rank); a full-width sketch
    spans everything, so the identity replaces an (occasionally ill-conditioned) square Gaussian"""
    from .engine import sketch_matrix

    if k > r:
        raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {r}).")
    l_req = k + n_oversamples
    l = min(l_req, r)
    if omega is None:
        omega = sketch_matrix(r, l_req, random_state)
    elif hasattr(omega, "result"):          # an engine.SketchFuture: joined here
        omega = omega.result()
    omega = np.ascontiguousarray(omega[:, :l], dtype=np.float32)
    if l == r:
        omega = np.eye(r, dtype=np.float32)
    return omega, l


def _orth_tall(tall_total: int, L: int, prec_power: str = "f16x3") -> bool:
    """the rule of `rsvd_core` (eofx_orth_tall_rule, one definition in the C library): re-normalise the tall panel
    inside the power iterations while that is cheap, and always in the float64 mode"""
    from . import _lib

    rows_pad = (int(tall_total) + 511) // 512 * 512
    return bool(_lib.load().eofx_orth_tall_rule(rows_pad, int(L), _lib.PREC[prec_power]))


def _peaked(G, l: int) -> bool:
    """eofx_peaked_spectrum on the (globally reduced) small-side Gram matrix of the first iteration"""
    from . import _lib

    Gh = np.ascontiguousarray(G.detach().cpu().numpy() if hasattr(G, "detach") else np.asarray(G), dtype=np.float64)
    return bool(_lib.load().eofx_peaked_spectrum(_lib.ptr(Gh), int(Gh.shape[1]), int(l)))


def _prec_power(ops) -> str:
    ctx = getattr(ops, "ctx", None)
    return getattr(ctx, "precision", ("f16x3", "f16x3"))[0]


def _fix_null_columns(la, P, gram, k, first, rows):
    """Columns [first, k) of a factor panel whose modes are numerically null (value <= 1e-5 of the leading one): rounding noise on
    the small side, zeroed dead columns of a Cholesky-QR on the tall side, where scikit-learn returns orthonormal factors whatever
    the values.  The panel-level form of `fix_null_columns` (csrc/eofx_abi.hip): block Gram-Schmidt through the (globally
    reduced) float64 Gram matrix, N <- (N - G G^T N) R^-1 on the host's L x L algebra, two rounds; a column that is zero, not
    finite or inside the span of the others is first replaced by a fixed pseudo-random vector.  The columns before `first`
    keep their bits; rows beyond `rows` (padding) and rows that are zero in all the columns before `first` (masked
    features) stay zero."""
    import torch

    if first >= k or rows <= first:
        return P
    L = P.shape[1]
    m = k - first

    def refill(cols):
        r = torch.arange(P.shape[0], device=P.device, dtype=torch.float64)
        live = r < rows
        if first > 0:
            live = live & (P[:, :first] != 0).any(dim=1)
        for c in cols:
            v = torch.frac(torch.sin(r * 12.9898 + (c + 1) * 78.233) * 43758.5453) * 2.0 - 1.0
            P[:, c] = torch.where(live, v, torch.zeros_like(v)).to(P.dtype)

    rounds = 0
    for _attempt in range(8):
        if rounds >= 2:
            break
        G = gram(P)
        G = (G.detach().cpu().numpy() if hasattr(G, "detach") else np.asarray(G)).astype(np.float64)
        bad = [j for j in range(first, k) if not np.isfinite(G[:k, j]).all() or not G[j, j] > 1e-30]
        if bad and rounds == 0:
            refill(bad)
            continue
        B = G[:first, first:k]
        S = G[first:k, first:k] - B.T @ B
        A = S.copy()
        dep = []
        for j in range(m):          # right-looking Cholesky with a pivot floor (a dependent column drops out, is refilled)
            d = A[j, j]
            if not np.isfinite(d) or not d > 1e-6 * max(S[j, j], 1e-300):
                dep.append(first + j)
                A[j, j:] = 0.0
                A[j, j] = 1.0
                continue
            A[j, j] = np.sqrt(d)
            A[j, j + 1:] /= A[j, j]
            for r_ in range(j + 1, m):
                A[r_, r_:] -= A[j, r_] * A[j, r_:]
        if dep:
            if rounds > 0:
                break
            refill(dep)
            continue
        Ri = np.linalg.inv(np.triu(A))
        Mx = np.zeros((L, L))
        Mx[np.arange(first), np.arange(first)] = 1.0
        Mx[first:k, first:k] = Ri
        Mx[:first, first:k] = -B @ Ri
        Q = la.matmul(P, Mx)
        P[:, first:k] = Q[:, first:k]
        rounds += 1
    return P


def _first_null(s, k):
    s0 = float(s[0]) if len(s) else 0.0
    first = k
    while first > 0 and not (float(s[first - 1]) > 1e-5 * s0):
        first -= 1
    return first if np.isfinite(s0) and s0 >= 0.0 else k


def _rsvd_panels(la, to_tall, to_small, gram_small, gram_tall, Z, l, k, n_iter, orth_tall=False, first_tall=None,
                 rows_tall=None, rows_small=None):
    """The pass sequence of `rsvd_core` (csrc/eofx_abi.hip) on abstract products.

    `to_tall(P, final)` / `to_small(P, final)` apply A / A^T to a panel (including whatever reduction
    the sharding needs), `gram_*` return the globally reduced L x L float64 Gram matrix of a panel on
    that side, `la` supplies the matrix-independent steps (cholqr, matmul, eigh).  Returns the
    singular-vector panels (tall side, small side) and the k singular values (float64).
    `first_tall`: the first product A Z when the caller already has it (the statistics-carrying first pass of the fused
    fit, `engine.fit_first`): it replaces the first `to_tall` of the run.
    """
    _first = [first_tall]

    def tall(P):
        y, _first[0] = _first[0], None
        return y if y is not None else to_tall(P, False)

    orth_rest = bool(orth_tall)
    for it in range(int(n_iter)):
        Yt = tall(Z)
        if it == 0 or orth_rest:       # the first iteration always re-normalises the tall panel (rsvd_core)
            Yt = la.cholqr(Yt, l, gram_tall(Yt))
        W = to_small(Yt, False)
        Gs = gram_small(W)
        if it == 0 and not orth_tall and int(n_iter) > 1:   # peaked spectrum?  same rule, same code as the C++ driver
            orth_rest = _peaked(Gs, l)
        Z = la.cholqr(W, l, Gs)
    Yt = tall(Z)                                 # range basis: a subspace only, power-pass precision
    Q = la.cholqr(Yt, l, gram_tall(Yt))
    # CholeskyQR2 as in rsvd_core: the second factor R2 = chol(Q^T Q) is applied on the SMALL side, B^T = (A^T Q) R2^-1,
    # and folded into the final rotation of the tall panel -- one pass over the tall panel less
    R2 = la.rinv(gram_tall(Q), l)
    Bt = la.matmul(to_small(Q, True), R2)
    G = gram_small(Bt)
    Gh = G.detach().cpu().numpy()[:l, :l]
    Gh = 0.5 * (Gh + Gh.T)
    if not np.isfinite(Gh).all():
        raise np.linalg.LinAlgError("SVD failed. This may be due to isolated NaN values in the data.")
    w, Uh = la.eigh(Gh)
    s = np.sqrt(np.maximum(w[:k], 0.0))
    L = Z.shape[1]
    Lo = (k + 31) // 32 * 32
    M1 = np.zeros((L, Lo))
    M2 = np.zeros((L, Lo))
    R2h = R2.detach().cpu().numpy() if hasattr(R2, "detach") else np.asarray(R2)
    M1[:l, :k] = _matmul_rowwise(R2h[:l, :l], Uh[:, :k])
    with np.errstate(divide="ignore"):
        inv = np.where(s > 0, 1.0 / s, 0.0)
    M2[:l, :k] = Uh[:, :k] * inv
    Tv = la.matmul(Q, M1)       # singular vectors on the tall side
    Sv = la.matmul(Bt, M2)      # singular vectors on the small side
    if rows_tall is not None and rows_small is not None:      # numerically null modes: both factors stay orthonormal
        first = _first_null(s, k)
        if first < k:
            Tv = _fix_null_columns(la, Tv, gram_tall, k, first, rows_tall)
            Sv = _fix_null_columns(la, Sv, gram_small, k, first, rows_small)
    return Tv, Sv, s


def _matmul_rowwise(A, B):
    """A @ B summed over the inner index in ascending order with one accumulator per entry (the loop of rsvd_core:
    numpy's BLAS would block the sum differently and the two drivers must agree bit for bit)"""
    out = np.zeros((A.shape[0], B.shape[1]))
    for q in range(A.shape[1]):
        out += A[:, q:q + 1] * B[q:q + 1, :]
    return out


def _sign_from_extrema(comm, ops, Vp, rows, k):
    """xeofs sign rule (utils/xarray_utils.py:273-301): global per-mode max / min over all shards"""
    mx, mn = ops.colminmax(Vp, rows)
    if getattr(comm, "active", False) and hasattr(mx, "detach"):      # one collective: max over [max | -min]
        import torch

        both = comm.max_(torch.cat([mx.reshape(-1), -mn.reshape(-1)]))
        h = both.detach().cpu().numpy()
        mxh, mnh = h[: h.size // 2][:k], -h[h.size // 2:][:k]
    else:
        mx, mn = comm.max_(mx), comm.min_(mn)
        mxh = (mx.detach().cpu().numpy() if hasattr(mx, "detach") else np.asarray(mx))[:k]
        mnh = (mn.detach().cpu().numpy() if hasattr(mn, "detach") else np.asarray(mn))[:k]
    return np.where(np.abs(mxh) >= np.abs(mnh), 1.0, -1.0)


def sharded_rsvd(ops, comm: Comm, k: int, p_total: int, p_offset: int, n_oversamples: int = 10,
                 n_iter="auto", random_state=None, flip: bool = True, omega=None, device_out: bool = False,
                 first=None):
    """Randomized SVD of X = [X_0 | X_1 | ...] with the feature axis sharded over ranks.

    Returns (U[n, k] replicated, s[k] replicated, V_local[p_g, k]) as float32 numpy arrays.
    `omega` is the global sketch matrix (min(n, p_total) x (k + n_oversamples)), identical
    on every rank (same seed), drawn as scikit-learn does.
    `first` = (Z panel, X_g^T Z panel) from `sharded_fit_first`: the sketch is already imported and the first product
    already taken (with the statistics of the shard); only for the transposed case n < p_total.
    """
    n, p_loc = ops.n, ops.p
    omega, l = _resolve_sketch(k, min(n, p_total), n_oversamples, omega, random_state)
    if n_iter == "auto" or n_iter is None or (isinstance(n_iter, int) and n_iter < 0):
        n_iter = rsvd_auto_iters(k, n, p_total)
    transposed = n < p_total   # A = X^T: tall side = features (sharded), small side = samples

    # side bookkeeping: "n" panels are replicated, "p" panels are sharded by rows
    first_tall = None
    if first is not None and not transposed:
        raise ValueError("a precomputed first product needs the sketch on the sample side (n < p_total)")
    if transposed:
        small, tall = "n", "p"
        if first is not None:
            Z, first_tall = first
        else:
            Z = ops.import_panel(omega, "n")
    else:
        small, tall = "p", "n"
        Z = ops.import_panel(omega[p_offset:p_offset + p_loc], "p")

    # rows of a feature-side panel that can carry data (a masked in-place matrix keeps its physical rows, zero where masked)
    p_rows = int(getattr(getattr(ops, "mat", None), "p_phys", p_loc))

    def to_side(P, side, final):
        """product that lands on `side` from a panel on the other side; `final` selects the
        precision of the last two passes (eofx_ctx_set_precision)"""
        if side == "p":
            return ops.tmul(P, final)               # local, no communication
        return comm.sum_(ops.mul(P, final))         # partial sums over the feature shards

    def gram(P, side):
        G = ops.gram(P)
        return comm.sum_(G) if side == "p" else G

    Tv, Sv, s = _rsvd_panels(ops, lambda P, f: to_side(P, tall, f), lambda P, f: to_side(P, small, f),
                             lambda P: gram(P, small), lambda P: gram(P, tall), Z, l, k, n_iter,
                             _orth_tall(n if tall == "n" else p_total, Z.shape[1], _prec_power(ops)), first_tall=first_tall,
                             rows_tall=(p_rows if tall == "p" else n), rows_small=(p_rows if small == "p" else n))
    Vp, Up = (Tv, Sv) if transposed else (Sv, Tv)
    sign = _sign_from_extrema(comm, ops, Vp, p_loc, k) if flip else None
    if device_out:   # results stay in HBM (torch tensors); nothing crosses PCIe
        U = ops.export(Up, n, k, sign, True)
        V = ops.export(Vp, p_loc, k, sign, True)
    else:
        U = ops.export(Up, n, k, sign)
        V = ops.export(Vp, p_loc, k, sign)
    return U, s.astype(np.float32), V


def sharded_crosscov_rsvd(opsx, opsy, comm: Comm, k: int, p1_total: int, p1_offset: int, p2_total: int,
                          p2_offset: int, n_oversamples: int = 10, n_iter="auto", random_state=None,
                          flip: bool = True, omega=None, want_tsc: bool = True):
    """Matrix-free randomized SVD of C = X^T Y / (n - 1) (cross/cpcca.py:168-225, `eofx_crosscov_rsvd_f32`)
    with X and Y each sharded along their own feature axis.

    Rank g holds X_g (n x p1_g) and Y_g (n x p2_g).  C Z = X^T (Y Z): the inner product Y Z is a sum
    over Y's shards (all-reduce of the n x L panel), the outer product is local to X's shards; the
    same for C^T.  Both sides of C are sharded, so every Gram matrix is all-reduced (L x L float64).
    Returns a dict like `engine.crosscov_rsvd`: Q1 / Q2 are this rank's rows, everything else is
    replicated.
    """
    n = opsx.n
    if opsy.n != n:
        raise ValueError("Both data matrices must have the same number of samples but found "
                         f"{n} in the first and {opsy.n} in the second.")
    omega, l = _resolve_sketch(k, min(p1_total, p2_total), n_oversamples, omega, random_state)
    if n_iter == "auto" or n_iter is None or (isinstance(n_iter, int) and n_iter < 0):
        n_iter = rsvd_auto_iters(k, p1_total, p2_total)
    transposed = p1_total < p2_total            # C is p1 x p2: sklearn works on C^T when rows < cols

    def c_mul(Z2, final):                        # C Z2 -> p1 side
        return opsx.tmul(comm.sum_(opsy.mul(Z2, final)), final)

    def ct_mul(Z1, final):                       # C^T Z1 -> p2 side
        return opsy.tmul(comm.sum_(opsx.mul(Z1, final)), final)

    def gram(P):
        return comm.sum_(opsx.gram(P))

    if transposed:       # A = C^T (p2 x p1): small side = p1 (X's features)
        Z = opsx.import_panel(omega[p1_offset:p1_offset + opsx.p], "p")
        Tv, Sv, s = _rsvd_panels(opsx, ct_mul, c_mul, gram, gram, Z, l, k, n_iter, _orth_tall(p2_total, Z.shape[1], _prec_power(opsx)))
        Q1p, Q2p = Sv, Tv
    else:                # A = C (p1 x p2): small side = p2 (Y's features)
        Z = opsy.import_panel(omega[p2_offset:p2_offset + opsy.p], "p")
        Tv, Sv, s = _rsvd_panels(opsx, c_mul, ct_mul, gram, gram, Z, l, k, n_iter, _orth_tall(p1_total, Z.shape[1], _prec_power(opsx)))
        Q1p, Q2p = Tv, Sv
    sign = _sign_from_extrema(comm, opsy, Q2p, opsy.p, k) if flip else None
    out = dict(s=(s / (n - 1)).astype(np.float32), Q1=opsx.export(Q1p, opsx.p, k, sign),
               Q2=opsy.export(Q2p, opsy.p, k, sign))
    # scores = X Q1, Y Q2 (sum over the feature shards) and their norms (cpcca.py:204-208)
    for name, ops, Qp in (("1", opsx, Q1p), ("2", opsy, Q2p)):
        Sn = comm.sum_(ops.mul(Qp, True))
        out["scores" + name] = ops.export(Sn, n, k, sign)
        g = ops.gram(Sn).detach().cpu().numpy()
        out["norm" + name] = np.sqrt(np.diag(g)[:k]).astype(np.float32)
    if want_tsc:
        # sum |C|^2 = <X X^T, Y Y^T> / (n-1)^2 ; X X^T = sum_g X_g X_g^T: one n x n all-reduce,
        # then <G_x, G_y,g> locally and a scalar all-reduce (cpcca.py:991-1000)
        import torch

        Gx = comm.sum_(opsx.sample_gram())
        t = torch.tensor([opsy.dot(Gx, opsy.sample_gram())], dtype=torch.float64, device=Gx.device)
        out["total_squared_covariance"] = float(comm.sum_(t).cpu()[0]) / float(n - 1) ** 2
    return out


def combine_sample_masks(comm: Comm, valid_sample, n_valid_features: int, check_nans: bool = True):
    """Global NaN bookkeeping of the Sanitizer (preprocessing/sanitizer.py:58-126) when the feature
    axis is sharded: a sample is valid if it is valid in any shard; with `check_nans` a sample that is
    all-NaN in one shard's valid features but not in another's has isolated NaNs globally, which the
    reference rejects.  `valid_sample` is this rank's boolean mask (from `eofx_preprocess_f32` with
    check_nans=0 semantics: any valid feature non-NaN).  Returns the global mask (numpy bool)."""
    import torch

    vs = np.asarray(valid_sample, dtype=bool)
    dev = "cpu"
    if comm.active and comm.dist.get_backend(comm.group) == "nccl":
        dev = f"cuda:{torch.cuda.current_device()}"
    cnt = torch.zeros(vs.size + 1, dtype=torch.int32, device=dev)
    has = n_valid_features > 0
    cnt[:-1] = torch.from_numpy((vs & has).astype(np.int32)).to(dev)
    cnt[-1] = int(has)
    comm.sum_(cnt)
    cnt = cnt.cpu().numpy()
    shards, votes = int(cnt[-1]), cnt[:-1]
    if check_nans and np.any((votes != 0) & (votes != shards)):
        raise ValueError("Input data contains partial NaN entries, which will cause the the SVD to fail.")
    return votes > 0


def _gather_counts(comm: Comm, value: int):
    """all-gather of one integer per rank (as an all-reduce of a one-hot vector)"""
    import torch

    dev = "cpu"
    if comm.active and comm.dist.get_backend(comm.group) == "nccl":
        dev = f"cuda:{torch.cuda.current_device()}"
    v = torch.zeros(max(comm.world, 1), dtype=torch.int64, device=dev)
    v[comm.rank] = int(value)
    return comm.sum_(v).cpu().numpy()


def _sum_scalar(comm: Comm, value: float) -> float:
    import torch

    dev = "cpu"
    if comm.active and comm.dist.get_backend(comm.group) == "nccl":
        dev = f"cuda:{torch.cuda.current_device()}"
    return float(comm.sum_(torch.tensor([value], dtype=torch.float64, device=dev)).cpu()[0])


def use_native(ctx, native=None) -> bool:
    """does this call issue its collectives through the engine's own communicator?  `native=None`: yes when one is attached
    to the context (`attach_native`); `native=True` insists on it."""
    from . import engine

    att = ctx is not None and engine.comm_attached(ctx) is not None
    if native is None:
        return att
    if native and not att:
        raise ValueError("no communicator is attached to the engine context (xeofs_amd.sharded.attach_native)")
    return bool(native)


def _sum_host(comm: Comm, buf, native_ctx=None):
    """all-reduce(sum) of a small float64 host vector: through the engine's communicator (eofx_ctx_comm_allreduce_f64) when
    `native_ctx` is given, else through torch.distributed"""
    buf = np.ascontiguousarray(buf, dtype=np.float64)
    if native_ctx is not None:
        from . import engine

        return engine.comm_allreduce_host(native_ctx, buf)
    if getattr(comm, "active", False):
        import torch

        dev = f"cuda:{torch.cuda.current_device()}" if comm.dist.get_backend(comm.group) == "nccl" else "cpu"
        return comm.sum_(torch.from_numpy(buf.copy()).to(dev)).cpu().numpy()
    return buf


def global_facts(comm: Comm, p_local: int, valid_sample, check_nans: bool, total_variance: float, bad: float = 0.0,
                 native_ctx=None):
    """The global facts of a feature-sharded preprocess in ONE all-reduce instead of four small ones
    (each is a collective launch plus a host round trip): the number of valid features of every rank (one-hot slots),
    the votes of `combine_sample_masks` (same rule, same error), the total variance and a veto flag.
    -> (counts[world] int64, global valid-sample mask, total variance, sum of the `bad` flags)"""
    import torch

    vs = np.asarray(valid_sample, dtype=bool)
    n, W = vs.size, max(comm.world, 1)
    has = p_local > 0
    buf = np.zeros(W + n + 3, np.float64)          # float64 sums of small integers are exact
    buf[comm.rank] = float(p_local)
    buf[W:W + n] = vs & has
    buf[W + n], buf[W + n + 1], buf[W + n + 2] = float(has), float(total_variance), float(bad)
    buf = _sum_host(comm, buf, native_ctx)
    counts = np.rint(buf[:W]).astype(np.int64)
    votes, shards = np.rint(buf[W:W + n]).astype(np.int64), int(round(buf[W + n]))
    if check_nans and np.any((votes != 0) & (votes != shards)):
        raise ValueError("Input data contains partial NaN entries, which will cause the the SVD to fail.")
    return counts, votes > 0, float(buf[W + n + 1]), float(buf[W + n + 2])


def sharded_preprocess(ctx, X_local, comm: Comm, center=True, standardize=False, feature_weights=None,
                       check_nans=True, want_stats=True, keep_raw=False, in_place=False, allow_masked=False, for_hilbert=False,
                       native=False):
    """Scaler + Sanitizer + total variance (rows R1-R6) of this rank's slice of the stacked feature axis.

    Per-feature statistics, masks and the compaction are local (`eofx_preprocess_f32`); the global
    facts are combined here: the valid-sample mask / isolated-NaN check
    (`combine_sample_masks`), the number of valid features of every rank (-> this rank's offset on the
    global valid-feature axis) and the total variance.  Returns (ResidentMatrix, stats) like
    `engine.preprocess`, with `p_total`, `p_offset` and the global `total_variance` added.
    """
    from . import engine

    mat, st = engine.preprocess(ctx, X_local, center=center, standardize=standardize, feature_weights=feature_weights,
                                check_nans=check_nans, want_stats=want_stats, keep_raw=keep_raw, in_place=in_place,
                                allow_masked=allow_masked, for_hilbert=for_hilbert)
    counts, vs, tv, _ = global_facts(comm, mat.p, st["valid_sample"], check_nans, st["total_variance"],
                                     native_ctx=ctx if native else None)
    st["p_total"] = int(counts.sum())
    st["p_offset"] = int(counts[:comm.rank].sum())
    st["valid_sample"] = vs
    st["total_variance_local"] = st["total_variance"]
    st["total_variance"] = tv
    return mat, st


def sharded_fit_first(ctx, X_local, comm: Comm, k: int, p_total: int, center=True, standardize=False, feature_weights=None,
                      check_nans=True, want_stats=True, n_oversamples: int = 10, omega=None, random_state=None):
    """`sharded_preprocess` and the FIRST product of `sharded_rsvd` in one engine call per rank (engine.fit_first /
    eofx_fit_first_f32): the statistics of the shard ride on X_g^T Omega, which is local to the rank -- no communication,
    one read of the field less.  Only where the fused pass applies on EVERY shard's shape (n < p_total, sketch narrower than
    its panel); the engine falls back by itself on a rank whose shard holds NaNs, and if that changed the global shape
    (dropped features or samples) the precomputed product is discarded.
    -> (ResidentMatrix, stats as `sharded_preprocess`, first = (Z, Yt) for `sharded_rsvd(..., first=first)` or None)"""
    from . import engine

    X_local = engine._f32c(X_local)
    n, p_loc = X_local.shape
    l_req = int(k) + int(n_oversamples)
    usable = n < p_total and l_req < n and l_req % 32 != 0 and l_req < 64 and ctx.precision[0] == "f16x3"
    if not usable:
        mat, st = sharded_preprocess(ctx, X_local, comm, center, standardize, feature_weights, check_nans, want_stats)
        return mat, st, None
    if omega is None:
        omega = engine.sketch_matrix(n, l_req, random_state)
    elif hasattr(omega, "result"):
        omega = omega.result()
    n_pad = (n + 511) // 512 * 512
    Z = engine.panel_import(ctx, np.ascontiguousarray(omega[:n], dtype=np.float32), n_pad, engine.panel_width(l_req))
    mat, st, Yt = engine.fit_first(ctx, X_local, Z, l_req, center, standardize, feature_weights, check_nans, want_stats)
    # one collective for all global facts; the precomputed product stands only if no rank dropped anything (same n
    # everywhere: a local fact every rank votes on) and n is still below the global p (known to all after the reduction)
    counts, vs, tv, bad = global_facts(comm, mat.p, st["valid_sample"], check_nans, st["total_variance"],
                                       0.0 if (Yt is not None and mat.n == n) else 1.0)
    st["p_total"] = int(counts.sum())
    st["p_offset"] = int(counts[:comm.rank].sum())
    st["valid_sample"] = vs
    st["total_variance_local"] = st["total_variance"]
    st["total_variance"] = tv
    first = (Z, Yt) if (bad == 0.0 and st["p_total"] > n) else None
    return mat, st, first


def sharded_eof_fit(ctx, X_local, comm: Comm, n_modes: int, center=True, standardize=False, feature_weights=None,
                    check_nans=True, random_state=None, n_oversamples: int = 10, n_iter="auto", omega=None,
                    device_out: bool = False, native=None):
    """`EOF.fit` (single/eof.py:85-118) with the space axis sharded: X_local is this rank's
    (n, P_g) slice of the stacked raw field.  Returns the DataContainer entries as a dict; `components`
    holds this rank's rows, everything else is replicated.  Where the shapes allow, every rank takes the statistics of its
    shard during the first product X_g^T Omega (`sharded_fit_first`: one read of the field less, no extra communication)."""
    from . import engine

    n = X_local.shape[0]
    nat = use_native(ctx, native)
    if nat:
        p_raw_total = int(round(_sum_host(comm, [float(X_local.shape[1])], ctx)[0]))
    else:
        p_raw_total = int(_gather_counts(comm, X_local.shape[1]).sum())
    if omega is None and n < p_raw_total:     # one draw for both steps (scikit-learn's stream for this seed)
        omega = engine.sketch_matrix(n, int(n_modes) + int(n_oversamples), random_state)
    if nat and n < p_raw_total and int(n_modes) <= n:
        # ONE engine call per rank (eofx_fit_sharded_f32): statistics during the first product, every collective issued by the
        # engine on its own stream; a land / sea mask stays in place as zero columns of every slice.  None = the ranks voted for
        # the panel-level driver below (isolated NaNs, shapes outside the fused pass): nothing was built.
        res = engine.fit_sharded(ctx, X_local, int(n_modes), p_raw_total, center, standardize, feature_weights, n_oversamples,
                                 n_iter, random_state, omega=omega, device_out=device_out, allow_masked=True)
        if res is not None:
            mat, st, U, s, V = res
            counts = _sum_host(comm, np.eye(1, comm.world, comm.rank).ravel() * float(mat.p), ctx)
            st["p_total"] = int(round(counts.sum()))
            st["p_offset"] = int(round(counts[:comm.rank].sum()))
            st["native"] = True
            s64 = np.asarray(s, dtype=np.float64)
            sc = U * (s if not device_out else engine._torch().as_tensor(s, device=U.device))
            return dict(input_data=mat, components=V, scores=sc, norms=s64, explained_variance=s64 ** 2 / (mat.n - 1),
                        total_variance=st["total_variance"], U=U, stats=st)
    mat, st, first = sharded_fit_first(ctx, X_local, comm, n_modes, p_raw_total, center, standardize, feature_weights,
                                       check_nans, True, n_oversamples, omega, random_state)
    ops = HipPanelOps(ctx, mat)
    if omega is not None:     # an array or a SketchFuture (rows / size): its row count must match the compacted problem
        om_rows = omega.rows if hasattr(omega, "result") else omega.shape[0]
        if om_rows != min(mat.n, st["p_total"]):
            omega = None                        # samples / features dropped: the driver draws for the compacted shape
    U, s, V = sharded_rsvd(ops, comm, n_modes, st["p_total"], st["p_offset"], n_oversamples, n_iter,
                           random_state=random_state, omega=omega, device_out=device_out, first=first)
    s64 = np.asarray(s, dtype=np.float64)
    return dict(input_data=mat, components=V, scores=U * (s if not device_out else ops.e._torch().as_tensor(s, device=U.device)),
                norms=s64, explained_variance=s64 ** 2 / (mat.n - 1), total_variance=st["total_variance"],
                U=U, stats=st)


def sharded_mca_fit(ctx, X_local, Y_local, comm: Comm, n_modes: int, standardize=(False, False),
                    feature_weights=(None, None), check_nans=(True, True), random_state=None,
                    n_oversamples: int = 10, n_iter="auto", omega=None, want_tsc: bool = True, use_pca: bool = False,
                    n_pca_modes=0.999, pca_init_rank_reduction=0.3, native=None):
    """`MCA.fit` (cross/base_model_cross_set.py:269-321 + cross/cpcca.py:168-225) with both fields sharded
    along their own space axes.  With `use_pca` (the reference default) each field is first reduced by a
    feature-sharded `ResidentPCA` (all-reduced n x n Gram matrix); the analysis on the replicated PC scores
    needs no further communication and the components come back as this rank's rows of V Q."""
    nat = use_native(ctx, native) and not use_pca
    if nat:
        # the engine's own sharded cross-covariance entry (eofx_crosscov_rsvd_sharded_f32): both slices stay where they lie (the
        # in-place layout, a land / sea mask as zero columns), the global facts of the two preprocesses and every collective of
        # the decomposition go through the engine's communicator
        from . import engine

        mx, sx = sharded_preprocess(ctx, X_local, comm, True, standardize[0], feature_weights[0], check_nans[0], in_place=True,
                                    allow_masked=True, native=True)
        my, sy = sharded_preprocess(ctx, Y_local, comm, True, standardize[1], feature_weights[1], check_nans[1], in_place=True,
                                    allow_masked=True, native=True)
        out = engine.crosscov_rsvd_sharded(ctx, mx, my, n_modes, sx["p_total"], sx["p_offset"], sy["p_total"], sy["p_offset"],
                                           n_oversamples, n_iter, random_state=random_state, omega=omega, want_tsc=want_tsc)
        s = out["s"].astype(np.float64)
        return dict(input_data1=mx, input_data2=my, components1=out["Q1"], components2=out["Q2"],
                    scores1=out["scores1"], scores2=out["scores2"], singular_values=s, squared_covariance=s ** 2,
                    total_squared_covariance=out.get("total_squared_covariance") if want_tsc else None, norm1=out["norm1"],
                    norm2=out["norm2"], stats1=sx, stats2=sy, native=True)
    mx, sx = sharded_preprocess(ctx, X_local, comm, True, standardize[0], feature_weights[0], check_nans[0])
    my, sy = sharded_preprocess(ctx, Y_local, comm, True, standardize[1], feature_weights[1], check_nans[1])
    if use_pca:
        from . import engine
        from .pca import ResidentPCA

        pcas, work = [], []
        for mat, st in ((mx, sx), (my, sy)):
            pca = ResidentPCA(ctx, n_pca_modes, pca_init_rank_reduction).fit(mat, st["total_variance"], comm=comm,
                                                                                p_total=st["p_total"])
            pcas.append(pca)
            work.append(engine.from_dense(ctx, pca.scores().astype(np.float32)))
        k = int(n_modes)
        rank = min(work[0].p, work[1].p)
        if k > rank:
            raise ValueError(f"n_modes must be less than or equal to the rank of the dataset (rank = {rank}).")
        out = engine.crosscov_rsvd(ctx, work[0], work[1], k, min(n_oversamples, rank - k), n_iter,
                                   random_state=random_state, omega=omega, want_tsc=want_tsc)
        for w in work:
            w.free()
        s = out["s"].astype(np.float64)
        return dict(input_data1=mx, input_data2=my, components1=pcas[0].back_project(out["Q1"]),
                    components2=pcas[1].back_project(out["Q2"]), scores1=out["scores1"], scores2=out["scores2"],
                    singular_values=s, squared_covariance=s ** 2,
                    total_squared_covariance=out.get("total_squared_covariance"), norm1=out["norm1"],
                    norm2=out["norm2"], stats1=sx, stats2=sy, pca=pcas)
    out = sharded_crosscov_rsvd(HipPanelOps(ctx, mx), HipPanelOps(ctx, my), comm, n_modes, sx["p_total"],
                                sx["p_offset"], sy["p_total"], sy["p_offset"], n_oversamples, n_iter,
                                random_state=random_state, omega=omega, want_tsc=want_tsc)
    s = out["s"].astype(np.float64)
    return dict(input_data1=mx, input_data2=my, components1=out["Q1"], components2=out["Q2"],
                scores1=out["scores1"], scores2=out["scores2"], singular_values=s, squared_covariance=s ** 2,
                total_squared_covariance=out.get("total_squared_covariance"), norm1=out["norm1"], norm2=out["norm2"],
                stats1=sx, stats2=sy)


def sharded_hilbert_eof_fit(ctx, X_local, comm: Comm, n_modes: int, padding="exp", decay_factor: float = 0.2, standardize=False,
                            feature_weights=None, check_nans=True, random_state=None, n_oversamples: int = 10, n_iter="converge",
                            omega=None, operator=True, native=None):
    """`HilbertEOF.fit` (single/eof.py:449-560: centred field -> analytic signal along the samples, utils/hilbert_transform.py
    -> complex decomposition, linalg/decomposer.py:149-160) with the space axis sharded: X_local is this rank's (n, P_g) slice of
    the stacked raw field.  Returns the DataContainer entries as a dict; `components` holds this rank's rows (complex64),
    everything else is replicated.

    operator=True (BASELINE config 5 as its 8-GPU form): the OPERATOR route -- the Hilbert stage is one n x n matrix Hc along the
    samples, applied to the replicated sample-side panel; every pass streams the rank's REAL slice once and the imaginary part
    is never written (half the bytes of the two-part route, no second resident field).  With the engine's communicator attached
    (`attach_native`) that is ONE engine call per rank, `eofx_rsvd_hilbert_sharded_c64`, which issues its own collectives; without it
    the panel-level driver (`complex_svd.complex_rsvd` over `HilbertOperatorOps`) runs the same recurrence with torch.distributed
    all-reduces between engine calls.  operator=False: the two-part route (Im written per slice, `eofx_hilbert_f32`)."""
    from . import engine
    from .complex_svd import ComplexOps, HilbertOperatorOps, complex_rsvd

    nat = use_native(ctx, native)
    k = int(n_modes)
    n = X_local.shape[0]
    use_op = bool(operator) and n <= engine.HILBERT_OPERATOR_MAX_SAMPLES and k + int(n_oversamples) <= 64
    # the statistics pass leaves the slice in place (the raw field through the Scaler map) and, for the sum of squares of the
    # imaginary part, writes the transposed raw copy the transform kernel reads; a land / sea mask stays in place on the
    # engine's own route only (the panel-level operator driver works on compacted slices)
    mat, st = sharded_preprocess(ctx, X_local, comm, True, standardize, feature_weights, check_nans, in_place=True,
                                 allow_masked=nat and use_op, for_hilbert=True, native=nat)
    if mat.n != n:
        raise NotImplementedError("feature-sharded HilbertEOF with all-NaN samples (the slices would need a common compaction)")
    if not n < st["p_total"]:
        mat.free()
        raise ValueError("the feature-sharded complex decomposition needs more valid features over all ranks than samples")
    if omega is not None and hasattr(omega, "result"):
        omega = omega.result()
    if use_op:
        im2 = engine.hilbert_sumsq(ctx, mat, padding, decay_factor)
        tv = st["total_variance"] + float(_sum_host(comm, [im2], ctx if nat else None)[0]) / (n - 1)
        if nat:
            U, s, V = engine.rsvd_hilbert_sharded_c64(ctx, mat, k, st["p_total"], padding, decay_factor, n_oversamples, n_iter,
                                                      random_state, omega=omega)
            parts = (mat, None)
        else:
            Hop = engine.from_dense(ctx, engine.hilbert_operator(ctx, n, padding, decay_factor))
            try:
                U, s, V = complex_rsvd(ctx, mat, None, k, n_oversamples, n_iter, random_state, ops=HilbertOperatorOps(ctx, mat, Hop),
                                       comm=comm, p_total=st["p_total"], p_offset=st["p_offset"], omega=omega)
            finally:
                Hop.free()
            parts = (mat, None)
    else:
        B, _ = engine.hilbert(ctx, mat, padding, decay_factor)
        tv = st["total_variance"] + float(_sum_host(comm, [B.sumsq()], ctx if nat else None)[0]) / (n - 1)
        if nat:
            U, s, V = engine.rsvd_sharded_c64(ctx, mat, B, k, st["p_total"], n_oversamples, n_iter, random_state, omega=omega)
        else:
            U, s, V = complex_rsvd(ctx, mat, B, k, n_oversamples, n_iter, random_state, ops=ComplexOps(ctx, mat, B), comm=comm,
                                   p_total=st["p_total"], p_offset=st["p_offset"], omega=omega)
        parts = (mat, B)
    s64 = np.asarray(s, dtype=np.float64)
    return dict(input_data=parts, components=V, scores=U * s, norms=s64, explained_variance=s64 ** 2 / (n - 1),
                total_variance=tv, stats=st, native=bool(nat), operator=bool(use_op),
                products=engine.last_iterations(ctx) if (nat or comm is None or getattr(comm, "world", 1) == 1) else None)


def shard_bounds(p_total: int, world: int, rank: int):
    """Contiguous, balanced split of the stacked feature axis."""
    base, rem = divmod(p_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
