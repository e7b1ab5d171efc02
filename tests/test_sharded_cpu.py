"""world_size-2 gloo test of the feature-sharded randomized SVD (xeofs_amd/sharded.py) on CPU.

The orchestration (which panel is replicated / sharded, where the all-reduces go, the sign rule's
global max/min, the omega slicing) is the product code; the panel arithmetic is supplied here by a
numpy stand-in with the same interface as HipPanelOps (the HIP kernels need a GPU and are covered
by the -m gpu tests).  The result must match the single-matrix oracle.
"""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NumpyPanelOps:
    """Stand-in for HipPanelOps: same steps, float32 panels as CPU torch tensors."""

    def __init__(self, X):
        import torch

        self.torch = torch
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.n, self.p = X.shape
        self.n_pad = (self.n + 511) // 512 * 512
        self.p_pad = (self.p + 511) // 512 * 512

    def _t(self, a, dt=np.float32):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dt))

    def import_panel(self, src, side):
        rows_pad = self.n_pad if side == "n" else self.p_pad
        L = (src.shape[1] + 31) // 32 * 32
        P = np.zeros((rows_pad, L), np.float32)
        P[:src.shape[0], :src.shape[1]] = src
        return self._t(P)

    def tmul(self, Zn, final=False):
        out = np.zeros((self.p_pad, Zn.shape[1]), np.float32)
        out[:self.p] = self.X.T @ Zn.numpy()[:self.n]
        return self._t(out)

    def mul(self, Yp, final=False):
        out = np.zeros((self.n_pad, Yp.shape[1]), np.float32)
        out[:self.n] = self.X @ Yp.numpy()[:self.p]
        return self._t(out)

    def gram(self, P):
        a = P.numpy().astype(np.float64)
        return self._t(a.T @ a, np.float64)

    def cholqr(self, P, l, G):
        g = G.numpy()[:l, :l]
        R = np.linalg.cholesky(g).T
        out = np.zeros_like(P.numpy())
        out[:, :l] = (P.numpy()[:, :l].astype(np.float64) @ np.linalg.inv(R)).astype(np.float32)
        return self._t(out)

    def rinv(self, G, l):
        out = np.zeros_like(G.numpy())
        out[:l, :l] = np.linalg.inv(np.linalg.cholesky(G.numpy()[:l, :l]).T)
        return self._t(out, np.float64)

    def matmul(self, P, M):
        M = M.numpy() if hasattr(M, "numpy") else M
        return self._t((P.numpy().astype(np.float64) @ M).astype(np.float32))

    def colminmax(self, P, rows):
        a = P.numpy()[:rows]
        if rows == 0:
            L = P.shape[1]
            return self._t(np.full(L, -np.inf)), self._t(np.full(L, np.inf))
        return self._t(a.max(axis=0)), self._t(a.min(axis=0))

    def export(self, P, rows, k, sign=None):
        out = P.numpy()[:rows, :k].copy()
        if sign is not None:
            out *= np.asarray(sign, dtype=np.float32)
        return out

    def eigh(self, G):
        w, V = np.linalg.eigh(G)
        return w[::-1], V[:, ::-1]

    def sample_gram(self):
        G = np.zeros((self.n_pad, self.n_pad), np.float32)
        G[:self.n, :self.n] = self.X.astype(np.float64) @ self.X.T.astype(np.float64)
        return self._t(G)

    def dot(self, a, b):
        return float(np.dot(a.numpy().ravel().astype(np.float64), b.numpy().ravel().astype(np.float64)))


def _field(n, p, seed):
    rng = np.random.default_rng(seed)
    amp = 6.0 * 0.75 ** np.arange(8)
    X = (rng.standard_normal((n, 8)) * amp) @ rng.standard_normal((8, p)) + rng.standard_normal((n, p))
    return (X - X.mean(axis=0)).astype(np.float32)


def _worker(rank, world, port, n, p, k, seed, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded

    X = _field(n, p, 5)
    lo, hi = sharded.shard_bounds(p, world, rank)
    ops = NumpyPanelOps(X[:, lo:hi])
    comm = sharded.Comm()
    U, s, V = sharded.sharded_rsvd(ops, comm, k, p, lo, random_state=seed)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), U=U, s=s, V=V, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("n,p,k", [(120, 700, 6), (600, 90, 5)])
def test_sharded_rsvd_two_ranks_gloo(tmp_path, n, p, k):
    import torch.multiprocessing as mp

    from oracle import eof_oracle as orc

    world, seed = 2, 11
    mp.spawn(_worker, args=(world, _free_port(), n, p, k, seed, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # replicated outputs agree bitwise across ranks
    assert np.array_equal(parts[0]["U"], parts[1]["U"]) and np.array_equal(parts[0]["s"], parts[1]["s"])
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == p
    V = np.concatenate([q["V"] for q in parts], axis=0)
    U, s = parts[0]["U"], parts[0]["s"]
    X = _field(n, p, 5).astype(np.float64)
    Uo, so, Vo = orc.decomposer_fit(X, k, random_state=seed, solver="randomized")
    assert np.all(np.abs(s - so) <= 2e-5 * so + 2e-6 * so[0])
    for j in range(k):
        assert np.dot(V[:, j].astype(np.float64), Vo[:, j]) >= 1 - 1e-4, j   # same sign convention too
        assert np.dot(U[:, j].astype(np.float64), Uo[:, j]) >= 1 - 1e-4, j


def test_sharded_world1_matches_two_ranks(tmp_path):
    """The split must not change the answer beyond float32 rounding."""
    import torch.multiprocessing as mp

    n, p, k, seed = 100, 520, 4, 3
    mp.spawn(_worker, args=(2, _free_port(), n, p, k, seed, str(tmp_path)), nprocs=2, join=True)
    two = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    from xeofs_amd import sharded

    ops = NumpyPanelOps(_field(n, p, 5))

    class NoComm:
        def sum_(self, t): return t
        def max_(self, t): return t
        def min_(self, t): return t

    U, s, V = sharded.sharded_rsvd(ops, NoComm(), k, p, 0, random_state=seed)
    assert np.allclose(s, two[0]["s"], rtol=2e-5)
    V2 = np.concatenate([q["V"] for q in two], axis=0)
    for j in range(k):
        assert np.dot(V[:, j], V2[:, j]) >= 1 - 1e-4


# --------------------------------------------------------------------------- complex path, 2 ranks
class NumpyComplexOps:
    """numpy stand-in for complex_svd.ComplexOps ([Re|Im] float32 panels as CPU torch tensors)."""

    def __init__(self, Z):
        import torch

        self.torch = torch
        self.Z = np.asarray(Z, dtype=np.complex128)
        self.n, self.p = Z.shape
        self.n_pad = (self.n + 511) // 512 * 512
        self.p_pad = (self.p + 511) // 512 * 512

    def _c(self, P, rows):
        a = P.numpy()[:rows].astype(np.float64)
        return a[:, :32] + 1j * a[:, 32:]

    def _p(self, C, rows_pad):
        out = np.zeros((rows_pad, 64), np.float32)
        out[:C.shape[0], :32] = C.real
        out[:C.shape[0], 32:] = C.imag
        return self.torch.from_numpy(out)

    def import_panel(self, host, side):
        out = np.zeros((self.n_pad if side == "n" else self.p_pad, 64), np.float32)
        out[:host.shape[0]] = host
        return self.torch.from_numpy(out)

    def zh_mul(self, Wn, final=False):
        return self._p(self.Z.conj().T @ self._c(Wn, self.n), self.p_pad)

    def z_mul(self, Yp, final=False):
        return self._p(self.Z @ self._c(Yp, self.p), self.n_pad)

    def gram_real(self, P):
        a = P.numpy().astype(np.float64)
        return self.torch.from_numpy(a.T @ a)

    def right_mul(self, P, M):
        from xeofs_amd.complex_svd import _embed_right

        return self.torch.from_numpy((P.numpy().astype(np.float64) @ _embed_right(M)).astype(np.float32))

    def matmul_real(self, P, E):
        return self.torch.from_numpy((P.numpy().astype(np.float64) @ np.asarray(E, dtype=np.float64)).astype(np.float32))

    def argminmax(self, P, rows):
        a = P.numpy()[:rows]
        return self.torch.from_numpy(a.argmax(axis=0)), self.torch.from_numpy(a.argmin(axis=0))

    def export(self, P, rows, sign):
        return P.numpy()[:rows] * np.asarray(sign, dtype=np.float32)


def _cfield(n, p, seed):
    rng = np.random.default_rng(seed)
    amp = 6.0 * 0.6 ** np.arange(5)
    L = (rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5))) * amp
    R = rng.standard_normal((5, p)) + 1j * rng.standard_normal((5, p))
    Z = L @ R + 0.3 * (rng.standard_normal((n, p)) + 1j * rng.standard_normal((n, p)))
    return Z - Z.mean(axis=0)


def _cworker(rank, world, port, n, p, k, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded
    from xeofs_amd.complex_svd import complex_rsvd

    Z = _cfield(n, p, 9)
    lo, hi = sharded.shard_bounds(p, world, rank)
    U, s, V = complex_rsvd(None, None, None, k, random_state=4, ops=NumpyComplexOps(Z[:, lo:hi]),
                           comm=sharded.Comm(), p_total=p, p_offset=lo)
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), U=U, s=s, V=V)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,p,k", [(90, 400, 4), (420, 80, 3), (90, 400, 12), (420, 300, 14)])
def test_sharded_complex_rsvd_two_ranks_gloo(tmp_path, n, p, k):
    """(k = 12, 14: most wanted modes are noise modes -- the block Lanczos recurrence of round 5 resolves them, and on the
    90-sample field its Krylov space exhausts the sample space after two products)"""
    import torch.multiprocessing as mp

    from oracle import eof_oracle as orc

    mp.spawn(_cworker, args=(2, _free_port(), n, p, k, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"c{r}.npz") for r in range(2)]
    assert np.array_equal(parts[0]["U"], parts[1]["U"]) and np.array_equal(parts[0]["s"], parts[1]["s"])
    V = np.concatenate([q["V"] for q in parts], axis=0)
    U, s = parts[0]["U"], parts[0]["s"]
    Z = _cfield(n, p, 9)
    Ue, se, Vhe = np.linalg.svd(Z, full_matrices=False)
    assert np.all(np.abs(s - se[:k]) <= 2e-5 * se[:k] + 2e-6 * se[0])
    for j in range(k):
        if min(se[j - 1] - se[j] if j else np.inf, se[j] - se[j + 1]) < 2e-2 * se[j]:
            continue                       # (noise modes without a gap: the pair is defined up to a rotation)
        assert abs(np.vdot(Vhe[j].conj(), V[:, j])) >= 1 - 1e-4
        assert abs(np.vdot(Ue[:, j], U[:, j])) >= 1 - 1e-4
    # global (cross-rank) sign rule of the reference: already satisfied by the assembled V
    assert (orc.deterministic_sign_multiplier(V.conj().T) == 1).all()


class NumpyHilbertOperatorOps(NumpyComplexOps):
    """numpy stand-in for complex_svd.HilbertOperatorOps: this rank's REAL slice A and the n x n operator Hc of the Hilbert stage;
    Z^H W = A^T (W - i Hc^T W), Z Y = (I + i Hc)(A Y) -- the imaginary part is never formed, and (as in the product ops) the
    operator is applied to the rank's PARTIAL sum A_g Y_g before the driver's all-reduce (linearity)."""

    def __init__(self, A, Hc):
        NumpyComplexOps.__init__(self, A.astype(np.complex128))
        self.A = np.asarray(A, dtype=np.float64)
        self.Hc = np.asarray(Hc, dtype=np.float64)

    def zh_mul(self, Wn, final=False):
        W = self._c(Wn, self.n)
        return self._p(self.A.T @ (W - 1j * (self.Hc.T @ W)), self.p_pad)

    def z_mul(self, Yp, final=False):
        T = self.A @ self._c(Yp, self.p)
        return self._p(T + 1j * (self.Hc @ T), self.n_pad)


def _hilbert_case(n, p, seed):
    """a centred real field with a few propagating patterns and the operator of the oracle's Hilbert stage (exp padding): the
    stage is linear along the samples, so its matrix is the transform of the identity"""
    from oracle import eof_oracle as orc

    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    x = np.linspace(0, 2 * np.pi, p)[None, :]
    A = sum(a * np.cos(2 * np.pi * t / per - m * x + ph) for a, per, m, ph in
            [(5.0, 17.0, 1, 0.3), (3.0, 9.0, 2, 1.1), (2.0, 29.0, 3, 2.0), (1.2, 6.0, 4, 0.7)])
    A = A + 0.2 * rng.standard_normal((n, p))
    A = A - A.mean(axis=0)
    Hc = orc.hilbert_transform(np.eye(n), "exp", 0.2).imag      # column j = the (re-centred, eof.py:546-555) transform of e_j
    return A, Hc


def _hworker(rank, world, port, n, p, k, out_dir, rule="auto"):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded
    from xeofs_amd.complex_svd import complex_rsvd

    A, Hc = _hilbert_case(n, p, 13)
    lo, hi = sharded.shard_bounds(p, world, rank)
    U, s, V = complex_rsvd(None, None, None, k, random_state=4, ops=NumpyHilbertOperatorOps(A[:, lo:hi], Hc),
                           comm=sharded.Comm(), p_total=p, p_offset=lo, n_iter=rule)
    np.savez(os.path.join(out_dir, f"h{rank}.npz"), U=U, s=s, V=V)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,p,k,rule", [(96, 500, 5, "auto"), (120, 333, 8, "auto"), (120, 333, 12, "converge")])
def test_sharded_hilbert_operator_route_two_ranks_gloo(tmp_path, n, p, k, rule):
    """The operator route of the analytic signal, feature-sharded (the panel-level form of eofx_rsvd_hilbert_sharded_c64): every
    rank holds its REAL slice and the n x n Hilbert operator; result = exact SVD of the oracle's analytic signal of the whole
    field (single/eof.py:546-555 -> utils/hilbert_transform.py -> decomposer.py:149-160)."""
    import torch.multiprocessing as mp

    from oracle import eof_oracle as orc

    mp.spawn(_hworker, args=(2, _free_port(), n, p, k, str(tmp_path), rule), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"h{r}.npz") for r in range(2)]
    assert np.array_equal(parts[0]["U"], parts[1]["U"]) and np.array_equal(parts[0]["s"], parts[1]["s"])
    V = np.concatenate([q["V"] for q in parts], axis=0)
    U, s = parts[0]["U"], parts[0]["s"]
    A, _ = _hilbert_case(n, p, 13)
    Z = orc.hilbert_transform(A, "exp", 0.2)
    Z = Z - Z.mean(axis=0)
    Ue, se, Vhe = np.linalg.svd(Z, full_matrices=False)
    assert np.all(np.abs(s - se[:k]) <= 2e-5 * se[:k] + 2e-6 * se[0])
    for j in range(k):
        if min(se[j - 1] - se[j] if j else np.inf, se[j] - se[j + 1]) < 2e-2 * se[j]:
            continue
        assert abs(np.vdot(Vhe[j].conj(), V[:, j])) >= 1 - 1e-4
        assert abs(np.vdot(Ue[:, j], U[:, j])) >= 1 - 1e-4
    assert (orc.deterministic_sign_multiplier(V.conj().T) == 1).all()


def _cnull_worker(rank, world, port, n, p, r, k, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded
    from xeofs_amd.complex_svd import complex_rsvd

    rng = np.random.default_rng(21)
    Z = (rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r))) @ (rng.standard_normal((r, p)) + 1j * rng.standard_normal((r, p)))
    lo, hi = sharded.shard_bounds(p, world, rank)
    U, s, V = complex_rsvd(None, None, None, k, random_state=4, ops=NumpyComplexOps(Z[:, lo:hi]),
                           comm=sharded.Comm(), p_total=p, p_offset=lo)
    np.savez(os.path.join(out_dir, f"n{rank}.npz"), U=U, s=s, V=V)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_complex_more_modes_than_rank_two_ranks_gloo(tmp_path):
    """k = 8 modes of an exactly rank-3 complex field: the reference's solver (scipy svds ends in a dense SVD of A V) returns
    orthonormal left vectors whatever the values; the panel-level complex driver re-orthonormalises the null columns of its
    replicated sample-side factor on the host (round 6; the engine entry has done so since round 5)."""
    import torch.multiprocessing as mp

    n, p, r, k = 80, 300, 3, 8
    mp.spawn(_cnull_worker, args=(2, _free_port(), n, p, r, k, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"n{q}.npz") for q in range(2)]
    assert np.array_equal(parts[0]["U"], parts[1]["U"])
    U, s = parts[0]["U"].astype(np.complex128), parts[0]["s"]
    assert np.all(s[r:] <= 1e-4 * s[0]) and np.all(s[:r] > 1e-2 * s[0])
    assert np.abs(U.conj().T @ U - np.eye(k)).max() <= 1e-5


# --------------------------------------------------------------------------- cross-covariance path, 2 ranks
def _xy(n, p1, p2):
    rng = np.random.default_rng(9)
    t = rng.standard_normal((n, 6)) * (5.0 * 0.7 ** np.arange(6))
    X = t @ rng.standard_normal((6, p1)) + 0.5 * rng.standard_normal((n, p1))
    Y = t @ rng.standard_normal((6, p2)) + 0.5 * rng.standard_normal((n, p2))
    return (X - X.mean(0)).astype(np.float32), (Y - Y.mean(0)).astype(np.float32)


def _cross_worker(rank, world, port, n, p1, p2, k, seed, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded

    X, Y = _xy(n, p1, p2)
    lo1, hi1 = sharded.shard_bounds(p1, world, rank)
    lo2, hi2 = sharded.shard_bounds(p2, world, rank)
    out = sharded.sharded_crosscov_rsvd(NumpyPanelOps(X[:, lo1:hi1]), NumpyPanelOps(Y[:, lo2:hi2]), sharded.Comm(),
                                        k, p1, lo1, p2, lo2, random_state=seed)
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,p1,p2,k", [(80, 300, 420, 4), (80, 420, 300, 4)])
def test_sharded_crosscov_two_ranks_gloo(tmp_path, n, p1, p2, k):
    """X and Y sharded on their own feature axes; result = the oracle's MCA."""
    import torch.multiprocessing as mp

    from oracle import eof_oracle as orc

    seed = 4
    mp.spawn(_cross_worker, args=(2, _free_port(), n, p1, p2, k, seed, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"c{r}.npz") for r in range(2)]
    for key in ("s", "scores1", "scores2", "norm1", "norm2", "total_squared_covariance"):
        assert np.array_equal(parts[0][key], parts[1][key]), key        # replicated bitwise
    Q1 = np.concatenate([q["Q1"] for q in parts], axis=0)
    Q2 = np.concatenate([q["Q2"] for q in parts], axis=0)
    X, Y = _xy(n, p1, p2)
    ref = orc.mca_fit(X.astype(np.float64), Y.astype(np.float64), k, random_state=seed, solver="randomized")
    so = ref["singular_values"]
    assert np.all(np.abs(parts[0]["s"] - so) <= 5e-5 * so[0])
    tsc = float(parts[0]["total_squared_covariance"])
    assert abs(tsc - ref["total_squared_covariance"]) <= 1e-4 * ref["total_squared_covariance"]
    for j in range(k):
        assert np.dot(Q1[:, j].astype(np.float64), ref["components1"][:, j]) >= 1 - 1e-4, j
        assert np.dot(Q2[:, j].astype(np.float64), ref["components2"][:, j]) >= 1 - 1e-4, j
    assert np.allclose(parts[0]["scores1"], ref["scores1"], atol=2e-3 * np.abs(ref["scores1"]).max())
    assert np.allclose(parts[0]["scores2"], ref["scores2"], atol=2e-3 * np.abs(ref["scores2"]).max())
    assert np.allclose(parts[0]["norm1"], ref["norm1"], rtol=1e-3)
    assert np.allclose(parts[0]["norm2"], ref["norm2"], rtol=1e-3)


def _mask_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded

    comm = sharded.Comm()
    res = {}
    base = np.array([True, True, False, True])
    res["same"] = sharded.combine_sample_masks(comm, base, 5)
    # rank 1 owns only all-NaN features: it does not vote
    res["empty_shard"] = sharded.combine_sample_masks(comm, base if rank == 0 else np.zeros(4, bool),
                                                      5 if rank == 0 else 0)
    other = base.copy()
    if rank == 1:
        other[1] = False        # sample 1 is all-NaN in rank 1's features only -> isolated NaNs globally
    try:
        sharded.combine_sample_masks(comm, other, 5)
        res["partial"] = "no error"
    except ValueError as e:
        res["partial"] = str(e)
    res["partial_unchecked"] = sharded.combine_sample_masks(comm, other, 5, check_nans=False)
    # the same facts through the single-collective form the sharded fit uses
    counts, vs, tv, bad = sharded.global_facts(comm, 5 if rank == 0 else 7, base, True, 1.5 + rank, float(rank == 1))
    res["gf_counts"], res["gf_vs"], res["gf_tv"], res["gf_bad"] = counts, vs, tv, bad
    counts, vs, _, _ = sharded.global_facts(comm, 5 if rank == 0 else 0, base if rank == 0 else np.zeros(4, bool), True, 0.0)
    res["gf_empty_counts"], res["gf_empty_vs"] = counts, vs
    try:
        sharded.global_facts(comm, 5, other, True, 0.0)
        res["gf_partial"] = "no error"
    except ValueError as e:
        res["gf_partial"] = str(e)
    res["gf_partial_unchecked"] = sharded.global_facts(comm, 5, other, False, 0.0)[1]
    np.savez(os.path.join(out_dir, f"m{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_combine_sample_masks_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_mask_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        m = np.load(tmp_path / f"m{r}.npz")
        assert m["same"].tolist() == [True, True, False, True]
        assert m["empty_shard"].tolist() == [True, True, False, True]
        assert "partial NaN entries" in str(m["partial"])
        assert m["partial_unchecked"].tolist() == [True, True, False, True]
        assert m["gf_counts"].tolist() == [5, 7] and m["gf_vs"].tolist() == [True, True, False, True]
        assert float(m["gf_tv"]) == 4.0 and float(m["gf_bad"]) == 1.0
        assert m["gf_empty_counts"].tolist() == [5, 0] and m["gf_empty_vs"].tolist() == [True, True, False, True]
        assert "partial NaN entries" in str(m["gf_partial"])
        assert m["gf_partial_unchecked"].tolist() == [True, True, False, True]


class _DeadColumnOps(NumpyPanelOps):
    """NumpyPanelOps whose Cholesky steps behave like the device kernels on a rank-deficient panel: a column left with nothing
    after the ones before it is dropped (zero column of Q, zero row / column of R^-1) instead of raising."""

    @staticmethod
    def _rinv(g, l):
        A = np.array(g[:l, :l], dtype=np.float64)
        d0 = np.diag(A).copy()
        R = np.zeros((l, l))
        dead = np.zeros(l, bool)
        for j in range(l):
            d = A[j, j]
            if not (d > 1e-13 * d0[j]) or not d0[j] > 0:
                dead[j] = True
                A[j, :] = 0.0
                A[:, j] = 0.0
                continue
            R[j, j] = np.sqrt(d)
            R[j, j + 1:] = A[j, j + 1:] / R[j, j]
            A[j + 1:, j + 1:] -= np.outer(R[j, j + 1:], R[j, j + 1:])
        Ri = np.zeros((l, l))
        live = ~dead
        if live.any():
            Ri[np.ix_(live, live)] = np.linalg.inv(R[np.ix_(live, live)])
        return Ri

    def cholqr(self, P, l, G):
        out = np.zeros_like(P.numpy())
        out[:, :l] = (P.numpy()[:, :l].astype(np.float64) @ self._rinv(G.numpy(), l)).astype(np.float32)
        return self._t(out)

    def rinv(self, G, l):
        out = np.zeros_like(G.numpy())
        out[:l, :l] = self._rinv(G.numpy(), l)
        return self._t(out, np.float64)


def _lowrank_worker(rank, world, port, n, p, r, k, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xeofs_amd import sharded

    rng = np.random.default_rng(7)
    X = ((rng.standard_normal((n, r)) * 2.0 ** -np.arange(r)) @ rng.standard_normal((r, p))).astype(np.float32)
    X = (X - X.mean(axis=0)).astype(np.float32)
    lo, hi = sharded.shard_bounds(p, world, rank)
    U, s, V = sharded.sharded_rsvd(_DeadColumnOps(X[:, lo:hi]), sharded.Comm(), k, p, lo, random_state=3)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), U=U, s=s, V=V, X=X)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,p,r,k", [(120, 700, 4, 9), (600, 90, 3, 7)])
def test_sharded_rsvd_more_modes_than_rank_two_ranks_gloo(tmp_path, n, p, r, k):
    """the panel-level (feature-sharded) driver on exactly low-rank data with k > rank: both factors orthonormal over ALL shards
    (scikit-learn returns orthonormal factors whatever the values; `_fix_null_columns` through the all-reduced Gram matrices)"""
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_lowrank_worker, args=(world, _free_port(), n, p, r, k, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{q}.npz") for q in range(world)]
    assert np.array_equal(parts[0]["U"], parts[1]["U"]) and np.array_equal(parts[0]["s"], parts[1]["s"])
    U, s = parts[0]["U"].astype(np.float64), parts[0]["s"].astype(np.float64)
    V = np.concatenate([q["V"] for q in parts], axis=0).astype(np.float64)
    assert np.isfinite(U).all() and np.isfinite(V).all()
    assert np.abs(U.T @ U - np.eye(k)).max() < 1e-5 and np.abs(V.T @ V - np.eye(k)).max() < 1e-5
    se = np.linalg.svd(parts[0]["X"].astype(np.float64), compute_uv=False)[:k]
    assert np.all(np.abs(s[:r] - se[:r]) <= 2e-5 * se[0]) and np.all(s[r:] <= 1e-5 * se[0])
    R = parts[0]["X"].astype(np.float64) @ V[:, :r] - U[:, :r] * s[:r]
    assert np.abs(R).max() <= 1e-4 * se[0]
