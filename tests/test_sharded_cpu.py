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

    def matmul(self, P, M):
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
