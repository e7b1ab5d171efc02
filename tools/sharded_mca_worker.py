"""Multi-rank functional check of the sharded EOF / MCA fits with the real HIP kernels.

Launched by torch.distributed.run; with `--same-gpu --backend gloo` all ranks share cuda:0 (RCCL needs
one GPU per rank).  Every rank builds the same synthetic pair of fields, keeps its slice of each space
axis, runs `sharded_mca_fit` / `sharded_eof_fit`; rank 0 compares with the single-GPU drivers on the whole
fields and prints one JSON line."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--nsamples", dest="n", type=int, default=700)
    ap.add_argument("--p1", type=int, default=5000)
    ap.add_argument("--p2", type=int, default=3600)
    ap.add_argument("--modes", type=int, default=8)
    ap.add_argument("--nan", action="store_true", help="mask some grid points / time steps with NaN")
    ap.add_argument("--pca", action="store_true", help="MCA with the PCA pre-reduction (reference default)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = 0 if a.same_gpu else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(dev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    from xeofs_amd import engine, sharded

    rng = np.random.default_rng(3)
    t = rng.standard_normal((a.n, 10)) * (6.0 * 0.8 ** np.arange(10))
    X = (t @ rng.standard_normal((10, a.p1)) + rng.standard_normal((a.n, a.p1)) + 3.0).astype(np.float32)
    Y = (t @ rng.standard_normal((10, a.p2)) + rng.standard_normal((a.n, a.p2)) - 1.0).astype(np.float32)
    if a.nan:
        X[:, rng.random(a.p1) < 0.2] = np.nan
        Y[:, rng.random(a.p2) < 0.3] = np.nan
        X[5], Y[5] = np.nan, np.nan          # one missing time step in both fields
    ctx = engine.Context(dev)
    comm = sharded.Comm()
    lo1, hi1 = sharded.shard_bounds(a.p1, world, rank)
    lo2, hi2 = sharded.shard_bounds(a.p2, world, rank)
    k, seed = a.modes, 5
    mca = sharded.sharded_mca_fit(ctx, X[:, lo1:hi1], Y[:, lo2:hi2], comm, k, random_state=seed, use_pca=a.pca,
                                  n_pca_modes=0.98)
    eof = sharded.sharded_eof_fit(ctx, X[:, lo1:hi1], comm, k, random_state=seed)

    def gather_rows(local):
        parts = [None] * world
        dist.all_gather_object(parts, np.asarray(local))
        return np.concatenate(parts, axis=0)

    Q1, Q2, V = gather_rows(mca["components1"]), gather_rows(mca["components2"]), gather_rows(eof["components"])
    res = None
    if rank == 0:
        mx, stx = engine.preprocess(ctx, X)
        my, sty = engine.preprocess(ctx, Y)
        if a.pca:
            from xeofs_amd.pca import ResidentPCA

            p1 = ResidentPCA(ctx, 0.98).fit(mx, stx["total_variance"])
            p2 = ResidentPCA(ctx, 0.98).fit(my, sty["total_variance"])
            w1, w2 = engine.from_dense(ctx, p1.scores().astype(np.float32)), engine.from_dense(ctx, p2.scores().astype(np.float32))
            ref = engine.crosscov_rsvd(ctx, w1, w2, k, min(10, min(w1.p, w2.p) - k), "auto", random_state=seed)
            ref["Q1"], ref["Q2"] = p1.back_project(ref["Q1"]), p2.back_project(ref["Q2"])
        else:
            ref = engine.crosscov_rsvd(ctx, mx, my, k, 10, "auto", random_state=seed)
        U1, s1, V1 = engine.rsvd(ctx, mx, k, 10, "auto", random_state=seed)
        rel = lambda x, y: float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).max() / np.abs(np.asarray(y, np.float64)).max())
        cosmin = lambda A, B: float(np.min(np.sum(A.astype(np.float64) * B.astype(np.float64), axis=0)))
        res = dict(world=world, n_valid=int(mx.n), p1=int(mx.p), p2=int(my.p),
                   mca_s=rel(mca["singular_values"], ref["s"]), mca_q1_cos=cosmin(Q1, ref["Q1"]), mca_q2_cos=cosmin(Q2, ref["Q2"]),
                   mca_scores1=rel(mca["scores1"], ref["scores1"]), mca_scores2=rel(mca["scores2"], ref["scores2"]),
                   mca_norm1=rel(mca["norm1"], ref["norm1"]),
                   mca_tsc=abs(mca["total_squared_covariance"] / ref["total_squared_covariance"] - 1.0),
                   eof_s=rel(eof["norms"], s1), eof_v_cos=cosmin(V, V1), eof_scores=rel(eof["scores"], U1 * s1),
                   eof_tv=abs(eof["total_variance"] / stx["total_variance"] - 1.0),
                   p_total=[int(mca["stats1"]["p_total"]), int(mca["stats2"]["p_total"])])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
