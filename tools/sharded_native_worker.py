"""Multi-rank functional check of the ENGINE-OWNED sharded entries (round 6: the 8-GPU forms of BASELINE configs 3, 4 and 5)
with the real HIP kernels: every collective of a fit is issued by the engine through the communicator attached to its context
(`xeofs_amd.sharded.attach_native`: RCCL with backend nccl, a host callback over gloo otherwise) -- no torch.distributed call
between engine calls.

    eofx_fit_sharded_f32               EOF, optionally with a land / sea mask kept in place as zero columns (--mask)
    eofx_crosscov_rsvd_sharded_f32     MCA (use_pca=False), both fields sharded along their own feature axes
    eofx_rsvd_hilbert_sharded_c64      HilbertEOF, operator route (the imaginary part is never written)
    eofx_rsvd_sharded_c64              HilbertEOF, two-part route (Im written per slice)

Launched by torch.distributed.run; `--same-gpu --backend gloo`: all ranks share cuda:0.  Every rank builds the same synthetic
fields and keeps its slice; rank 0 compares with the single-GPU entries on the whole fields and prints one JSON line."""
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
    ap.add_argument("--mask", action="store_true", help="all-NaN grid points (a land / sea mask) in both fields")
    ap.add_argument("--lowrank", action="store_true", help="exactly rank-4 fields and more modes than that (null modes)")
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
    n, k, seed = a.n, a.modes, 5
    if a.lowrank:
        t = rng.standard_normal((n, 4)) * np.array([6.0, 4.0, 2.5, 1.5])
        X = (t @ rng.standard_normal((4, a.p1))).astype(np.float32)
        Y = (t @ rng.standard_normal((4, a.p2))).astype(np.float32)
    else:
        t = rng.standard_normal((n, 10)) * (6.0 * 0.8 ** np.arange(10))
        X = (t @ rng.standard_normal((10, a.p1)) + rng.standard_normal((n, a.p1)) + 3.0).astype(np.float32)
        Y = (t @ rng.standard_normal((10, a.p2)) + rng.standard_normal((n, a.p2)) - 1.0).astype(np.float32)
    # a propagating-wave field for the Hilbert model (as tools/sharded_complex_worker.py)
    tt = np.arange(n)[:, None]
    xx = np.linspace(0, 2 * np.pi, a.p1)[None, :]
    H = sum(amp * np.cos(w * tt - m * xx + ph) for amp, w, m, ph in
            ((3.0, 0.21, 2, 0.0), (1.7, 0.37, -3, 0.4), (0.9, 0.11, 1, 1.0), (0.5, 0.53, 5, 2.0)))
    H = (H + 0.3 * rng.standard_normal((n, a.p1)) + 0.002 * tt).astype(np.float32)
    if a.mask:
        X[:, rng.random(a.p1) < 0.2] = np.nan
        Y[:, rng.random(a.p2) < 0.3] = np.nan
        H[:, rng.random(a.p1) < 0.25] = np.nan
    ctx = engine.Context(dev)
    comm = sharded.Comm()
    attached = sharded.attach_native(ctx, comm)
    lo1, hi1 = sharded.shard_bounds(a.p1, world, rank)
    lo2, hi2 = sharded.shard_bounds(a.p2, world, rank)
    Xl, Yl, Hl = (np.ascontiguousarray(v) for v in (X[:, lo1:hi1], Y[:, lo2:hi2], H[:, lo1:hi1]))

    engine.comm_stats(ctx)
    eof = sharded.sharded_eof_fit(ctx, Xl, comm, k, random_state=seed, native=True)
    calls_eof = engine.comm_stats(ctx)["calls"]
    mca = sharded.sharded_mca_fit(ctx, Xl, Yl, comm, k, random_state=seed, native=True)
    calls_mca = engine.comm_stats(ctx)["calls"]
    hop = sharded.sharded_hilbert_eof_fit(ctx, Hl, comm, k, random_state=seed, operator=True, native=True)
    calls_hop = engine.comm_stats(ctx)["calls"]
    h2p = sharded.sharded_hilbert_eof_fit(ctx, Hl, comm, k, random_state=seed, operator=False, native=True)
    calls_h2p = engine.comm_stats(ctx)["calls"]
    # the panel-level form of the operator route (HilbertOperatorOps over the panel ABI, torch.distributed collectives between
    # engine calls): the fallback when no engine communicator is attached; on a land-masked field it works on compacted slices
    hpy = sharded.sharded_hilbert_eof_fit(ctx, Hl, comm, k, random_state=seed, operator=True, native=False)

    def gather_rows(local):
        parts = [None] * world
        dist.all_gather_object(parts, np.asarray(local))
        return np.concatenate(parts, axis=0)

    V, Q1, Q2 = gather_rows(eof["components"]), gather_rows(mca["components1"]), gather_rows(mca["components2"])
    Vh, Vh2, Vhp = gather_rows(hop["components"]), gather_rows(h2p["components"]), gather_rows(hpy["components"])
    res = None
    if rank == 0:
        engine.comm_clear(ctx)      # the references below are single-GPU entries on the whole fields
        rel = lambda x, y: float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).max() / np.abs(np.asarray(y, np.float64)).max())
        cosmin = lambda A, B: float(np.min(np.abs(np.sum(np.conj(A.astype(np.complex128)) * B.astype(np.complex128), axis=0))))
        orth = lambda A: float(np.abs(np.conj(A.astype(np.complex128)).T @ A.astype(np.complex128) - np.eye(A.shape[1])).max())
        matf, stf, U1, s1, V1 = engine.fit(ctx, X, k, random_state=seed, allow_masked=True)
        mx, stx = engine.preprocess(ctx, X, in_place=True, allow_masked=True)
        my, sty = engine.preprocess(ctx, Y, in_place=True, allow_masked=True)
        try:
            ref = engine.crosscov_rsvd(ctx, mx, my, k, 10, "auto", random_state=seed)
        except NotImplementedError:      # (a single-GPU limit of masked in-place pairs, engine.crosscov_rsvd: compacted matrices instead)
            mx.free(); my.free()
            mx, stx = engine.preprocess(ctx, X)
            my, sty = engine.preprocess(ctx, Y)
            ref = engine.crosscov_rsvd(ctx, mx, my, k, 10, "auto", random_state=seed)
        mh, sth = engine.preprocess(ctx, H, in_place=True, allow_masked=True, for_hilbert=True)
        tv_h = sth["total_variance"] + engine.hilbert_sumsq(ctx, mh, "exp", 0.2) / (n - 1)
        Uh, sh, Vhr = engine.rsvd_hilbert_c64(ctx, mh, k, "exp", 0.2, random_state=seed, n_iter="converge")   # the sharded drivers' default rule
        nres = 4 if a.lowrank else k          # modes that carry a value (directions of null modes are arbitrary)
        res = dict(world=world, attached=bool(attached), mask=bool(a.mask), lowrank=bool(a.lowrank),
                   native=[bool(eof["stats"].get("native")), bool(mca.get("native")), bool(hop["native"]), bool(h2p["native"])],
                   calls=[calls_eof, calls_mca, calls_hop, calls_h2p],
                   p_valid=[int(matf.p), int(mx.p), int(my.p), int(mh.p)],
                   p_total=[int(eof["stats"]["p_total"]), int(mca["stats1"]["p_total"]), int(mca["stats2"]["p_total"]), int(hop["stats"]["p_total"])],
                   eof_s=rel(eof["norms"][:nres], s1[:nres]), eof_v_cos=cosmin(V[:, :nres], V1[:, :nres]),
                   eof_scores=rel(eof["scores"][:, :nres], (U1 * s1)[:, :nres]),
                   eof_tv=abs(eof["total_variance"] / stf["total_variance"] - 1.0), eof_orth_v=orth(V), eof_orth_u=orth(eof["U"]),
                   mca_s=rel(mca["singular_values"][:nres], ref["s"][:nres]), mca_q1_cos=cosmin(Q1[:, :nres], ref["Q1"][:, :nres]),
                   mca_q2_cos=cosmin(Q2[:, :nres], ref["Q2"][:, :nres]),
                   mca_scores1=rel(mca["scores1"][:, :nres], ref["scores1"][:, :nres]),
                   mca_scores2=rel(mca["scores2"][:, :nres], ref["scores2"][:, :nres]),
                   mca_norm1=rel(mca["norm1"][:nres], ref["norm1"][:nres]),
                   mca_tsc=abs(mca["total_squared_covariance"] / ref["total_squared_covariance"] - 1.0),
                   mca_orth_q1=orth(Q1), mca_orth_q2=orth(Q2),
                   hop_s=rel(hop["norms"], sh), hop_v_cos=cosmin(Vh[:, :4], Vhr[:, :4]), hop_tv=abs(hop["total_variance"] / tv_h - 1.0),
                   hop_orth_v=orth(Vh), hop_orth_u=orth(hop["scores"] / hop["norms"]),
                   hpy_s=rel(hpy["norms"], sh), hpy_v_cos=cosmin(Vhp[:, :4], Vhr[:, :4]), hpy_tv=abs(hpy["total_variance"] / tv_h - 1.0),
                   hpy_orth_v=orth(Vhp), hpy_native=bool(hpy["native"]), hpy_operator=bool(hpy["operator"]),
                   h2p_s=rel(h2p["norms"], sh), h2p_v_cos=cosmin(Vh2[:, :4], Vhr[:, :4]), h2p_tv=abs(h2p["total_variance"] / tv_h - 1.0),
                   h2p_orth_v=orth(Vh2))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
