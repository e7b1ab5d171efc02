"""Multi-rank functional check of the Hilbert / complex path (BASELINE config 5 is a multi-GPU config) with the real HIP
kernels: every rank keeps its slice of the feature axis, preprocesses it, runs the Hilbert stage on its own features
(per-feature operation, no communication) and the feature-sharded complex randomized SVD
(`xeofs_amd.complex_svd.complex_rsvd` with a communicator: all-reduce of the sample-side panel and of the small Gram
matrices).  Rank 0 compares with the single-rank engine entry (`eofx_rsvd_c64`) on the whole field and prints one JSON line.
Launched by torch.distributed.run; `--same-gpu --backend gloo`: all ranks share cuda:0."""
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
    ap.add_argument("--nsamples", dest="n", type=int, default=900)
    ap.add_argument("--p", type=int, default=7001)
    ap.add_argument("--modes", type=int, default=12)
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
    from xeofs_amd.complex_svd import complex_rsvd

    rng = np.random.default_rng(11)
    n, p, k = a.n, a.p, a.modes
    t = np.arange(n)[:, None]
    x = np.linspace(0, 2 * np.pi, p)[None, :]
    X = sum(amp * np.cos(w * t - m * x + ph) for amp, w, m, ph in
            ((3.0, 0.21, 2, 0.0), (1.7, 0.37, -3, 0.4), (0.9, 0.11, 1, 1.0), (0.5, 0.53, 5, 2.0)))
    X = (X + 0.3 * rng.standard_normal((n, p)) + 0.002 * t).astype(np.float32)
    ctx = engine.Context(dev)
    comm = sharded.Comm()
    lo, hi = sharded.shard_bounds(p, world, rank)
    A, _ = engine.preprocess(ctx, np.ascontiguousarray(X[:, lo:hi]))
    B, _ = engine.hilbert(ctx, A, "exp", 0.2)
    U, s, V = complex_rsvd(ctx, A, B, k, random_state=5, comm=comm, p_total=p, p_offset=lo)
    # global identity Z V = U s: partial products over this rank's features, summed over the ranks
    Vr, Vi = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
    part = (engine.project(ctx, A, Vr) - engine.project(ctx, B, Vi)) + 1j * (engine.project(ctx, A, Vi) + engine.project(ctx, B, Vr))
    parts = [None] * world
    dist.all_gather_object(parts, part)
    Vs = [None] * world
    dist.all_gather_object(Vs, V)
    res = None
    if rank == 0:
        ZV = sum(parts)
        Vall = np.concatenate(Vs, axis=0)
        Af, _ = engine.preprocess(ctx, X)
        Bf, _ = engine.hilbert(ctx, Af, "exp", 0.2)
        U1, s1, V1 = engine.rsvd_c64(ctx, Af, Bf, k, random_state=5)
        Us = U.astype(np.complex128) * s.astype(np.float64)
        cos = np.abs(np.sum(Vall.conj().astype(np.complex128) * V1.astype(np.complex128), axis=0))
        res = dict(world=world, n=n, p=p, s_rel=float(np.abs(s - s1).max() / s1[0]),
                   zv_us=float(np.linalg.norm(ZV - Us) / np.linalg.norm(Us)), v_cos_min=float(cos.min()),
                   orth_v=float(np.abs(Vall.conj().T @ Vall - np.eye(k)).max()),
                   orth_u=float(np.abs(U.conj().T @ U - np.eye(k)).max()))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
