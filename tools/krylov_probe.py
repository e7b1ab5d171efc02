"""Complex branch (row R9): the block-Krylov engine rule against the exact complex SVD AND the reference's own solver
(scipy svds(lobpcg) through oracle.complex_svds) on bulk-heavy samples, next to the subspace iteration of rounds 1-4
(EOFX_C64_KRYLOV=0).  python tools/krylov_probe.py [n nlat nlon k] ..."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import eof_oracle as orc
from xeofs_amd import engine

np.set_printoptions(linewidth=220)
ctx = engine.Context(0)
cases = [(2000, 40, 80, 20)]
if len(sys.argv) > 4:
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)]
for n, nlat, nlon, k in cases:
    X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, "cuda:0", seed=51_000)
    x64 = X.cpu().numpy().astype(np.float64)
    z = orc.hilbert_transform(x64 - x64.mean(axis=0), padding="exp", decay_factor=0.2)
    if min(z.shape) * 4 < max(z.shape):
        g = z @ z.conj().T if z.shape[0] < z.shape[1] else z.conj().T @ z
        sz = np.sqrt(np.maximum(np.linalg.eigvalsh(g)[::-1], 0))
    else:
        sz = np.linalg.svd(z, compute_uv=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.time(); _, sl, _ = orc.complex_svds(z, k, random_state=5); tl = time.time() - t0
    el = np.abs(sl - sz[:k]) / sz[:k]
    print(f"== {n} x ({nlat}x{nlon}) k={k}: s1 {sz[0]:.2f} s_k {sz[k-1]:.2f} s_(k+11) {sz[k+10]:.2f}")
    print(f"lobpcg (reference solver, {tl:.1f} s)   max {el.max():.2e}", np.array2string(el, precision=1))
    for env, it in (("1", "auto"), ("0", "auto"), ("1", "converge"), ("0", "converge")):
        os.environ["EOFX_C64_KRYLOV"] = env
        A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
        B, _ = engine.hilbert(ctx, A, "exp", 0.2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        U, s, V = engine.rsvd_c64(ctx, A, B, k, random_state=5, n_iter=it)
        torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        its = engine.last_iterations(ctx)
        A.free(); B.free()
        e = np.abs(s.astype(np.float64) - sz[:k]) / sz[:k]
        orthu = np.abs(U.conj().T @ U - np.eye(k)).max(); orthv = np.abs(V.conj().T @ V - np.eye(k)).max()
        ok = bool(np.all(e <= np.maximum(1e-5, el)))
        print(f"{'krylov  ' if env == '1' else 'subspace'} n_iter={it:8s} products {its:2d} {dt:7.1f} ms  max {e.max():.2e}  <= max(1e-5, lobpcg) per mode: {ok}  orth {orthu:.1e} {orthv:.1e}",
              np.array2string(e, precision=1))
    del X
