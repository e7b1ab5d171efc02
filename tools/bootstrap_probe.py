"""Time one bootstrap member (validation/bootstrapper.py loop body) at config 2 / config 4 size on the resident
matrix: row-gather resample + re-centre, randomized SVD, projection of the original matrix."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench

n, nlat, nlon, k = int(os.environ.get("N", 5000)), int(os.environ.get("NLAT", 360)), int(os.environ.get("NLON", 720)), 50
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
mat, st = engine.preprocess(ctx, X, want_stats=False)
del X
rng = np.random.default_rng(0)
sync = torch.cuda.synchronize
for rep in range(3):
    idx = rng.choice(n, n, replace=True)
    sync(); t0 = time.perf_counter()
    bm, mean, tv = engine.resample(ctx, mat, idx)
    sync(); t1 = time.perf_counter()
    U, s, V = engine.rsvd(ctx, bm, k, random_state=rep, device_out=True)
    sync(); t2 = time.perf_counter()
    proj = engine.project(ctx, mat, V)
    sync(); t3 = time.perf_counter()
    bm.free()
    gb = n * nlat * nlon * 4 / 1e9
    print(f"member {rep}: resample {1e3*(t1-t0):.1f} ms ({3*gb/(t1-t0):.0f} GB/s alg: 2 reads + 2 writes... ) rsvd {1e3*(t2-t1):.1f} ms  "
          f"project {1e3*(t3-t2):.1f} ms  total {1e3*(t3-t0):.1f} ms; tv ratio {tv/st['total_variance']:.4f}", flush=True)
