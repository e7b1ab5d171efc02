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

# the resample-free member (validation.BootstrapOps): both products on the ORIGINAL matrix, in place
from xeofs_amd.validation.bootstrapper import BootstrapOps, _Solo
from xeofs_amd.sharded import sharded_rsvd

mat.free()
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
mat, st = engine.preprocess(ctx, X, want_stats=False, keep_raw=True, in_place=True)
sync(); t0 = time.perf_counter()
r2 = engine.sample_norms(ctx, mat) ** 2
sync(); print(f"row norms once: {1e3*(time.perf_counter()-t0):.1f} ms; layouts {mat.layout()}", flush=True)
for rep in range(3):
    idx = rng.choice(n, n, replace=True)
    sync(); t0 = time.perf_counter()
    ops = BootstrapOps(ctx, mat, idx)
    U, s, V = sharded_rsvd(ops, _Solo(), k, nlat * nlon, 0, random_state=rep, device_out=True)
    sync(); t1 = time.perf_counter()
    c = ops.counts.cpu().numpy()
    tv = (float(c @ r2) - n * ops.mean_sumsq()) / (n - 1)
    sync(); t2 = time.perf_counter()
    proj = engine.project(ctx, mat, V)
    sync(); t3 = time.perf_counter()
    print(f"in-place member {rep}: rsvd {1e3*(t1-t0):.1f} ms  total variance {1e3*(t2-t1):.1f} ms  project {1e3*(t3-t2):.1f} ms  "
          f"total {1e3*(t3-t0):.1f} ms; tv ratio {tv/st['total_variance']:.4f}; layouts {mat.layout()}; "
          f"HBM in use {torch.cuda.mem_get_info()[1]/1e9 - torch.cuda.mem_get_info()[0]/1e9:.0f} GB", flush=True)
from xeofs_amd.sharded import HipPanelOps
for rep in range(2):
    sync(); t0 = time.perf_counter()
    U, s, V = sharded_rsvd(HipPanelOps(ctx, mat), _Solo(), k, nlat * nlon, 0, random_state=rep, device_out=True)
    sync(); print(f"panel-level driver on the plain matrix: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
    sync(); t0 = time.perf_counter()
    U, s, V = engine.rsvd(ctx, mat, k, random_state=rep, device_out=True)
    sync(); print(f"engine driver on the plain matrix: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
# the same member with the sample-contiguous layout built ahead (EOFBootstrapper does this where HBM has room)
sync(); t0 = time.perf_counter()
built = mat.ensure_sample_layout(only_if_room=True)
sync(); print(f"ensure_sample_layout: built={built} in {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
for rep in range(3):
    idx = rng.choice(n, n, replace=True)
    sync(); t0 = time.perf_counter()
    ops = BootstrapOps(ctx, mat, idx)
    U, s, V = sharded_rsvd(ops, _Solo(), k, nlat * nlon, 0, random_state=rep, device_out=True)
    sync(); t1 = time.perf_counter()
    proj = engine.project(ctx, mat, V)
    sync(); t3 = time.perf_counter()
    print(f"member with the layout {rep}: rsvd {1e3*(t1-t0):.1f} ms  project {1e3*(t3-t1):.1f} ms  total {1e3*(t3-t0):.1f} ms; "
          f"HBM in use {torch.cuda.mem_get_info()[1]/1e9 - torch.cuda.mem_get_info()[0]/1e9:.0f} GB", flush=True)
