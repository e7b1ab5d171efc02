"""Time config 3: xe.cross.MCA n_modes=20 on two 5000 x (360 x 360) halves (matrix-free cross-covariance)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench
n, nlat, nlon, k = 5000, 360, 720, 20
ctx = engine.Context(0)
INPLACE = os.environ.get("LAYOUT", "inplace") == "inplace"
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = F[:, :, :360].reshape(n, -1).contiguous(); Y = F[:, :, 360:].reshape(n, -1).contiguous()
for rep in range(3):
    for tsc in (False, True):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        om = engine.SketchFuture(min(X.shape[1], Y.shape[1]), k + 10, 5)
        mx, _ = engine.preprocess(ctx, X, want_stats=False, in_place=INPLACE); my, _ = engine.preprocess(ctx, Y, want_stats=False, in_place=INPLACE)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = engine.crosscov_rsvd(ctx, mx, my, k, random_state=5, want_tsc=tsc, omega=om)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        p1 = p2 = nlat * 360
        alg = 16 * n * (p1 + p2) * 4.0
        print(f"rep{rep} tsc={tsc}: preprocess {1e3*(t1-t0):.1f} ms  crosscov rsvd {1e3*(t2-t1):.1f} ms  "
              f"alg {alg/(t2-t1)/1e9:.0f} GB/s  s[:3]={out['s'][:3]}  scf_sum={float((out['s'].astype(np.float64)**2).sum()/out['total_squared_covariance']) if tsc else None}")
        mx.free(); my.free()
