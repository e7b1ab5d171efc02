"""rSVD alone (sketch drawn beforehand) on the three layouts of the same field: python tools/layout_probe.py n nlat nlon k"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench

n, nlat, nlon, k = (int(a) for a in sys.argv[1:5])
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
om = engine.sketch_matrix(min(n, nlat * nlon), k + 10, 5)
for name, kw in (("copy", {}), ("raw", {"keep_raw": True}), ("inplace", {"in_place": True})):
    ts, tp = [], []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mat, _ = engine.preprocess(ctx, X, want_stats=False, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        U, s, V = engine.rsvd(ctx, mat, k, omega=om, device_out=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        mat.free()
        if rep >= 2:
            tp.append(t1 - t0); ts.append(t2 - t1)
    print(f"{name:8s} preprocess {1e3*np.mean(tp):7.3f} ms   rsvd {1e3*np.mean(ts):7.3f} ms   s[0]={float(s[0]):.4f}")
