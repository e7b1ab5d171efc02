"""EOF.transform of new data at config 4 (field resident as a torch tensor): python tools/transform_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
import bench

n, nlat, nlon, k = 10000, 720, 1440, 50
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F, dims=("time", "lat", "lon"))
m = xe.single.EOF(n_modes=k, random_state=5).fit(X, "time")
sc = m.scores().values
for in_place in (True, False):
    m.preprocessor.in_place = in_place
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t = m.transform(X)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    err = np.abs(t.values - sc).max() / np.abs(sc).max()
    print(f"transform of the {n} x {nlat * nlon} training field, in_place={in_place}: {1e3 * dt:.1f} ms; max diff to the fitted scores {err:.1e}", flush=True)
