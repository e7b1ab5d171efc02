"""Model-level timing as a user calls it: xe.single.EOF(n_modes).fit(DataArray(host numpy), "time") at the reference's
CPU-runnable config 1 shape (2920 x 25 x 53, k = 10) and config 2 (5000 x 360 x 720, k = 50), against the oracle on the
same host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xeofs_amd as xe
from oracle import eof_oracle as orc

for name, (n, nlat, nlon, k), reps in (("config1", (2920, 25, 53, 10), 20), ("config2", (5000, 360, 720, 50), 3)):
    vals, lat = orc.synthetic_field(n, nlat, nlon, rank=20 if n < 3000 else 30, seed=0)
    vals = vals.reshape(n, nlat, nlon)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"), coords={"lat": lat})
    xe.single.EOF(n_modes=k, random_state=5).fit(X, "time")
    t0 = time.perf_counter()
    for _ in range(reps):
        m = xe.single.EOF(n_modes=k, random_state=5).fit(X, "time")
        c = m.components(); s = m.scores()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    ref = orc.eof_fit(vals.reshape(n, -1), k, random_state=5)
    dc = time.perf_counter() - t0
    err = np.abs(m.singular_values().values - ref["norms"]).max() / ref["norms"][0]
    print(f"{name}: GPU model fit + accessors {1e3 * dt:.1f} ms (host numpy in / out, PCIe included); CPU oracle fit {1e3 * dc:.0f} ms; "
          f"max rel diff of s {err:.1e}", flush=True)

# config 4 with the field resident in HBM (torch CUDA tensor inside the DataArray)
import torch
import bench
n, nlat, nlon, k = 10000, 720, 1440, 50
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F, dims=("time", "lat", "lon"))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = xe.single.EOF(n_modes=k, random_state=5).fit(X, "time")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    c = m.components(); s = m.scores()
    t2 = time.perf_counter()
    print(f"config4 resident: model fit {1e3 * (t1 - t0):.1f} ms; components() + scores() (207 MB download, unstack) {1e3 * (t2 - t1):.1f} ms", flush=True)
    del m, c, s
