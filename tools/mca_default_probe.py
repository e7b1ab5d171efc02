import sys, time, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, xeofs_amd as xe
from xeofs_amd import engine
n, nlat, nlon, k = 5000, 360, 720, 20
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, "cuda:0").reshape(n, nlat, nlon)
X = F[:, :, :360].contiguous(); Y = F[:, :, 360:].contiguous(); del F
Xd = xe.DataArray(X, dims=("time", "lat", "lon")); Yd = xe.DataArray(Y, dims=("time", "lat", "lon"))
def run(cls):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return cls(n_modes=k, random_state=5).fit(Xd, Yd, "time")
for cls in (xe.cross.MCA, xe.cross.CCA):
    run(cls)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m = run(cls); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(cls.__name__, "ms", [round(1e3 * t, 1) for t in ts], "pca modes", m.pca[0].m, m.pca[1].m, "spectrum known", m.pca[0].spectrum_known,
          "s", np.asarray(m.singular_values().values)[:3], "tsc", float(np.asarray(m.data["total_squared_covariance"])) if "total_squared_covariance" in m.data else None)
