"""Time the whitened cross models at config-3 size: xe.cross.CCA / RDA / CPCCA(alpha=0.5) with the reference's default
arguments (use_pca=True, n_pca_modes=0.999) on two 5000 x (360 x 360) halves -- PCA pre-reduction, whitener
(preprocessing/whitener.py:86-133), cross-covariance rSVD, total squared covariance of the unwhitened matrices."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
import bench

n, nlat, nlon, k = int(os.environ.get("N", 5000)), 360, 720, 20
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F[:, :, :360].contiguous(), dims=("time", "lat", "lon"))
Y = xe.DataArray(F[:, :, 360:].contiguous(), dims=("time", "lat", "lon"))
for name, mk in (("MCA  (alpha 1)", lambda: xe.cross.MCA(n_modes=k, random_state=5)),
                 ("CCA  (alpha 0)", lambda: xe.cross.CCA(n_modes=k, random_state=5)),
                 ("RDA  (alpha [0, 1])", lambda: xe.cross.RDA(n_modes=k, random_state=5)),
                 ("CPCCA(alpha 0.5)", lambda: xe.cross.CPCCA(n_modes=k, alpha=0.5, random_state=5))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = mk().fit(X, Y, "time")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = np.asarray(m.singular_values().values)
    print(f"{name}: fit {1e3 * dt:.1f} ms; s[:3] = {s[:3]}; TSC = {m.total_squared_covariance():.6g}", flush=True)
