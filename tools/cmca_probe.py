"""Complex cross models at config-3 size (two 5000 x 129 600 halves), as a user calls them: HilbertMCA on real fields,
ComplexMCA on complex ones (default arguments: PCA pre-reduction).  python tools/cmca_probe.py [n nlat nlon]"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xeofs_amd as xe
from oracle import eof_oracle as orc

warnings.simplefilter("ignore")
n, nlat, nlon = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 360, 360)
vals, lat = orc.synthetic_field(n, nlat, 2 * nlon, rank=30, seed=0)
vals = vals.reshape(n, nlat, 2 * nlon)
A, B = np.ascontiguousarray(vals[:, :, :nlon]), np.ascontiguousarray(vals[:, :, nlon:])
X = xe.DataArray(A, dims=("time", "lat", "lon"))
Y = xe.DataArray(B, dims=("time", "lat", "lon"))
for rep in range(2):
    t0 = time.perf_counter()
    m = xe.cross.HilbertMCA(n_modes=10, random_state=5).fit(X, Y, "time")
    t1 = time.perf_counter()
    c = m.components_amplitude(); s = m.scores_phase()
    t2 = time.perf_counter()
    print(f"HilbertMCA {n} x ({nlat} x {nlon}) x 2, k = 10: fit {1e3 * (t1 - t0):.0f} ms (PCA modes kept: "
          f"{m.field[0].pca.m}, {m.field[1].pca.m}), accessors {1e3 * (t2 - t1):.0f} ms; s[:3] = {m.singular_values().values[:3]}", flush=True)
    del m
Zx = xe.DataArray(A + 1j * np.roll(A, 7, axis=0), dims=("time", "lat", "lon"))
Zy = xe.DataArray(B + 1j * np.roll(B, 7, axis=0), dims=("time", "lat", "lon"))
for rep in range(2):
    t0 = time.perf_counter()
    m = xe.cross.ComplexMCA(n_modes=10, random_state=5).fit(Zx, Zy, "time")
    t1 = time.perf_counter()
    print(f"ComplexMCA same size: fit {1e3 * (t1 - t0):.0f} ms (PCA modes kept: {m.field[0].pca.m}, {m.field[1].pca.m}); "
          f"s[:3] = {m.singular_values().values[:3]}", flush=True)
    del m
