"""cProfile of ComplexMCA.fit / HilbertMCA.fit at config-3 size (host-side view: which step the wall time sits in)."""
import sys, os, cProfile, pstats, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xeofs_amd as xe
from oracle import eof_oracle as orc
warnings.simplefilter("ignore")
n, nlat, nlon = 5000, 360, 360
vals, lat = orc.synthetic_field(n, nlat, 2 * nlon, rank=30, seed=0)
vals = vals.reshape(n, nlat, 2 * nlon)
A, B = np.ascontiguousarray(vals[:, :, :nlon]), np.ascontiguousarray(vals[:, :, nlon:])
which = sys.argv[1] if len(sys.argv) > 1 else "complex"
if which == "complex":
    X = xe.DataArray(A + 1j * np.roll(A, 7, axis=0), dims=("time", "lat", "lon"))
    Y = xe.DataArray(B + 1j * np.roll(B, 7, axis=0), dims=("time", "lat", "lon"))
    mk = lambda: xe.cross.ComplexMCA(n_modes=10, random_state=5)
else:
    X = xe.DataArray(A, dims=("time", "lat", "lon")); Y = xe.DataArray(B, dims=("time", "lat", "lon"))
    mk = lambda: xe.cross.HilbertMCA(n_modes=10, random_state=5)
mk().fit(X, Y, "time")
pr = cProfile.Profile(); pr.enable()
mk().fit(X, Y, "time")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
