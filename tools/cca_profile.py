"""cProfile of one default-argument CCA fit at config-3 size (which host step does the time sit in)."""
import sys, os, cProfile, pstats, io, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
import bench

n, nlat, nlon, k = 5000, 360, 720, 20
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F[:, :, :360].contiguous(), dims=("time", "lat", "lon"))
Y = xe.DataArray(F[:, :, 360:].contiguous(), dims=("time", "lat", "lon"))
warnings.simplefilter("ignore")
for _ in range(2):
    xe.cross.CCA(n_modes=k, random_state=5).fit(X, Y, "time")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
xe.cross.CCA(n_modes=k, random_state=5).fit(X, Y, "time")
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
