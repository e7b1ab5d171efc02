"""Three default-argument CCA fits at config-3 size for rocprofv3 --kernel-trace (which kernels besides rocSOLVER's eigh?)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
import bench
n, nlat, nlon, k = 5000, 360, 720, 20
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F[:, :, :360].contiguous(), dims=("time", "lat", "lon"))
Y = xe.DataArray(F[:, :, 360:].contiguous(), dims=("time", "lat", "lon"))
warnings.simplefilter("ignore")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = xe.cross.CCA(n_modes=k, random_state=5).fit(X, Y, "time")
    torch.cuda.synchronize(); print("CCA fit ms", 1e3 * (time.perf_counter() - t0), flush=True)
