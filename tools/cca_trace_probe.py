"""Default-argument CCA at config-3 size, four fits (rocprofv3 --kernel-trace + tools/trace_tail.py for the last one)."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
import bench
n, nlat, nlon, k = 5000, 360, 720, 20
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = xe.DataArray(F[:, :, :360].contiguous(), dims=("time", "lat", "lon"))
Y = xe.DataArray(F[:, :, 360:].contiguous(), dims=("time", "lat", "lon"))
warnings.simplefilter("ignore")
which = os.environ.get("MODEL", "CCA")
for rep in range(int(os.environ.get("REPS", 4))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = (xe.cross.CCA(n_modes=k, random_state=5) if which == "CCA" else xe.cross.MCA(n_modes=k, random_state=5)).fit(X, Y, "time")
    torch.cuda.synchronize(); print(f"{which} fit {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    time.sleep(0.05)
