"""Five config-3 MCA calls (TSC=1: with the total squared covariance, the Gram route; LAYOUT=inplace|copy) for
`rocprofv3 --kernel-trace` + tools/trace_gaps.py with the marker panel_import_kernel: one sketch import per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench
n, nlat, nlon, k = 5000, 360, 720, 20
ctx = engine.Context(0)
TSC = os.environ.get('TSC', '0') == '1'
INPLACE = os.environ.get('LAYOUT', 'copy') == 'inplace'
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = F[:, :, :360].reshape(n, -1).contiguous(); Y = F[:, :, 360:].reshape(n, -1).contiguous()
for rep in range(5):
    om = engine.SketchFuture(X.shape[1], k + 10, 5)
    mx, _ = engine.preprocess(ctx, X, want_stats=False, in_place=INPLACE); my, _ = engine.preprocess(ctx, Y, want_stats=False, in_place=INPLACE)
    out = engine.crosscov_rsvd(ctx, mx, my, k, random_state=5, want_tsc=TSC, omega=om)
    mx.free(); my.free()
torch.cuda.synchronize()
