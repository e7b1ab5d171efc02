"""Three `n_iter="converge"` decompositions of the config-5 field on the operator route (for rocprofv3 --kernel-trace +
tools/trace_gaps.py with the marker panel_import_kernel: one start-panel import per call)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from xeofs_amd import engine

ctx = engine.default_context(0)
n, nlat, nlon, k = 8000, 720, 1440, 20
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=1)
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    U, s, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=os.environ.get("RULE", "converge"), device_out=True)
    torch.cuda.synchronize()
    print("rule", os.environ.get("RULE", "converge"), "products", engine.last_iterations(ctx), "ms", 1e3 * (time.perf_counter() - t0), flush=True)
