"""PCIe-inclusive rate of the boundary when it is handed host (numpy) buffers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine
ctx = engine.Context(0)
n, p, k = 5000, 259200, 50
X = np.random.default_rng(0).standard_normal((n, p), dtype=np.float32)
for rep in range(2):
    t0 = time.perf_counter()
    mat, st = engine.preprocess(ctx, X, want_stats=False)
    t1 = time.perf_counter()
    U, s, V = engine.rsvd(ctx, mat, k, random_state=5)      # numpy outputs (D2H of U, V)
    t2 = time.perf_counter()
    mat.free()
    print(f"host-buffer fit {n}x{p}: upload+preprocess {1e3*(t1-t0):.0f} ms ({n*p*4/(t1-t0)/1e9:.1f} GB/s of input), "
          f"rsvd+download {1e3*(t2-t1):.0f} ms, whole fit {1e3*(t2-t0):.0f} ms -> "
          f"{16*n*p*4/(t2-t0)/1e9:.0f} GB/s algorithmic")
