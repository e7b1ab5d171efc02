"""Where the host time of one headline step goes (config 4, one engine call): wall time per step against the GPU's busy time
(HIP events around the call), and a cProfile of 20 steps.  usage: python tools/step_host_profile.py [nlon]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench

n, nlat, nlon, k = 10000, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 1440, 50
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))

def step(marks=None):
    t0 = time.perf_counter()
    om = engine.sketch_matrix(n, k + 10, 5)
    t1 = time.perf_counter()
    mat, st, U, s, V = engine.fit(ctx, X, k, center=True, standardize=False, feature_weights=None, n_oversamples=10, n_iter="auto",
                                  omega=om, want_stats=False, device_out=True)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    mat.free()
    del U, V
    t4 = time.perf_counter()
    if marks is not None:
        marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))

for _ in range(3):
    step()
marks = []
torch.cuda.synchronize(); a = time.perf_counter()
for _ in range(20):
    step(marks)
torch.cuda.synchronize(); b = time.perf_counter()
m = 1e3 * np.array(marks).mean(0)
print(f"20 steps: {1e3 * (b - a) / 20:.3f} ms per step; host phases (ms): sketch {m[0]:.3f}  engine.fit call {m[1]:.3f}  wait for the GPU {m[2]:.3f}  free {m[3]:.3f}")
print("engine fit_info:", engine.fit_info(ctx))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
