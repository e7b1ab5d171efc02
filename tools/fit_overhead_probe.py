"""Where the host time of one `engine.fit` goes (python tools/fit_overhead_probe.py [nlon]): wall time of the engine
call itself (ctypes -> eofx_fit_f32) next to everything Python does around it, plus a cProfile of 20 steps."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench

nlon = int(sys.argv[1]) if len(sys.argv) > 1 else 180
n, nlat, k = 10000, 720, 50
P = nlat * nlon
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, P, torch.device("cuda:0"))
torch.cuda.synchronize()
raw = ctx.lib.eofx_fit_f32
spent = [0.0]


class Timed:
    def __call__(self, *a):
        t = time.perf_counter()
        r = raw(*a)
        spent[0] += time.perf_counter() - t
        return r


ctx.lib.eofx_fit_f32 = Timed()


def step(last):
    if last is not None:
        last.free()
    omega = engine.SketchFuture(min(n, P), k + 10, 5)
    mat, st, U, s, V = engine.fit(ctx, X, k, omega=omega, want_stats=False, device_out=True)
    torch.cuda.synchronize()
    return mat


last = None
for _ in range(5):
    last = step(last)
spent[0] = 0.0
t0 = time.perf_counter()
for _ in range(20):
    last = step(last)
t1 = time.perf_counter()
print(f"nlon={nlon}: {1e3 * (t1 - t0) / 20:.3f} ms per step, {1e3 * spent[0] / 20:.3f} ms inside eofx_fit_f32, "
      f"{1e3 * ((t1 - t0) - spent[0]) / 20:.3f} ms of Python around it; fit_info {engine.fit_info(ctx)}")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    last = step(last)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue())
