"""Preprocess cost with a 30 % land mask (all-NaN grid points) + cos-lat weights at config-4 size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import bench
n, nlat, nlon, k = 10000, 720, 1440, 50
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
if os.environ.get("MASK", "random") == "blobs":      # "continents": rectangles in lat x lon, ~30 % of the grid in long runs
    m2 = torch.zeros((nlat, nlon), dtype=torch.bool, device="cuda")
    for (a, b, c, d) in ((120, 400, 100, 420), (320, 620, 620, 860), (60, 250, 980, 1300), (520, 680, 1040, 1180), (0, 48, 0, 1440)):
        m2[a:b, c:d] = True
    mask = m2.reshape(-1)
else:
    mask = torch.rand(nlat * nlon, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) < 0.3
print(f"mask: {os.environ.get('MASK', 'random')}, {float(mask.float().mean()):.3f} of the grid points")
X[:, mask] = float("nan")
lat = np.linspace(-89.75, 89.75, nlat)
w = np.repeat(np.sqrt(np.cos(np.deg2rad(lat)).clip(0, 1)), nlon)
for rep in range(6):
    masked = rep >= 3     # layout mode 3: the masked grid points stay in place as zero columns (1x the field in HBM)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mat, st = engine.preprocess(ctx, X, True, False, w, want_stats=False, in_place=masked, allow_masked=masked)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    U, s, V = engine.rsvd(ctx, mat, k, random_state=5, device_out=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep{rep} [{'masked in place' if mat.masked else 'compacted, two layouts'}]: valid features {st['p']} of {nlat*nlon}; preprocess {1e3*(t1-t0):.1f} ms, rsvd {1e3*(t2-t1):.1f} ms, s0={s[0]:.3f}")
    mat.free()
# the one-call fit on the masked field: statistics during the first pass, masked in-place layout (round 3)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mat, st, U, s, V = engine.fit(ctx, X, k, True, False, w, random_state=5, want_stats=False, device_out=True, allow_masked=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"fit rep{rep} [{'one call, masked in place' if st['fused'] and mat.masked else 'fallback ' + str(engine.fit_info(ctx))}]: "
          f"valid features {st['p']} of {nlat*nlon}; whole fit {1e3*(t1-t0):.1f} ms, s0={s[0]:.3f}")
    mat.free()
