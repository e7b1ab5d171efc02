"""Time the Hilbert / complex-EOF path: python tools/complex_probe.py n nlat nlon k
LAYOUT=written: the four written layouts + two-matrix launches (rounds 1-2); default: the lean layout (Re in place, Im^T)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
from xeofs_amd.complex_svd import complex_rsvd
import bench

n, nlat, nlon, k = (int(a) for a in sys.argv[1:5])
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    A, st = engine.preprocess(ctx, X, want_stats=False, in_place=os.environ.get("LAYOUT", "lean") != "written",
                              for_hilbert=not os.environ.get("NO_RAWT"))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    B, _ = engine.hilbert(ctx, A, "exp", 0.2)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    p = nlat * nlon
    passes = 16
    om = engine.sketch_matrix(min(n, p), k + 10, 5)
    for name, fn in (("engine eofx_rsvd_c64, outputs left on the device", lambda: engine.rsvd_c64(ctx, A, B, k, random_state=5, omega=om, device_out=True)),
                     ("panel-level (python) driver", lambda: complex_rsvd(ctx, A, B, k, random_state=5))):
        if name.startswith("panel") and (k + 10 > 32 or os.environ.get("ENGINE_ONLY")):
            continue
        torch.cuda.synchronize(); t2b = time.perf_counter()
        ctx.profile(True)
        U, s, V = fn()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        pr = ctx.profile_read(); ctx.profile(False)
        print(f"rep{rep} n={n} p={p} k={k} layout={os.environ.get('LAYOUT', 'lean')} HBM {torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9:.0f} GB in use [{name}]: preprocess {1e3*(t1-t0):.1f} ms  hilbert {1e3*(t2-t1):.1f} ms  "
              f"complex rsvd {1e3*(t3-t2b):.1f} ms  (atb launches {pr['launches']}, {pr['ms']:.1f} ms in atb)  "
              f"alg complex64 GB/s {passes*n*p*8.0/(t3-t2b)/1e9:.0f}  s[:3]={s[:3]}")
    A.free(); B.free()
