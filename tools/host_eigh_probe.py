"""Wall time of the host eigen-solver of the small Rayleigh-Ritz problems (eofx_host_eigh_f64) next to numpy's LAPACK call."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import _lib
lib = _lib.load()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for n in (30, 60, 64, 128, 240):
    B = np.random.default_rng(0).standard_normal((n, 400)); A = np.ascontiguousarray(B @ B.T)
    w = np.empty(n); V = np.empty((n, n))
    lib.eofx_host_eigh_f64(p(A), n, p(w), p(V))
    t = time.perf_counter()
    for _ in range(200): lib.eofx_host_eigh_f64(p(A), n, p(w), p(V))
    dt = (time.perf_counter() - t) / 200
    t = time.perf_counter()
    for _ in range(200): np.linalg.eigh(A)
    dn = (time.perf_counter() - t) / 200
    print(f"n {n}: eofx_host_eigh_f64 {dt * 1e6:.1f} us   numpy.linalg.eigh {dn * 1e6:.1f} us   max |w - w_numpy| / w_max {np.abs(np.sort(w) - np.linalg.eigvalsh(A)).max() / w.max():.1e}")
