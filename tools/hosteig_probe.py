"""Host Rayleigh-Ritz solver (eofx_host_zheigh_top_f64) on this box: order 240 / 360, 1..8 threads of the tridiagonalisation team."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
for m, nev in ((240, 30), (360, 30)):
    A = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)); A = A + A.conj().T
    Hr, Hi = np.ascontiguousarray(A.real), np.ascontiguousarray(A.imag)
    for nt in ("1", "2", "4", "8"):
        os.environ["EOFX_HOSTEIG_THREADS"] = nt
        w = np.zeros(nev); Xr = np.zeros((m, nev)); Xi = np.zeros((m, nev)); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); rc = lib.eofx_host_zheigh_top_f64(Hr.ctypes.data, Hi.ctypes.data, m, nev, w.ctypes.data, Xr.ctypes.data, Xi.ctypes.data); ts.append(time.perf_counter() - t0)
        print(f"order {m} threads {nt}: rc {rc} min {1e3*min(ts):.2f} ms median {1e3*sorted(ts)[3]:.2f} ms")
