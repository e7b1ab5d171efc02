"""G8: pin the oracle's varimax / promax to outputs of the REFERENCE's own routine.

`/root/reference/xeofs/linalg/_numpy/_rotation.py` needs only numpy + dask, which /opt/conda/bin/python3.9 of
this image has (the build interpreter has no dask).  The file is loaded BY PATH and executed where it lies;
nothing of it is copied.  Run in this container:

    /opt/conda/bin/python3.9 oracle/make_golden_rotation.py

Writes tests/golden/g8_rotation.npz: seeded loadings (real and complex) and, per case, the reference's
rotated loadings, rotation matrix and phi for Varimax (power 1) and Promax (power 2, 4).
"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/xeofs/linalg/_numpy/_rotation.py"
spec = importlib.util.spec_from_file_location("ref_rotation", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def loadings(p, m, seed, cplx=False):
    """EOF-like loadings: smooth-ish structured columns times decaying amplitudes plus noise."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 1, p)[:, None]
    base = np.sin(2 * np.pi * t * (1 + np.arange(m))[None, :] + rng.uniform(0, 6, m)[None, :])
    A = (base + 0.3 * rng.standard_normal((p, m))) * (3.0 * 0.8 ** np.arange(m))[None, :]
    if cplx:
        B = (np.cos(2 * np.pi * t * (1 + np.arange(m))[None, :]) + 0.3 * rng.standard_normal((p, m)))
        A = A + 1j * B * (3.0 * 0.8 ** np.arange(m))[None, :]
    return A


out = {}
cases = [("real_a", 400, 5, 1, False), ("real_b", 1500, 12, 2, False), ("cplx_a", 300, 4, 3, True),
         ("cplx_b", 900, 8, 4, True)]
for name, p, m, seed, cplx in cases:
    X = loadings(p, m, seed, cplx)
    out[f"{name}_X"] = X
    for power in (1, 2, 4):
        Xrot, R, phi = ref._promax(X, power=power, max_iter=1000, rtol=1e-8)
        out[f"{name}_p{power}_Xrot"] = Xrot
        out[f"{name}_p{power}_R"] = R
        out[f"{name}_p{power}_phi"] = phi
    Xv, Rv = ref._varimax(X, gamma=1, max_iter=1000, rtol=1e-8)
    out[f"{name}_varimax_Xrot"] = Xv
    out[f"{name}_varimax_R"] = Rv
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g8_rotation.npz"), numpy=np.__version__, **out)
print("wrote g8_rotation.npz:", sorted(k for k in out if k.endswith("_R"))[:4], "...")
