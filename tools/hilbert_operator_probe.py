"""Operator route (eofx_rsvd_hilbert_c64) against the two-part route and the exact float64 SVD of the oracle's analytic
signal, as the series gets longer.  usage: python tools/hilbert_operator_probe.py [p]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import eof_oracle as orc
from xeofs_amd import engine

p = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = engine.default_context(0)
k = 20
for n in (1000, 2000, 4000, 8000):
    rng = np.random.default_rng(n)
    # AR(1) series with a few shared patterns (like bench.make_field): red spectra along the samples
    e = rng.standard_normal((n, p))
    X = np.empty((n, p))
    X[0] = e[0]
    for i in range(1, n):
        X[i] = 0.8 * X[i - 1] + 0.6 * e[i]
    for j in range(24):
        t = np.cumsum(rng.standard_normal(n)) * 0.05 + np.sin(2 * np.pi * (j + 1) * np.arange(n) / n * 3.3 + j)
        X += (24 - j) * 0.6 * np.outer(t, rng.standard_normal(p))
    X = X.astype(np.float32)
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    Z = orc.hilbert_transform(Xc, padding="exp", decay_factor=0.2)
    se = np.linalg.svd(Z, compute_uv=False)[:k]
    A0, _ = engine.preprocess(ctx, X, True, False, None)
    B0, _ = engine.hilbert(ctx, A0, "exp", 0.2)
    b0 = B0.download().astype(np.float64)
    err_im = np.abs(b0 - Z.imag).max() / np.abs(Z.imag).max()
    out = {}
    for rule in ("auto", "converge"):
        _, s2, _ = engine.rsvd_c64(ctx, A0, B0, k, random_state=5, n_iter=rule)
        _, s1, _ = engine.rsvd_hilbert_c64(ctx, A0, k, "exp", 0.2, random_state=5, n_iter=rule)
        out[rule] = (np.abs(s2 - se).max() / se[0], np.abs(s1 - se).max() / se[0], np.abs(s1 - s2).max() / se[0])
    print(f"n={n} p={p} Im stage err {err_im:.2e} | auto: two-part {out['auto'][0]:.2e} operator {out['auto'][1]:.2e} diff {out['auto'][2]:.2e}"
          f" | converge: two-part {out['converge'][0]:.2e} operator {out['converge'][1]:.2e} diff {out['converge'][2]:.2e}"
          f" | s0/s1/s19 {se[0]:.1f} {se[1]:.1f} {se[19]:.1f}", flush=True)
    A0.free(); B0.free()
