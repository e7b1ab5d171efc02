"""Extreme magnitudes and mixed feature scales through the fused fit against the float64 oracle fit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

ctx = engine.Context(0)
rng = np.random.default_rng(5)
n, p, k = 400, 3000, 8
base = (rng.standard_normal((n, 10)) * 2.0 ** -np.arange(10)) @ rng.standard_normal((10, p)) + 0.05 * rng.standard_normal((n, p))
cases = {"x1": 1.0, "x1e-20": 1e-20, "x1e-30": 1e-30, "x1e15": 1e15, "x1e30": 1e30,
         "feature scales 1e-4..1e4": 10.0 ** rng.uniform(-4, 4, p), "feature scales 1e-8..1e8": 10.0 ** rng.uniform(-8, 8, p),
         "one feature x1e6": np.where(np.arange(p) == 7, 1e6, 1.0), "offset 1e6 (mean >> anomalies)": None}
for name, sc in cases.items():
    X = base * sc if sc is not None else base + 1e6
    X = X.astype(np.float32)
    for std in (False, True):
        try:
            mat, st, U, s, V = engine.fit(ctx, X, k, standardize=std, random_state=1)
            mat.free()
            ref = orc.eof_fit(X.astype(np.float64), k, standardize=std, random_state=1)
            es = np.abs(s - ref["norms"]).max() / ref["norms"][0]
            cos = min(abs(float(np.dot(V[:, j].astype(np.float64), ref["components"][:, j]))) for j in range(4))
            ou = np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max()
            print(f"{name:34s} standardize={std!s:5s}: s rel err {es:.2e}  min |cos| of 4 modes {cos:.7f}  |U^T U - I| {ou:.1e}  s0 {s[0]:.4g}", flush=True)
        except Exception as e:
            print(f"{name:34s} standardize={std!s:5s}: {type(e).__name__} {str(e)[:120]}", flush=True)

print("# two-step path (engine.preprocess in place + engine.rsvd) and the written-layout path on the mixed-scale cases, standardize=True")
for name in ("feature scales 1e-4..1e4", "feature scales 1e-8..1e8"):
    X = (base * cases[name]).astype(np.float32)
    ref = orc.eof_fit(X.astype(np.float64), k, standardize=True, random_state=1)
    for label, kw in (("in place", dict(in_place=True)), ("written", dict())):
        mat, st = engine.preprocess(ctx, X, True, True, None, **kw)
        U, s, V = engine.rsvd(ctx, mat, k, random_state=1)
        mat.free()
        es = np.abs(s - ref["norms"]).max() / ref["norms"][0]
        cos = min(abs(float(np.dot(V[:, j].astype(np.float64), ref["components"][:, j]))) for j in range(4))
        print(f"{name:34s} {label:9s}: s rel err {es:.2e}  min |cos| {cos:.7f}  fit_info {engine.fit_info(ctx)}", flush=True)
