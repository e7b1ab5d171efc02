"""Edge shapes and input kinds through the engine and the EOF model against the float64 oracle: tiny matrices, k = rank, k = 1,
one feature, two samples, float64 / integer / non-contiguous / Fortran-ordered input, zero weights, a constant feature."""
import sys, os, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine
import xeofs_amd as xe

warnings.simplefilter("ignore")
ctx = engine.Context(0)
rng = np.random.default_rng(1)
bad = 0
def check(name, X, k, **kw):
    global bad
    try:
        X32 = np.asarray(X)
        mat, st, U, s, V = engine.fit(ctx, X32, k, random_state=1, **kw)
        mat.free()
        X64 = np.asarray(X, dtype=np.float32).astype(np.float64)
        ref = orc.eof_fit(X64, k, random_state=1, **kw)
        es = np.abs(s - ref["norms"]).max() / max(ref["norms"][0], 1e-300)
        ou = np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max()
        ov = np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max()
        rec = np.linalg.norm((U.astype(np.float64) * s) @ V.T.astype(np.float64) - (ref["scores"] @ ref["components"].T if "scores" in ref else 0))
        ok = es <= 2e-5 and ou <= 3e-5 and ov <= 3e-5 and V.shape == ref["components"].shape
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {name:46s} shape {X32.shape} k {k}: s err {es:.1e} |U^TU-I| {ou:.1e} |V^TV-I| {ov:.1e} V {V.shape} fused {engine.fit_info(ctx)}", flush=True)
    except Exception as e:
        try:
            orc.eof_fit(np.asarray(X, dtype=np.float64), k, random_state=1, **kw)
            bad += 1
            print(f"BAD {name:46s} engine raised {type(e).__name__}: {str(e)[:100]} but the oracle did not", flush=True)
        except Exception as e2:
            same = type(e).__name__ == type(e2).__name__
            print(f"{'ok ' if same else '?? '} {name:46s} both raise: engine {type(e).__name__}: {str(e)[:60]} | oracle {type(e2).__name__}: {str(e2)[:60]}", flush=True)

def field(n, p, r=4):
    return ((rng.standard_normal((n, r)) * 2.0 ** -np.arange(r)) @ rng.standard_normal((r, p)) + 0.1 * rng.standard_normal((n, p)) + 2.0).astype(np.float32)

check("tiny 3 x 5, k 2", field(3, 5, 2), 2)
check("2 samples, k 1", field(2, 40, 1), 1)
check("2 samples, k 2 (= rank before centring)", field(2, 40, 1), 2)
check("one feature", field(50, 1, 1), 1)
check("k = min(n, p) tall", field(40, 6), 6)
check("k = min(n, p) wide", field(6, 40), 6)
check("k = 1 wide", field(100, 3000), 1)
check("k = 63 (widest fused sketch is 64)", field(200, 4000, 8), 54)
check("float64 input", field(120, 700).astype(np.float64), 4)
check("int32 input", (field(120, 700) * 100).astype(np.int32), 4)
check("non-contiguous rows (every 2nd)", field(240, 700)[::2], 4)
check("non-contiguous columns", field(120, 1400)[:, ::2], 4)
check("Fortran order", np.asfortranarray(field(120, 700)), 4)
w = rng.uniform(0.2, 1.0, 700); w[:50] = 0.0
check("zero weights on 50 features", field(120, 700), 4, feature_weights=w)
Xc = field(120, 700); Xc[:, 10] = 7.5
check("a constant feature", Xc, 4)
check("a constant feature, standardize", Xc, 4, standardize=True)
Xn = field(120, 700); Xn[5, :] = np.nan
check("one all-NaN sample", Xn, 4)
Xn = field(120, 700); Xn[:, 5] = np.nan; Xn[7, :] = np.nan
check("all-NaN sample and all-NaN feature", Xn, 4)
Xi = field(120, 700); Xi[3, 3] = np.inf
check("an infinity", Xi, 4)
check("all zeros", np.zeros((50, 80), np.float32), 2)
check("all equal", np.full((50, 80), 3.25, np.float32), 2)
print("bad", bad)
