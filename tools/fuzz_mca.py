"""Randomised parity sweep of the matrix-free cross-covariance solver (MCA, use_pca=False) with many modes in the noise
bulk and peaked spectra, against the oracle (C formed explicitly in float64 + the scikit-learn restatement)."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

warnings.simplefilter("ignore")
ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(ncase):
    n = int(rng.integers(40, 500)); p1 = int(rng.integers(50, 2500)); p2 = int(rng.integers(50, 2500))
    rank = int(rng.integers(2, 8))
    peak = float(rng.choice([1.0, 5.0, 30.0]))
    T = rng.standard_normal((n, rank)) * (peak * rng.uniform(0.4, 0.9) ** np.arange(rank))
    X = (T @ rng.standard_normal((rank, p1)) + rng.standard_normal((n, p1)) + rng.uniform(-30, 30)).astype(np.float32)
    Y = (T @ rng.standard_normal((rank, p2)) + rng.standard_normal((n, p2)) + rng.uniform(-30, 30)).astype(np.float32)
    k = int(rng.integers(1, min(p1, p2, 46)))
    seed = int(rng.integers(0, 1000))
    try:
        ref = orc.mca_fit(X.astype(np.float64), Y.astype(np.float64), k, random_state=seed, solver="randomized")
        mx, _ = engine.preprocess(ctx, X); my, _ = engine.preprocess(ctx, Y)
        out = engine.crosscov_rsvd(ctx, mx, my, k, random_state=seed)
        mx.free(); my.free()
        so, s = ref["singular_values"], out["s"]
        ok = np.all(np.abs(s - so) <= 1e-5 * so + 3e-6 * so[0])
        ok &= abs(out["total_squared_covariance"] - ref["total_squared_covariance"]) <= 1e-5 * ref["total_squared_covariance"]
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(n=n, p1=p1, p2=p2, k=k, rank=rank, peak=peak, seed=seed), "max rel",
                  float(np.max(np.abs(s - so) / so)))
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p1=p1, p2=p2, k=k), type(e).__name__, str(e)[:160])
print("cases", ncase, "bad", bad)
