"""Second randomised parity sweep: tall panels above the 16 MB size rule, peaked and flat spectra, sketch widths at and
beyond the limits of the device factorisation (k + 10 = 64), the host factorisation (256) and the wide route."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine
from xeofs_amd.linalg import Decomposer

warnings.simplefilter("ignore")
ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for case in range(ncase):
    n = int(rng.integers(60, 700)); p = int(rng.integers(20000, 150000))
    if rng.random() < 0.3:
        n, p = p // 40, n * 3          # tall in samples instead (n > p)
    r = min(n, p)
    k = int(rng.choice([3, 20, 44, 54, 55, 100, 246, 300]))
    k = max(1, min(k, r - 1))
    rank = int(rng.integers(2, 10))
    peak = float(rng.choice([1.0, 5.0, 40.0]))
    amp = peak * rng.uniform(0.4, 0.9) ** np.arange(rank)
    X = ((rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p)) + rng.standard_normal((n, p))).astype(np.float32)
    X -= X.mean(0)
    seed = int(rng.integers(0, 1000))
    try:
        Uo, so, Vo = orc.decomposer_fit(X.astype(np.float64), k, random_state=seed)
        d = Decomposer(n_modes=k, random_state=seed, ctx=ctx).fit(X, total_variance=1.0)
        s = d.s_
        se = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:k]
        wide = k + 10 > 256
        tol = 1e-5 * so + 3e-6 * so[0]
        if wide:      # exact route against an unconverged randomized oracle: compare with the exact SVD instead
            ok = np.all(np.abs(s - se) <= 2e-5 * se + 3e-6 * se[0])
        else:
            ok = np.all(np.abs(s - so) <= tol)
        ok &= d.V_.shape == (p, k) and d.U_.shape == (n, k)
        ok &= np.abs(d.V_.astype(np.float64).T @ d.V_.astype(np.float64) - np.eye(k)).max() < 5e-5
        if not ok:
            bad += 1
            ref = se if wide else so
            print("MISMATCH case", case, dict(n=n, p=p, k=k, rank=rank, peak=peak, seed=seed, wide=wide), "max rel",
                  float(np.max(np.abs(s - ref) / ref)))
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p=p, k=k), type(e).__name__, str(e)[:160])
print("cases", ncase, "bad", bad)
