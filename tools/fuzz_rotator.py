"""Randomised parity sweep of EOFRotator / EOFBootstrapper against the oracle (random shapes, NaN masks, power)."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xeofs_amd as xe
from oracle import eof_oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
warnings.simplefilter("ignore")
for case in range(ncase):
    n = int(rng.integers(40, 300))
    shp = (int(rng.integers(4, 16)), int(rng.integers(4, 20)))
    p = shp[0] * shp[1]
    r = int(rng.integers(4, 9))
    # simple-structure patterns so that the varimax optimum is well defined
    S = 0.05 * rng.standard_normal((p, r))
    S[np.arange(p), rng.integers(0, r, p)] += rng.uniform(1, 3, p)
    T = rng.standard_normal((n, r)) * (4.0 * rng.uniform(0.75, 0.95) ** np.arange(r))
    A = T @ S.T + 0.05 * rng.standard_normal((n, p)) + rng.uniform(-20, 20)
    if rng.random() < 0.4:
        A[:, rng.choice(p, size=max(1, p // 10), replace=False)] = np.nan
    k = int(rng.integers(2, r + 1))
    power = int(rng.choice([1, 1, 2, 3]))
    seed = int(rng.integers(0, 1000))
    ok = True
    try:
        X = xe.DataArray(A.reshape((n,) + shp), dims=("time", "lat", "lon"))
        m = xe.single.EOF(n_modes=r, random_state=seed, solver="full").fit(X, "time")
        rot = xe.single.EOFRotator(n_modes=k, power=power).fit(m)
        eof = orc.eof_fit(A, r, random_state=seed, solver="full")
        eof["input_data"] = eof["input_data"] if "input_data" in eof else None
        Ac = A[:, ~np.isnan(A).all(axis=0)]
        eof["input_data"] = Ac - Ac.mean(0)
        ref = orc.eof_rotator_fit(eof, k, power=power)
        ev = rot.explained_variance().values
        if not np.allclose(ev, ref["explained_variance"], rtol=2e-3):
            ok = False
        comps = rot.components().values.reshape(k, -1).T
        comps = comps[~np.isnan(comps).any(axis=1)]
        gaps = np.abs(np.diff(ref["explained_variance"])) / ref["explained_variance"][0]
        for j in range(k):
            g = min(gaps[j - 1] if j else 1.0, gaps[j] if j < k - 1 else 1.0)
            c = np.dot(comps[:, j], ref["components"][:, j]) / np.linalg.norm(comps[:, j]) / np.linalg.norm(ref["components"][:, j])
            if g > 0.02 and c < 1 - 5e-3:
                ok = False
        bs = xe.validation.EOFBootstrapper(n_bootstraps=3, seed=seed).fit(m, random_state=1)
        refb = orc.eof_bootstrap(dict(eof, scores=eof["scores"]), r, n_bootstraps=3, seed=seed, random_state=1)
        if not np.allclose(bs.data["explained_variance"][:, :3], refb["explained_variance"][:, :3], rtol=2e-3):
            ok = False
    except Exception as e:
        ok = False
        print("  exception:", type(e).__name__, str(e)[:200])
    if not ok:
        bad += 1
        print(f"case {case}: n={n} p={p} r={r} k={k} power={power} seed={seed} FAILED")
print("cases", ncase, "bad", bad)
