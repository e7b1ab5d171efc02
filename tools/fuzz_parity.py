"""Randomised parity sweep: HIP path vs oracle on random shapes / options.  Prints failures.
FUZZ_LAYOUT=inplace|raw: ask for that layout policy (half of the cases then have P % 4 == 0 and no NaN, so that it applies;
the others fall back to the two-layout mode as the engine documents)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine
ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
layout = os.environ.get("FUZZ_LAYOUT", "")
lkw = {"inplace": {"in_place": True}, "raw": {"keep_raw": True}}.get(layout, {})
applied = 0
for case in range(ncase):
    n = int(rng.integers(12, 700)); p = int(rng.integers(12, 3000))
    clean = bool(lkw) and rng.random() < 0.5
    if clean:
        p = max(12, p // 4 * 4)
    r = min(n, p)
    k = int(rng.integers(1, max(2, min(r - 1, 40))))
    rank = int(rng.integers(2, 12))
    amp = 5.0 * rng.uniform(0.5, 0.95) ** np.arange(rank)
    X = ((rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p)) + rng.uniform(0.05, 1.0) * rng.standard_normal((n, p))
         + rng.uniform(-300, 300)).astype(np.float32)
    center = True; std = bool(rng.integers(0, 2)); use_w = bool(rng.integers(0, 2))
    w = rng.uniform(0.2, 1.5, size=p) if use_w else None
    if rng.random() < 0.5 and not clean: X[:, rng.choice(p, size=max(1, p // 7), replace=False)] = np.nan
    if rng.random() < 0.3 and not clean: X[rng.choice(n, size=max(1, n // 20), replace=False), :] = np.nan
    seed = int(rng.integers(0, 1000))
    try:
        ref = orc.eof_fit(X.astype(np.float64), k, center, std, w, random_state=seed, solver="randomized")
        mat, st = engine.preprocess(ctx, X, center, std, w, **lkw)
        applied += int(mat.layout()[1])
        U, s, V = engine.rsvd(ctx, mat, k, random_state=seed)
        mat.free()
        so = ref["norms"]
        se = np.linalg.svd(ref["input_data"], compute_uv=False)[:k]
        ok = np.all(np.abs(s - so) <= 1e-5 * so + 3e-6 * so[0])
        ok &= abs(st["total_variance"] - ref["total_variance"]) <= 1e-5 * ref["total_variance"]
        ok &= V.shape == ref["components"].shape and U.shape == ref["U"].shape
        for j in range(k):
            gap = min(abs(so[j] - so[j + 1]) / so[j] if j + 1 < k else 1, abs(so[j - 1] - so[j]) / so[j] if j else 1)
            if gap > 1e-2 and abs(so[j] - se[j]) < 1e-4 * se[j]:      # separated AND converged in the oracle itself
                ok &= abs(np.dot(V[:, j].astype(np.float64), ref["components"][:, j])) >= 1 - 1e-5
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(n=n, p=p, k=k, std=std, w=use_w, seed=seed), "max rel", np.max(np.abs(s - so) / so))
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p=p, k=k), type(e).__name__, str(e)[:120])
print("cases", ncase, "bad", bad, f"(layout {layout}: applied in {applied} cases)" if lkw else "")
