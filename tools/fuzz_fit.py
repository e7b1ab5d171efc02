"""Randomised parity sweep of the ONE-CALL fit (engine.fit / eofx_fit_f32: Scaler statistics during the first pass of the
randomized SVD) against the oracle: random shapes with n < p, standardisation, weights, land masks (all-NaN columns, with
and without allow_masked), NaN rows, uncentred offsets up to the fp16-overflow fallback, sketch widths on both sides of
the fused limit.  Prints failures and how often the fused pass / each fallback applied.  python tools/fuzz_fit.py seed ncases"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
how = collections.Counter()
for case in range(ncase):
    n = int(rng.integers(70, 900))
    p = int(rng.integers(n + 1, 6000)) // 4 * 4
    k = int(rng.integers(1, 53))
    if k + 10 >= n:
        k = max(1, n - 12)
    rank = int(rng.integers(2, 14))
    amp = 5.0 * rng.uniform(0.5, 0.95) ** np.arange(rank)
    offset = rng.choice([0.0, 3.0, 300.0, 3e4])
    X = ((rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p)) + rng.uniform(0.05, 1.0) * rng.standard_normal((n, p))
         + offset * rng.uniform(0.5, 1.5, size=p)).astype(np.float32)
    # round 5: the degenerate classes the robustness pass found -- no noise floor (more modes than numerical rank), mixed-unit
    # feature scales (8 .. 16 orders of magnitude; with standardize the fused pass must hand over to the two-step path)
    lowrank = rng.random() < 0.15
    if lowrank:
        X = (((rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p))) + offset * rng.uniform(0.5, 1.5, size=p)).astype(np.float32)
    mixed = rng.random() < 0.2
    if mixed:
        X = (X * 10.0 ** rng.uniform(-rng.choice([4.0, 6.0, 8.0]), 4.0, size=p)).astype(np.float32)
    std = bool(rng.integers(0, 2)); use_w = bool(rng.integers(0, 2))
    w = rng.uniform(0.2, 1.5, size=p) if use_w else None
    mask = rng.random() < 0.4
    if mask:
        X[:, rng.choice(p, size=max(1, int(p * rng.uniform(0.02, 0.5))), replace=False)] = np.nan
    rows = rng.random() < 0.15
    if rows:
        X[rng.choice(n, size=max(1, n // 25), replace=False), :] = np.nan
    allow = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1000))
    try:
        ref = orc.eof_fit(X.astype(np.float64), k, True, std, w, random_state=seed, solver="randomized")
        mat, st, U, s, V = engine.fit(ctx, X, k, True, std, w, random_state=seed, allow_masked=allow)
        info = engine.fit_info(ctx)
        how[("fused" if info["fused"] else f"fallback{info['reason']}") + ("+masked" if mat.masked else "")] += 1
        mat.free()
        so = ref["norms"]
        ok = np.all(np.abs(s - so) <= 1e-5 * so + 3e-6 * so[0])
        ok &= bool(np.isfinite(U).all() and np.isfinite(V).all())
        ok &= np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max() <= 3e-5 and np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max() <= 3e-5
        how["low rank" if lowrank else "-"] += 0
        if lowrank: how["low rank (k %s rank)" % (">" if k > rank else "<=")] += 1
        if mixed: how["mixed scales" + (" + standardize" if std else "")] += 1
        ok &= abs(st["total_variance"] - ref["total_variance"]) <= 1e-5 * ref["total_variance"]
        ok &= V.shape == ref["components"].shape and U.shape == ref["U"].shape
        ok &= bool(np.array_equal(st["valid_feature"], ~np.isnan(X).all(axis=0)))
        se = np.linalg.svd(ref["input_data"], compute_uv=False)[:k]
        for j in range(k):
            gap = min(abs(so[j] - so[j + 1]) / so[j] if j + 1 < k else 1, abs(so[j - 1] - so[j]) / so[j] if j else 1)
            if gap > 1e-2 and abs(so[j] - se[j]) < 1e-4 * se[j] and so[j] > 1e-4 * so[0]:      # (a numerically null mode has no direction)
                ok &= abs(np.dot(V[:, j].astype(np.float64), ref["components"][:, j])) >= 1 - 1e-5
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(n=n, p=p, k=k, std=std, w=use_w, mask=mask, rows=rows, allow=allow, offset=offset, seed=seed, lowrank=lowrank, mixed=mixed),
                  "max rel", np.max(np.abs(s - so) / so), "max |s - so| / s0", float(np.max(np.abs(s - so)) / so[0]),
                  "orth U V", float(np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max()), float(np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max()),
                  "shapes", V.shape, ref["components"].shape, "rank", rank, "s/s0 tail", (s[-3:] / so[0]).tolist(), (so[-3:] / so[0]).tolist(),
                  "tv", st["total_variance"], ref["total_variance"], info)
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p=p, k=k, mask=mask, rows=rows, allow=allow), type(e).__name__, str(e)[:160])
print("cases", ncase, "bad", bad, dict(how))
