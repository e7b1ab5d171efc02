"""Randomised parity sweep of the complex path in the LEAN layout (Re = raw field in place through the Scaler map, Im = Hilbert
output in its sample-contiguous layout only, or both parts of a complex input in place): eofx_hilbert_f32 + eofx_rsvd_c64
against the exact complex SVD of the analytic signal the oracle builds, random shapes on both sides of n = p, weights,
standardisation, sketch widths up to 32 complex columns (the lean route) and beyond (written layouts on demand).
python tools/fuzz_complex.py seed ncases"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
how = collections.Counter()
for case in range(ncase):
    n = int(rng.integers(40, 900))
    p = int(rng.integers(40, 3000)) // 4 * 4
    r = min(n, p)
    k = int(rng.integers(1, min(r - 11, 40)))
    std = bool(rng.integers(0, 2)); use_w = bool(rng.integers(0, 2))
    w = rng.uniform(0.3, 1.5, size=p) if use_w else None
    padding = "exp" if rng.random() < 0.6 else None
    t = np.arange(n)[:, None]; x = np.linspace(0, 2 * np.pi, p)[None, :]
    X = 0.002 * rng.standard_normal((n, p))
    for j in range(k + 4):
        X += 6.0 * rng.uniform(0.75, 0.9) ** j * np.cos((0.05 + 0.043 * j) * t - (1 + j % 7) * x + 0.3 * j)
    X = (X + rng.uniform(-50, 50) + rng.standard_normal(p)).astype(np.float32)
    seed = int(rng.integers(0, 1000))
    try:
        pre = orc.preprocess(X.astype(np.float64), True, std, w)
        Z = pre["X"] + 1j * orc.hilbert_transform(pre["X"], padding=padding, decay_factor=0.2).imag
        A, st = engine.preprocess(ctx, X, True, std, w, in_place=True)
        B, _ = engine.hilbert(ctx, A, padding, 0.2)
        lean = A.layout() == (False, True) and not B.layout()[0]
        U, s, V = engine.rsvd_c64(ctx, A, B, k, random_state=seed)
        kept = A.layout() == (False, True) and not B.layout()[0]
        how[("lean" if kept else "written on demand") + (" k+10>32" if k + 10 > 32 else "")] += 1
        A.free(); B.free()
        sall = np.linalg.svd(Z, compute_uv=False)
        se = sall[:k]
        # the randomized solver resolves a mode to 1e-5 once it stands clear of the spectrum beyond the sketch (4 or 7 power
        # iterations, sklearn's rule); modes inside a flat noise bulk are only as good as the iteration count (the
        # reference's svds(lobpcg) would polish them): checked loosely
        clear = se > 4.0 * sall[min(k + 10, len(sall) - 1)]
        ok = np.all(np.abs(s - se)[clear] <= 1e-5 * se[clear] + 3e-6 * se[0]) and np.all(np.abs(s - se) <= 0.1 * se)
        ok &= np.abs(U.conj().T @ U - np.eye(k)).max() < 3e-5 and np.abs(V.conj().T @ V - np.eye(k)).max() < 3e-5
        rec = (U.astype(np.complex128) * s) @ V.astype(np.complex128).conj().T
        Ue, sf, Vhe = np.linalg.svd(Z, full_matrices=False)
        best = (Ue[:, :k] * sf[:k]) @ Vhe[:k]
        ok &= np.linalg.norm(Z - rec) <= np.linalg.norm(Z - best) * (1 + (1e-4 if clear.all() else 5e-2)) + 1e-6 * np.linalg.norm(Z)
        ok &= bool(lean) and (kept == (k + 10 <= 32))
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(n=n, p=p, k=k, std=std, w=use_w, padding=padding, seed=seed), "max rel", float(np.max(np.abs(s - se) / se)), lean, kept)
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p=p, k=k, padding=padding), type(e).__name__, str(e)[:160])
print("cases", ncase, "bad", bad, dict(how))
