"""Soak: a few hundred fits of random shapes and models on one context, one after the other -- no error, no drift of the results
of a reference fit that is repeated in between, device memory bounded."""
import sys, os, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
import xeofs_amd as xe

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = engine.default_context(0)
Xref = (rng.standard_normal((500, 6)) @ rng.standard_normal((6, 4096)) + 0.2 * rng.standard_normal((500, 4096))).astype(np.float32)
def ref_fit():
    mat, st, U, s, V = engine.fit(ctx, Xref, 5, random_state=3)
    mat.free()
    return s.copy(), V.copy()
s0, V0 = ref_fit()
bad = 0
t0 = time.time()
peak = 0
for i in range(N):
    n = int(rng.integers(8, 1500)); a = int(rng.integers(1, 40)); b = int(rng.integers(1, 120)); k = int(rng.integers(1, max(2, min(n, a * b, 40) - 1)))
    X = (rng.standard_normal((n, 4)) @ rng.standard_normal((4, a * b)) + 0.3 * rng.standard_normal((n, a * b)) + rng.uniform(-5, 5)).astype(np.float32)
    if rng.random() < 0.2:
        X[:, rng.integers(0, a * b)] = np.nan
    d = xe.DataArray(X.reshape(n, a, b), dims=("time", "lat", "lon"))
    kind = int(rng.integers(0, 5))
    try:
        if kind == 0:
            xe.single.EOF(n_modes=k, standardize=bool(rng.integers(0, 2)), random_state=1).fit(d, "time").components()
        elif kind == 1:
            xe.single.HilbertEOF(n_modes=min(k, 30), random_state=1).fit(d, "time").scores()
        elif kind == 2:
            Y = xe.DataArray((X[:, ::-1] * 0.5 + 1).reshape(n, a, b).copy(), dims=("time", "lat", "lon"))
            xe.cross.MCA(n_modes=min(k, 20), use_pca=bool(rng.integers(0, 2)), n_pca_modes=0.9, random_state=1).fit(d, Y, "time").singular_values()
        elif kind == 3:
            kk, pw = max(2, min(k, 12)), int(rng.integers(1, 3))
            m = xe.single.EOF(n_modes=kk, random_state=1).fit(d, "time")
            try:
                xe.single.EOFRotator(n_modes=kk, power=pw).fit(m).components()
            except RuntimeError as e:       # "Rotation process did not converge.": does the reference's loop on the same loadings?
                from oracle import eof_oracle as orc
                Xv = X[:, ~np.isnan(X).all(axis=0)].astype(np.float64)
                ref = orc.eof_fit(Xv, kk, random_state=1)
                try:
                    orc.eof_rotator_fit(ref, kk, power=pw)
                    bad += 1; print("case", i, "rotator raised", str(e)[:60], "but the oracle's loop converged", (n, a, b, kk, pw), flush=True)
                except RuntimeError as e2:
                    print("case", i, "rotator and oracle both:", str(e2)[:60], (n, a, b, kk, pw), flush=True)
        else:
            mat, st = engine.preprocess(ctx, np.nan_to_num(X, nan=0.5), True, False, None, in_place=bool(rng.integers(0, 2)))
            engine.rsvd(ctx, mat, k, random_state=2); mat.free()
    except ValueError as e:
        if "rank" not in str(e) and "modes" not in str(e):
            bad += 1; print("case", i, kind, (n, a, b, k), type(e).__name__, str(e)[:120], flush=True)
    except Exception as e:
        bad += 1; print("case", i, kind, (n, a, b, k), type(e).__name__, str(e)[:120], flush=True)
    if i % 25 == 24:
        s1, V1 = ref_fit()
        same = np.array_equal(s0, s1) and np.array_equal(V0, V1)
        free, total = torch.cuda.mem_get_info()
        peak = max(peak, total - free)
        if not same:
            bad += 1
        print(f"after {i + 1} fits: reference fit bitwise {'same' if same else 'DIFFERENT'}; device memory in use {(total - free) / 2**30:.2f} GiB; {time.time() - t0:.0f} s", flush=True)
print("fits", N, "bad", bad, f"peak device memory {peak / 2**30:.2f} GiB")
