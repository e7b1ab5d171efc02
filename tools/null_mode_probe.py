"""More modes asked for than the matrix has numerical rank: are both factors still orthonormal?  Real path (eofx_rsvd_f32 /
eofx_fit_f32), tall and wide, against what scikit-learn's randomized_svd returns (the oracle's bit-identical restatement)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

ctx = engine.Context(0)
rng = np.random.default_rng(3)
for n, p, r, k in ((300, 2000, 5, 10), (2000, 300, 5, 10), (300, 2000, 12, 30), (64, 5000, 3, 8), (1000, 1000, 7, 20)):
    X = (rng.standard_normal((n, r)) * 3.0 ** -np.arange(r)) @ rng.standard_normal((r, p)) + 4.0
    X = X.astype(np.float32)
    for name, run in (("rsvd", lambda: engine.rsvd(ctx, engine.preprocess(ctx, X, True, False, None)[0], k, random_state=1)),
                      ("fit ", lambda: engine.fit(ctx, X, k, random_state=1)[2:])):
        U, s, V = run()
        ou = np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max()
        ov = np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max()
        Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
        ref = orc.randomized_svd(Xc, k, random_state=1)
        ru = np.abs(ref[0].T @ ref[0] - np.eye(k)).max()
        rv = np.abs(ref[2] @ ref[2].T - np.eye(k)).max()
        print(f"{name} n {n} p {p} rank {r} k {k}: |U^T U - I| {ou:.2e} |V^T V - I| {ov:.2e}   (sklearn restatement: {ru:.1e} {rv:.1e})   s/s0 tail {(s[-2:] / s[0]).tolist()} ref {(ref[1][-2:] / ref[1][0]).tolist()}", flush=True)

print("# complex path (eofx_rsvd_c64): exactly low-rank complex matrices, tall and wide")
for n, p, r, k in ((300, 2000, 3, 8), (2000, 300, 3, 8), (200, 1500, 10, 24)):
    A_ = (rng.standard_normal((n, r)) * 2.0 ** -np.arange(r))
    Zc = (A_ @ (rng.standard_normal((r, p)) + 1j * rng.standard_normal((r, p))))
    Re, Im = np.ascontiguousarray(Zc.real, dtype=np.float32), np.ascontiguousarray(Zc.imag, dtype=np.float32)
    A = engine.from_dense(ctx, Re); B = engine.from_dense(ctx, Im)
    for rule in ("auto", "converge"):
        U, s, V = engine.rsvd_c64(ctx, A, B, k, random_state=1, n_iter=rule)
        ou = np.abs(U.conj().T @ U - np.eye(k)).max(); ov = np.abs(V.conj().T @ V - np.eye(k)).max()
        se = np.linalg.svd(Re.astype(np.float64) + 1j * Im.astype(np.float64), compute_uv=False)[:k]
        print(f"c64 {rule:8s} n {n} p {p} rank {r} k {k}: |U^H U - I| {ou:.2e} |V^H V - I| {ov:.2e}  s err (leading {r}) {np.abs(s - se)[:r].max() / se[0]:.1e}  tail {(s[-2:] / se[0]).tolist()}", flush=True)
    A.free(); B.free()
