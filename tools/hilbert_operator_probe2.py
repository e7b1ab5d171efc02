"""n < p, in place (the layout of config 5): operator route vs two-part route vs exact (eigenvalues of Z Z^H in float64)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import eof_oracle as orc
from xeofs_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ctx = engine.default_context(0)
k = 20
rng = np.random.default_rng(n)
e = rng.standard_normal((n, p))
X = np.empty((n, p))
X[0] = e[0]
for i in range(1, n):
    X[i] = 0.8 * X[i - 1] + 0.6 * e[i]
del e
for j in range(24):
    t = np.cumsum(rng.standard_normal(n)) * 0.05 + np.sin(2 * np.pi * (j + 1) * np.arange(n) / n * 3.3 + j)
    X += (24 - j) * 0.6 * np.outer(t, rng.standard_normal(p))
X = X.astype(np.float32)
Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
Z = orc.hilbert_transform(Xc, padding="exp", decay_factor=0.2)
w = np.linalg.eigvalsh(Z @ Z.conj().T)[::-1][:k]
se = np.sqrt(w)
import torch
Xd = torch.as_tensor(X, device="cuda")
for in_place in (False, True):
    A0, _ = engine.preprocess(ctx, Xd, True, False, None, in_place=in_place, for_hilbert=in_place)
    B0, _ = engine.hilbert(ctx, A0, "exp", 0.2)
    for rule in ("auto", "converge"):
        _, s2, _ = engine.rsvd_c64(ctx, A0, B0, k, random_state=5, n_iter=rule)
        _, s1, _ = engine.rsvd_hilbert_c64(ctx, A0, k, "exp", 0.2, random_state=5, n_iter=rule)
        print(f"n={n} p={p} in_place={in_place} {rule}: two-part {np.abs(s2 - se).max() / se[0]:.2e} (per mode {(np.abs(s2 - se) / se).max():.2e}) "
              f"operator {np.abs(s1 - se).max() / se[0]:.2e} (per mode {(np.abs(s1 - se) / se).max():.2e}) diff {np.abs(s1 - s2).max() / se[0]:.2e}", flush=True)
    A0.free(); B0.free()
print("s", se[:3], se[-1])
