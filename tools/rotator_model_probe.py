"""Model-level timing of EOFRotator.fit at config-4 scale (p = 1,036,800 features, 50 modes) on a model whose loadings
have simple structure (the noise-dominated benchmark field has no rotation optimum to converge to)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeofs_amd as xe
from xeofs_amd import engine

p, k, n = 720 * 1440, 50, 10000
rng = np.random.default_rng(0)
S = 0.1 * rng.standard_normal((p, k)).astype(np.float32)
S[np.arange(p), rng.integers(0, k, p)] += rng.uniform(1, 3, p).astype(np.float32)
Q = np.linalg.qr(rng.standard_normal((k, k)))[0].astype(np.float32)
V = np.linalg.qr(S @ Q)[0].astype(np.float32)                     # orthonormal "components" hiding simple structure
s = np.linspace(2000, 500, k)
ctx = engine.default_context()


class _Mat:                      # stands in for the resident input matrix (only .n is used by the rotator)
    n = 10000


m = xe.single.EOF(n_modes=k)
m.ctx = ctx
m.preprocessor = type("P", (), {})()
m.data = dict(input_data=_Mat(), components=V, scores=(rng.standard_normal((n, k)) * s).astype(np.float32), norms=s,
              explained_variance=s ** 2 / (n - 1), total_variance=float((s ** 2).sum() / (n - 1) * 1.2))
for power in (1, 2):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rot = xe.single.EOFRotator(n_modes=k, power=power).fit(m)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"EOFRotator(power={power}).fit at p = {p}, {k} modes: {1e3 * (t1 - t0):.0f} ms", flush=True)
