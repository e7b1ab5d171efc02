import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine
ctx = engine.Context(0)
rng = np.random.default_rng(7)
X = rng.normal(5, 3, size=(25, 20)) + 2 * np.sin(np.linspace(0, 2 * np.pi, 25))[:, None]
X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
se = np.linalg.svd(X.astype(np.float64), compute_uv=False)
ref = None; bad = 0
for it in range(300):
    mat = engine.from_dense(ctx, X)
    U, s, V = engine.rsvd(ctx, mat, 20, 0, 0, random_state=None if False else 3)
    mat.free()
    if ref is None: ref = (U, s, V)
    same = all(np.array_equal(a, b) for a, b in zip(ref, (U, s, V)))
    err = np.abs(s - se).max() / se[0]
    if not same or err > 1e-5:
        bad += 1
        if bad < 6: print("iter", it, "bitwise same:", same, "rel err vs exact:", err, s[:3], se[:3])
print("bad", bad, "of 300; err of ref", np.abs(ref[1]-se).max()/se[0])
