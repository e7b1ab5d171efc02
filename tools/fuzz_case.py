import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine
ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for case in range((int(sys.argv[2]) if len(sys.argv) > 2 else 40) + 1):
    n = int(rng.integers(12, 700)); p = int(rng.integers(12, 3000)); r = min(n, p)
    k = int(rng.integers(1, max(2, min(r - 1, 40)))); rank = int(rng.integers(2, 12))
    amp = 5.0 * rng.uniform(0.5, 0.95) ** np.arange(rank)
    X = ((rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p)) + rng.uniform(0.05, 1.0) * rng.standard_normal((n, p)) + rng.uniform(-300, 300)).astype(np.float32)
    std = bool(rng.integers(0, 2)); use_w = bool(rng.integers(0, 2))
    w = rng.uniform(0.2, 1.5, size=p) if use_w else None
    if rng.random() < 0.5: X[:, rng.choice(p, size=max(1, p // 7), replace=False)] = np.nan
    if rng.random() < 0.3: X[rng.choice(n, size=max(1, n // 20), replace=False), :] = np.nan
    seed = int(rng.integers(0, 1000))
print(n, p, k, rank, std, use_w, seed)
ref = orc.eof_fit(X.astype(np.float64), k, True, std, w, random_state=seed, solver="randomized")
so = ref["norms"]; se = np.linalg.svd(ref["input_data"], compute_uv=False)[:k]
print("oracle vs exact  max rel", np.max(np.abs(so - se) / se), "per mode", np.round(np.abs(so - se) / se, 6)[-8:])
o32 = orc.randomized_svd(ref["input_data"].astype(np.float32), k, random_state=seed)[1]
print("sklearn-fp32 vs oracle-fp64 max rel", np.max(np.abs(o32 - so) / so))
for prec in [("f32", "f32"), ("bf16x3", "bf16x6"), ("bf16x6", "bf16x6"), ("f16x3", "f16x3")]:
    ctx.set_precision(*prec)
    mat, st = engine.preprocess(ctx, X, True, std, w)
    U, s, V = engine.rsvd(ctx, mat, k, random_state=seed)
    print(prec, "gpu vs oracle max rel", np.max(np.abs(s - so) / so), "gpu vs exact", np.max(np.abs(s - se) / se), np.round(np.abs(s - so) / so, 6)[-6:])
