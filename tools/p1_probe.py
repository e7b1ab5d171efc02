import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine
ctx = engine.Context(0)
rng = np.random.default_rng(1)
for shape, k in (((50, 1), 1), ((50, 2), 1), ((50, 3), 2), ((1, 30), 1)):
    X = rng.standard_normal(shape).astype(np.float32)
    print("trying", shape, k, flush=True)
    try:
        mat, st, U, s, V = engine.fit(ctx, X, k, random_state=1)
        print("  ok", s, engine.fit_info(ctx), flush=True)
        mat.free()
    except Exception as e:
        print("  raised", type(e).__name__, str(e)[:150], flush=True)
    try:
        mat, st = engine.preprocess(ctx, X)
        U, s, V = engine.rsvd(ctx, mat, k, random_state=1)
        print("  two-step ok", s, flush=True)
        mat.free()
    except Exception as e:
        print("  two-step raised", type(e).__name__, str(e)[:150], flush=True)
