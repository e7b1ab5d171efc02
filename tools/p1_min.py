import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine
ctx = engine.Context(0)
X = np.random.default_rng(1).standard_normal((50, 1)).astype(np.float32)
mat, st, U, s, V = engine.fit(ctx, X, 1, random_state=1)
print(s)
