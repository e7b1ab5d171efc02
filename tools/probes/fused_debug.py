"""Mapping check of the fused product on a structured input (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
n, p = 3000, 1024
ctx = engine.Context(0)
X = torch.zeros((n, p), device="cuda")
srow = int(sys.argv[1]) if len(sys.argv) > 1 else 0
X[srow] = torch.arange(1, p + 1, device="cuda", dtype=torch.float32)
mat = engine.from_dense(ctx, X)
Z = torch.zeros((mat.n_pad, 64), device="cuda")
Z[srow] = torch.arange(1, 65, device="cuda", dtype=torch.float32)
W, Y = engine.panel_fused(ctx, mat, Z, want_y=True)
Yr = engine.panel_tmul(ctx, mat, Z, prec="f32")
print("Y fused[:40,:20]\n", Y[:40, :20])
print("Y ref[:40,:20]\n", Yr[:40, :20])
bad = (Y - Yr).abs() > 1e-3 * Yr.abs().max()
print("bad entries:", int(bad.sum()), "of", Y.numel())
idx = bad.nonzero()[:20]
print(idx.tolist())
Wr = engine.panel_mul(ctx, mat, Yr, prec="f32")
print("W err", float((W - Wr).abs().max()) / float(Wr.abs().max()))
