"""Check and time the experimental fused power-iteration product W = X (X^T Z) against mul(tmul(Z))."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine

n, p = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = engine.Context(0)
X = torch.randn((n, p), device="cuda", dtype=torch.float32)
mat = engine.from_dense(ctx, X)
del X
Z = torch.randn((mat.n_pad, 64), device="cuda"); Z[n:] = 0
Z = Z / Z.norm(dim=0)
ref = engine.panel_mul(ctx, mat, engine.panel_tmul(ctx, mat, Z, prec="f32"), prec="f32")
two = engine.panel_mul(ctx, mat, engine.panel_tmul(ctx, mat, Z, prec="f16x3"), prec="f16x3")
got, Yg = engine.panel_fused(ctx, mat, Z, want_y=True)
Yr = engine.panel_tmul(ctx, mat, Z, prec="f32")
print(f"max |Y fused - Y ref| / max = {float((Yg - Yr).abs().max()) / float(Yr.abs().max()):.3e}")
torch.cuda.synchronize()
sc = float(ref.abs().max())
print(f"max |fused - f32 ref| / max = {float((got - ref).abs().max()) / sc:.3e};  two-pass f16x3 vs ref = {float((two - ref).abs().max()) / sc:.3e}")
for name, fn in (("two-pass", lambda: engine.panel_mul(ctx, mat, engine.panel_tmul(ctx, mat, Z, prec="f16x3"), prec="f16x3")),
                 ("fused   ", lambda: engine.panel_fused(ctx, mat, Z))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {1e3 * dt:.3f} ms per W = X (X^T Z)   ({n * p * 4.0 / dt / 1e9:.0f} GB/s of matrix per product pair)")
