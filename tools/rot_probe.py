"""Time Varimax / Promax on C4-sized loadings (p = 1,036,800 features, m modes): HIP path vs the oracle.
`rot_probe.py p m steps`: only the per-iteration step kernel (any m <= 256), next to the same step as three float64
library GEMMs over blocks of rows (what panels wider than 64 used before round 3)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xeofs_amd import engine, rotation
from oracle import eof_oracle as orc

p = int(sys.argv[1]) if len(sys.argv) > 1 else 1036800
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(0)
S = 0.1 * rng.standard_normal((p, m))
S[np.arange(p), rng.integers(0, m, p)] += rng.uniform(1, 3, p)
Q = np.linalg.qr(rng.standard_normal((m, m)))[0]
L = (S @ np.diag(np.linspace(2, 1, m)) @ Q).astype(np.float32)
ctx = engine.default_context()
import torch
steps_only = len(sys.argv) > 3 and sys.argv[3] == "steps"
for power in (() if steps_only else (1, 2)):
    rotation.promax(ctx, L, power=power)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Xr, R, phi = rotation.promax(ctx, L, power=power)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"gpu promax power={power}: {1e3*(t1-t0):.1f} ms", flush=True)
    t0 = time.perf_counter()
    Xo, Ro, phio = orc.promax(L.astype(np.float64), power=power)
    t1 = time.perf_counter()
    print(f"cpu oracle power={power}: {1e3*(t1-t0):.1f} ms; max|dR|={np.abs(R-Ro).max():.2e} max|dX|/max={np.abs(Xr-Xo).max()/np.abs(Xo).max():.2e}", flush=True)
# iteration count + per-step kernel time
Lw = rotation._rot_width(m)
P = engine.panel_import(ctx, L, (p + 511) // 512 * 512, Lw)
Xn = engine.panel_row_normalize(ctx, P)
Rd = torch.eye(Lw, dtype=torch.float64, device="cuda"); aux = torch.zeros(Lw, dtype=torch.float64, device="cuda")
engine.panel_rot_step(ctx, Xn, Rd, aux, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    engine.panel_rot_step(ctx, Xn, Rd, aux, 0)
torch.cuda.synchronize(); t1 = time.perf_counter()
dt = (t1 - t0) / 20
print(f"rot_step: {1e3*dt:.3f} ms/iter; panel {p*Lw*4/1e6:.0f} MB -> {p*Lw*4/dt/1e9:.0f} GB/s; fp64 {4*p*Lw*Lw/dt/1e12:.2f} TFLOP/s")


def library_step(X, R, aux, block=1 << 17):
    G = torch.zeros((Lw, Lw), dtype=torch.float64, device=X.device)
    for r0 in range(0, X.shape[0], block):
        x = X[r0:r0 + block].double()
        b = x @ R
        G += x.T @ (b * (b * b - aux))
    return G

if steps_only:
    Rd = torch.as_tensor(np.linalg.qr(rng.standard_normal((Lw, Lw)))[0], device="cuda")
    aux = torch.full((Lw,), 1e-7, dtype=torch.float64, device="cuda")
    G = engine.panel_rot_step(ctx, Xn, Rd, aux, 0)
    Gl = library_step(Xn, Rd, aux)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        Gl = library_step(Xn, Rd, aux)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"library GEMM step: {1e3*(t1-t0)/5:.3f} ms/iter; max|G - G_lib| / max|G| = {float((G-Gl).abs().max()/Gl.abs().max()):.2e}")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        engine.panel_rot_step(ctx, Xn, Rd, aux, 2)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"rot_step complex (mode 2, {Lw//2} | {Lw//2}): {1e2*(t1-t0):.3f} ms/iter")
