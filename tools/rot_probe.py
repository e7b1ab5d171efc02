"""Time Varimax / Promax on C4-sized loadings (p = 1,036,800 features, m modes): HIP path vs the oracle."""
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
for power in (1, 2):
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
Lw = engine.panel_width(m)
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
