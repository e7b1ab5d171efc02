"""Accuracy of the split-fp16 passes on a field dominated by one mode (coherent sums), in place vs written layouts.
X = T W S^T (as tools/hilbert_operator_probe3.py); products against float64 torch references on the device."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from xeofs_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 720 * 1440
r = 60
rng = np.random.default_rng(7)
T = np.empty((n, r))
e = rng.standard_normal((n, r))
T[0] = e[0]
for i in range(1, n):
    T[i] = 0.8 * T[i - 1] + 0.6 * e[i]
T[:, :4] += np.cumsum(rng.standard_normal((n, 4)), axis=0) * 0.05
W = 0.93 ** np.arange(r) * 3.0
dev = "cuda"
Td = torch.as_tensor((T * W).astype(np.float32), device=dev)
Sd = torch.randn((p, r), device=dev, dtype=torch.float32, generator=torch.Generator(dev).manual_seed(3))
X = torch.empty((n, p), dtype=torch.float32, device=dev)
for c0 in range(0, p, 65536):
    X[:, c0:c0 + 65536] = Td @ Sd[c0:c0 + 65536].T
ctx = engine.default_context(0)
L = 64
# panels: the leading right / left singular directions (coherent sums) + random columns
Tc = (Td.double() - Td.double().mean(0))
Yp_small = torch.linalg.qr(Sd.double())[0][:, :L // 2]                     # p x 32 in the row space
Yr = torch.randn((p, L // 2), device=dev, dtype=torch.float64) / p ** 0.5
def refs(A):
    Yp = torch.zeros((A.p_pad, L), device=dev, dtype=torch.float32)
    Yp[:p, :L // 2] = Yp_small.float()
    Yp[:p, L // 2:] = Yr.float()
    # exact X Y = Tc (S^T Y) through the factors, float64
    XY = Tc @ (Sd.double().T @ Yp[:p].double())
    Zn = torch.zeros((A.n_pad, L), device=dev, dtype=torch.float32)
    Zn[:n] = XY.float() / XY.abs().max().float()
    XtZ = Sd.double() @ (Tc.T @ Zn[:n].double())
    return Yp, XY, Zn, XtZ
for in_place in (True, False):
    A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=in_place)
    Yp, XY, Zn, XtZ = refs(A)
    for prec in ("f16x3", "f32"):
        Wn = engine.panel_mul(ctx, A, Yp, prec=prec)[:n].double()
        Zt = engine.panel_tmul(ctx, A, Zn, prec=prec)[:p].double()
        ew = ((Wn - XY).norm(dim=0) / XY.norm(dim=0))
        ez = ((Zt - XtZ).norm(dim=0) / XtZ.norm(dim=0))
        # signed bias of the coherent columns: <computed, exact> / <exact, exact> - 1
        bw = ((Wn * XY).sum(0) / (XY * XY).sum(0) - 1)
        bz = ((Zt * XtZ).sum(0) / (XtZ * XtZ).sum(0) - 1)
        print(f"in_place={in_place} {prec}: X Y   relerr coherent cols max {ew[:32].max():.2e} random cols max {ew[32:].max():.2e} | bias col0 {bw[0]:+.2e} mean {bw[:32].mean():+.2e}")
        print(f"in_place={in_place} {prec}: X^T Z relerr coherent cols max {ez[:32].max():.2e} random cols max {ez[32:].max():.2e} | bias col0 {bz[0]:+.2e} mean {bz[:32].mean():+.2e}", flush=True)
    A.free()
