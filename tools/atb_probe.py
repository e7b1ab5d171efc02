"""Micro-probe of the dominant kernel: times X^T Z (tmul) and X Y (mul) on a random resident
matrix with the in-library HIP-event hook.  Usage: python tools/atb_probe.py n p L [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine

n, p, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = engine.Context(0)
X = torch.randn((n, p), device="cuda", dtype=torch.float32)
if os.environ.get("PROBE_LAYOUT", "") in ("raw", "inplace"):   # the raw view / in-place kernels (f16x3 only)
    mat, _ = engine.preprocess(ctx, X, center=True, want_stats=False, keep_raw=True, in_place=os.environ["PROBE_LAYOUT"] == "inplace")
else:
    mat = engine.from_dense(ctx, X)
    del X
Z = torch.randn((mat.n_pad, L), device="cuda"); Z[n:] = 0
Y = torch.randn((mat.p_pad, L), device="cuda"); Y[p:] = 0
precs = sys.argv[5].split(",") if len(sys.argv) > 5 else ["f32", "bf16x3", "bf16x6"]
for prec in precs:
  for name, fn, arg in (("tmul X^T Z", engine.panel_tmul, Z), ("mul  X Y  ", engine.panel_mul, Y)):
    fn(ctx, mat, arg, prec=prec); torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(reps):
        fn(ctx, mat, arg, prec=prec)
    pr = ctx.profile_read(); ctx.profile(False)
    ms = pr["ms"] / pr["launches"]
    print(f"{prec:7s} {name}: n={n} p={p} L={L}  {ms:.3f} ms/launch  alg(l=L) {2.0*n*p*L/ms/1e9:.1f} TF/s  "
          f"A-stream {n*p*4.0/ms/1e6:.0f} GB/s")
