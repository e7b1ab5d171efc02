#!/bin/bash
# round 6, job b: the value-based "converge" rule -- R9 evidence table on the 8000 x 65536 field, and config 5 at full size
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
EOFX_C64_TRACE=1 python tools/r9_evidence.py --exact-from profiles/r06_r9_exact_8000x65536.json --json gpurun_out/r06_r9_evidence_b.json > gpurun_out/r06_r9_evidence_b.txt 2> gpurun_out/r06_r9_evidence_b.err
grep "after\|Rayleigh-Ritz over\|thick" gpurun_out/r06_r9_evidence_b.err | tail -30; cat gpurun_out/r06_r9_evidence_b.txt
python - <<'PY' 2>&1 | grep -v Warning | tail -20
import time, torch, numpy as np, os
import bench
from xeofs_amd import engine
ctx = engine.default_context(0)
n, nlat, nlon, k = 8000, 720, 1440, 20
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=1)
for rule in ("auto", "converge", "converge"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    U, s, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=rule, device_out=True)
    torch.cuda.synchronize(); t = 1e3 * (time.perf_counter() - t0)
    print(rule, "products", engine.last_iterations(ctx), f"{t:.1f} ms", "s[:3]", s[:3], "s[-3:]", s[-3:])
    if rule == "auto": s_auto = s.copy()
    else: print("   max rel change vs auto", float(np.max(np.abs(s - s_auto) / s)))
U, s20, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=20, device_out=True)
print("20 products: max rel diff of converge vs 20 products", float(np.max(np.abs(s - s20) / s20)), "per mode", np.abs(s - s20) / s20)
PY
