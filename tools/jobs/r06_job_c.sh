#!/bin/bash
# round 6, job c: kernel trace of one `n_iter="converge"` call at config-5 size (where do the ~15 ms per product go?)
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
mkdir -p gpurun_out
cat > /tmp/c5conv.py <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import torch, numpy as np
import bench
from xeofs_amd import engine
ctx = engine.default_context(0)
n, nlat, nlon, k = 8000, 720, 1440, 20
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=1)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    U, s, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter="converge", device_out=True)
    torch.cuda.synchronize(); print("converge", engine.last_iterations(ctx), 1e3 * (time.perf_counter() - t0), flush=True)
PY
EOFX_C64_TRACE=1 python /tmp/c5conv.py 2>&1 | grep "converge\|after\|rate\|thick\|host Rayleigh" | tail -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 -- python /tmp/c5conv.py 2>&1 | grep "^converge"
cd "$ROOT"
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
head -40 "$f" > gpurun_out/r06_c5_converge_kernel_stats.csv
cut -c1-170 gpurun_out/r06_c5_converge_kernel_stats.csv
