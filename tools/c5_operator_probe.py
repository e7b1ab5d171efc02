"""Config 5 on the operator route, five calls (preprocess for the Hilbert stage + sum of squares of Im + eofx_rsvd_hilbert_c64):
python tools/c5_operator_probe.py [n nlat nlon k].  Under rocprofv3 --kernel-trace, tools/trace_gaps.py <dir> colstats_tr_kernel
prints the timeline of the last call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xeofs_amd import engine
import bench

n, nlat, nlon, k = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8000, 720, 1440, 20)
ctx = engine.Context(0)
X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
om = engine.sketch_matrix(n, k + 10, 5)
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True, for_hilbert=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sq = engine.hilbert_sumsq(ctx, A, "exp", 0.2)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    U, s, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, omega=om, device_out=True)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep{rep}: pre {1e3*(t1-t0):.2f} sumsq {1e3*(t2-t1):.2f} rsvd {1e3*(t3-t2):.2f} total {1e3*(t3-t0):.2f} ms  sq={sq!r} s[:3]={s[:3]}", flush=True)
    A.free()
