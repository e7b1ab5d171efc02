import time, numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(0)
A = rng.standard_normal((n, 2 * n)).astype(np.float32)
G = torch.as_tensor(A @ A.T, device="cuda")
for dt in (torch.float32, torch.float64):
    Gd = G.to(dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, U = torch.linalg.eigh(Gd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    w, U = torch.linalg.eigh(Gd)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(dt, "gpu eigh first %.2f s, second %.2f s" % (t1 - t0, t2 - t1), flush=True)
Gh = G.double().cpu().numpy()
t0 = time.perf_counter(); w2, U2 = np.linalg.eigh(Gh); t1 = time.perf_counter()
print("numpy eigh fp64 %.2f s" % (t1 - t0), flush=True)
import scipy.linalg as sla
t0 = time.perf_counter(); w3, U3 = sla.eigh(Gh, subset_by_index=[n - int(0.3 * n), n - 1]); t1 = time.perf_counter()
print("scipy eigh subset 30%% fp64 %.2f s" % (t1 - t0), flush=True)
t0 = time.perf_counter(); w3, U3 = sla.eigh(Gh.astype(np.float32), subset_by_index=[n - int(0.3 * n), n - 1]); t1 = time.perf_counter()
print("scipy eigh subset 30%% fp32 %.2f s" % (t1 - t0), flush=True)
print("max rel diff gpu64 vs numpy:", float(np.abs(w.cpu().numpy() - w2).max() / w2.max()))
import os; print("cpus", os.cpu_count())
