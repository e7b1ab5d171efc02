# GPU check of the blocked device Cholesky-inverse against numpy, and timing
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from xeofs_amd import engine
ctx = engine.Context(0)
rng = np.random.default_rng(0)
for (rows, l) in ((3000, 60), (3000, 100), (4000, 200), (5120, 700), (5120, 1510)):
    L = (l + 31) // 32 * 32
    P = np.zeros((rows, L), np.float32)
    P[:, :l] = rng.standard_normal((rows, l)) * (1.0 + 9.0 * rng.random(l))
    if l > 80:
        P[:, 77] = P[:, 3] + P[:, 5]          # an exactly dependent column
    Pd = torch.as_tensor(P, device="cuda")
    G = engine.panel_gram(ctx, Pd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = engine.panel_rinv(ctx, G, l)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    Q = engine.panel_cholqr(ctx, Pd, l, G)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    Qh = Q.double().cpu().numpy()[:, :l]
    QtQ = Qh.T @ Qh
    dead = np.where(np.abs(np.diag(QtQ)) < 0.5)[0]
    keep = np.setdiff1d(np.arange(l), dead)
    err = np.abs(QtQ[np.ix_(keep, keep)] - np.eye(keep.size)).max()
    # same span
    resid = P[:, keep].astype(np.float64) - Qh[:, keep] @ (Qh[:, keep].T @ P[:, keep].astype(np.float64))
    print(f"rows {rows} l {l}: rinv {1e3*(t1-t0):.2f} ms, cholqr {1e3*(t2-t1):.2f} ms, dead columns {dead.tolist()}, max |Q^T Q - I| {err:.2e}, span residual {np.abs(resid).max():.2e}")
