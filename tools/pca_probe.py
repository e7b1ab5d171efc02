"""Time config 3 with the reference's *default* arguments: xe.cross.MCA(n_modes=20, use_pca=True,
n_pca_modes=0.999) on two 5000 x (360 x 360) halves -- PCA pre-reduction (xeofs_amd/pca.py) + the analysis
on the PC scores, step by step; and the CPU oracle's PCA (randomized SVD of width 0.3 n) on a bounded sample."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
from xeofs_amd.pca import ResidentPCA
import bench

n, nlat, nlon, k = int(os.environ.get("N", 5000)), 360, 720, 20
ctx = engine.Context(0)
F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0")).reshape(n, nlat, nlon)
X = F[:, :, :360].reshape(n, -1).contiguous(); Y = F[:, :, 360:].reshape(n, -1).contiguous()
sync = torch.cuda.synchronize


def T(label, fn):
    sync(); t0 = time.perf_counter(); out = fn(); sync()
    print(f"  {label}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    return out


for rep in range(2):
    print(f"rep {rep}")
    sync(); t00 = time.perf_counter()
    (mx, sx), (my, sy) = T("preprocess x2", lambda: (engine.preprocess(ctx, X, want_stats=False), engine.preprocess(ctx, Y, want_stats=False)))
    p1 = T("ResidentPCA.fit X (all steps)", lambda: ResidentPCA(ctx, 0.999).fit(mx, sx["total_variance"]))
    p2 = T("ResidentPCA.fit Y (all steps)", lambda: ResidentPCA(ctx, 0.999).fit(my, sy["total_variance"]))
    print(f"  kept modes: {p1.m}, {p2.m} of n_pre={int(0.3 * min(n, mx.p))}; explained {float((p1.s**2).sum()/(n-1)/sx['total_variance']):.5f}")
    wx = T("scores -> resident", lambda: (engine.from_dense(ctx, p1.scores().astype(np.float32)), engine.from_dense(ctx, p2.scores().astype(np.float32))))
    out = T("crosscov rsvd on PC scores", lambda: engine.crosscov_rsvd(ctx, wx[0], wx[1], k, random_state=5))
    c = T("back-projection V Q x2", lambda: (p1.back_project(out["Q1"]), p2.back_project(out["Q2"])))
    sync(); print(f"  TOTAL default-args MCA fit: {1e3 * (time.perf_counter() - t00):.1f} ms; s[:3]={out['s'][:3]}", flush=True)
    # the two dominant steps inside ResidentPCA.fit, timed on their own (NOT part of the total above)
    G = T("[inside PCA.fit] gram X X^T (5120^2, f16x3)", lambda: mx.gram(0))
    if getattr(p1, "solver_used", "") == "randomized":
        pr = ResidentPCA(ctx, 0.999)
        Qr, GQ, Gm = T("[inside PCA.fit] randomized range finder on G: 5 x (G Q, Gram, blocked Cholesky-QR) + G Q", lambda: pr._range_randomized(G, n, 1510))
        Bw = T("[inside PCA.fit] B = X^T Q (129600 x 1536, NT kernel over transposed planes)", lambda: engine.panel_tmul(ctx, mx, Qr, prec="f16x3"))
        Mw = T("[inside PCA.fit] Gram of B (fp64 MFMA)", lambda: engine.panel_gram(ctx, Bw))
        T("[inside PCA.fit] eigh of order 1510 (rocSOLVER: the one library call)", lambda: torch.linalg.eigh(Mw[:1510, :1510]))
        Gm.free(); del Qr, GQ, Bw, Mw
    lam = T("[inside PCA.fit] eigh n x n fp64 (rocSOLVER)", lambda: torch.linalg.eigh(G[:n, :n].double()))
    del G, lam
    ref = engine.crosscov_rsvd(ctx, mx, my, k, random_state=5)
    print(f"  vs use_pca=False: s rel diff {np.abs(out['s'] - ref['s']).max() / ref['s'][0]:.2e}; min |cos| comps1 "
          f"{np.abs(np.sum(c[0] * ref['Q1'], axis=0)).min():.6f}")
    for w in wx:
        w.free()
    mx.free(); my.free(); del p1, p2

if os.environ.get("CPU", "1") == "1":
    from oracle import eof_oracle as orc
    ns, ps = 1000, 129600 // 5          # bounded sample: cost ~ n^2 p -> (5000/1000)^2 * 5 = 125x smaller
    Xs = X[:ns, :ps].cpu().numpy().astype(np.float64); Xs -= Xs.mean(0)
    t0 = time.perf_counter(); orc.pca_fit(Xs, 0.999, 0.3, random_state=0); dt = time.perf_counter() - t0
    print(f"CPU oracle PCA (randomized SVD k=0.3n, fp64) on {ns}x{ps}: {dt:.2f} s -> x125 = {125 * dt:.0f} s per field at config 3")
