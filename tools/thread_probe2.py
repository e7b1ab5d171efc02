"""Which part of the Hilbert stage differs under concurrent contexts: download Im from worker threads and compare with serial."""
import sys, os, threading, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
warnings.simplefilter("ignore")
rng = np.random.default_rng(0)
fields = [(rng.standard_normal((n, 5)) @ rng.standard_normal((5, p)) + 0.2 * rng.standard_normal((n, p)) + 1.0).astype(np.float32)
          for n, p in ((400, 4096), (700, 2048), (300, 8192), (1000, 1000))]
MODE = os.environ.get("MODE", "rawT")
def run(ctx, X):
    A, _ = engine.preprocess(ctx, X, True, False, None, in_place=True, for_hilbert=(MODE == "rawT"))
    B, _ = engine.hilbert(ctx, A, "exp", 0.2)
    im = B.download()
    A.free(); B.free()
    return im
ctx0 = engine.Context(0)
serial = [run(ctx0, X) for X in fields]
bad = []
def worker(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = engine.Context(0)
        VICTIM = os.environ.get("VICTIM", "all")
        kept = {}
        for rep in range(40):
            i = (rep + tid) % len(fields)
            if VICTIM == "fft":          # the matrix is made once: only the transform kernel and the download run under the load
                if i not in kept:
                    kept[i] = engine.preprocess(ctx, fields[i], True, False, None)[0]
                    ctx.synchronize()
                B, _ = engine.hilbert(ctx, kept[i], "exp", 0.2)
                im = B.download(); B.free()
            elif VICTIM == "apply":      # statistics + apply kernel + download, no transform
                A = engine.preprocess(ctx, fields[i], True, False, None)[0]
                im = A.download(); A.free()
                if "ap" not in kept:
                    kept["ap"] = {}
                if i not in kept["ap"]:
                    kept["ap"][i] = im
                    continue
                if not np.array_equal(im, kept["ap"][i]):
                    bad.append((tid, rep, i, int((im != kept["ap"][i]).sum())))
                continue
            else:
                im = run(ctx, fields[i])
            d = im != serial[i]
            if d.any():
                rows, cols = np.nonzero(d)
                bad.append((tid, rep, i, int(d.sum()), (int(rows.min()), int(rows.max())), (int(cols.min()), int(cols.max())), len(np.unique(cols)),
                            float(np.abs(im - serial[i]).max())))
def fitter(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        if os.environ.get("LOAD") == "torch":      # a load that is not the engine: matrix products and host-device copies
            a = torch.randn(2048, 2048, device="cuda")
            for rep in range(400):
                b = (a @ a).cpu()
                a = torch.randn(2048, 2048, device="cuda") * 1e-2
            return
        ctx = engine.Context(0)
        kind = os.environ.get("LOAD", "fit")
        for rep in range(60):
            if kind == "pre":         # statistics pass only (two-step preprocess, in place), no decomposition
                mat, _ = engine.preprocess(ctx, fields[(rep + tid) % 4], True, False, None, in_place=True)
                mat.free()
                continue
            if kind.startswith("rsvd"):        # decomposition only, on a matrix made once
                if rep == 0:
                    if "f32" in kind:
                        ctx.set_precision("f32", "f32")
                    keep = engine.preprocess(ctx, fields[tid % 4], True, False, None, in_place="copy" not in kind)[0]
                    om = engine.sketch_matrix(min(keep.n, keep.p), 16, 3)
                engine.rsvd(ctx, keep, 6, n_iter=(0 if "it0" in kind else "auto"), omega=om, device_out=("dev" in kind))
                continue
            if kind in ("tmul", "mul"):       # ONE streaming kernel of the in-place layout, again and again
                if rep == 0:
                    keep = engine.preprocess(ctx, fields[tid % 4], True, False, None, in_place=True)[0]
                    Zn = torch.randn(keep.n_pad, 64, device="cuda"); Zn[keep.n:] = 0
                    Yp = torch.randn(keep.p_pad, 64, device="cuda"); Yp[keep.p:] = 0
                for _ in range(10):
                    if kind == "tmul":
                        engine.panel_tmul(ctx, keep, Zn, prec="f16x3")
                    else:
                        engine.panel_mul(ctx, keep, Yp, prec="f16x3")
                continue
            if kind == "panels":      # small-side kernels only: Gram matrix + Cholesky-QR of a panel, no pass over a field
                if rep == 0:
                    P = torch.randn(1024, 64, device="cuda")
                for _ in range(20):
                    G = engine.panel_gram(ctx, P)
                    Q = engine.panel_cholqr(ctx, P, 16, G)
                continue
            mat, st_, U, s, V = engine.fit(ctx, fields[(rep + tid) % 4], 6, random_state=3)
            mat.free()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)] + [threading.Thread(target=fitter, args=(t,)) for t in range(2)]
for t in ths: t.start()
for t in ths: t.join()
print("MODE", MODE, "mismatches:", len(bad))
for b in bad[:8]: print("  (thread, rep, field, n differing, row range, col range, distinct cols, max abs diff)", b)
