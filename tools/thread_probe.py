"""Two Python threads, each with its own Context (its own stream) on the same GPU, fitting different fields at the same time:
results must equal the serial ones bit for bit (process-wide state of the library: the sketch generator's worker team, cached
launch attributes)."""
import sys, os, threading, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine

warnings.simplefilter("ignore")
rng = np.random.default_rng(0)
fields = [(rng.standard_normal((n, 5)) @ rng.standard_normal((5, p)) + 0.2 * rng.standard_normal((n, p)) + 1.0).astype(np.float32)
          for n, p in ((400, 4096), (700, 2048), (300, 8192), (1000, 1000))]
def run(ctx, X, seed, hil=False):
    if hil:
        A, _ = engine.preprocess(ctx, X, True, False, None, in_place=True, for_hilbert=True)
        sq = engine.hilbert_sumsq(ctx, A, "exp", 0.2)
        if os.environ.get("NO_RSVD"):
            A.free()
            return np.zeros(1), np.float64(sq)
        U, s, V = engine.rsvd_hilbert_c64(ctx, A, 4, "exp", 0.2, random_state=seed)
        A.free()
        return s.copy(), np.float64(sq)
    mat, st, U, s, V = engine.fit(ctx, X, 6, random_state=seed)
    mat.free()
    return s.copy(), V.copy()
ctx0 = engine.Context(0)
serial = {}
for i, X in enumerate(fields):
    for hil in (False, True):
        serial[(i, hil)] = run(ctx0, X, 10 + i, hil)
bad = []
def worker(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = engine.Context(0)
        for rep in range(40):
            i = (rep + tid) % len(fields)
            hil = {"mixed": (rep // 2 + tid) % 2 == 1, "hilbert": True, "fit": False}[os.environ.get("JOBS", "mixed")]
            out = run(ctx, fields[i], 10 + i, hil)
            ref = serial[(i, hil)]
            if not (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])):
                bad.append((tid, rep, i, hil, float(np.abs(out[0] - ref[0]).max()),
                            (float(out[1]), float(ref[1]), float(out[1]) - float(ref[1])) if hil else float(np.abs(out[1] - ref[1]).max())))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
for t in ths: t.start()
for t in ths: t.join()
print("threads 3 x 40 fits; mismatches:", len(bad), bad[:5])
