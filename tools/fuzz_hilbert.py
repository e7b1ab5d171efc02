"""Randomised parity sweep of the Hilbert stage (random lengths incl. odd / prime, decay factors, padding on/off)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
from xeofs_amd import engine

ctx = engine.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(ncase):
    n = int(rng.choice([2, 3, 5, 7, 16, 31, 64, 97, 100, 255, 256, 257, 500, 511, 512, 513, 1021, 1500, 2048, 2049, 4099, 8191, 8192, 8193, 16383, 16384, 16385,
                        int(rng.integers(2, 2500)), int(rng.integers(2500, 9000)), int(rng.integers(9000, 17000))]))
    p = int(rng.integers(1, 700 if n < 2500 else 40))
    padding = "exp" if rng.random() < 0.6 else None
    decay = float(rng.uniform(0.05, 1.5))
    t = np.arange(n)[:, None]
    y = (np.sin(0.3 * t * rng.uniform(0.1, 2, p)) + 0.3 * rng.standard_normal((n, p)) + rng.uniform(-2, 2, p) * t / max(n, 1)).astype(np.float32)
    yc = y - y.mean(0)
    try:
        m = engine.from_dense(ctx, yc)
        im, _ = engine.hilbert(ctx, m, padding=padding, decay_factor=decay)
        got = im.download()
        ssq = im.sumsq()          # over the padded buffer: also sees what the stage left unwritten in recycled memory
        m.free(); im.free()
        ref = orc.hilbert_transform(yc.astype(np.float64), padding=padding, decay_factor=decay).imag
        scale = max(np.abs(ref).max(), np.abs(yc).max(), 1e-30)
        err = np.abs(got - ref).max() / scale
        tol = 3e-5 if n > 16384 else 2e-6    # (series longer than 16384 samples: hipFFT route)
        if not (err < tol and np.isclose(ssq, (ref ** 2).sum(), rtol=1e-4, atol=1e-10 * scale * scale * n * p)):
            bad += 1
            print("MISMATCH case", case, dict(n=n, p=p, padding=padding, decay=round(decay, 3)), "max err / scale", float(err), "sumsq", ssq, float((ref ** 2).sum()))
    except Exception as e:
        bad += 1
        print("EXC case", case, dict(n=n, p=p, padding=padding), type(e).__name__, str(e)[:160])
print("cases", ncase, "bad", bad)
