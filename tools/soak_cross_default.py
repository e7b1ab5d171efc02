"""soak of default-argument cross models -- `MCA(...).fit` (PCA pre-reduction with `basis_only`, the two
fields' library calls side by side in `_run_two`) and `CCA(...).fit` (the same plus the two whiteners' eigen-problems) -- for
bitwise-equal outputs over N fits each.  The overlap in `_run_two` puts rocSOLVER's eigh on a side stream under the other field's
fp16 Gram / streaming kernels; tools/thread_probe5.py found eigh unaffected by such neighbours (0 of 240 results differ where
rocFFT differs in 72 %): this is the end-to-end check.  Usage: python tools/soak_cross_default.py [N=300]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import xeofs_amd as xe

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
n, p1, p2 = 1500, 9000, 7000
t = rng.standard_normal((n, 12)) * (6.0 * 0.8 ** np.arange(12))
X = torch.as_tensor((t @ rng.standard_normal((12, p1)) + rng.standard_normal((n, p1)) + 2.0).astype(np.float32), device="cuda")
Y = torch.as_tensor((t @ rng.standard_normal((12, p2)) + rng.standard_normal((n, p2)) - 1.0).astype(np.float32), device="cuda")
Xd = xe.DataArray(X.reshape(n, 90, 100), dims=("time", "lat", "lon"))
Yd = xe.DataArray(Y.reshape(n, 70, 100), dims=("time", "lat", "lon"))
warnings.simplefilter("ignore")
from xeofs_amd import pca as _pca


def outputs(model):
    _pca._unseeded_fits[0] = 0          # the PCA's unseeded sketches: the same streams for every fit of the soak
    m = model(n_modes=8, random_state=3).fit(Xd, Yd, "time")
    c1, c2 = m.components()
    s1, s2 = m.scores()
    return [np.asarray(m.singular_values().values), np.asarray(c1.values), np.asarray(c2.values), np.asarray(s1.values), np.asarray(s2.values)]


for name, model in (("MCA", xe.cross.MCA), ("CCA", xe.cross.CCA)):
    ref = outputs(model)
    bad, t0 = 0, time.perf_counter()
    for i in range(N):
        got = outputs(model)
        if not all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, ref)):
            bad += 1
            worst = max(float(np.nanmax(np.abs(a - b))) for a, b in zip(got, ref))
            print(f"{name} fit {i}: differs from the first (max abs difference {worst:.3e})", flush=True)
    print(f"{name}: {N} default-argument fits at {n} x ({p1}, {p2}), {bad} differ from the first bit for bit; {time.perf_counter() - t0:.0f} s", flush=True)
