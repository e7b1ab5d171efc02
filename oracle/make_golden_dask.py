"""G7: the dask branch of the reference (`linalg/decomposer.py:163-171`) calls
`dask.array.linalg.svd_compressed(X, k, seed=..., n_power_iter=4)`.  dask is not installed for the build
interpreter, but /opt/conda/bin/python3.9 in this image has dask 2021.10.0 (older than the reference's pin
`>=2023.0.1`; `svd_compressed`'s algorithm is unchanged).  This script runs the REAL dask routine there:

    /opt/conda/bin/python3.9 oracle/make_golden_dask.py

and writes tests/golden/g7_dask_svd_compressed.npz (input recipe, singular values, |v| heads).
"""
import os

import dask
import dask.array as da
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(12)
amp = 9.0 * 0.65 ** np.arange(7)
X = (rng.standard_normal((200, 7)) * amp) @ rng.standard_normal((7, 1000)) + 0.05 * rng.standard_normal((200, 1000))
X -= X.mean(axis=0)
Xd = da.from_array(X, chunks=(50, 250))
out = {}
for k, seed in ((3, 1), (5, 7)):
    u, s, v = da.linalg.svd_compressed(Xd, k, seed=seed, n_power_iter=4, compute=True)
    u, s, v = dask.compute(u, s, v)
    out[f"s_k{k}"] = s
    out[f"absv_head_k{k}"] = np.abs(v[:, :32])
    out[f"shape_u_k{k}"] = np.array(u.shape)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_dask_svd_compressed.npz"), X=X.astype(np.float32),
                    s_exact=np.linalg.svd(X, compute_uv=False)[:8], dask=dask.__version__, numpy=np.__version__, **out)
print("wrote g7_dask_svd_compressed.npz", {k: v for k, v in out.items() if k.startswith("s_")})
