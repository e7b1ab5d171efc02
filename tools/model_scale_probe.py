"""Mixed-unit fields (feature scales spread over 8 and 14 orders of magnitude) with standardize=True through the MODEL classes:
EOF, MCA (direct and PCA route), CCA, HilbertEOF -- singular values against the float64 oracle / exact references."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
import xeofs_amd as xe

warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
n, nlat, nlon, k = 300, 20, 60, 5
p = nlat * nlon
T = rng.standard_normal((n, 8)) * 2.0 ** -np.arange(8)
base1 = T @ rng.standard_normal((8, p)) + 0.05 * rng.standard_normal((n, p))
base2 = T[:, :5] @ rng.standard_normal((5, p)) + 0.05 * rng.standard_normal((n, p))
for spread in (0.0, 8.0, 14.0):
    sc1 = 10.0 ** rng.uniform(-spread / 2, spread / 2, p)
    sc2 = 10.0 ** rng.uniform(-spread / 2, spread / 2, p)
    X = (base1 * sc1 + 3.0 * sc1).astype(np.float32)
    Y = (base2 * sc2 - 1.0 * sc2).astype(np.float32)
    Xd = xe.DataArray(X.reshape(n, nlat, nlon), dims=("time", "lat", "lon"))
    Yd = xe.DataArray(Y.reshape(n, nlat, nlon), dims=("time", "lat", "lon"))
    X64, Y64 = X.astype(np.float64), Y.astype(np.float64)
    def rel(a, b):
        return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max())
    out = []
    m = xe.single.EOF(n_modes=k, standardize=True, random_state=1).fit(Xd, "time")
    out.append(("EOF", rel(m.singular_values().values, orc.eof_fit(X64, k, standardize=True, random_state=1)["norms"])))
    m = xe.cross.MCA(n_modes=k, standardize=True, use_pca=False, random_state=1).fit(Xd, Yd, "time")
    ref = orc.mca_fit(X64, Y64, k, standardize=True, random_state=1)
    out.append(("MCA direct", rel(m.singular_values().values, ref["singular_values"])))
    out.append(("MCA direct TSC", abs(m.total_squared_covariance() - ref["total_squared_covariance"]) / ref["total_squared_covariance"]))
    m = xe.cross.MCA(n_modes=k, standardize=True, use_pca=True, n_pca_modes=20, random_state=1).fit(Xd, Yd, "time")
    ref = orc.mca_fit(X64, Y64, k, standardize=True, random_state=1, use_pca=True, n_pca_modes=20, pca_random_state=1)
    out.append(("MCA PCA route", rel(m.singular_values().values, ref["singular_values"])))
    m = xe.cross.CCA(n_modes=k, standardize=True, use_pca=True, n_pca_modes=20, random_state=1).fit(Xd, Yd, "time")
    ref = orc.cpcca_fit(X64, Y64, k, alpha=0.0, standardize=True, use_pca=True, n_pca_modes=20, pca_solver="full", random_state=1)
    out.append(("CCA", rel(m.singular_values().values, ref["singular_values"])))
    m = xe.single.HilbertEOF(n_modes=k, standardize=True, padding="exp", decay_factor=0.2, random_state=1).fit(Xd, "time")
    pre = orc.preprocess(X64, True, True, None)
    Z = orc.hilbert_transform(pre["X"], "exp", 0.2)
    se = np.linalg.svd(Z, compute_uv=False)[:k]
    out.append(("HilbertEOF", rel(m.singular_values().values, se)))
    out.append(("HilbertEOF total variance", abs(m.data["total_variance"] - orc.total_variance(Z).real) / orc.total_variance(Z).real))
    print(f"spread 1e{spread:.0f}: " + "  ".join(f"{a} {b:.1e}" for a, b in out), flush=True)
