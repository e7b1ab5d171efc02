"""Model-level edge shapes: every model class on tiny / degenerate inputs.  Each case prints its outcome at once (a hard crash shows
as the last 'trying' line); values are checked against exact references where one exists.  START=<i> resumes after a crash."""
import sys, os, warnings, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eof_oracle as orc
import xeofs_amd as xe

warnings.simplefilter("ignore")
rng = np.random.default_rng(2)
def da(n, a, b, r=3, off=1.0):
    X = ((rng.standard_normal((n, r)) * 2.0 ** -np.arange(r)) @ rng.standard_normal((r, a * b)) + 0.1 * rng.standard_normal((n, a * b)) + off).astype(np.float32)
    return xe.DataArray(X.reshape(n, a, b), dims=("time", "lat", "lon")), X

cases = []
def case(name):
    def deco(fn):
        cases.append((name, fn)); return fn
    return deco

def svals(X, k, complex_=False):
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    if complex_:
        Xc = orc.hilbert_transform(Xc, "exp", 0.2)
    return np.linalg.svd(Xc, compute_uv=False)[:k]

def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))

for (n, a, b, k) in ((3, 1, 5, 2), (2, 1, 3, 1), (40, 1, 1, 1), (40, 1, 2, 2), (40, 1, 3, 3), (5, 2, 50, 4), (300, 3, 4, 12), (12, 3, 4, 12)):
    @case(f"EOF n={n} grid={a}x{b} k={k}")
    def _(n=n, a=a, b=b, k=k):
        d, X = da(n, a, b)
        m = xe.single.EOF(n_modes=k, random_state=1).fit(d, "time")
        s = m.singular_values().values
        t = m.transform(d).values
        r = m.inverse_transform(m.scores()).values
        return f"s err {rel(s, svals(X, k)):.1e} transform==scores {rel(t, m.scores().values):.1e} recon shape {r.shape}"
for (n, a, b, k) in ((4, 1, 6, 2), (3, 1, 3, 1), (30, 1, 1, 1), (30, 1, 2, 2), (64, 2, 5, 8), (9, 2, 50, 4)):
    @case(f"HilbertEOF n={n} grid={a}x{b} k={k}")
    def _(n=n, a=a, b=b, k=k):
        d, X = da(n, a, b)
        m = xe.single.HilbertEOF(n_modes=k, random_state=1).fit(d, "time")
        s = m.singular_values().values
        return f"s err {rel(s, svals(X, k, True)):.1e} amp shape {m.components_amplitude().values.shape}"
for (n, a1, b1, a2, b2, k, pca) in ((5, 1, 4, 1, 3, 2, False), (40, 1, 1, 1, 1, 1, False), (40, 1, 2, 1, 30, 2, False), (40, 1, 2, 1, 30, 2, True), (6, 2, 40, 3, 30, 3, True), (100, 2, 3, 2, 2, 4, True)):
    @case(f"MCA n={n} X={a1}x{b1} Y={a2}x{b2} k={k} use_pca={pca}")
    def _(n=n, a1=a1, b1=b1, a2=a2, b2=b2, k=k, pca=pca):
        d1, X = da(n, a1, b1); d2, Y = da(n, a2, b2)
        m = xe.cross.MCA(n_modes=k, use_pca=pca, n_pca_modes=3, random_state=1).fit(d1, d2, "time")
        s = m.singular_values().values
        Xc, Yc = X.astype(np.float64) - X.astype(np.float64).mean(0), Y.astype(np.float64) - Y.astype(np.float64).mean(0)
        se = np.linalg.svd(Xc.T @ Yc / (n - 1), compute_uv=False)[:k]
        out = f"s err vs exact C {rel(s, se):.1e}" if not pca else f"s {s[:2]}"
        c1, c2 = m.components()
        return out + f" comps {c1.values.shape} {c2.values.shape} scf {m.squared_covariance_fraction().values[:2]}"
for (n, k, alpha) in ((50, 2, 0.0), (50, 2, 0.5), (8, 2, 0.0)):
    @case(f"CPCCA n={n} k={k} alpha={alpha}")
    def _(n=n, k=k, alpha=alpha):
        d1, X = da(n, 2, 6); d2, Y = da(n, 3, 4)
        m = xe.cross.CPCCA(n_modes=k, alpha=alpha, use_pca=True, n_pca_modes=4, random_state=1).fit(d1, d2, "time")
        ref = orc.cpcca_fit(X.reshape(n, -1).astype(np.float64), Y.reshape(n, -1).astype(np.float64), k, alpha=alpha, use_pca=True, n_pca_modes=4, pca_solver="full", random_state=1)
        return f"s err {rel(m.singular_values().values, ref['singular_values']):.1e}"
for (k, power) in ((1, 1), (2, 1), (2, 2), (5, 4)):
    @case(f"EOFRotator k={k} power={power}")
    def _(k=k, power=power):
        d, X = da(60, 4, 10, r=6)
        m = xe.single.EOF(n_modes=max(k, 2), random_state=1).fit(d, "time")
        r = xe.single.EOFRotator(n_modes=k, power=power).fit(m)
        return f"explained variance {r.explained_variance().values[:2]} comps {r.components().values.shape}"
@case("EOFBootstrapper tiny")
def _():
    d, X = da(30, 2, 5)
    m = xe.single.EOF(n_modes=2, random_state=1).fit(d, "time")
    bs = xe.validation.EOFBootstrapper(n_bootstraps=3, seed=1).fit(m)
    return f"ok {bs.explained_variance().values.shape}"
@case("EOF with coslat weights, standardize, a NaN column")
def _():
    d, X = da(50, 4, 6)
    v = d.values.copy(); v[:, 1, 2] = np.nan
    lat = np.linspace(-60, 60, 4)
    d2 = xe.DataArray(v, dims=("time", "lat", "lon"), coords={"lat": lat})
    m = xe.single.EOF(n_modes=3, standardize=True, use_coslat=True, random_state=1).fit(d2, "time")
    c = m.components().values
    return f"comps {c.shape} nan at masked {bool(np.isnan(c[:, 1, 2]).all())} s {m.singular_values().values}"

start = int(os.environ.get("START", 0))
for i, (name, fn) in enumerate(cases):
    if i < start:
        continue
    print(f"[{i}] trying {name}", flush=True)
    try:
        print(f"[{i}]   -> {fn()}", flush=True)
    except Exception as e:
        print(f"[{i}]   raised {type(e).__name__}: {str(e)[:160]}", flush=True)
print("done")
