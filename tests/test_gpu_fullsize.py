"""GPU tests at the BASELINE.json config sizes, through size-independent properties (the oracle
cannot run at these sizes in seconds): orthonormality, X V = U diag(s), linearity, centring,
bitwise determinism; plus the reference's own CPU-runnable config 1
shape against the oracle and degenerate tiny shapes."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402


def _device_field(n, nlat, nlon):
    import torch

    import bench

    return bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))


@pytest.mark.parametrize("n,nlat,nlon,k", [(5000, 360, 720, 50), (10000, 720, 1440, 50)],
                         ids=["config2_5000x259200", "config4_10000x1036800"])
def test_eof_properties_at_baseline_sizes(ctx, n, nlat, nlon, k):
    import torch

    from xeofs_amd import engine

    X = _device_field(n, nlat, nlon)
    mat, st = engine.preprocess(ctx, X, want_stats=False)
    assert (st["n"], st["p"]) == (n, nlat * nlon)
    U, s, V = engine.rsvd(ctx, mat, k, random_state=5, device_out=True)
    Ud, Vd = U.double(), V.double()
    sd = torch.as_tensor(s.astype(np.float64), device=Ud.device)
    eye = torch.eye(k, dtype=torch.float64, device=Ud.device)
    assert float((Ud.T @ Ud - eye).abs().max()) < 1e-6
    assert float((Vd.T @ Vd - eye).abs().max()) < 1e-6
    assert np.all(np.diff(s) <= 0) and s[-1] > 0
    # X V = U diag(s) through the projection kernel (one more pass over the matrix)
    XV = torch.as_tensor(engine.project(ctx, mat, V), device=Ud.device).double()
    assert float((XV - Ud * sd).norm() / (Ud * sd).norm()) < 1e-5
    # explained variance cannot exceed the total variance (reference tests/models/single/test_eof.py:85-100)
    assert (s.astype(np.float64) ** 2).sum() / (n - 1) <= st["total_variance"] * (1 + 1e-5)
    # bitwise determinism under the seed (tests/linalg/test_decomposer.py:164-192)
    U2, s2, V2 = engine.rsvd(ctx, mat, k, random_state=5, device_out=True)
    assert np.array_equal(s, s2) and torch.equal(U, U2) and torch.equal(V, V2)
    # the resident matrix is centred: projecting it on V gives zero-mean scores
    assert float(XV.mean(dim=0).abs().max()) <= 1e-6 * float(sd[0])
    mat.free()
    # linearity: decomposing 2 X doubles the singular values and leaves the vectors
    mat2, _ = engine.preprocess(ctx, X * 2.0, want_stats=False)
    U3, s3, V3 = engine.rsvd(ctx, mat2, k, random_state=5, device_out=True)
    assert np.allclose(s3, 2.0 * s, rtol=1e-6)
    assert float((V3.double() - Vd).abs().max()) < 1e-5
    mat2.free()
    ctx.trim()


def test_config1_shape_vs_oracle(ctx):
    """BASELINE config 1: xe.single.EOF(n_modes=10) on the air_temperature shape 2920 x (25 x 53)
    (synthetic stand-in of the same shape), use_coslat as in the README quickstart."""
    import xeofs_amd as xe

    n, nlat, nlon, k = 2920, 25, 53, 10
    X, lat = orc.synthetic_field(n, nlat, nlon, rank=30, seed=0)
    lat = np.linspace(75.0, 15.0, nlat)
    da = xe.DataArray(X.reshape(n, nlat, nlon) + 270.0, dims=("time", "lat", "lon"),
                      coords={"lat": lat, "lon": np.linspace(200, 330, nlon)}, name="air")
    m = xe.single.EOF(n_modes=k, use_coslat=True, random_state=5).fit(da, "time")
    w = np.repeat(orc.sqrt_cos_lat_weights(lat), nlon)
    ref = orc.eof_fit((X + 270.0).astype(np.float64), k, feature_weights=w, random_state=5)
    s = m.singular_values().values
    assert np.all(np.abs(s - ref["norms"]) <= 1e-5 * ref["norms"])
    assert np.allclose(m.explained_variance_ratio().values, ref["explained_variance_ratio"], rtol=3e-5)
    c = m.components().values.reshape(k, -1)
    for j in range(k):
        assert abs(np.dot(c[j], ref["components"][:, j])) >= 1 - 1e-5, j


@pytest.mark.parametrize("n,p,k", [(2, 3, 1), (3, 2, 2), (2, 2, 2), (5, 1, 1), (1, 4, 1)])
def test_tiny_shapes(ctx, n, p, k):
    from xeofs_amd import engine

    X = np.random.default_rng(n * 10 + p).standard_normal((n, p)).astype(np.float32)
    mat = engine.from_dense(ctx, X)
    U, s, V = engine.rsvd(ctx, mat, k, random_state=0)
    se = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:k]
    assert np.allclose(s, se, rtol=2e-5, atol=1e-6)
    rec = (U * s) @ V.T
    best = np.linalg.svd(X.astype(np.float64))
    approx = (best[0][:, :k] * best[1][:k]) @ best[2][:k]
    assert np.abs(rec - approx).max() <= 1e-4


def test_all_nan_and_constant_inputs(ctx):
    from xeofs_amd import engine

    X = np.full((10, 6), np.nan, dtype=np.float32)
    with pytest.raises(ValueError, match="no valid"):
        engine.preprocess(ctx, X)
    X = np.ones((10, 6), dtype=np.float32)           # constant field: zero matrix after centring
    mat, st = engine.preprocess(ctx, X, True, True)   # std clipped at float32 eps (scaler.py:106-108)
    assert not mat.download().any() and st["total_variance"] == 0.0
    U, s, V = engine.rsvd(ctx, mat, 2, random_state=0)
    assert np.all(s == 0) and np.isfinite(U).all() and np.isfinite(V).all()


def test_two_rank_sharded_path_on_one_gpu(ctx):
    """End-to-end multi-rank path with the real HIP kernels: two processes share cuda:0, each holds half
    of the feature axis, collectives over gloo (RCCL needs one GPU per rank; the single-rank RCCL calls are
    covered by test_sharded_world1_equals_driver_bitwise).  The 2-rank singular values must equal the
    1-rank ones to float32 rounding and the global X V = U s identity must hold."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "1", "--nlat", "90", "--nlon", "180", "--nsamples", "1500", "--modes", "12",
              "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--same-gpu"] + common,
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                         env=env, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    d2 = json.loads(two.stdout.strip().splitlines()[-1])
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1
    assert len(two.stdout.strip().splitlines()) == 1 and len(one.stdout.strip().splitlines()) == 1  # ONE JSON line
    assert np.allclose(d2["parity"]["s_head"], d1["parity"]["s_head"], rtol=2e-6)
    assert d2["parity"]["XV_eq_Us_relerr"] < 1e-5 and d2["parity"]["orth_V_maxabs"] < 1e-6


@pytest.mark.parametrize("nan,pca", [(False, False), (True, False), (False, True)])
def test_two_rank_sharded_mca_and_eof_on_one_gpu(ctx, nan, pca):
    """SURVEY.md §8e (rows C3 + preprocess facts): two processes share cuda:0, each holds half of each
    field's space axis; `sharded_mca_fit` / `sharded_eof_fit` (HIP kernels + gloo all-reduces) against the
    single-GPU drivers on the whole fields.  Tolerances: float32 summation-order differences only."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29519", os.path.join(root, "tools", "sharded_mca_worker.py"),
           "--backend", "gloo", "--same-gpu"] + (["--nan"] if nan else []) + (["--pca"] if pca else [])
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    d = json.loads([ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["world"] == 2 and d["p_total"] == [d["p1"], d["p2"]]
    if nan:
        assert d["n_valid"] == 699 and d["p1"] < 5000 and d["p2"] < 3600
    assert d["mca_s"] < 2e-5 and d["eof_s"] < 2e-5
    assert d["mca_q1_cos"] > 1 - 1e-5 and d["mca_q2_cos"] > 1 - 1e-5 and d["eof_v_cos"] > 1 - 1e-5
    assert d["mca_scores1"] < 1e-4 and d["mca_scores2"] < 1e-4 and d["eof_scores"] < 1e-4
    assert d["mca_norm1"] < 1e-4 and d["mca_tsc"] < 1e-5 and d["eof_tv"] < 1e-6
