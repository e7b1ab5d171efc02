"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances (stated once):

  singular values          rel 1e-5 vs the float64 oracle
  gap-separated vectors    |cos| >= 1 - 1e-5
  elementwise GEMM panels  1e-5 * sum|a||b| (float32 fma-chain class)
  preprocessing            float32 rounding of the float64 oracle value (rel 2e-6)
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def _field(n, p, rank=8, seed=0, scale=3.0, noise=1.0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    amp = scale * 0.8 ** np.arange(rank)
    X = (rng.standard_normal((n, rank)) * amp) @ rng.standard_normal((rank, p)) / np.sqrt(rank)
    X = X * np.sqrt(max(n, p)) ** 0.5 + noise * rng.standard_normal((n, p)) + 5.0
    return X.astype(dtype)


def _relgap(s, j):
    lo = abs(s[j] - s[j + 1]) / s[j] if j + 1 < len(s) else 1.0
    hi = abs(s[j - 1] - s[j]) / s[j] if j > 0 else 1.0
    return min(lo, hi)


def _gap_ok(s, j, tol=1e-3):
    return _relgap(s, j) > tol


def _cos_tol(s, j):
    """1 - |cos| allowed for mode j: 1e-5 for gap-separated modes; for modes whose
    relative spectral gap g is small, float32 perturbations eps*s_0 rotate the vector by about
    eps*s_0/(s_j*g) (Davis-Kahan), so the bound grows as that angle squared (eps = 4e-7)."""
    ang = 4e-7 * s[0] / (s[j] * _relgap(s, j))
    return max(1e-5, 0.5 * ang * ang)


# --------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("prec,tol", [("f32", 1e-5), ("bf16x6", 1e-5), ("f16x3", 1e-5), ("bf16x3", 4e-5), ("f64", 2e-7)])
@pytest.mark.parametrize("n,p,L", [(100, 700, 32), (300, 1500, 64), (1000, 520, 64), (64, 3000, 96)])
def test_panel_tmul_mul(ctx, n, p, L, prec, tol):
    import torch
    from xeofs_amd import engine

    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, p)).astype(np.float32)
    mat = engine.from_dense(ctx, X)
    assert np.array_equal(mat.download(), X)
    Z = rng.standard_normal((n, L)).astype(np.float32)
    Y = rng.standard_normal((p, L)).astype(np.float32)
    Zp = engine.panel_import(ctx, Z, mat.n_pad, L)
    Yp = engine.panel_import(ctx, Y, mat.p_pad, L)
    out_t = engine.panel_tmul(ctx, mat, Zp, prec=prec)
    out_m = engine.panel_mul(ctx, mat, Yp, prec=prec)
    torch.cuda.synchronize()
    got_t = out_t.cpu().numpy()
    got_m = out_m.cpu().numpy()
    ref_t = X.astype(np.float64).T @ Z.astype(np.float64)
    ref_m = X.astype(np.float64) @ Y.astype(np.float64)
    bound_t = np.abs(X).astype(np.float64).T @ np.abs(Z)
    bound_m = np.abs(X).astype(np.float64) @ np.abs(Y)
    assert np.all(np.abs(got_t[:p] - ref_t) <= tol * bound_t + 1e-30)
    assert np.all(np.abs(got_m[:n] - ref_m) <= tol * bound_m + 1e-30)
    if prec != "bf16x3":  # f32-class accuracy in the RMS sense as well
        assert np.sqrt(np.mean((got_t[:p] - ref_t) ** 2)) <= 2e-6 * np.sqrt(np.mean(bound_t ** 2))
    # padding rows of the outputs must be exact zeros (they feed the next product)
    assert not got_t[p:].any() and not got_m[n:].any()


def test_panel_gram_cholqr(ctx):
    import torch
    from xeofs_amd import engine

    rng = np.random.default_rng(2)
    rows, l, L = 5000, 60, 64
    W = (rng.standard_normal((rows, l)) * (10.0 ** rng.uniform(-2, 2, size=l))).astype(np.float32)
    W[:, 7] = W[:, 3] * 2.0  # exactly dependent column -> must come out as zeros
    rows_pad = 5120
    P = engine.panel_import(ctx, W, rows_pad, L)
    G = engine.panel_gram(ctx, P)
    Q = engine.panel_cholqr(ctx, P, l, G)
    torch.cuda.synchronize()
    Gh = G.cpu().numpy()
    ref = W.astype(np.float64).T @ W.astype(np.float64)
    assert np.allclose(Gh[:l, :l], ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    Qh = Q.cpu().numpy().astype(np.float64)
    assert not Qh[:, 7].any()
    keep = [j for j in range(l) if j != 7]
    QtQ = Qh[:, keep].T @ Qh[:, keep]
    assert np.abs(QtQ - np.eye(len(keep))).max() < 5e-5
    # same column space
    coef = np.linalg.lstsq(Qh[:rows, keep], W[:, keep].astype(np.float64), rcond=None)[0]
    assert np.abs(Qh[:rows, keep] @ coef - W[:, keep]).max() < 1e-4 * np.abs(W).max()


# --------------------------------------------------------------------------- preprocess
@pytest.mark.parametrize("center,standardize,weights", [(True, False, False), (True, True, True), (False, False, True)])
@pytest.mark.parametrize("nan_kind", ["none", "features", "features+samples"])
def test_preprocess_vs_oracle(ctx, center, standardize, weights, nan_kind):
    from xeofs_amd import engine

    n, nlat, nlon = 120, 9, 14
    P = nlat * nlon
    X = _field(n, P, seed=3)
    X += 250.0  # temperatures: large mean, small variance
    if nan_kind != "none":
        X[:, [3, 17, 50, 51, 125]] = np.nan
    if nan_kind == "features+samples":
        X[[0, 77], :] = np.nan
    w = None
    if weights:
        lat = np.linspace(-80, 80, nlat)
        w = np.repeat(orc.sqrt_cos_lat_weights(lat), nlon)
    ref = orc.preprocess(X, center, standardize, w)
    mat, st = engine.preprocess(ctx, X, center, standardize, w)
    assert np.array_equal(st["valid_feature"], ref["valid_feature"])
    assert np.array_equal(st["valid_sample"], ref["valid_sample"])
    got = mat.download()
    assert got.shape == ref["X"].shape
    # `ref` ran on the float32 field (numpy nanmean / nanstd in float32): its own statistics carry eps32 * |mean| / std of
    # rounding (250 / 0.1-ish here), so it only pins masks and shapes; the VALUES are gated against the float64-statistics
    # oracle below -- the arithmetic xeofs itself runs (it promotes the field, utils/xarray_utils.py:78-100) -- at 2e-6
    vf = ref["valid_feature"]
    if center:
        assert np.allclose(st["mean"][vf], ref["mean"][vf], rtol=2e-6)
        assert np.isnan(st["mean"][~vf]).all()
    exact = orc.preprocess(X.astype(np.float64), center, standardize, w)
    # against the float64-statistics oracle only the final float32 rounding remains
    assert np.abs(got - exact["X"]).max() <= 2e-6 * max(np.abs(exact["X"]).max(), 1.0)
    tv = orc.total_variance(exact["X"])
    assert abs(st["total_variance"] - tv) <= 1e-6 * tv


def test_preprocess_isolated_nan_raises(ctx):
    from xeofs_amd import engine

    X = _field(50, 40, seed=4)
    X[3, 7] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        engine.preprocess(ctx, X)
    X = _field(50, 40, seed=4)
    X[:, 5] = np.nan
    X[10, :] = np.nan
    X[11, 3] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        engine.preprocess(ctx, X)


def test_apply_new_data(ctx):
    from xeofs_amd import engine

    X = _field(80, 60, seed=5)
    X[:, [4, 9]] = np.nan
    mat, st = engine.preprocess(ctx, X, True, True, None)
    Xn = _field(30, 60, seed=6)
    Xn[:, [4, 9]] = np.nan
    m2, vs = engine.apply(ctx, Xn, st["mean"], st["std"], None, st["valid_feature"])
    vf = st["valid_feature"]
    ref = (Xn[:, vf].astype(np.float64) - st["mean"][vf]) / st["std"][vf]
    assert np.abs(m2.download() - ref).max() < 1e-5 * np.abs(ref).max()
    Xbad = Xn.copy()
    Xbad[:, 20] = np.nan
    with pytest.raises(ValueError, match="different locations"):
        engine.apply(ctx, Xbad, st["mean"], st["std"], None, st["valid_feature"])


# --------------------------------------------------------------------------- rSVD
def _check_svd(U, s, V, Uo, so, Vo, X64, k):
    assert np.abs(s - so).max() <= 1e-5 * so[0], (s, so)
    assert np.all(np.abs(s - so) <= 1e-5 * so + 2e-6 * so[0])
    for j in range(k):
        if _gap_ok(so, j):
            assert abs(np.dot(V[:, j].astype(np.float64), Vo[:, j])) >= 1 - _cos_tol(so, j), j
            assert abs(np.dot(U[:, j].astype(np.float64), Uo[:, j])) >= 1 - _cos_tol(so, j), j
            # identical sign convention (the rule |max| >= |min| is discontinuous: only checked where the
            # oracle's own margin between |max| and |min| is not a rounding-level tie)
            mx, mn = abs(Vo[:, j].max()), abs(Vo[:, j].min())
            if abs(mx - mn) > 1e-3 * max(mx, mn):
                assert np.dot(V[:, j].astype(np.float64), Vo[:, j]) > 0, j
    rec = (U.astype(np.float64) * s) @ V.astype(np.float64).T
    rec_o = (Uo * so) @ Vo.T
    e, eo = np.linalg.norm(X64 - rec), np.linalg.norm(X64 - rec_o)
    assert e <= eo * (1 + 1e-4)
    # orthonormality
    assert np.abs(U.astype(np.float64).T @ U - np.eye(k)).max() < 2e-5
    assert np.abs(V.astype(np.float64).T @ V - np.eye(k)).max() < 2e-5


@pytest.fixture(params=[("f16x3", "f16x3"), ("f32", "f32"), ("bf16x3", "bf16x6"), ("bf16x3", "bf16x3"), ("f64", "f64")],
                ids=["f16x3", "f32", "bf16mixed", "bf16x3", "f64"])
def precision(request, ctx):
    ctx.set_precision(*request.param)
    yield request.param
    ctx.set_precision("f16x3", "f16x3")


@pytest.mark.parametrize("n,p,k", [(512, 2048, 10), (300, 4000, 40), (2500, 700, 20), (600, 600, 5),
                                   (700, 3000, 100), (1500, 900, 200)])   # last two: sketch wider than 64
def test_rsvd_vs_oracle(ctx, n, p, k, precision):
    from xeofs_amd import engine

    X = _field(n, p, rank=12, seed=10 + k)
    X = X - X.mean(axis=0, dtype=np.float64).astype(np.float32)
    mat = engine.from_dense(ctx, X)
    U, s, V = engine.rsvd(ctx, mat, k, random_state=42)
    X64 = X.astype(np.float64)
    Uo, so, Vo = orc.decomposer_fit(X64, k, random_state=42, solver="randomized")
    _check_svd(U, s, V, Uo, so, Vo, X64, k)
    # vs the exact SVD as well, on the signal modes (the noise-bulk modes of a 4/7-iteration
    # randomized SVD are not converged in the reference algorithm either)
    se = np.linalg.svd(X64, compute_uv=False)[:k]
    m = min(k, 10)
    assert np.abs(s[:m] - se[:m]).max() <= 1e-4 * se[0]


def test_rsvd_bitwise_deterministic(ctx):
    from xeofs_amd import engine

    X = _field(700, 3000, seed=21)
    mat = engine.from_dense(ctx, X)
    a = engine.rsvd(ctx, mat, 12, random_state=5)
    b = engine.rsvd(ctx, mat, 12, random_state=5)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sketch_width_limit(ctx):
    from xeofs_amd import engine

    mat = engine.from_dense(ctx, _field(600, 700, seed=3))
    with pytest.raises(ValueError, match="sketch width"):
        engine.rsvd(ctx, mat, 300, random_state=0)


def test_rsvd_rank_error_and_wide_sketch(ctx):
    from xeofs_amd import engine

    X = _field(25, 20, rank=20, seed=22)
    mat = engine.from_dense(ctx, X)
    with pytest.raises(ValueError, match="rank"):
        engine.rsvd(ctx, mat, 21, random_state=0)
    # sketch wider than the rank (k + 10 > 20): results equal the exact SVD
    U, s, V = engine.rsvd(ctx, mat, 15, random_state=0)
    se = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:15]
    assert np.abs(s - se).max() <= 2e-5 * se[0]


def test_eof_fit_pipeline_vs_oracle(ctx):
    """raw field with land-mask NaNs -> preprocess -> rSVD -> EOF quantities."""
    from xeofs_amd import engine

    n, nlat, nlon = 400, 24, 40
    X, lat = orc.synthetic_field(n, nlat, nlon, rank=20, seed=0, nan_frac=0.3)
    w = np.repeat(orc.sqrt_cos_lat_weights(lat), nlon)
    k = 10
    ref = orc.eof_fit(X.astype(np.float64), k, feature_weights=w, random_state=5)
    mat, st = engine.preprocess(ctx, X, True, False, w)
    U, s, V = engine.rsvd(ctx, mat, k, random_state=5)
    assert abs(st["total_variance"] - ref["total_variance"]) <= 1e-5 * ref["total_variance"]
    Xc = ref["input_data"]
    _check_svd(U, s, V, ref["U"], ref["norms"], ref["components"], Xc, k)
    scores = U * s
    for j in range(k):
        if _gap_ok(ref["norms"], j):
            assert np.abs(scores[:, j] - ref["scores"][:, j]).max() <= 2e-4 * np.abs(ref["scores"][:, j]).max()
    # projection of the training data reproduces the scores (reference test_eof.py:364-391, rtol 1e-3)
    proj = engine.project(ctx, mat, V)
    assert np.allclose(proj, scores, rtol=1e-3, atol=1e-3 * np.abs(scores).max())
    # reconstruction kernel
    rec = engine.reconstruct(ctx, scores, V)
    assert np.abs(rec - (scores.astype(np.float64) @ V.astype(np.float64).T)).max() <= 1e-5 * np.abs(rec).max()


# --------------------------------------------------------------------------- cross-covariance / MCA
@pytest.mark.parametrize("p1,p2", [(900, 1400), (1300, 800)])
def test_crosscov_vs_oracle(ctx, p1, p2):
    from xeofs_amd import engine

    n, k = 300, 6
    rng = np.random.default_rng(31)
    T = rng.standard_normal((n, 8)) * (4.0 * 0.7 ** np.arange(8))
    X = (T @ rng.standard_normal((8, p1)) + rng.standard_normal((n, p1))).astype(np.float32)
    Y = (T @ rng.standard_normal((8, p2)) + rng.standard_normal((n, p2))).astype(np.float32)
    ref = orc.mca_fit(X.astype(np.float64), Y.astype(np.float64), k, random_state=7, solver="randomized")
    mx, _ = engine.preprocess(ctx, X)
    my, _ = engine.preprocess(ctx, Y)
    out = engine.crosscov_rsvd(ctx, mx, my, k, random_state=7)
    so = ref["singular_values"]
    assert np.all(np.abs(out["s"] - so) <= 1e-5 * so + 2e-6 * so[0])
    for j in range(k):
        if _gap_ok(so, j):
            c1 = np.dot(out["Q1"][:, j].astype(np.float64), ref["components1"][:, j])
            c2 = np.dot(out["Q2"][:, j].astype(np.float64), ref["components2"][:, j])
            assert c1 >= 1 - 1e-5 and c2 >= 1 - 1e-5, (j, c1, c2)
            assert np.abs(out["scores1"][:, j] - ref["scores1"][:, j]).max() <= 3e-4 * np.abs(ref["scores1"][:, j]).max()
            assert np.abs(out["scores2"][:, j] - ref["scores2"][:, j]).max() <= 3e-4 * np.abs(ref["scores2"][:, j]).max()
            assert abs(out["norm1"][j] - ref["norm1"][j]) <= 1e-4 * ref["norm1"][j]
            assert abs(out["norm2"][j] - ref["norm2"][j]) <= 1e-4 * ref["norm2"][j]
    tsc = ref["total_squared_covariance"]
    assert abs(out["total_squared_covariance"] - tsc) <= 1e-5 * tsc
    # reference invariant tests/models/cross/test_cpcca.py:152-164
    assert out["total_squared_covariance"] >= (out["s"].astype(np.float64) ** 2).sum() * (1 - 1e-6)


def test_crosscov_sample_mismatch(ctx):
    from xeofs_amd import engine

    mx, _ = engine.preprocess(ctx, _field(40, 600, seed=1))
    my, _ = engine.preprocess(ctx, _field(41, 600, seed=2))
    with pytest.raises(ValueError, match="same number of samples"):
        engine.crosscov_rsvd(ctx, mx, my, 3, random_state=0)


# --------------------------------------------------------------------------- sharded path on one GPU
def test_sharded_world1_equals_driver_bitwise(ctx):
    """The Python orchestration over the panel-level ABI (the multi-GPU path) must be the C++ driver's
    step sequence: at world size 1 the results are bitwise identical.  The RCCL collectives are
    issued for real (single-rank nccl group, Comm(force=True))."""
    import os

    import torch
    import torch.distributed as dist

    from xeofs_amd import engine, sharded

    X = _field(400, 3000, seed=41)
    X = X - X.mean(axis=0, dtype=np.float64).astype(np.float32)
    mat = engine.from_dense(ctx, X)
    ref = engine.rsvd(ctx, mat, 12, random_state=9)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        comm = sharded.Comm(force=True)
        assert comm.active
        got = sharded.sharded_rsvd(sharded.HipPanelOps(ctx, mat), comm, 12, 3000, 0, random_state=9)
    finally:
        if created:
            dist.destroy_process_group()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    # tall orientation (n >= p): the sharded side is the small side
    Xt = np.ascontiguousarray(X.T)
    mat2 = engine.from_dense(ctx, Xt)
    ref2 = engine.rsvd(ctx, mat2, 12, random_state=9)
    got2 = sharded.sharded_rsvd(sharded.HipPanelOps(ctx, mat2), sharded.Comm(), 12, 400, 0, random_state=9)
    for a, b in zip(ref2, got2):
        assert np.array_equal(a, b)


def test_context_on_a_non_default_stream():
    """A context created inside `torch.cuda.stream(s)` runs on s (eofx_ctx_create takes torch's current stream), so torch
    work issued on s between engine calls is ordered with the engine's kernels; `use_stream` re-binds.  Same results as on
    the default stream, bit for bit."""
    import torch
    from xeofs_amd import engine

    X = torch.as_tensor(_field(300, 2048, seed=31), device="cuda")
    ref = engine.fit(engine.default_context(0), X, 6, random_state=1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c2 = engine.Context(0)
        Y = X * 1.0                                   # produced on `side`: only ordered with a context that runs there
        got = engine.fit(c2, Y, 6, random_state=1)
        assert np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4])
    c2.use_stream(torch.cuda.default_stream())
    got2 = engine.fit(c2, X, 6, random_state=1)
    assert np.array_equal(got2[3], ref[3])
    for r in (ref, got, got2):
        r[0].free()
    c2.close()


def test_sharded_fused_first_pass_world1(ctx):
    """The feature-sharded fit with the statistics taken during each rank's first product (`sharded_fit_first` ->
    `eofx_fit_first_f32`, then `sharded_rsvd(first=...)`): at world size 1 the panel-level driver and the one-call engine
    fit run the same sequence and must agree bit for bit; a shard with a NaN mask falls back on its own."""
    import torch
    from xeofs_amd import engine, sharded

    n, p, k = 400, 3072, 12
    X = torch.as_tensor(_field(n, p, rank=9, seed=21), device="cuda")
    comm = sharded.Comm()
    mat, st, first = sharded.sharded_fit_first(ctx, X, comm, k, p, random_state=9)
    assert first is not None and st["fused"] and (st["p_total"], st["p_offset"]) == (p, 0)
    U, s, V = sharded.sharded_rsvd(sharded.HipPanelOps(ctx, mat), comm, k, p, 0, random_state=9, first=first)
    mat2, st2, U2, s2, V2 = engine.fit(ctx, X, k, random_state=9)
    assert st2["fused"]
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    assert np.array_equal(st["mean"], st2["mean"]) and st["total_variance"] == st2["total_variance"]
    out = sharded.sharded_eof_fit(ctx, X, comm, k, random_state=9)
    assert np.array_equal(out["components"], V2) and np.array_equal(out["norms"], s2.astype(np.float64))
    # a sketch handed over as a future (drawn on a worker thread) is the same fit
    outf = sharded.sharded_eof_fit(ctx, X, comm, k, random_state=9, omega=engine.SketchFuture(n, k + 10, 9))
    assert np.array_equal(outf["components"], V2) and np.array_equal(outf["norms"], out["norms"])
    mat.free(); mat2.free(); out["input_data"].free(); outf["input_data"].free()
    Xn = X.clone()
    Xn[:, 100:140] = float("nan")
    mat3, st3, first3 = sharded.sharded_fit_first(ctx, Xn, comm, k, p, random_state=9)
    # the engine fell back to the statistics pass + compaction by itself and still handed over the first product
    assert not st3["fused"] and st3["p_total"] == p - 40 == mat3.p and first3 is not None and first3[1].shape[0] == mat3.p_pad
    U3, s3, V3 = sharded.sharded_rsvd(sharded.HipPanelOps(ctx, mat3), comm, k, st3["p_total"], 0, random_state=9, first=first3)
    ref = orc.eof_fit(Xn.cpu().numpy().astype(np.float64), k, random_state=9)
    _check_factors(U3, s3, V3, ref, k)
    mat3.free()


def _check_factors(U, s, V, ref, k, tol=1e-5):
    assert np.all(np.abs(s - ref["norms"]) <= tol * ref["norms"][0]), (s, ref["norms"])
    sv = ref["norms"]
    for j in range(k):
        if _gap_ok(sv, j):
            c = float(np.dot(V[:, j].astype(np.float64), ref["components"][:, j]))
            assert c >= 1 - _cos_tol(sv, j), (j, c)          # same sign, same direction
            cu = float(np.dot(U[:, j].astype(np.float64), ref["scores"][:, j])) / float(np.linalg.norm(ref["scores"][:, j]))
            assert cu >= 1 - _cos_tol(sv, j), (j, cu)


@pytest.mark.parametrize("n,p,k,opts", [
    (300, 2048, 6, {}),
    (200, 5000, 20, {"standardize": True}),
    (517, 4100, 12, {"weights": True}),                 # n % 16 != 0: the tail rows come from the finalize kernel
    (1000, 9000, 50, {"standardize": True, "weights": True}),
    (333, 1036, 8, {"center": False}),
    (96, 40000, 10, {}),                                 # several column blocks, one slab pair per split
    (5000, 20480, 20, {}),                               # L = 32 panel (k + 10 = 30): the one-sub-tile variant
])
def test_fused_fit_vs_two_step_and_oracle(ctx, n, p, k, opts):
    """eofx_fit_f32: the column statistics ride on the first pass of the randomized SVD (eofx_fit.hpp).  The result must
    be the two-step result (statistics pass, then the decomposition) to float32 rounding, and match the float64
    oracle within the same tolerances as the two-step path."""
    from xeofs_amd import engine

    rng = np.random.default_rng(n + p)
    X = _field(n, p, rank=10, seed=n + k)
    X += (10.0 * rng.standard_normal(p)).astype(np.float32)          # per-feature offsets well above the noise
    w = (0.2 + rng.random(p)) if opts.get("weights") else None
    center, standardize = opts.get("center", True), opts.get("standardize", False)
    mat, st, U, s, V = engine.fit(ctx, X, k, center=center, standardize=standardize, feature_weights=w, random_state=7)
    assert st["fused"], engine.fit_info(ctx)
    assert mat.layout() == (False, True) and not mat.has_sample_layout()     # in place: nothing was written
    mat2, st2 = engine.preprocess(ctx, X, center, standardize, w, in_place=True)
    U2, s2, V2 = engine.rsvd(ctx, mat2, k, random_state=7)
    # statistics: float64 sums of float32-exact terms on both paths
    np.testing.assert_allclose(st["mean"], st2["mean"], rtol=0, atol=2e-6 * (np.abs(st2["mean"]).max() + st2["std"].max()))
    np.testing.assert_allclose(st["std"], st2["std"], rtol=2e-6)
    assert abs(st["total_variance"] - st2["total_variance"]) <= 2e-6 * st2["total_variance"]
    assert st["n"] == n and st["p"] == p and st["valid_feature"].all() and st["valid_sample"].all()
    assert np.all(np.abs(s - s2) <= 2e-6 * s2[0]), (s, s2)
    ref = orc.eof_fit(X.astype(np.float64), k, center=center, standardize=standardize, feature_weights=w, random_state=7)
    _check_factors(U, s, V, ref, k)
    assert abs(st["total_variance"] - ref["total_variance"]) <= 1e-6 * ref["total_variance"]
    # the matrix it leaves behind is the two-step matrix: same projection
    P1 = engine.project(ctx, mat, V2)
    P2 = engine.project(ctx, mat2, V2)
    assert np.abs(P1 - P2).max() <= 2e-6 * np.abs(P2).max()
    # bitwise reproducible
    mat3, st3, U3, s3, V3 = engine.fit(ctx, X, k, center=center, standardize=standardize, feature_weights=w, random_state=7)
    assert np.array_equal(U, U3) and np.array_equal(s, s3) and np.array_equal(V, V3)
    assert np.array_equal(st["mean"], st3["mean"])
    for m_ in (mat, mat2, mat3):
        m_.free()


@pytest.mark.parametrize("standardize,use_w", [(False, False), (True, True)])
def test_fused_fit_with_land_mask(ctx, standardize, use_w):
    """The one-call fit on a field with all-NaN grid points (land / sea mask; SURVEY §8d's NaN variant) when the caller
    allows the masked in-place layout: the statistics still ride on the first pass -- the probe marks the columns that
    are NaN in all sampled rows, the pass keeps their NaNs confined to their own rows and verifies that they hold no
    finite value -- and the matrix ends in layout mode 3.  Same factors as the two-step masked path and the oracle;
    a column that only LOOKS masked in the sampled rows, a mask that is too large, and allow_masked=False fall back."""
    from xeofs_amd import engine

    n, nlat, nlon, k = 300, 24, 48, 6
    X, lat = orc.synthetic_field(n, nlat, nlon, rank=8, seed=4, nan_frac=0.3)
    X = np.ascontiguousarray(X.reshape(n, -1), dtype=np.float32)
    P = X.shape[1]
    w = np.repeat(np.sqrt(np.cos(np.deg2rad(lat)).clip(0, 1)), nlon) if use_w else None
    mat, st, U, s, V = engine.fit(ctx, X, k, standardize=standardize, feature_weights=w, random_state=2, allow_masked=True)
    info = engine.fit_info(ctx)
    assert st["fused"] and info["fused"] and info["reason"] == 0, info
    assert mat.masked and mat.layout() == (False, True)
    vf = ~np.isnan(X).all(axis=0)
    pv = int(vf.sum())
    assert np.array_equal(st["valid_feature"], vf) and st["p"] == pv and mat.p == pv and mat.p_phys == P
    assert U.shape == (n, k) and V.shape == (pv, k)
    ref = orc.eof_fit(X.astype(np.float64), k, True, standardize, w, random_state=2)
    _check_factors(U, s, V, ref, k)
    assert abs(st["total_variance"] - ref["total_variance"]) <= 1e-6 * ref["total_variance"]
    if st["mean"] is not None:
        assert np.allclose(st["mean"][vf], np.nanmean(X.astype(np.float64), axis=0)[vf], rtol=1e-6, atol=1e-6)
        assert np.isnan(st["mean"][~vf]).all()
    # the two-step masked path (statistics pass + masked in-place layout): same spectrum to rounding
    mat2, st2 = engine.preprocess(ctx, X, standardize=standardize, feature_weights=w, in_place=True, allow_masked=True)
    U2, s2, V2 = engine.rsvd(ctx, mat2, k, random_state=2)
    assert mat2.masked and np.allclose(s, s2, rtol=2e-6)
    # later passes over the matrix see zero columns at the mask: projection of the components gives the scores
    sc = engine.project(ctx, mat, V)
    assert np.allclose(sc, U * s, atol=3e-5 * s[0])
    mat.free(); mat2.free()
    # not allowed: the call falls back to the Sanitizer's compaction (reason 1: NaN in the sampled rows)
    mat, st, U3, s3, V3 = engine.fit(ctx, X, k, standardize=standardize, feature_weights=w, random_state=2)
    assert not st["fused"] and not mat.masked and engine.fit_info(ctx)["reason"] == 1 and V3.shape == (pv, k)
    assert np.allclose(s3, s, rtol=2e-6)
    mat.free()
    # a column that is NaN in every SAMPLED row but holds data elsewhere: the pass notices (its maximum becomes inf)
    Y = X.copy()
    cols = np.flatnonzero(vf)[:3]
    rows9 = sorted({min((n * t) // 8, n - 1) for t in range(8)} | {n - 1})
    Y[np.ix_(rows9, cols)] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        engine.fit(ctx, Y, k, random_state=2, allow_masked=True)
    assert engine.fit_info(ctx)["reason"] in (4, 5)
    # more than 40 % of the grid points masked: outside the in-place range, compacted instead
    Z = X.copy()
    Z[:, np.flatnonzero(vf)[: int(0.3 * P)]] = np.nan
    mat, st, U4, s4, V4 = engine.fit(ctx, Z, k, random_state=2, allow_masked=True)
    assert not st["fused"] and not mat.masked and engine.fit_info(ctx)["reason"] == 6
    ref4 = orc.eof_fit(Z.astype(np.float64), k, random_state=2)
    _check_factors(U4, s4, V4, ref4, k)
    mat.free()


@pytest.mark.parametrize("one_call", [False, True])
def test_masked_in_place_skips_masked_runs(ctx, one_call):
    """Masked grid points that come in RUNS (land in an ocean field): the MASK kernels do not read them -- a wave of
    atb_f16 whose 128 features are all masked only helps staging B, a workgroup of four such waves leaves at once, and
    axb_f16 walks the list of 64-feature slab pairs that hold at least one valid feature (eofx_mat::act).  Runs that
    cover whole workgroups, single waves and single slab pairs, next to scattered masked points; results against the
    oracle and the compacted matrix."""
    from xeofs_amd import engine

    n, P, k = 260, 4608, 7
    rng = np.random.default_rng(12)
    X = _field(n, P, seed=9)
    dead = np.zeros(P, bool)
    for a, b in ((0, 1024), (1500, 1700), (2048, 2176), (2240, 2304), (3000, 3300), (4544, 4608)):
        dead[a:b] = True
    dead[rng.choice(np.flatnonzero(~dead), 40, replace=False)] = True           # scattered points inside valid slabs
    assert 0.6 * P < (~dead).sum()
    X[:, dead] = np.nan
    w = rng.uniform(0.5, 1.5, P)
    ref = orc.eof_fit(X.astype(np.float64), k, True, False, w, random_state=3)
    if one_call:
        mat, st, U, s, V = engine.fit(ctx, X, k, feature_weights=w, random_state=3, allow_masked=True)
        assert st["fused"]
    else:
        mat, st = engine.preprocess(ctx, X, True, False, w, in_place=True, allow_masked=True)
        U, s, V = engine.rsvd(ctx, mat, k, random_state=3)
    assert mat.masked and np.array_equal(st["valid_feature"], ~dead) and V.shape == (int((~dead).sum()), k)
    _check_factors(U, s, V, ref, k)
    mat2, st2 = engine.preprocess(ctx, X, True, False, w)                          # compacted, two layouts
    U2, s2, V2 = engine.rsvd(ctx, mat2, k, random_state=3)
    assert np.all(np.abs(s - s2) <= 2e-6 * s2[0])
    P1, P2 = engine.project(ctx, mat, V2), engine.project(ctx, mat2, V2)         # one more axb pass over the list
    assert np.abs(P1 - P2).max() <= 2e-6 * np.abs(P2).max()
    U3, s3, V3 = engine.rsvd(ctx, mat, k, random_state=3)                         # reproducible bit for bit
    if not one_call:
        assert np.array_equal(s, s3) and np.array_equal(V, V3)
    mat.free(); mat2.free()


def test_wide_sketch_on_an_in_place_matrix(ctx, monkeypatch):
    """Sketches of 65+ columns (EOF with 55+ modes) on an in-place matrix: the X^T Z passes run in the 128-column tile over
    the raw field; for the X Y passes the engine builds the sample-contiguous layout once where HBM has room (128-column
    tile again) and otherwise (EOFX_NO_WIDE_XT=1 stands in for a full HBM) keeps streaming the field through axb_f16, 64
    columns per launch.  Same factors either way, against the oracle."""
    from xeofs_amd import engine

    n, p = 400, 6000
    X = _field(n, p, rank=110, seed=21)
    for k in (70, 90, 200):        # 80 / 100 / 210 sketch columns: panels of 96 (one partial wide tile), 128, 224 (128 + partial)
        _wide_case(ctx, monkeypatch, X, k)
    # 64 columns and fewer never build anything
    mat, st = engine.preprocess(ctx, X, in_place=True)
    engine.rsvd(ctx, mat, 50, random_state=6)
    assert not mat.has_sample_layout()
    mat.free()


def _wide_case(ctx, monkeypatch, X, k):
    from xeofs_amd import engine

    ref = orc.eof_fit(X.astype(np.float64), k, random_state=6)
    out = {}
    for no_xt in ("1", None):
        if no_xt:
            monkeypatch.setenv("EOFX_NO_WIDE_XT", no_xt)
        else:
            monkeypatch.delenv("EOFX_NO_WIDE_XT", raising=False)
        mat, st = engine.preprocess(ctx, X, in_place=True)
        assert mat.layout() == (False, True) and not mat.has_sample_layout()
        U, s, V = engine.rsvd(ctx, mat, k, random_state=6)
        assert mat.has_sample_layout() == (no_xt is None)              # built only when allowed; the raw field stays the X^T operand
        assert mat.layout() == (False, True)
        _check_factors(U, s, V, ref, k)
        out[no_xt] = s
        mat.free()
    assert np.all(np.abs(out["1"] - out[None]) <= 2e-6 * out[None][0])


def test_fused_fit_falls_back(ctx):
    """NaN fields, sketches wider than 64 columns and n >= P take the two-step path inside the same call -- with the
    Sanitizer's policies and error messages -- and say so."""
    from xeofs_amd import engine

    X, lat = orc.synthetic_field(240, 16, 32, rank=6, seed=3, nan_frac=0.25)     # all-NaN grid points (land mask)
    X = np.ascontiguousarray(X.reshape(240, -1), dtype=np.float32)
    mat, st, U, s, V = engine.fit(ctx, X, 5, random_state=2)
    assert not st["fused"] and engine.fit_info(ctx)["reason"] in (1, 3)
    pv = int(st["valid_feature"].sum())
    assert 0 < pv < X.shape[1] and V.shape == (pv, 5) and U.shape == (240, 5)
    mat2, st2 = engine.preprocess(ctx, X, in_place=True)
    U2, s2, V2 = engine.rsvd(ctx, mat2, 5, random_state=2)
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    mat.free(); mat2.free()
    # a NaN that the eight sampled rows do not see: found by the statistics of the first pass
    Y = _field(400, 1024, seed=5)
    Y[137, 900] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        engine.fit(ctx, Y, 4, random_state=1)
    assert engine.fit_info(ctx)["reason"] == 3
    Y[:, 900] = np.nan                              # the whole feature: dropped, as by the Sanitizer
    mat, st, U, s, V = engine.fit(ctx, Y, 4, random_state=1)
    assert not st["fused"] and V.shape == (1023, 4) and not st["valid_feature"][900]
    mat.free()
    Y = _field(400, 2048, seed=5)                   # k + 10 = 32 fills the panel: no spare column for the ones
    mat, st, U, s, V = engine.fit(ctx, Y, 22, random_state=1)
    assert not st["fused"] and engine.fit_info(ctx)["reason"] == -1
    mat.free()
    Z = _field(600, 400, seed=6)                    # n >= P: the sketch lives on the feature side
    mat, st, U, s, V = engine.fit(ctx, Z, 4, random_state=1)
    assert not st["fused"] and engine.fit_info(ctx)["reason"] == -1
    ref = orc.eof_fit(Z.astype(np.float64), 4, random_state=1)
    _check_factors(U, s, V, ref, 4)
    mat.free()
    # an outlier far outside what the sampled rows suggest: the provisional fp16 range overflows, the call recovers
    Wd = _field(512, 2048, seed=8)
    Wd[300, 77] = 3.0e7
    mat, st, U, s, V = engine.fit(ctx, Wd, 3, random_state=1)
    # (4: the range check caught it; 5: the converted value also overflowed to infinity -- the split rounds to nearest since
    #  round 5 -- which the statistics cannot tell from an infinity in the field; either way the call falls back and recovers)
    assert not st["fused"] and engine.fit_info(ctx)["reason"] in (4, 5)
    ref = orc.eof_fit(Wd.astype(np.float64), 3, random_state=1)
    assert np.all(np.abs(s - ref["norms"]) <= 1e-5 * ref["norms"][0])
    mat.free()


@pytest.mark.parametrize("spread,fused", [(2.0, True), (8.0, False), (16.0, False)])
def test_standardize_with_mixed_feature_scales(ctx, spread, fused):
    """Mixed-unit fields (pressure in Pa next to specific humidity) under standardize=True.  The fused first pass splits the
    RAW values against one scale: a feature whose standard deviation is below 2^-14 of the field's largest value would reach
    the matrix cores with a fraction of its bits and 1 / std would magnify the loss (round 5, tools/scale_probe.py: 7e-6 at
    8 orders of magnitude between features, nonsense at 16) -- such a fit goes back to the two-step path (reason 7), which
    maps every feature to unit variance before the split.  Either way: the float64 oracle's values to 1e-6."""
    from xeofs_amd import engine

    rng = np.random.default_rng(5)
    n, p, k = 400, 3000, 8
    base = (rng.standard_normal((n, 10)) * 2.0 ** -np.arange(10)) @ rng.standard_normal((10, p)) + 0.05 * rng.standard_normal((n, p))
    X = (base * 10.0 ** rng.uniform(-spread / 2, spread / 2, p)).astype(np.float32)
    mat, st, U, s, V = engine.fit(ctx, X, k, standardize=True, random_state=1)
    mat.free()
    info = engine.fit_info(ctx)
    assert info["fused"] == fused and (fused or info["reason"] == 7), info
    ref = orc.eof_fit(X.astype(np.float64), k, standardize=True, random_state=1)
    assert np.all(np.abs(s - ref["norms"]) <= 1e-6 * ref["norms"][0]), np.abs(s - ref["norms"]).max() / ref["norms"][0]
    for j in range(4):
        assert abs(float(np.dot(V[:, j].astype(np.float64), ref["components"][:, j]))) >= 1 - 1e-6, j
    # without standardize the small features carry no weight: the fused pass stays, and is exact
    mat, st, U, s, V = engine.fit(ctx, X, k, standardize=False, random_state=1)
    mat.free()
    assert engine.fit_info(ctx)["fused"]
    ref = orc.eof_fit(X.astype(np.float64), k, standardize=False, random_state=1)
    assert np.all(np.abs(s - ref["norms"]) <= 1e-6 * ref["norms"][0])


@pytest.mark.parametrize("opts", [{}, {"standardize": True}, {"weights": True}])
def test_masked_in_place_layout(ctx, opts):
    """Layout mode 3: a field with all-NaN grid points (land / sea mask, sanitizer.py:80-126) stays IN PLACE -- the masked
    features are zero columns of the engine's matrix (scale 0, bits ANDed to +0 by the MASK kernels) instead of being
    compacted into a second copy.  Same factors as the compacted matrix, same oracle tolerances, nothing written."""
    from xeofs_amd import engine

    n, nlat, nlon, k = 300, 24, 40, 8
    X, lat = orc.synthetic_field(n, nlat, nlon, rank=8, seed=11, nan_frac=0.3)
    X = np.ascontiguousarray(X, dtype=np.float32)
    w = np.repeat(orc.sqrt_cos_lat_weights(lat), nlon) if opts.get("weights") else None
    std = opts.get("standardize", False)
    mat, st = engine.preprocess(ctx, X, True, std, w, in_place=True, allow_masked=True)
    pv = int(st["valid_feature"].sum())
    assert mat.masked and mat.p == pv == st["p"] and mat.p_phys == X.shape[1] and 0.6 * X.shape[1] < pv < X.shape[1]
    assert mat.layout() == (False, True) and not mat.has_sample_layout()          # nothing was written
    U, s, V = engine.rsvd(ctx, mat, k, random_state=4)
    assert V.shape == (pv, k) and U.shape == (n, k)
    assert not mat.has_sample_layout()                                               # the passes streamed the field
    ref = orc.eof_fit(X.astype(np.float64), k, standardize=std, feature_weights=w, random_state=4)
    assert np.array_equal(st["valid_feature"], ref["valid_feature"])
    _check_factors(U, s, V, ref, k)
    assert abs(st["total_variance"] - ref["total_variance"]) <= 1e-6 * ref["total_variance"]
    # against the compacted two-layout matrix of the same field
    mat2, st2 = engine.preprocess(ctx, X, True, std, w)
    assert not mat2.masked and mat2.p == pv
    U2, s2, V2 = engine.rsvd(ctx, mat2, k, random_state=4)
    assert np.all(np.abs(s - s2) <= 2e-6 * s2[0])
    D1, D2 = mat.download(), mat2.download()            # the masked view, compacted by the engine wrapper
    assert D1.shape == D2.shape and np.abs(D1 - D2).max() <= 2e-6 * np.abs(D2).max()
    P1, P2 = engine.project(ctx, mat, V2), engine.project(ctx, mat2, V2)
    assert np.abs(P1 - P2).max() <= 2e-6 * np.abs(P2).max()
    assert np.allclose(engine.feature_norms(ctx, mat), engine.feature_norms(ctx, mat2), rtol=2e-6)
    assert np.allclose(engine.sample_norms(ctx, mat), engine.sample_norms(ctx, mat2), rtol=2e-6)
    # the one-call fit ends in the same layout when it meets the mask -- since round 3 with the statistics still taken
    # during the first pass (test_fused_fit_with_land_mask); its provisional shift differs from the exact mean by rounding
    mat3, st3, U3, s3, V3 = engine.fit(ctx, X, k, standardize=std, feature_weights=w, random_state=4, allow_masked=True)
    assert st3["fused"] and mat3.masked and V3.shape == (pv, k) and np.array_equal(st3["valid_feature"], st["valid_feature"])
    assert np.all(np.abs(s3 - s) <= 2e-6 * s[0])
    _check_factors(U3, s3, V3, ref, k)
    # transform of new data with the same mask: in place again, scores of the training data come back
    mat4, vs4 = engine.apply(ctx, X, st["mean"], st["std"] if std else None, w, st["valid_feature"], in_place=True,
                             allow_masked=True)
    assert mat4.masked and not mat4.has_sample_layout()
    S4 = engine.project(ctx, mat4, V)
    assert np.abs(S4 - U * s).max() <= 1e-5 * s[0]
    for m_ in (mat, mat2, mat3, mat4):
        m_.free()
    # an isolated NaN inside a valid feature is still the Sanitizer's error
    Y = X.copy()
    Y[17, int(np.flatnonzero(st["valid_feature"])[5])] = np.nan
    with pytest.raises(ValueError, match="partial NaN"):
        engine.preprocess(ctx, Y, in_place=True, allow_masked=True)
    # more than 40 % masked (or fewer valid features than samples): compaction, as before
    Z, _ = orc.synthetic_field(n, nlat, nlon, rank=8, seed=12, nan_frac=0.55)
    mat5, st5 = engine.preprocess(ctx, np.ascontiguousarray(Z, dtype=np.float32), in_place=True, allow_masked=True)
    assert not mat5.masked and mat5.p == st5["p"] == mat5.p_phys
    mat5.free()


def test_documented_size_limits_fail_loudly(ctx):
    """The limits DESIGN.md §10 lists raise instead of degrading silently."""
    import torch
    from xeofs_amd import engine, rotation
    from xeofs_amd.complex_svd import complex_rsvd

    with pytest.raises(NotImplementedError, match="more than 256 modes"):
        rotation.promax(ctx, np.ones((600, 257), np.float32))
    rng = np.random.default_rng(0)
    A = engine.from_dense(ctx, rng.standard_normal((300, 400)).astype(np.float32))
    B = engine.from_dense(ctx, rng.standard_normal((300, 400)).astype(np.float32))
    with pytest.raises(NotImplementedError, match="complex sketch width"):
        complex_rsvd(ctx, A, B, 60)                      # panel-level (sharded) driver: 60 + 10 oversamples > 64
    with pytest.raises(ValueError, match="complex sketch width"):
        engine.rsvd_c64(ctx, A, B, 60)                   # engine entry: 60 + 10 oversamples > 64
    with pytest.raises(ValueError, match="rank of the dataset"):
        engine.rsvd(ctx, A, 301)
    A.free(); B.free()


def test_peaked_spectrum_on_a_large_tall_panel(ctx):
    """A peaked spectrum (sigma_1 / sigma_k ~ 6) with many unconverged noise-bulk modes on a tall panel above the
    16 MB size rule (no re-normalisation of the tall panel inside the iterations): the result matches the float64
    oracle to the strict tolerance, and the sharded driver is bitwise equal at world size 1."""
    from xeofs_amd import engine, sharded

    rng = np.random.default_rng(3)
    n, p, k = 160, 70000, 36
    amp = 60.0 * 0.45 ** np.arange(5)
    X = ((rng.standard_normal((n, 5)) * amp) @ rng.standard_normal((5, p)) / np.sqrt(p) * 40 + rng.standard_normal((n, p))).astype(np.float32)
    X -= X.mean(0)
    mat = engine.from_dense(ctx, X)
    assert not sharded._orth_tall(mat.p, 64)          # above the size rule: the tall panel is not re-normalised
    U, s, V = engine.rsvd(ctx, mat, k, random_state=9)
    Uo, so, Vo = orc.decomposer_fit(X.astype(np.float64), k, random_state=9, solver="randomized")
    assert so[0] / so[-1] > 5
    assert np.all(np.abs(s - so) <= 1e-5 * so + 3e-6 * so[0])
    ops = sharded.HipPanelOps(ctx, mat)

    class NoComm:
        active = False
        def sum_(self, t): return t
        def max_(self, t): return t
        def min_(self, t): return t

    U2, s2, V2 = sharded.sharded_rsvd(ops, NoComm(), k, p, 0, random_state=9)
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    mat.free()


@pytest.mark.parametrize("n,P,std,wts", [(300, 1024, False, False), (517, 2500, True, True), (1000, 7300, False, True),
                                         (96, 516, True, False)])
def test_raw_mode_equals_two_layout_mode(ctx, n, P, std, wts):
    """eofx_ctx_set_layout(1): the feature-contiguous layout is never written, X^T Z streams the raw field through the
    Scaler map (atb_f16_kernel<NB, true>).  The map is the expression the apply kernel writes; the only difference is
    that the kernel's low fp16 term is taken from the unrounded product (x - mean) * scale (a fused multiply-subtract),
    so with scale == 1 the product is BITWISE that of the two-layout mode and otherwise equal to 2^-24 per element --
    for row counts that are not a multiple of 32 and column counts that are not a multiple of 512 (clamped reads) too."""
    import torch
    from xeofs_amd import engine

    def close(a, b, tol):
        return float((a.double() - b.double()).norm() / b.double().norm()) <= tol

    g = torch.Generator(device="cuda").manual_seed(n * 7 + P)
    X = torch.randn((n, P), device="cuda", generator=g) * 3 + torch.linspace(-40, 250, P, device="cuda")
    w = np.linspace(0.2, 1.7, P) if wts else None
    exact = not std and not wts
    m2, st2 = engine.preprocess(ctx, X, True, std, w)
    m1, st1 = engine.preprocess(ctx, X, True, std, w, keep_raw=True)
    assert m2.layout() == (True, False) and m1.layout() == (False, True)
    assert st1["total_variance"] == st2["total_variance"] and np.array_equal(st1["mean"], st2["mean"])
    Z = torch.randn((m1.n_pad, 64), device="cuda", generator=g)
    Z[n:] = 0
    T1, T2 = engine.panel_tmul(ctx, m1, Z, prec="f16x3"), engine.panel_tmul(ctx, m2, Z, prec="f16x3")
    assert torch.equal(T1, T2) if exact else close(T1, T2, 2e-7)
    k = 7
    for a, b in zip(engine.rsvd(ctx, m1, k, random_state=3), engine.rsvd(ctx, m2, k, random_state=3)):
        assert np.array_equal(a, b) if exact else np.allclose(np.abs(a), np.abs(b), rtol=0, atol=2e-5 * np.abs(b).max())
    assert m1.layout() == (False, True)          # still streaming the raw field
    # other precisions, the Gram matrix and the download need the layout itself: rebuilt from the sample-contiguous one
    assert torch.equal(engine.panel_tmul(ctx, m1, Z, prec="f32"), engine.panel_tmul(ctx, m2, Z, prec="f32"))
    assert m1.layout()[0]
    assert np.array_equal(m1.download(), m2.download())
    m1.release_raw()
    assert m1.layout() == (True, False)
    assert torch.equal(engine.panel_tmul(ctx, m1, Z, prec="f16x3"), T2)
    # a host field: the staged copy is owned by the matrix
    Xh = X.cpu().numpy()
    m3, _ = engine.preprocess(ctx, Xh, True, std, w, keep_raw=True)
    assert m3.layout() == (False, True)
    assert torch.equal(engine.panel_tmul(ctx, m3, Z, prec="f16x3"), T1)
    # anything that drops features or samples falls back to the two-layout mode
    Xn = X.clone()
    Xn[:, 5] = float("nan")
    m4, st4 = engine.preprocess(ctx, Xn, True, std, w, keep_raw=True)
    assert m4.layout() == (True, False) and st4["p"] == P - 1
    for m in (m1, m2, m3, m4):
        m.free()


@pytest.mark.parametrize("n,P,std,wts", [(300, 1024, False, False), (517, 2500, True, True), (1000, 7300, False, True),
                                         (96, 516, True, False), (2100, 640, True, True), (63, 100000, False, False)])
def test_in_place_mode(ctx, n, P, std, wts, monkeypatch):
    """eofx_ctx_set_layout(2): the preprocessor writes NO copy of the matrix; X^T Z streams the field through the Scaler
    map exactly as in raw mode and X Y streams it along its rows (axb_f16_kernel: another summation order, so equal to
    rounding, and checked against a float64 product); the randomized SVD built on them
    meets the float64 oracle at the usual 1e-5.  Row / column counts off every tile size; sketch widths 32, 64 and 96.
    Layouts appear only when an entry point needs them and are bitwise what the apply kernel writes.  (EOFX_NO_WIDE_XT:
    panels of 96+ columns stay on axb_f16 here, as they do when HBM has no room for the sample-contiguous layout;
    the other branch is test_wide_sketch_on_an_in_place_matrix.)"""
    import torch
    monkeypatch.setenv("EOFX_NO_WIDE_XT", "1")
    from oracle import eof_oracle as orc
    from xeofs_amd import engine

    g = torch.Generator(device="cuda").manual_seed(n * 7 + P)
    X = torch.randn((n, P), device="cuda", generator=g) * (1 + 3 * torch.rand(P, device="cuda", generator=g)) \
        + torch.linspace(-40, 250, P, device="cuda")
    w = np.linspace(0.2, 1.7, P) if wts else None
    m2, st2 = engine.preprocess(ctx, X, True, std, w)
    m0, st0 = engine.preprocess(ctx, X, True, std, w, in_place=True)
    assert m0.layout() == (False, True) and not m0.has_sample_layout()
    assert st0["total_variance"] == st2["total_variance"] and np.array_equal(st0["mean"], st2["mean"])
    Z = torch.randn((m0.n_pad, 64), device="cuda", generator=g)
    Z[n:] = 0
    m1, _ = engine.preprocess(ctx, X, True, std, w, keep_raw=True)
    assert torch.equal(engine.panel_tmul(ctx, m0, Z, prec="f16x3"), engine.panel_tmul(ctx, m1, Z, prec="f16x3"))
    m1.free()
    Xp = torch.as_tensor(m2.download(), device="cuda").double()
    for L in (64, 32, 96):
        Y = torch.randn((m0.p_pad, L), device="cuda", generator=g)
        Y[P:] = 0
        W0, W2 = engine.panel_mul(ctx, m0, Y, prec="f16x3"), engine.panel_mul(ctx, m2, Y, prec="f16x3")
        ref = Xp @ Y[:P].double()
        e0 = float((W0[:n].double() - ref).norm() / ref.norm())
        e2 = float((W2[:n].double() - ref).norm() / ref.norm())
        assert e0 < 2 * e2 + 1e-7, (L, e0, e2)
        assert m0.n_pad == n or float(W0[n:].abs().max()) == 0.0        # padded rows of the panel stay zero
    assert not m0.has_sample_layout()                                   # nothing was materialised so far
    k = min(7, n - 1, P - 1)
    U, s, V = engine.rsvd(ctx, m0, k, random_state=3)
    Xh = m2.download().astype(np.float64)
    Uo, so, Vo = orc.decomposer_fit(Xh, k, random_state=3, solver="randomized")
    assert np.max(np.abs(s - so) / so) < 1e-5
    assert np.min(np.abs(np.sum(V * Vo, axis=0))) > 1 - 1e-4
    assert not m0.has_sample_layout()
    # the on-demand layouts: bitwise the ones the apply kernel writes in the two-layout mode
    assert np.array_equal(m0.download(), m2.download()) and m0.has_sample_layout()
    m0.free()
    # release_raw() on an in-place matrix materialises first (the field was the only copy)
    m5, _ = engine.preprocess(ctx, X.cpu().numpy(), True, std, w, in_place=True)
    m5.release_raw()
    assert m5.layout()[1] is False and m5.has_sample_layout()
    assert torch.equal(engine.panel_mul(ctx, m5, Y, prec="f16x3"), engine.panel_mul(ctx, m2, Y, prec="f16x3"))
    m5.free()
    m2.free()


@pytest.mark.parametrize("peak", [100.0, 300.0, 1000.0, 3000.0])
def test_peaked_spectrum_float64_passes(ctx, peak):
    """SURVEY H2 / tools/cond_study.py: the reference promotes the field to float64 and normalises after every product.
    k = 40 modes on 200 samples, 36 of them in the unconverged noise bulk, leading modes up to 41 000x above them.
    Round 1 left the tall panel un-normalised between the two products of an iteration: 6e-5 at sigma_1 / sigma_k = 1370,
    lost beyond 4000.  Now the first iteration always takes that step and a peaked spectrum keeps it: the default
    split-fp16 passes stay within 1e-5 up to 41 000, and `set_precision("f64", "f64")` -- exact products, float64 sums on
    the fp64 matrix cores -- within 1e-7."""
    from xeofs_amd import engine

    rng = np.random.default_rng(5)
    n, p, k = 200, 70000, 40
    amp = peak * 0.6 ** np.arange(4)
    X = ((rng.standard_normal((n, 4)) * amp) @ rng.standard_normal((4, p)) + rng.standard_normal((n, p))).astype(np.float32)
    X -= X.mean(0)
    Uo, so, Vo = orc.decomposer_fit(X.astype(np.float64), k, random_state=2, solver="randomized")
    mat = engine.from_dense(ctx, X)
    ctx.set_precision("f64", "f64")
    try:
        U, s, V = engine.rsvd(ctx, mat, k, random_state=2)
    finally:
        ctx.set_precision("f16x3", "f16x3")
    Ud, sd, Vd = engine.rsvd(ctx, mat, k, random_state=2)
    mat.free()
    err64, err_def = float(np.max(np.abs(s - so) / so)), float(np.max(np.abs(sd - so) / so))
    print(f"sigma_1/sigma_k = {so[0] / so[-1]:.0f}: float64 passes {err64:.1e}, default passes {err_def:.1e}")
    assert err64 <= 2e-7, (so[0] / so[-1], err64, err_def)
    # the default split-fp16 passes stay inside the tolerance too: the drivers detect the peaked spectrum after the
    # first iteration and keep re-normalising the tall panel (eofx_peaked_spectrum)
    assert err_def <= 1e-5, (so[0] / so[-1], err64, err_def)


@pytest.mark.gpu
@pytest.mark.parametrize("n,P,L,masked", [(1000, 4096, 64, False), (700, 3000, 96, False), (515, 8200, 32, False),
                                          (900, 6400, 64, True), (333, 5000, 96, True)])
def test_axb_dma_kernel_equals_the_register_path_bit_for_bit(monkeypatch, n, P, L, masked):
    """The in-place X Y product with the B slab moved by LDS-DMA (eofx_axb_dma.hpp: every load and LDS access of its pair
    loop is hand-scheduled inline assembly) against axb_f16_kernel, which leaves the scheduling to the compiler: same
    arithmetic in the same order, so the SAME BITS -- full and partial row tiles, a 32-column remainder, feature counts that
    are not multiples of 64, and a land mask (zero columns + the active-pair list)."""
    import torch
    from xeofs_amd import engine

    rng = np.random.default_rng(n + P)
    X = (280.0 + 5.0 * rng.standard_normal((n, P))).astype(np.float32)
    if masked:
        land = np.zeros(P, bool)
        land[500:1900] = True            # whole 64-feature pairs without a valid feature, and ragged edges
        land[rng.integers(0, P, 200)] = True
        X[:, land] = np.nan
    Xd = torch.as_tensor(X, device="cuda")
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("EOFX_AXB_DMA", flag)
        c = engine.Context(0)            # the switch is read by a context at its first in-place product
        mat, st = engine.preprocess(c, Xd, in_place=True, allow_masked=masked)
        assert mat.masked == masked
        Y = torch.zeros((mat.p_pad, L), dtype=torch.float32, device="cuda")
        Yh = np.random.default_rng(7).standard_normal((mat.p, L)).astype(np.float32)     # the same panel both times
        Y[: mat.p_phys if masked else mat.p] = torch.as_tensor(mat.scatter_rows(Yh) if masked else Yh, device="cuda")
        W = engine.panel_mul(c, mat, Y, prec="f16x3")
        torch.cuda.synchronize()
        outs.append(W.cpu().numpy().view(np.uint32).copy())
        if flag == "1":      # the panel planes are a per-context cache: trimming it and running again changes nothing
            c.trim()
            W2 = engine.panel_mul(c, mat, Y, prec="f16x3")
            torch.cuda.synchronize()
            assert np.array_equal(W2.cpu().numpy().view(np.uint32), outs[-1])
        mat.free()
    assert np.array_equal(outs[0], outs[1])
    assert np.isfinite(outs[1].view(np.float32)).all() and np.abs(outs[1].view(np.float32)[:n]).max() > 0


_NULL_SHAPES = [(300, 2000, 5, 10), (2000, 300, 5, 10), (300, 2000, 12, 30), (64, 5000, 3, 8), (1000, 1000, 7, 20)]


@pytest.mark.parametrize("n,p,r,k,entry", [sh + (e,) for sh in _NULL_SHAPES for e in ("rsvd", "fit", "fit_masked")
                                           if not (e == "fit_masked" and sh[0] >= sh[1])])
def test_more_modes_than_numerical_rank_real_path(ctx, n, p, r, k, entry):
    """Exactly low-rank data with more modes asked for than it has rank: scikit-learn's randomized_svd (a QR and a dense SVD,
    sklearn/utils/extmath.py) returns orthonormal factors whatever the values; so does the engine (fix_null_columns: the
    numerically null columns of both factors are re-orthonormalised against the others; round 5 -- before it they were
    rounding noise on the small side and zeroed columns on the tall side).  Leading values against the float64 oracle, null
    values at rounding level, the resolved modes' vectors unchanged by the repair."""
    from xeofs_amd import engine

    rng = np.random.default_rng(n + p + r)
    X = (rng.standard_normal((n, r)) * 2.0 ** -np.arange(r)) @ rng.standard_normal((r, p)) + 4.0
    X = X.astype(np.float32)
    if entry == "fit_masked":        # (a land mask kept in place: needs more valid features than samples)
        X[:, rng.choice(p, size=p // 4, replace=False)] = np.nan
    if entry == "rsvd":
        mat, _ = engine.preprocess(ctx, X, True, False, None)
        U, s, V = engine.rsvd(ctx, mat, k, random_state=1)
        mat.free()
    else:
        mat, st, U, s, V = engine.fit(ctx, X, k, random_state=1)
        mat.free()
    assert np.isfinite(U).all() and np.isfinite(V).all() and np.isfinite(s).all()
    assert np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max() < 2e-5
    assert np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max() < 2e-5
    Xv = X[:, ~np.isnan(X).all(axis=0)].astype(np.float64)
    Xc = Xv - Xv.mean(0)
    se = np.linalg.svd(Xc, compute_uv=False)[:k]
    good = se > 1e-4 * se[0]
    assert good.sum() == min(r, k)
    assert np.all(np.abs(s - se)[good] <= 2e-5 * se[0])
    assert np.all(s[~good] <= 1e-5 * se[0])
    # the resolved modes still are singular pairs of the matrix: X v = s u
    g = int(good.sum())
    assert V.shape[0] == Xv.shape[1]
    R = Xc @ V[:, :g].astype(np.float64) - U[:, :g].astype(np.float64) * s[:g]
    assert np.abs(R).max() <= 5e-5 * se[0]


def test_more_modes_than_the_cross_covariance_has_rank(ctx):
    """MCA on two fields that share three signals only, ten modes asked for: the singular vectors of the (rank-3 plus rounding)
    cross-covariance matrix stay orthonormal for the null modes too (fix_null_modes in eofx_crosscov_rsvd_f32)."""
    from xeofs_amd import engine

    rng = np.random.default_rng(8)
    n, p1, p2, k = 400, 900, 700, 10
    T = rng.standard_normal((n, 3))
    X = (T @ rng.standard_normal((3, p1))).astype(np.float32) + 2.0
    Y = (T @ rng.standard_normal((3, p2))).astype(np.float32) - 1.0
    mx, _ = engine.preprocess(ctx, X, True, False, None)
    my, _ = engine.preprocess(ctx, Y, True, False, None)
    out = engine.crosscov_rsvd(ctx, mx, my, k, random_state=2)
    Q1, s, Q2 = out["Q1"], out["s"], out["Q2"]
    mx.free(); my.free()
    assert np.isfinite(Q1).all() and np.isfinite(Q2).all()
    assert np.abs(Q1.T.astype(np.float64) @ Q1 - np.eye(k)).max() < 2e-5
    assert np.abs(Q2.T.astype(np.float64) @ Q2 - np.eye(k)).max() < 2e-5
    Xc, Yc = X.astype(np.float64) - X.astype(np.float64).mean(0), Y.astype(np.float64) - Y.astype(np.float64).mean(0)
    se = np.linalg.svd(Xc.T @ Yc / (n - 1), compute_uv=False)[:k]
    assert np.all(np.abs(s[:3] - se[:3]) <= 2e-5 * se[0]) and np.all(s[3:] <= 1e-4 * se[0])


def test_fewer_than_four_features_and_other_edge_inputs(ctx):
    """Round 5 (tools/edge_shape_probe.py): a field with one to three features divided by zero in the statistics pass's launch
    arithmetic (SIGFPE); a constant field came back with zero vectors where scikit-learn returns orthonormal ones; an infinity
    raised the decomposition's error where the reference's Scaler -> Sanitizer order raises the partial-NaN one."""
    from xeofs_amd import engine

    rng = np.random.default_rng(1)
    for shape, k in (((50, 1), 1), ((50, 2), 1), ((50, 3), 2), ((3, 5), 2), ((2, 40), 1)):
        X = rng.standard_normal(shape).astype(np.float32) + 1.0
        mat, st, U, s, V = engine.fit(ctx, X, k, random_state=1)
        mat.free()
        Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
        se = np.linalg.svd(Xc, compute_uv=False)[:k]
        assert np.all(np.abs(s - se) <= 1e-5 * max(se[0], 1e-30)), (shape, s, se)
        assert np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max() < 1e-5 and np.abs(V.T.astype(np.float64) @ V - np.eye(k)).max() < 1e-5
    for X in (np.zeros((50, 80), np.float32), np.full((50, 80), 3.25, np.float32)):
        mat, st, U, s, V = engine.fit(ctx, X, 2, random_state=1)
        mat.free()
        assert np.all(s == 0)
        assert np.abs(U.T.astype(np.float64) @ U - np.eye(2)).max() < 1e-5 and np.abs(V.T.astype(np.float64) @ V - np.eye(2)).max() < 1e-5
    Xi = (rng.standard_normal((120, 700)) + 2.0).astype(np.float32)
    Xi[3, 3] = np.inf
    with pytest.raises(ValueError, match="partial NaN"):          # scaler.py:128-154 then sanitizer.py:109-122
        engine.fit(ctx, Xi, 4, random_state=1)
    with pytest.raises(np.linalg.LinAlgError):                    # nothing subtracts: the infinity reaches the decomposition
        engine.fit(ctx, Xi, 4, center=False, random_state=1)


def test_three_threads_with_their_own_contexts_fit_bitwise_like_one(ctx):
    """Several contexts (each on its own stream) driven from several threads on one GPU: fused in-place fits of different fields at
    the same time equal the serial results bit for bit (process-wide pieces: the sketch generator's worker team, cached launch
    attributes, the library's statics).  (FFT-type work beside the passes is a different matter: DESIGN.md section 10.)"""
    import threading

    import torch

    from xeofs_amd import engine

    rng = np.random.default_rng(0)
    fields = [(rng.standard_normal((n, 5)) @ rng.standard_normal((5, p)) + 0.2 * rng.standard_normal((n, p)) + 1.0).astype(np.float32)
              for n, p in ((400, 4096), (700, 2048), (300, 8192), (1000, 1000))]

    def run(c, X, seed):
        mat, st, U, s, V = engine.fit(c, X, 6, random_state=seed)
        mat.free()
        return s, V

    serial = [run(ctx, X, 10 + i) for i, X in enumerate(fields)]
    bad = []

    def worker(tid):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                c = engine.Context(0)
                for rep in range(12):
                    i = (rep + tid) % len(fields)
                    s, V = run(c, fields[i], 10 + i)
                    if not (np.array_equal(s, serial[i][0]) and np.array_equal(V, serial[i][1])):
                        bad.append((tid, rep, i))
        except BaseException as e:      # noqa: BLE001  (reported by the assertion below)
            bad.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad
