"""GPU parity of the bootstrap path: `eofx_resample_f32` and
`xeofs_amd.validation.EOFBootstrapper` against the oracle restatement of xeofs/validation/bootstrapper.py."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


@pytest.mark.parametrize("n,p", [(70, 300), (600, 1100), (513, 64)])
def test_resample_matches_numpy(ctx, n, p):
    from xeofs_amd import engine

    rng = np.random.default_rng(0)
    X = (rng.standard_normal((n, p)) + np.linspace(-2, 2, p)).astype(np.float32)
    mat = engine.from_dense(ctx, X)
    idx = rng.choice(n, n, replace=True)
    for center in (True, False):
        bm, mean, tv = engine.resample(ctx, mat, idx, center=center)
        Xb = X[idx].astype(np.float64)
        ref = Xb - Xb.mean(0) if center else Xb
        got = bm.download()
        assert got.shape == (n, p)
        assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max()
        assert np.allclose(mean, Xb.mean(0), atol=1e-6)
        assert np.isclose(tv, Xb.var(axis=0, ddof=1).sum(), rtol=1e-6)
        bm.free()
    with pytest.raises(ValueError):
        engine.resample(ctx, mat, np.array([0, n]))
    mat.free()


def test_eof_bootstrapper_vs_oracle(ctx):
    import xeofs_amd as xe

    vals = orc.synthetic_field(120, 8, 10, rank=6, seed=3)[0].reshape(120, 8, 10)
    X = xe.DataArray(vals, dims=("time", "lat", "lon"))
    model = xe.single.EOF(n_modes=3, random_state=1).fit(X, "time")
    had_layout = model.data["input_data"].has_sample_layout()
    bs = xe.validation.EOFBootstrapper(n_bootstraps=5, seed=11).fit(model, random_state=0)
    Xs = vals.reshape(120, -1).astype(np.float64)
    eof = orc.eof_fit(Xs, 3, random_state=1)
    eof["input_data"] = Xs - Xs.mean(0)
    ref = orc.eof_bootstrap(eof, 3, n_bootstraps=5, seed=11, random_state=0)
    assert bs.get_params() == {"n_bootstraps": 5, "seed": 11}
    ev = bs.explained_variance()
    assert ev.dims == ("n", "mode") and ev.shape == (5, 3) and list(ev.coords["n"]) == [1, 2, 3, 4, 5]
    assert np.allclose(ev.values, ref["explained_variance"], rtol=2e-4)
    assert np.allclose(bs.total_variance().values, ref["total_variance"], rtol=1e-5)
    comps = bs.components()
    assert comps.dims == ("n", "mode", "lat", "lon") and comps.shape == (5, 3, 8, 10)
    C = comps.values.reshape(5, 3, -1).transpose(0, 2, 1)
    for b in range(5):
        for j in range(3):
            assert np.dot(C[b, :, j], ref["components"][b, :, j]) > 1 - 1e-4, (b, j)   # incl. the aligned sign
    sc = bs.scores()
    assert sc.dims == ("n", "mode", "time") and sc.shape == (5, 3, 120)
    S = sc.values.transpose(0, 2, 1)
    assert np.allclose(S, ref["scores"], atol=2e-3 * np.abs(ref["scores"]).max())
    # seed determinism of the member selection
    bs2 = xe.validation.EOFBootstrapper(n_bootstraps=5, seed=11).fit(model, random_state=0)
    assert np.array_equal(bs2.data["scores"], bs.data["scores"])
    assert (bs.explained_variance_ratio().values <= 1).all()
    # the sample-contiguous copy the members run over is released again: the model's matrix keeps its footprint
    assert model.data["input_data"].has_sample_layout() == had_layout


def test_bootstrap_member_operator(ctx):
    """`BootstrapOps`: the member X_b = H X through panel products on the ORIGINAL matrix -- against the resampled,
    re-centred matrix itself; H / H^T are the engine's gather / segment-sum kernels (`eofx_panel_bootstrap_f32`, no
    library GEMM), on an in-place matrix (no layout is built), plus the member's total variance from the row norms."""
    import torch
    from xeofs_amd import engine
    from xeofs_amd.validation.bootstrapper import BootstrapOps

    n, p = 333, 1100
    rng = np.random.default_rng(4)
    X = (rng.standard_normal((n, p)) * rng.uniform(0.5, 3, p) + rng.standard_normal(p)).astype(np.float32)
    mat, _ = engine.preprocess(ctx, torch.as_tensor(X, device="cuda"), want_stats=False, keep_raw=True, in_place=True)
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    idx = rng.integers(0, n, n)
    Xb = Xc[idx] - Xc[idx].mean(0)
    Z = torch.zeros((mat.n_pad, 32), device="cuda"); Z[:n] = torch.randn((n, 32), device="cuda")
    Y = torch.zeros((mat.p_pad, 32), device="cuda"); Y[:p] = torch.randn((p, 32), device="cuda")
    outs = []
    for rep in range(2):
        ops = BootstrapOps(ctx, mat, idx)
        # H and H^T on their own against dense float64 algebra
        Hd = -np.bincount(idx, minlength=n)[None, :].repeat(n, 0) / n
        Hd[np.arange(n), idx] += 1.0
        Zh = Z[:n].double().cpu().numpy()
        assert np.abs(ops._h(Z)[:n].double().cpu().numpy() - Hd @ Zh).max() <= 1e-6 * np.abs(Zh).max()
        assert np.abs(ops._ht(Z)[:n].double().cpu().numpy() - Hd.T @ Zh).max() <= 1e-6 * np.abs(Zh).max()
        assert not bool(ops._h(Z)[n:].any()) and not bool(ops._ht(Z)[n:].any())      # padding rows stay zero
        t = ops.tmul(Z)[:p].double().cpu().numpy()
        m = ops.mul(Y)[:n].double().cpu().numpy()
        want_t = Xb.T @ Z[:n].double().cpu().numpy()
        want_m = Xb @ Y[:p].double().cpu().numpy()
        assert np.abs(t - want_t).max() <= 2e-5 * np.abs(want_t).max()
        assert np.abs(m - want_m).max() <= 2e-5 * np.abs(want_m).max()
        c = ops.counts.cpu().numpy()
        tv = (c @ engine.sample_norms(ctx, mat) ** 2 - n * ops.mean_sumsq()) / (n - 1)
        assert np.isclose(tv, (Xb ** 2).sum() / (n - 1), rtol=1e-5)
        outs.append((t, m))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])     # bitwise reproducible
    assert mat.layout()[0] in (False, 0)          # still in place: no layout was materialised
