"""The sample-space Gram matrix on the fp16 matrix cores (xeofs_amd/csrc/eofx_gram.hpp: planes_split_kernel ->
gram_nt_kernel -> gram_finish_kernel) through the C ABI (eofx_mat_gram_f32) against float64 numpy on the oracle's
preprocessed matrix: every layout (in place, masked in place, written layouts), shapes that leave partial tiles /
stages / splits, and its two consumers -- the total squared covariance and the sample-space power iterations of
eofx_crosscov_rsvd_f32 (xeofs/cross/cpcca.py:186-221, 991-1000)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


@pytest.mark.parametrize("n,nlat,nlon,layout,std", [
    (300, 20, 37, "inplace", False),       # P % 4 != 0: the scalar path of the split kernel
    (300, 24, 40, "inplace", True),
    (300, 24, 40, "masked", False),
    (700, 30, 50, "copy", True),           # n_pad = 1024: 4 x 4 tiles, several splits
    (1100, 16, 90, "inplace", False),      # n_pad = 1536
])
def test_sample_gram_vs_float64(ctx, n, nlat, nlon, layout, std):
    from xeofs_amd import engine

    X, lat = orc.synthetic_field(n, nlat, nlon, rank=10, seed=n)
    w = np.repeat(orc.sqrt_cos_lat_weights(lat), nlon) if std else None
    if layout == "masked":
        rng = np.random.default_rng(0)
        X[:, rng.choice(X.shape[1], X.shape[1] // 5, replace=False)] = np.nan
    mat, st = engine.preprocess(ctx, X, True, std, w, in_place=layout != "copy", allow_masked=layout == "masked")
    assert mat.masked == (layout == "masked")
    ref = orc.preprocess(X.astype(np.float64), True, std, w)["X"]
    Gref = ref @ ref.T
    G = mat.gram(0)[:n, :n].double().cpu().numpy()
    Gpad = mat.gram(0).cpu().numpy()
    assert not Gpad[n:].any() and not Gpad[:, n:].any()            # padding rows / columns are exact zeros
    assert np.array_equal(Gpad, Gpad.T)                               # mirrored, not recomputed
    scale = np.sqrt(np.outer(np.diag(Gref), np.diag(Gref)))
    assert np.max(np.abs(G - Gref) / scale) <= 3e-6, np.max(np.abs(G - Gref) / scale)
    # bitwise reproducible (fixed split order, no atomics)
    assert np.array_equal(mat.gram(0).cpu().numpy(), Gpad)
    mat.free()


def test_crosscov_gram_route_equals_matrix_free(ctx, monkeypatch):
    """The same call with and without the Gram route (EOFX_CROSS_NO_GRAM=1 keeps the power iterations on the fields):
    singular values / vectors / scores / TSC agree to float32 rounding, and both match the float64 oracle."""
    from xeofs_amd import engine

    n, k = 400, 6
    F, _ = orc.synthetic_field(n, 30, 80, rank=15, seed=5)
    F = F.reshape(n, 30, 80)
    X = np.ascontiguousarray(F[:, :, :44].reshape(n, -1))
    Y = np.ascontiguousarray(F[:, :, 44:].reshape(n, -1))
    outs = []
    for no_gram in ("", "1"):
        if no_gram:
            monkeypatch.setenv("EOFX_CROSS_NO_GRAM", "1")
        mx, _ = engine.preprocess(ctx, X, in_place=True)
        my, _ = engine.preprocess(ctx, Y, in_place=True)
        outs.append(engine.crosscov_rsvd(ctx, mx, my, k, random_state=7))
        mx.free(); my.free()
    a, b = outs
    ref = orc.mca_fit(X.astype(np.float64), Y.astype(np.float64), k, random_state=7, use_pca=False)
    for o in (a, b):
        assert np.all(np.abs(o["s"] - ref["singular_values"]) <= 1e-5 * ref["singular_values"][0])
        assert abs(o["total_squared_covariance"] - ref["total_squared_covariance"]) <= 1e-5 * ref["total_squared_covariance"]
        for j in range(k):
            for key, rk in (("Q1", "components1"), ("Q2", "components2")):
                assert abs(np.dot(o[key][:, j].astype(np.float64), ref[rk][:, j])) >= 1 - 1e-5
    assert np.allclose(a["s"], b["s"], rtol=2e-6)
    assert np.allclose(a["scores1"], b["scores1"], atol=2e-5 * np.abs(b["scores1"]).max())
    assert np.isclose(a["total_squared_covariance"], b["total_squared_covariance"], rtol=1e-6)


@pytest.mark.parametrize("n,P,L,layout", [(700, 3000, 512, "inplace"), (1100, 2500, 672, "copy"), (600, 2050, 1024, "masked")])
def test_wide_feature_side_product_through_the_nt_kernel(ctx, n, P, L, layout, monkeypatch):
    """eofx_panel_tmul_f32 with 512 columns and more goes through transposed fp16 planes + gram_nt_kernel (the PCA
    pre-reduction's wide panel): against float64 numpy on the oracle's preprocessed matrix, and against the streaming
    kernel's result for the same call (EOFX_NO_TMUL_NT=1)."""
    import torch
    from xeofs_amd import engine

    X, _ = orc.synthetic_field(n, 1, P, rank=10, seed=P)
    if layout == "masked":
        X[:, np.random.default_rng(1).choice(P, P // 6, replace=False)] = np.nan
    mat, st = engine.preprocess(ctx, X, True, True, None, in_place=layout != "copy", allow_masked=layout == "masked")
    ref = orc.preprocess(X.astype(np.float64), True, True, None)["X"]
    rng = np.random.default_rng(2)
    Z = np.zeros((mat.n_pad, L), np.float32)
    Z[:n] = rng.standard_normal((n, L)) * (1.0 + rng.random(L))
    Zd = torch.as_tensor(Z, device="cuda")
    Y = engine.panel_tmul(ctx, mat, Zd, prec="f16x3")
    Yh = mat.compact_rows(Y[:mat.p_phys].cpu().numpy()).astype(np.float64)
    want = ref.T @ Z[:n].astype(np.float64)
    scale = np.sqrt((ref ** 2).sum(0))[:, None] * np.sqrt((Z[:n].astype(np.float64) ** 2).sum(0))[None, :]
    assert np.max(np.abs(Yh - want) / scale) <= 3e-6
    assert not Y[mat.p_phys:].any()
    monkeypatch.setenv("EOFX_NO_TMUL_NT", "1")
    Y2 = engine.panel_tmul(ctx, mat, Zd, prec="f16x3")
    assert np.max(np.abs((Y - Y2).cpu().numpy()[:mat.p_phys]) / np.maximum(scale.max(), 1e-30)) <= 3e-6
    mat.free()


@pytest.mark.parametrize("rows,L,Lo", [(2048, 768, 512), (1280, 1536, 1504), (1024, 512, 260)])
def test_wide_panel_matmul_through_the_nt_kernel(ctx, rows, L, Lo, monkeypatch):
    """eofx_panel_matmul_f32 with a wide panel and a wide matrix (the PCA pre-reduction's V = B W) runs on the fp16 matrix
    cores (fp16 {hi, lo} planes of both operands, gram_nt_kernel): against float64, and against the float64 VALU kernel it
    replaces there (EOFX_NO_MATMUL_NT=1)."""
    import torch
    from xeofs_amd import engine

    rng = np.random.default_rng(rows + L)
    P = torch.as_tensor((rng.standard_normal((rows, L)) * np.exp(rng.uniform(-3, 3, size=L))).astype(np.float32), device="cuda")
    M = torch.as_tensor(rng.standard_normal((L, Lo)) / np.sqrt(L), device="cuda")
    ref = (P.double() @ M).cpu().numpy()
    out = engine.panel_matmul(ctx, P, M).cpu().numpy()
    scale = np.abs(ref).max()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 3e-6 * scale                      # 22-bit operands, float32 accumulation over L terms
    monkeypatch.setenv("EOFX_NO_MATMUL_NT", "1")
    old = engine.panel_matmul(ctx, P, M).cpu().numpy()
    assert np.abs(old - ref).max() <= 1e-6 * scale
    assert np.abs(out - old).max() <= 3e-6 * scale


@pytest.mark.parametrize("n,P,L", [(700, 3000, 512), (2000, 4096, 608)])
def test_wide_feature_side_product_on_a_fresh_context(n, P, L):
    """the NT route's scratch estimate left out the plan's split-K factor (700 x 3000 with 512 columns
    plans S = 3, 2000 x 4096 with 608 plans S = 4), so a context whose grow-only arena had not been enlarged by earlier calls
    ran out of it.  A FRESH context per case: the first engine call it sees is this product."""
    import torch
    from xeofs_amd import engine

    ctx2 = engine.Context(0)
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, P)).astype(np.float32)
    mat = engine.from_dense(ctx2, X)
    Z = np.zeros((mat.n_pad, L), np.float32)
    Z[:n] = rng.standard_normal((n, L))
    Y = engine.panel_tmul(ctx2, mat, torch.as_tensor(Z, device="cuda"), prec="f16x3")
    want = X.astype(np.float64).T @ Z[:n].astype(np.float64)
    assert np.abs(Y[:P].cpu().numpy() - want).max() <= 3e-6 * np.sqrt(n) * 4.0 * 4.0
    mat.free()
    del ctx2
