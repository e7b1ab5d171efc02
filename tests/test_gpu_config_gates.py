"""The parity gates of SURVEY.md §8d at BASELINE config sizes as pytest cases (until round 4 they lived only
inside bench.py's driver line, so GPUTEST carried properties at full size but no oracle comparison at a config size).

  * config 2 at its OWN size, 5000 x (360 x 720), k = 50: the whole float64 oracle fit (what xeofs itself computes in: it
    promotes the field) against one engine call -- singular values 1e-5, |cos| >= 1 - 1e-5 and identical sign for every
    gap-separated mode, reconstruction error within 1 + 1e-4 of the oracle's;
  * config 3's path (rSVD of X^T Y through the sample-space Gram route + total squared covariance) at 3000 x two (60 x 90)
    against the materialised-C oracle;
  * config 5's path (in-place preprocess + Hilbert stage + complex decomposition, DEFAULT rule) at 2000 x (40 x 80) against
    the Hilbert restatement + exact complex SVD, with the reference's own solver (svds(lobpcg)) on the same matrix beside it.
The fields are the bench's (`bench.make_field`, generated on the GPU from fixed seeds); `bench.py` keeps the same gates."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402


def test_config2_size_fit_vs_float64_oracle(ctx):
    import bench
    from xeofs_amd import engine

    n, nlat, nlon, k = 5000, 360, 720, 50
    X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, "cuda:0")
    mat, st, U, s, V = engine.fit(ctx, X, k, random_state=5, want_stats=False, device_out=True)
    mat.free()
    U, V = U.cpu().numpy(), V.cpu().numpy()
    X64 = X.cpu().numpy().astype(np.float64)
    del X
    ref = orc.eof_fit(X64, k, random_state=5)                     # Scaler + Sanitizer + randomized SVD + sign rule + scores, float64
    so = np.asarray(ref["norms"], dtype=np.float64)
    assert np.max(np.abs(np.asarray(s, dtype=np.float64) - so) / so) <= 1e-5
    vg = bench.vector_gate(so, V, ref["components"])
    assert vg["n_gap_modes"] >= 10 and vg["min_abs_cos"] >= 1 - 1e-5 and vg["sign_ok"], vg
    Xc = X64 - X64.mean(axis=0)
    del X64
    ratio = bench.recon_err(Xc, U, s, V) / bench.recon_err(Xc, ref["U"], so, ref["components"])
    assert ratio <= 1 + 1e-4, ratio
    assert abs(st["total_variance"] - ref["total_variance"]) <= 1e-6 * ref["total_variance"]


def test_config3_path_vs_materialised_oracle(ctx):
    import bench
    from xeofs_amd import engine

    n, nlat, nlon, k = 3000, 60, 180, 20
    F = bench.make_field(n, nlat, nlon, 0, nlat * nlon, "cuda:0", seed=31_000).reshape(n, nlat, nlon)
    X = F[:, :, :90].reshape(n, -1).contiguous()
    Y = F[:, :, 90:].reshape(n, -1).contiguous()
    del F
    mx, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
    my, _ = engine.preprocess(ctx, Y, want_stats=False, in_place=True)
    r = engine.crosscov_rsvd(ctx, mx, my, k, random_state=5)
    mx.free(); my.free()
    ref = orc.mca_fit(X.cpu().numpy().astype(np.float64), Y.cpu().numpy().astype(np.float64), k, random_state=5, use_pca=False)
    so = np.asarray(ref["singular_values"], dtype=np.float64)
    assert np.max(np.abs(r["s"] - so) / so[0]) <= 1e-5
    assert abs(r["total_squared_covariance"] - ref["total_squared_covariance"]) <= 1e-5 * ref["total_squared_covariance"]
    for got, want in ((r["Q1"], ref["components1"]), (r["Q2"], ref["components2"])):
        vg = bench.vector_gate(so, got, want)
        assert vg["min_abs_cos"] >= 1 - 1e-5 and vg["sign_ok"], vg


def test_config5_path_default_rule_vs_exact_and_reference_solver(ctx):
    import bench
    from xeofs_amd import engine

    n, nlat, nlon, k = 2000, 40, 80, 20
    X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, "cuda:0", seed=51_000)
    A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
    B, _ = engine.hilbert(ctx, A, "exp", 0.2)
    U, s, V = engine.rsvd_c64(ctx, A, B, k, random_state=5)          # the fixed count (n_iter="auto": scikit-learn's 7 products)
    assert engine.last_iterations(ctx) == 7
    # the rule bench.py times and the models default to since round 6: to convergence, on the operator route HilbertEOF takes
    Uc, sc, Vc = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter="converge")
    its_c = engine.last_iterations(ctx)
    A.free(); B.free()
    x64 = X.cpu().numpy().astype(np.float64)
    z = orc.hilbert_transform(x64 - x64.mean(axis=0), padding="exp", decay_factor=0.2)
    _, sz, vhz = np.linalg.svd(z, full_matrices=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, s_ref, _ = orc.complex_svds(z, k, random_state=5)            # xeofs/linalg/decomposer.py:149-160
    e = np.abs(np.asarray(s, dtype=np.float64) - sz[:k]) / sz[:k]
    e_ref = np.abs(s_ref - sz[:k]) / sz[:k]
    assert np.all(e <= np.maximum(1e-5, e_ref)), (e, e_ref)
    vg = bench.vector_gate(sz[:k], np.asarray(V), vhz[:k].conj().T, complex_phase=True)
    assert vg["min_abs_cos"] >= 1 - 1e-5, vg
    ec = np.abs(np.asarray(sc, dtype=np.float64) - sz[:k]) / sz[:k]
    assert np.all(ec <= np.maximum(1e-5, e_ref)) and ec.max() <= 2e-6 and 4 <= its_c <= 20, (ec, its_c)
    vgc = bench.vector_gate(sz[:k], np.asarray(Vc), vhz[:k].conj().T, complex_phase=True)
    assert vgc["min_abs_cos"] >= 1 - 1e-5, vgc
