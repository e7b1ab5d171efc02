"""The analytic-signal decomposition WITHOUT the imaginary part in memory (eofx_rsvd_hilbert_c64, eofx_hilbert_sumsq_f64).

The reference's HilbertEOF (single/eof.py:433-447) writes the complex field and hands it to svds(lobpcg)
(decomposer.py:149-160).  The Hilbert stage is linear along the samples, Im = Hc A, so the engine keeps the real field only
and applies the n x n operator to the sample-side panels.  These tests pin that route to the two-part route
(eofx_hilbert_f32 + eofx_rsvd_c64), to the oracle's analytic signal and to exact float64 SVDs.
"""
import numpy as np
import pytest

from oracle import eof_oracle as orc

pytestmark = pytest.mark.gpu


def _waves(n, p, seed=0, noise=0.3, nsig=5):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    x = np.arange(p)[None, :]
    X = np.zeros((n, p))
    for j in range(nsig):
        X += (nsig - j) * np.sin(2 * np.pi * ((1 + j) * t / n * 2.3 - (1 + j) * x / p) + rng.uniform(0, 6.28))
    X += noise * rng.standard_normal((n, p))
    X += 0.002 * t           # a trend: the padded transform removes and restores a linear fit
    return X.astype(np.float32)


def _check_modes(s0, U0, V0, s1, U1, V1, tol_s=2e-6, tol_v=1e-5):
    k = len(s0)
    assert np.all(np.abs(s1 - s0) <= tol_s * s0[0]), np.abs(s1 - s0).max() / s0[0]
    for j in range(k):
        gap = min(s0[j - 1] - s0[j] if j else np.inf, s0[j] - (s0[j + 1] if j + 1 < k else 0.0))
        if gap > 1e-3 * s0[0]:
            assert abs(np.vdot(V0[:, j], V1[:, j])) >= 1 - tol_v, j       # (a complex mode is defined up to a unit phase)
            assert abs(np.vdot(U0[:, j], U1[:, j])) >= 1 - tol_v, j


@pytest.mark.parametrize("padding", ["exp", None])
@pytest.mark.parametrize("n,p,k,in_place", [(300, 1536, 6, True), (257, 900, 5, False), (1000, 700, 8, False), (640, 2048, 40, True),
                                            (64, 3000, 4, True), (2100, 1300, 6, False)])
def test_operator_route_equals_the_two_part_route(ctx, n, p, k, in_place, padding):
    """same singular triplets as eofx_hilbert_f32 + eofx_rsvd_c64 (to the rounding of the float32 stage), same sign rule;
    and the singular values of the oracle's analytic signal in float64"""
    from xeofs_amd import engine

    X = _waves(n, p, seed=n + p)
    A0, _ = engine.preprocess(ctx, X, True, False, None)
    B0, _ = engine.hilbert(ctx, A0, padding, 0.2)
    U0, s0, V0 = engine.rsvd_c64(ctx, A0, B0, k, random_state=1)
    A1, _ = engine.preprocess(ctx, X, True, False, None, in_place=in_place)
    U1, s1, V1 = engine.rsvd_hilbert_c64(ctx, A1, k, padding, 0.2, random_state=1)
    if in_place:
        assert A1.layout() == (False, True)           # nothing was written: the field is streamed where it lies
    _check_modes(s0, U0, V0, s1, U1, V1)
    assert (orc.deterministic_sign_multiplier(V1.conj().T) == 1).all()
    # oracle: the analytic signal of the centred field in float64
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    Z = orc.hilbert_transform(Xc, padding=padding, decay_factor=0.2)
    se = np.linalg.svd(Z, compute_uv=False)[:k]
    assert np.all(np.abs(s1 - se) <= 2e-5 * se[0]), np.abs(s1 - se).max() / se[0]
    # total variance of the imaginary part without writing it
    sq = engine.hilbert_sumsq(ctx, A1, padding, 0.2)
    assert abs(sq - (Z.imag ** 2).sum()) <= 2e-6 * (Z.imag ** 2).sum()
    b0 = B0.download().astype(np.float64)
    assert abs(sq - (b0 ** 2).sum()) <= 1e-6 * (b0 ** 2).sum()
    for m in (A0, B0, A1):
        m.free()


def test_operator_route_default_rule_and_converge_on_a_decaying_spectrum(ctx):
    """the iteration rules go through the same block Krylov driver: default count and "converge" against the exact values"""
    from xeofs_amd import engine

    n, p, k = 700, 2600, 16
    rng = np.random.default_rng(4)
    t = np.arange(n)[:, None] / n
    X = np.zeros((n, p))
    for j in range(30):
        X += 0.86 ** j * 40.0 * np.sin(2 * np.pi * ((2 + j) * t * 3.1 - (1 + j) * np.arange(p)[None, :] / p) + rng.uniform(0, 6.28))
    X += 1.5 * rng.standard_normal((n, p))
    X = X.astype(np.float32)
    A, _ = engine.preprocess(ctx, X, True, False, None, in_place=True)
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    Z = orc.hilbert_transform(Xc, padding="exp", decay_factor=0.2)
    se = np.linalg.svd(Z, compute_uv=False)[:k]
    for rule, tol in (("auto", 1e-5), ("converge", 1e-5)):
        U, s, V = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, n_iter=rule, random_state=7)
        assert np.all(np.abs(s - se) <= tol * se), (rule, (np.abs(s - se) / se).max())
        # the factors reproduce the field's leading part: Z V = U diag(s)
        R = Z @ V.astype(np.complex128) - U.astype(np.complex128) * s
        assert np.linalg.norm(R, axis=0).max() <= 3e-4 * se[0], rule
    A.free()


@pytest.mark.parametrize("for_hilbert", [False, True])
def test_operator_route_on_a_masked_field_in_place(ctx, for_hilbert):
    """a land-masked field kept in place (zero columns): factors over the valid features, equal to the compaction route"""
    from xeofs_amd import engine

    n, P, k = 300, 1600, 6
    rng = np.random.default_rng(P)
    X = _waves(n, P, seed=n, noise=0.2) + 11.0
    dead = rng.choice(P, size=int(0.3 * P), replace=False)
    X[:, dead] = np.nan
    w = rng.uniform(0.4, 1.3, size=P)
    A0, _ = engine.preprocess(ctx, X, True, False, w)
    B0, _ = engine.hilbert(ctx, A0, "exp", 0.2)
    U0, s0, V0 = engine.rsvd_c64(ctx, A0, B0, k, random_state=3)
    A1, _ = engine.preprocess(ctx, X, True, False, w, in_place=True, allow_masked=True, for_hilbert=for_hilbert)
    assert A1.masked
    sq = engine.hilbert_sumsq(ctx, A1, "exp", 0.2)           # (consumes the transposed raw layout when there is one)
    U1, s1, V1 = engine.rsvd_hilbert_c64(ctx, A1, k, "exp", 0.2, random_state=3)
    assert A1.layout() == (False, True)                      # still in place, nothing written
    assert V1.shape == V0.shape == (A0.p, k)
    _check_modes(s0, U0, V0, s1, U1, V1)
    b0 = B0.download().astype(np.float64)
    assert abs(sq - (b0 ** 2).sum()) <= 1e-6 * (b0 ** 2).sum()
    for m in (A0, B0, A1):
        m.free()


def test_operator_route_argument_errors(ctx):
    from xeofs_amd import engine
    X = _waves(100, 300, seed=1)
    A, _ = engine.preprocess(ctx, X, True, False, None)
    with pytest.raises(ValueError, match="rank"):
        engine.rsvd_hilbert_c64(ctx, A, 101)
    with pytest.raises(ValueError, match="decay_factor"):
        engine.rsvd_hilbert_c64(ctx, A, 4, "exp", -1.0)
    with pytest.raises(ValueError, match="sketch width"):
        engine.rsvd_hilbert_c64(ctx, A, 60, n_oversamples=10)          # sketch wider than 64
    A.free()


def test_operator_cache_is_bounded(ctx):
    """the resident n x n operators are a two-entry cache: a third series length evicts the oldest, which is rebuilt (same
    bits) when it comes back"""
    from xeofs_amd import engine

    out = {}
    for rnd in range(2):
        for n in (120, 200, 260):
            X = _waves(n, 900, seed=n)
            A, _ = engine.preprocess(ctx, X, True, False, None)
            _, s, _ = engine.rsvd_hilbert_c64(ctx, A, 4, "exp", 0.2, random_state=2)
            A.free()
            if rnd == 0:
                out[n] = s
            else:
                assert np.array_equal(out[n], s), n
