"""The engine's own feature-sharded fit (include/eofx.h: eofx_fit_sharded_f32 -- SURVEY.md 8e): every collective of a fit is
issued by the engine on its own stream (RCCL, or a host callback in tests).  One GPU here, so: the RCCL binding at world
size 1 (ncclCommInitRank / ncclAllReduce really run) and the callback binding must both reproduce eofx_fit_f32 BIT FOR BIT;
two ranks that share the GPU run through bench.py (tests/test_gpu_fullsize.py::test_two_rank_sharded_path_on_one_gpu)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eof_oracle as orc  # noqa: E402  (checker only)


def _field(n=600, nlat=40, nlon=64, seed=3):
    X, _ = orc.synthetic_field(n, nlat, nlon, rank=12, seed=seed)
    return X


@pytest.mark.parametrize("binding", ["rccl", "callback"])
def test_world1_equals_single_gpu_fit_bitwise(ctx, binding):
    from xeofs_amd import engine

    X = _field()
    n, P = X.shape
    k = 8
    calls = []
    if binding == "rccl":
        engine.comm_init_rccl(ctx, engine.comm_unique_id(), 1, 0)
    else:
        engine.comm_set_callback(ctx, lambda buf, count, dtype, op, stream: calls.append((count, dtype, op)) or 0, 1, 0)
    try:
        res = engine.fit_sharded(ctx, X, k, P, random_state=4)
        assert res is not None
        mat, st, U, s, V = res
        stats = engine.comm_stats(ctx)
    finally:
        engine.comm_clear(ctx)
    mat2, st2, U2, s2, V2 = engine.fit(ctx, X, k, random_state=4)
    assert st2["fused"]
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    assert st["total_variance"] == st2["total_variance"]
    assert np.array_equal(st["mean"], st2["mean"])
    # what a fit sends: 2 votes, n_iter + 1 sample-side panels, (1 + 2) Gram matrices of feature-side panels (the first
    # iteration's and the two of CholeskyQR2 at the end; more on peaked spectra), the sign rule's extrema, the total variance
    n_iter = 7
    assert stats["calls"] >= 2 + (n_iter + 1) + 3 + 1 + 1
    assert stats["bytes"] >= (n_iter + 1) * mat.n_pad * 32 * 4
    if binding == "callback":
        assert len(calls) == stats["calls"]
        assert sum(1 for c in calls if c[1] == 0 and c[0] == mat.n_pad * 32) == n_iter + 1     # the n x L float32 panels
    ref = orc.eof_fit(X.astype(np.float64), k, random_state=4)
    assert np.all(np.abs(s - ref["norms"]) <= 1e-5 * ref["norms"][0])
    mat.free(); mat2.free()


def test_vote_sends_every_rank_to_the_fallback(ctx):
    """A NaN in the slice: the fused first pass is not available, the entry returns None (after the vote) and builds nothing."""
    from xeofs_amd import engine

    X = _field(seed=5)
    X[10, 100] = np.nan
    engine.comm_init_rccl(ctx, engine.comm_unique_id(), 1, 0)
    try:
        assert engine.fit_sharded(ctx, X, 6, X.shape[1], random_state=1) is None
    finally:
        engine.comm_clear(ctx)


def test_sharded_entry_needs_a_communicator(ctx):
    from xeofs_amd import engine
    from xeofs_amd._lib import EofxError

    with pytest.raises((EofxError, ValueError)):
        engine.fit_sharded(ctx, _field(), 4, 40 * 64, random_state=0)


def test_comm_selftest(ctx):
    """eofx_ctx_comm_selftest: every collective the sharded fit uses, on known values.  World 1 over RCCL passes; a callback
    that reduces nothing passes at world 1 too (identity), one that claims two ranks without reducing is caught."""
    from xeofs_amd import engine
    from xeofs_amd._lib import EofxError

    with pytest.raises((EofxError, ValueError)):
        engine.comm_selftest(ctx)                    # nothing attached
    engine.comm_init_rccl(ctx, engine.comm_unique_id(), 1, 0)
    try:
        assert engine.comm_selftest(ctx)
        assert engine.comm_stats(ctx)["calls"] == 5
    finally:
        engine.comm_clear(ctx)
    engine.comm_set_callback(ctx, lambda buf, count, dtype, op, stream: 0, 2, 0)    # "two ranks", no reduction
    try:
        assert not engine.comm_selftest(ctx)
    finally:
        engine.comm_clear(ctx)


@pytest.mark.parametrize("binding", ["rccl", "callback"])
def test_comm_probe_reports_ranks_and_latencies(ctx, binding):
    """eofx_ctx_comm_probe (what `bench.py --gpus N` prints as `comm.ranks_seen` before anything is timed): the rank count the
    attached communicator really reduces over and a latency per collective of a fit -- world size 1 here, through RCCL itself
    and through the host callback."""
    from xeofs_amd import engine

    if binding == "rccl":
        engine.comm_init_rccl(ctx, engine.comm_unique_id(), 1, 0)
    else:
        engine.comm_set_callback(ctx, lambda buf, count, dtype, op, stream: 0, 1, 0)
    try:
        seen, us = engine.comm_probe(ctx, [(10240 * 64, "f32"), (64 * 64, "f64"), (1, "i32")], reps=5)
    finally:
        engine.comm_clear(ctx)
    assert seen == 1.0
    assert len(us) == 3 and all(u >= 0.0 and np.isfinite(u) for u in us)
    with pytest.raises(Exception):
        engine.comm_probe(ctx, [(1, "f32")])          # no communicator attached any more


# ---- round 6: the engine-owned sharded forms of configs 3 and 5, masked fields and null modes on the sharded EOF entry --------
def _rccl1(ctx):
    from xeofs_amd import engine

    engine.comm_init_rccl(ctx, engine.comm_unique_id(), 1, 0)


def test_world1_crosscov_sharded_equals_single_gpu_bitwise(ctx):
    """eofx_crosscov_rsvd_sharded_f32 over a one-rank RCCL communicator (ncclAllReduce really runs on every sample-side panel,
    Gram matrix and on the two n x n sample-space Gram matrices) reproduces eofx_crosscov_rsvd_f32 bit for bit."""
    from xeofs_amd import engine

    rng = np.random.default_rng(2)
    n, p1, p2, k = 500, 3000, 2200, 6
    t = rng.standard_normal((n, 8)) * (5.0 * 0.8 ** np.arange(8))
    X = (t @ rng.standard_normal((8, p1)) + rng.standard_normal((n, p1))).astype(np.float32)
    Y = (t @ rng.standard_normal((8, p2)) + rng.standard_normal((n, p2))).astype(np.float32)
    mx, _ = engine.preprocess(ctx, X, in_place=True)
    my, _ = engine.preprocess(ctx, Y, in_place=True)
    ref = engine.crosscov_rsvd(ctx, mx, my, k, random_state=7)
    _rccl1(ctx)
    try:
        engine.comm_stats(ctx)
        got = engine.crosscov_rsvd_sharded(ctx, mx, my, k, p1, 0, p2, 0, random_state=7)
        stats = engine.comm_stats(ctx)
    finally:
        engine.comm_clear(ctx)
    for key in ("s", "Q1", "Q2", "scores1", "scores2", "norm1", "norm2"):
        assert np.array_equal(got[key], ref[key]), key
    assert got["total_squared_covariance"] == ref["total_squared_covariance"]
    # the Gram-route vote, 2 sample-space Gram matrices, the sketch product, the range basis (2 Grams), the projection
    # (panel + Gram), the sign rule, two score panels
    assert stats["calls"] >= 10 and stats["bytes"] >= 2 * mx.n_pad * mx.n_pad * 4
    mx.free(); my.free()


@pytest.mark.parametrize("route", ["operator", "two_part"])
def test_world1_complex_sharded_equals_single_gpu_bitwise(ctx, route):
    """eofx_rsvd_hilbert_sharded_c64 / eofx_rsvd_sharded_c64 over a one-rank RCCL communicator = the single-GPU entries, bit
    for bit (block-Krylov recurrence on the replicated sample side; all-reduce after every Z Y and every feature-side Gram)."""
    from xeofs_amd import engine

    rng = np.random.default_rng(4)
    n, p, k = 400, 2600, 6
    tt, xx = np.arange(n)[:, None], np.linspace(0, 2 * np.pi, p)[None, :]
    X = sum(a * np.cos(w * tt - m * xx + ph) for a, w, m, ph in ((3.0, 0.21, 2, 0.0), (1.7, 0.37, -3, 0.4), (0.9, 0.11, 1, 1.0)))
    X = (X + 0.3 * rng.standard_normal((n, p))).astype(np.float32)
    A, _ = engine.preprocess(ctx, X, in_place=True, for_hilbert=(route == "operator"))
    if route == "operator":
        engine.hilbert_sumsq(ctx, A, "exp", 0.2)         # (consumes the transposed raw copy, as the model does)
        ref = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=3)
    else:
        B, _ = engine.hilbert(ctx, A, "exp", 0.2)
        ref = engine.rsvd_c64(ctx, A, B, k, random_state=3)
    _rccl1(ctx)
    try:
        engine.comm_stats(ctx)
        if route == "operator":
            got = engine.rsvd_hilbert_sharded_c64(ctx, A, k, p, "exp", 0.2, random_state=3)
        else:
            got = engine.rsvd_sharded_c64(ctx, A, B, k, p, random_state=3)
        stats = engine.comm_stats(ctx)
    finally:
        engine.comm_clear(ctx)
    for g, r, name in zip(got, ref, "UsV"):
        assert np.array_equal(g, r), name
    assert stats["calls"] >= 8 + 3 + 2          # 8 sample-side panels, the feature-side Gram matrices, the sign rule
    A.free()
    if route != "operator":
        B.free()


def test_world1_masked_field_stays_on_the_sharded_entry(ctx):
    """A land / sea mask (all-NaN grid points, sanitizer.py:80-126) no longer sends the ranks to the panel-level fallback: the
    slice keeps the masked features as zero columns, the entry returns the valid rows -- bit for bit eofx_fit_f32 in layout 3."""
    from xeofs_amd import engine

    X = _field(seed=6)
    rng = np.random.default_rng(1)
    X[:, rng.random(X.shape[1]) < 0.25] = np.nan
    k = 7
    _rccl1(ctx)
    try:
        res = engine.fit_sharded(ctx, X, k, X.shape[1], random_state=4, allow_masked=True)
    finally:
        engine.comm_clear(ctx)
    assert res is not None
    mat, st, U, s, V = res
    assert mat.masked and mat.p == int(st["valid_feature"].sum()) < X.shape[1] and V.shape == (mat.p, k)
    mat2, st2, U2, s2, V2 = engine.fit(ctx, X, k, random_state=4, allow_masked=True)
    assert st2["fused"] and mat2.masked
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    assert st["total_variance"] == st2["total_variance"] and np.array_equal(st["valid_feature"], st2["valid_feature"])
    ref = orc.eof_fit(X.astype(np.float64), k, random_state=4)
    assert np.all(np.abs(s - ref["norms"]) <= 1e-5 * ref["norms"][0])
    mat.free(); mat2.free()


def test_world1_more_modes_than_rank_on_the_sharded_entry(ctx):
    """k = 9 modes of an exactly rank-4 field: both factors of the sharded entry are orthonormal (scikit-learn's QR + dense SVD
    return orthonormal factors whatever the values), as eofx_fit_f32's -- the feature-side factor through its all-reduced Gram."""
    from xeofs_amd import engine

    rng = np.random.default_rng(8)
    n, p, r, k = 300, 2048, 4, 9
    X = ((rng.standard_normal((n, r)) * [5, 3, 2, 1]) @ rng.standard_normal((r, p))).astype(np.float32)
    _rccl1(ctx)
    try:
        res = engine.fit_sharded(ctx, X, k, p, random_state=2)
    finally:
        engine.comm_clear(ctx)
    assert res is not None
    mat, st, U, s, V = res
    assert np.all(s[r:] <= 1e-5 * s[0])
    U64, V64 = U.astype(np.float64), V.astype(np.float64)
    assert np.abs(U64.T @ U64 - np.eye(k)).max() <= 1e-5 and np.abs(V64.T @ V64 - np.eye(k)).max() <= 1e-5
    mat2, st2, U2, s2, V2 = engine.fit(ctx, X, k, random_state=2)
    assert np.array_equal(s, s2) and np.array_equal(U, U2) and np.array_equal(V, V2)
    mat.free(); mat2.free()


@pytest.mark.parametrize("variant", ["plain", "mask", "lowrank"])
def test_two_rank_engine_owned_entries_on_one_gpu(ctx, variant):
    """tools/sharded_native_worker.py: two processes share cuda:0, each holds half of every field's space axis, the engine's
    communicator is the host-callback binding over gloo.  All four engine-owned sharded fits (EOF -- with a land / sea mask in
    place --, MCA, HilbertEOF on the operator route and on the two-part route) against the single-GPU entries on the whole
    fields; tolerances = float32 summation-order differences."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "tools", "sharded_native_worker.py"), "--backend", "gloo", "--same-gpu"]
    cmd += {"plain": [], "mask": ["--mask"], "lowrank": ["--lowrank", "--modes", "9"]}[variant]
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    d = json.loads([ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["world"] == 2 and d["attached"] and all(d["native"]), d
    assert d["p_total"] == [d["p_valid"][0], d["p_valid"][1], d["p_valid"][2], d["p_valid"][3]]
    if variant == "mask":
        assert d["p_valid"][0] < 5000 and d["p_valid"][2] < 3600 and d["p_valid"][3] < 5000
    assert min(d["calls"]) >= 10
    assert d["eof_s"] < 2e-5 and d["mca_s"] < 2e-5 and d["hop_s"] < 2e-5 and d["h2p_s"] < 2e-5, d
    assert d["eof_v_cos"] > 1 - 1e-5 and d["mca_q1_cos"] > 1 - 1e-5 and d["mca_q2_cos"] > 1 - 1e-5, d
    assert d["hop_v_cos"] > 1 - 1e-4 and d["h2p_v_cos"] > 1 - 1e-4, d
    assert d["eof_scores"] < 1e-4 and d["mca_scores1"] < 1e-4 and d["mca_scores2"] < 1e-4 and d["mca_norm1"] < 1e-4, d
    assert d["eof_tv"] < 1e-6 and d["mca_tsc"] < 1e-5 and d["hop_tv"] < 1e-6 and d["h2p_tv"] < 1e-6, d
    # orthonormal factors, also where more modes were asked for than the fields have rank (variant "lowrank")
    for key in ("eof_orth_v", "eof_orth_u", "mca_orth_q1", "mca_orth_q2", "hop_orth_v", "hop_orth_u", "h2p_orth_v", "hpy_orth_v"):
        assert d[key] < 2e-5, (key, d[key])
    # the panel-level operator route (no engine communicator: HilbertOperatorOps + torch.distributed) agrees as well
    assert d["hpy_operator"] and not d["hpy_native"]
    assert d["hpy_s"] < 2e-5 and d["hpy_v_cos"] > 1 - 1e-4 and d["hpy_tv"] < 1e-6, d
