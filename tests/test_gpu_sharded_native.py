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
