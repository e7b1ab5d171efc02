"""CPU tests of the drop-in boundary: libeofx.so loads, exports every symbol include/eofx.h
declares (no compute calls without a GPU), the ctypes table covers the header, host-side
helpers work, and the product fails loudly when no GPU is present."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "eofx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eofx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from xeofs_amd import _lib

    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/eofx.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert lib.eofx_abi_version() == 1


def test_host_eigh():
    from xeofs_amd import engine

    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 60):
        A = rng.standard_normal((n, n + 3))
        A = A @ A.T
        w, V = engine.host_eigh(A)
        assert np.all(np.diff(w) <= 1e-12 * abs(w[0]))
        np.testing.assert_allclose(w, np.linalg.eigvalsh(A)[::-1], rtol=1e-11, atol=1e-12 * abs(w[0]))
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)
        np.testing.assert_allclose(A @ V, V * w, atol=1e-10 * abs(w[0]))


def test_sketch_matrix_is_sklearns():
    from xeofs_amd.engine import sketch_matrix

    # native generator (eofx_sketch_gaussian_f32) vs numpy's legacy stream, bit for bit, including odd
    # totals (cached second deviate), sizes that span MT19937 refills and the multi-threaded tail
    for seed, shape in [(42, (50, 12)), (0, (7, 3)), (5, (3000, 60)), (2 ** 32 - 1, (1, 1)), (77, (1001, 61))]:
        a = sketch_matrix(shape[0], shape[1], seed)
        b = np.random.RandomState(seed).normal(size=shape).astype(np.float32)
        assert np.array_equal(a, b) and a.flags.c_contiguous, (seed, shape)
    rs = np.random.RandomState(9)
    assert np.array_equal(sketch_matrix(20, 5, rs), np.random.RandomState(9).normal(size=(20, 5)).astype(np.float32))
    with pytest.raises(ValueError):
        sketch_matrix(5, 2, np.random.default_rng(0))


def test_error_code_mapping():
    from xeofs_amd import _lib

    for code in (_lib.ERR_ARG, _lib.ERR_PARTIAL_NAN, _lib.ERR_NAN_MISMATCH, _lib.ERR_RANK, _lib.ERR_SHAPE):
        with pytest.raises(ValueError):
            _lib.raise_for(code)
    with pytest.raises(np.linalg.LinAlgError):
        _lib.raise_for(_lib.ERR_LINALG)
    with pytest.raises(MemoryError):
        _lib.raise_for(_lib.ERR_NOMEM)
    with pytest.raises(_lib.EofxError):
        _lib.raise_for(_lib.ERR_HIP)


def test_no_cpu_fallback():
    """Without a GPU the product refuses to run instead of computing on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from xeofs_amd import _lib, engine

    with pytest.raises(_lib.EofxError, match="no CPU fallback"):
        engine.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "xeofs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_shard_bounds():
    from xeofs_amd.sharded import shard_bounds

    for P, W in [(1036800, 8), (1325, 3), (7, 8)]:
        b = [shard_bounds(P, W, r) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == P
        assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
