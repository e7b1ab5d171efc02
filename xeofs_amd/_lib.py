"""ctypes binding of libeofx.so (C ABI declared in include/eofx.h).

The library is the product: there is NO CPU fallback.  If the shared object is
missing or a symbol is absent the import fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EOFX_LIB: another build of the same ABI -- same-box A/B runs of two library versions, tools/jobs/*; never set in tests)
LIB_PATH = os.environ.get("EOFX_LIB") or os.path.join(_HERE, "lib", "libeofx.so")

EOFX_OK = 0
ERR_ARG, ERR_HIP, ERR_PARTIAL_NAN, ERR_NAN_MISMATCH, ERR_RANK, ERR_LINALG, ERR_NOMEM, ERR_SHAPE = (
    -1, -2, -3, -4, -5, -6, -7, -8)

_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_pi64 = C.POINTER(C.c_int64)
_pd = C.POINTER(C.c_double)

# eofx_sketch_fn (include/eofx.h): const float *(*)(void *user)
SKETCH_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p)

# eofx_allreduce_fn (include/eofx.h): int (*)(void *user, void *device_buf, int64_t count, int dtype, int op, void *stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)

# name -> (restype, argtypes); must list every symbol of include/eofx.h
SIGNATURES = {
    "eofx_abi_version": (_int, []),
    "eofx_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "eofx_ctx_destroy": (_int, [_vp]),
    "eofx_ctx_synchronize": (_int, [_vp]),
    "eofx_ctx_set_stream": (_int, [_vp, _vp]),
    "eofx_last_error": (C.c_char_p, [_vp]),
    "eofx_ctx_trim": (_int, [_vp]),
    "eofx_ctx_set_precision": (_int, [_vp, _int, _int]),
    "eofx_ctx_profile": (_int, [_vp, _int]),
    "eofx_ctx_profile_read": (_int, [_vp, _pi64, _pd, _pd, _pd]),
    "eofx_ctx_profile_by_kernel": (_int, [_vp, _pi64, _pd]),
    "eofx_preprocess_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _vp, _int, C.POINTER(_vp),
                                   _vp, _vp, _vp, _vp, _pi64, _pi64, _pd]),
    "eofx_apply_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, C.POINTER(_vp), _vp, _pi64]),
    "eofx_mat_from_dense_f32": (_int, [_vp, _vp, _i64, _i64, _i64, C.POINTER(_vp)]),
    "eofx_mat_destroy": (_int, [_vp, _vp]),
    "eofx_mat_shape": (_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    "eofx_mat_download_f32": (_int, [_vp, _vp, _vp]),
    "eofx_rsvd_f32": (_int, [_vp, _vp, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "eofx_project_f32": (_int, [_vp, _vp, _vp, _int, _vp]),
    "eofx_reconstruct_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "eofx_crosscov_rsvd_f32": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _int, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _pd]),
    "eofx_crosscov_rsvd_lazy_f32": (_int, [_vp, _vp, _vp, _int, _int, _int, SKETCH_FN, _vp, _int, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _pd]),
    "eofx_panel_tmul_f32": (_int, [_vp, _vp, _vp, _vp, _int, _int]),
    "eofx_panel_mul_f32": (_int, [_vp, _vp, _vp, _vp, _int, _int]),
    "eofx_panel_gram_f64": (_int, [_vp, _vp, _i64, _int, _vp]),
    "eofx_panel_cholqr_f32": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp]),
    "eofx_panel_rinv_f64": (_int, [_vp, _vp, _int, _int, _vp]),
    "eofx_panel_matmul_f32": (_int, [_vp, _vp, _i64, _int, _vp, _int, _vp]),
    "eofx_panel_colminmax_f32": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "eofx_panel_export_f32": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp]),
    "eofx_panel_import_f32": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _int]),
    "eofx_hilbert_f32": (_int, [_vp, _vp, _int, C.c_double, C.POINTER(_vp), C.POINTER(_vp)]),
    "eofx_mat_sumsq_f64": (_int, [_vp, _vp, _pd]),
    "eofx_fit_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _vp, _int, _int, _int, _int, _vp, _i64, _int,
                            C.POINTER(_vp), _vp, _vp, _vp, _vp, _pi64, _pi64, _pd, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "eofx_fit_first_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _vp, _int, _vp, _int, _int, _vp, C.POINTER(_vp),
                                  _vp, _vp, _vp, _vp, _pi64, _pi64, _pd, C.POINTER(C.c_int)]),
    "eofx_ctx_fit_info": (_int, [_vp, _pd]),
    "eofx_ctx_last_iterations": (_int, [_vp, C.POINTER(_int)]),
    "eofx_comm_unique_id": (_int, [C.c_char_p]),
    "eofx_ctx_comm_init_rccl": (_int, [_vp, C.c_char_p, _int, _int]),
    "eofx_ctx_comm_set_callback": (_int, [_vp, ALLREDUCE_FN, _vp, _int, _int]),
    "eofx_ctx_comm_clear": (_int, [_vp]),
    "eofx_ctx_comm_stats": (_int, [_vp, _pi64, _pi64, _pd]),
    "eofx_ctx_comm_selftest": (_int, [_vp, C.POINTER(C.c_int)]),
    "eofx_ctx_comm_probe": (_int, [_vp, _int, _vp, _vp, _int, _vp, _vp]),
    "eofx_fit_sharded_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _vp, _int, _int, _int, _vp, _i64, _int,
                                    C.POINTER(_vp), _vp, _vp, _vp, _pd, _vp, _vp, _vp]),
    "eofx_ctx_comm_allreduce_f64": (_int, [_vp, _vp, _i64, _int]),
    "eofx_crosscov_rsvd_sharded_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _vp, _int, _vp, _vp,
                                              _vp, _vp, _vp, _vp, _vp, _pd]),
    "eofx_rsvd_sharded_c64": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "eofx_rsvd_hilbert_sharded_c64": (_int, [_vp, _vp, _i64, _int, C.c_double, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "eofx_rsvd_c64": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "eofx_rsvd_hilbert_c64": (_int, [_vp, _vp, _int, C.c_double, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "eofx_hilbert_operator_f32": (_int, [_vp, _i64, _int, C.c_double, _vp]),
    "eofx_hilbert_sumsq_f64": (_int, [_vp, _vp, _int, C.c_double, C.POINTER(C.c_double)]),
    "eofx_orth_tall_rule": (_int, [_i64, _int, _int]),
    "eofx_peaked_spectrum": (_int, [_vp, _int, _int]),
    "eofx_ctx_set_layout": (_int, [_vp, _int]),
    "eofx_ctx_set_sample_raw": (_int, [_vp, _int]),
    "eofx_mat_release_raw": (_int, [_vp, _vp]),
    "eofx_mat_ensure_sample_layout": (_int, [_vp, _vp, _int, _vp]),
    "eofx_mat_release_sample_layout": (_int, [_vp, _vp]),
    "eofx_mat_masked": (_int, [_vp, C.POINTER(C.c_int), _pi64]),
    "eofx_mat_layout": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "eofx_panel_rownorm_f64": (_int, [_vp, _vp, _i64, _int, _vp]),
    "eofx_mat_feature_norms_f64": (_int, [_vp, _vp, _vp]),
    "eofx_mat_sample_norms_f64": (_int, [_vp, _vp, _vp]),
    "eofx_panel_bootstrap_f32": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _int, _vp]),
    "eofx_resample_f32": (_int, [_vp, _vp, _vp, _i64, _int, C.POINTER(_vp), _vp, _pd]),
    "eofx_mat_gram_f32": (_int, [_vp, _vp, _int, _vp]),
    "eofx_mat_cross_gram_f32": (_int, [_vp, _vp, _vp, _int, _vp]),
    "eofx_vec_dot_f64": (_int, [_vp, _vp, _vp, _i64, _pd]),
    "eofx_cmat_mul_f32": (_int, [_vp, _vp, _vp, _int, _vp, _int, _int, _vp]),
    "eofx_cpanel_combine_f32": (_int, [_vp, _vp, _vp, _int, _i64, _int, _vp]),
    "eofx_panel_colargminmax_f32": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "eofx_panel_row_normalize_f32": (_int, [_vp, _vp, _i64, _int, _vp]),
    "eofx_panel_rot_step_f64": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _int, C.c_double, _vp]),
    "eofx_cpanel_colabsmax_f32": (_int, [_vp, _vp, _i64, _int, _vp]),
    "eofx_sketch_gaussian_f32": (_int, [C.c_uint32, _i64, _i64, _vp]),
    "eofx_host_eigh_f64": (_int, [_vp, _int, _vp, _vp]),
    "eofx_host_zheigh_top_f64": (_int, [_vp, _vp, _int, _int, _vp, _vp, _vp]),
}

PREC = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3, "f64": 4}

_lib = None


def load() -> C.CDLL:
    """Load libeofx.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C xeofs_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "xeofs_amd has no CPU fallback.")
    # torch ships its own HIP runtime: it has to be the one the process binds first, otherwise
    # torch later reports "No HIP GPUs are available" (two libamdhip64 copies in one process).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.eofx_abi_version() != 1:
        raise ImportError("libeofx.so ABI version mismatch")
    _lib = lib
    return lib


class EofxError(RuntimeError):
    pass


def raise_for(code: int, ctx=None):
    """Map a status code onto the exception type the reference raises at that condition."""
    if code == EOFX_OK:
        return
    msg = ""
    if ctx:
        msg = load().eofx_last_error(ctx).decode(errors="replace")
    if code in (ERR_ARG, ERR_PARTIAL_NAN, ERR_NAN_MISMATCH, ERR_RANK, ERR_SHAPE):
        raise ValueError(msg or f"eofx error {code}")
    if code == ERR_LINALG:
        raise np.linalg.LinAlgError(msg or "SVD failed.")
    if code == ERR_NOMEM:
        raise MemoryError(msg or "device allocation failed")
    raise EofxError(msg or f"eofx HIP failure ({code})")


def ptr(a):
    """Raw pointer of a numpy array / torch tensor / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))
