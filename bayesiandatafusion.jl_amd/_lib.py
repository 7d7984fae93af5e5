"""ctypes binding of libbdf_hip.so (include/bdf.h).

The library is the product: there is no Python/CPU fallback for anything it computes.  A missing
library raises at import of this module's `lib()`; a missing GPU raises at context creation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BDF_LIB_PATH selects another build of the same library (kernel tuning variants, tools/ab_k1.sh)
LIB_PATH = os.environ.get("BDF_LIB_PATH") or os.path.join(_HERE, "csrc", "libbdf_hip.so")

BDF_MAX_MODES = 4
BDF_MAX_TERMS = 4
BDF_MAX_D = 64

P_ROW, P_BETA_E1, P_BETA_E2, P_NW_NORMAL, P_GAMMA_N, P_GAMMA_U, P_NW_MEAN, P_BETA_REL1, P_BETA_REL2 = 1, 2, 3, 4, 5, 6, 7, 8, 9


class ArgumentError(ValueError):
    """The reference's ArgumentError."""


class DimensionMismatch(ValueError):
    """The reference's DimensionMismatch."""


class BoundsError(IndexError):
    """The reference's BoundsError."""


class HipError(RuntimeError):
    pass


class NotPositiveDefinite(ArithmeticError):
    pass


class NoGpuError(RuntimeError):
    pass


_ERR = {-1: ArgumentError, -2: BoundsError, -3: HipError, -4: NotPositiveDefinite, -5: NoGpuError}

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)


class Term(C.Structure):
    """bdf_term"""
    _fields_ = [("rel", C.c_void_p), ("mode", C.c_int32), ("_pad", C.c_int32), ("alpha", C.c_double),
                ("mean_value", C.c_double), ("linear_values", C.c_void_p), ("factors", C.c_void_p * BDF_MAX_MODES),
                ("alpha_dev", C.c_void_p)]


BDF_COMM_ID_BYTES = 128


class _GibbsTerm(C.Structure):
    _fields_ = [("rel", C.c_void_p), ("mode", C.c_int32), ("entity_of_mode", C.c_int32 * BDF_MAX_MODES),
                ("alpha", C.c_double), ("mean_value", C.c_double)]


class GibbsEntity(C.Structure):
    """bdf_gibbs_entity"""
    _fields_ = [("N", C.c_int64), ("n_real", C.c_int64), ("tag", C.c_uint32), ("n_terms", C.c_int32),
                ("terms", _GibbsTerm * BDF_MAX_TERMS), ("sample", C.c_void_p * 3),
                ("mu", C.c_void_p), ("Lambda", C.c_void_p), ("mu0", C.c_void_p), ("WI", C.c_void_p), ("sumU", C.c_void_p),
                ("UUt", C.c_void_p), ("params", C.c_void_p), ("prior_pack", C.c_void_p), ("draws", C.c_void_p),
                ("b0", C.c_double), ("nu0", C.c_double),
                ("feat", C.c_void_p), ("beta", C.c_void_p), ("uhat", C.c_void_p), ("mu_matrix", C.c_void_p), ("Tinv", C.c_void_p),
                ("lambda_beta", C.c_void_p), ("cg_iters", C.c_void_p), ("use_ff", C.c_int32), ("sample_lambda_beta", C.c_int32),
                ("full_lambda_u", C.c_int32), ("_pad", C.c_int32), ("tol", C.c_double), ("lb_nu", C.c_double), ("lb_mu", C.c_double)]


class GibbsRelation(C.Structure):
    """bdf_gibbs_relation"""
    _fields_ = [("rel", C.c_void_p), ("entity_of_mode", C.c_int32 * BDF_MAX_MODES), ("mean_value", C.c_double), ("alpha_dev", C.c_void_p),
                ("alpha_sample", C.c_int32), ("rel_tag", C.c_uint32), ("alpha_lambda0", C.c_double), ("alpha_nu0", C.c_double),
                ("nnz", C.c_int64), ("train", C.c_void_p), ("first_obs", C.c_int64), ("obs_block", C.c_int64), ("feat", C.c_void_p),
                ("beta", C.c_void_p), ("linear", C.c_void_p), ("lambda_beta", C.c_double), ("feat_test", C.c_void_p),
                ("test_baseline", C.c_void_p)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

_SIGS = {
    # name: (restype, argtypes)
    "bdf_last_error": (C.c_char_p, []),
    "bdf_version": (C.c_int, []),
    "bdf_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bdf_ctx_destroy": (C.c_int, [C.c_void_p]),
    "bdf_ctx_set_sweep": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bdf_ctx_advance_sweep": (C.c_int, [C.c_void_p]),
    "bdf_ctx_sync": (C.c_int, [C.c_void_p]),
    "bdf_ctx_warnings": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "bdf_ctx_set_piece_size": (C.c_int, [C.c_void_p, C.c_int]),
    "bdf_ctx_set_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "bdf_rows_unfinished": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "bdf_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bdf_event_destroy": (C.c_int, [C.c_void_p]),
    "bdf_event_elapsed_us": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "bdf_ctx_time_next_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_ctx_time_next_hyper": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_ctx_set_item_size": (C.c_int, [C.c_void_p, C.c_int]),
    "bdf_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "bdf_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdf_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bdf_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bdf_index_build": (C.c_int, [C.c_int, c_i64p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(c_i64p), C.POINTER(c_i64p)]),
    "bdf_relation_create": (C.c_int, [C.c_void_p, C.c_int, c_i64p, C.c_int64, C.c_void_p, C.c_int, c_dp, C.POINTER(C.c_void_p)]),
    "bdf_relation_destroy": (C.c_int, [C.c_void_p]),
    "bdf_relation_index": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(c_i64p), C.POINTER(c_i64p)]),
    "bdf_relation_value_mean": (C.c_int, [C.c_void_p, c_dp]),
    "bdf_relation_order": (C.c_int, [C.c_void_p, C.c_int, c_i32p]),
    "bdf_sample_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(Term), C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bdf_prior_pack_doubles": (C.c_int, [C.c_int]),
    "bdf_sample_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                   C.c_void_p, C.c_uint32, C.c_void_p]),
    "bdf_row_system": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(Term), C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "bdf_normals": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "bdf_philox": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bdf_hyper_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_hyper_sums_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_hyper_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                   C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_hyper_draws": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint32, C.c_void_p]),
    "bdf_pairs_sort": (C.c_int, [C.c_void_p, C.c_int]),
    "bdf_pairs_order": (C.c_int, [C.c_void_p, c_i64p]),
    "bdf_pairs_set_baseline": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdf_feat_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "bdf_predict_sse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_void_p, C.c_void_p]),
    "bdf_sample_alpha": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bdf_sample_beta_rel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_double,
                                      C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_sample_beta_rel_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p),
                                            C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_sum_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "bdf_gibbs_recorded": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bdf_gibbs_set_recorded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "bdf_ctx_set_small_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "bdf_ctx_set_lowrank": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "bdf_ctx_set_col_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "bdf_ctx_rows_dispatch": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_int64)]),
    "bdf_pairs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, c_dp, C.POINTER(C.c_void_p)]),
    "bdf_pairs_destroy": (C.c_int, [C.c_void_p]),
    "bdf_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_void_p]),
    "bdf_predict_all": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_void_p]),
    "bdf_predict_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_void_p]),
    "bdf_pairs_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_i64p]),
    "bdf_feat_create_dense": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_dp, C.POINTER(C.c_void_p)]),
    "bdf_feat_create_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_i32p, c_i32p, c_dp, C.POINTER(C.c_void_p)]),
    "bdf_feat_create_bin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_i32p, c_i32p, C.POINTER(C.c_void_p)]),
    "bdf_feat_destroy": (C.c_int, [C.c_void_p]),
    "bdf_feat_set_row_ids": (C.c_int, [C.c_void_p, c_i32p]),
    "bdf_feat_size": (C.c_int, [C.c_void_p, c_i64p, c_i64p, c_i64p]),
    "bdf_feat_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "bdf_feat_AtA_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "bdf_uhat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_hyper_feature_terms": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdf_sample_beta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "bdf_sample_beta_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "bdf_allgather_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bdf_ctx_create_side": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bdf_ctx_create_rows": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "bdf_ctx_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bdf_layout_build": (C.c_int, [C.c_int64, c_i64p, C.c_int, C.c_int, c_i32p, c_i64p]),
    "bdf_relation_create_sharded": (C.c_int, [C.c_void_p, C.c_int, c_i64p, C.c_int64, C.c_void_p, C.c_int, c_dp, C.POINTER(c_i32p),
                                              c_i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bdf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bdf_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "bdf_comm_create_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, EXCHANGE_FN, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bdf_comm_enable_peer": (C.c_int, [C.c_void_p, EXCHANGE_FN, C.c_void_p, C.c_size_t]),
    "bdf_comm_disable_peer": (C.c_int, [C.c_void_p]),
    "bdf_comm_peer_selftest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bdf_comm_peer_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bdf_comm_destroy": (C.c_int, [C.c_void_p]),
    "bdf_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bdf_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int]),
    "bdf_allgather_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdf_gibbs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(GibbsEntity), C.POINTER(C.c_void_p)]),
    "bdf_gibbs_destroy": (C.c_int, [C.c_void_p]),
    "bdf_gibbs_contexts": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "bdf_gibbs_set_test": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_void_p]),
    "bdf_gibbs_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdf_gibbs_set_relations": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bdf_gibbs_sweep": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int]),
    "bdf_gibbs_current": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "bdf_gibbs_time_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bdf_gibbs_span_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bdf_ctx_span_next_rows": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdf_gibbs_sync": (C.c_int, [C.c_void_p]),
    "bdf_gibbs_rows_only": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "bdf_gibbs_warm_device": (C.c_int, [C.c_void_p, C.c_double]),
    "bdf_synth_ratings": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, c_i32p, c_i32p,
                                    c_dp, C.c_void_p]),
}

_LIB = None


def _preload_torch_hip_runtime():
    """torch ships its own libamdhip64.so (soname libamdhip64.so.7, the same soname as /opt/rocm's).  If this
    library pulled in the system copy first, a later `import torch` would load a second HIP runtime into the
    process and one of the two would see no device.  Loading torch's copy first makes both resolve to it; hosts
    without torch (e.g. Julia) simply use the system runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded library; raises if it has not been built (python __graft_entry__.py build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP library is the product and has no fallback. "
                              "Build it with `make -C bayesiandatafusion.jl_amd/csrc` (hipcc --offload-arch=gfx950).")
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def declared_symbols():
    return sorted(_SIGS)


def check(rc):
    if rc != 0:
        msg = lib().bdf_last_error().decode("utf-8", "replace")
        exc = _ERR.get(rc, RuntimeError)
        if exc is ArgumentError and "DimensionMismatch" in msg:
            exc = DimensionMismatch
        raise exc(msg)
    return rc
