"""ctypes binding of libifr.so (the C ABI declared in include/ifr.h).

The library is built in-tree by ``recommenders_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is no CPU fallback: if the shared object is
missing, loading raises, and every compute entry point needs a GPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libifr.so")

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class CsrSplit(ctypes.Structure):
    """ifr_csr_split (include/ifr.h): one rating split in device memory."""
    _fields_ = [("indptr", vp), ("indices", vp), ("ratings", vp), ("n_rows", ctypes.c_int64)]


# name -> (restype, argtypes); one row per declaration in include/ifr.h
SIGNATURES = {
    "ifr_version": (ctypes.c_char_p, []),
    "ifr_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_int]),
    "ifr_destroy": (ctypes.c_int, [vp]),
    "ifr_last_error": (ctypes.c_char_p, [vp]),
    "ifr_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_int64]),
    "ifr_set_factors": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, vp, vp]),
    "ifr_set_user_factors": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, vp]),
    "ifr_set_item_matrix_dense": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, vp]),
    "ifr_set_item_matrix_dense_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, vp]),
    "ifr_score_dense_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]),
    "ifr_select_topk_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp]),
    "ifr_rank_full_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp, vp]),
    "ifr_rank_items_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]),
    "ifr_auc_dense_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]),
    "ifr_eval_dense_metrics_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int, vp, vp, vp, vp]),
    "ifr_pad_scores_min_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, vp]),
    "ifr_eval_dense_metrics": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int, vp, vp, vp, vp]),
    "ifr_select_topk": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp]),
    "ifr_fit_rp3beta": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, vp]),
    "ifr_fit_ease": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                    ctypes.c_int, vp]),
    "ifr_set_item_matrix_csr": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]),
    "ifr_set_history_csr": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64]),
    "ifr_item_fixed_point_shift": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int)]),
    "ifr_item_csr_layout": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "ifr_set_exclusion_csr": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64]),
    "ifr_set_truth_csr": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64]),
    "ifr_build_eval_csrs": (ctypes.c_int, [vp, ctypes.POINTER(CsrSplit), ctypes.POINTER(CsrSplit), ctypes.POINTER(CsrSplit),
                                           ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           vp, c_i64p, c_i64p, ctypes.POINTER(ctypes.c_int)]),
    "ifr_read_eval_csr": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, c_i64p, c_i64p]),
    "ifr_rank_topk": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp]),
    "ifr_rank_topk_noalloc": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp]),
    "ifr_rank_topk_dense": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, ctypes.c_int64, vp]),
    "ifr_rank_topk_dense_f64": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, ctypes.c_int64, vp]),
    "ifr_eval_topk_metrics": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int, c_i32p, ctypes.c_int,
                                             vp, vp, vp, vp]),
    "ifr_eval_ranked_metrics": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, c_i32p, ctypes.c_int,
                                               vp, vp, vp, vp]),
    "ifr_rank_items": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]),
    "ifr_score_dense": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]),
    "ifr_auc_dense": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]),
    "ifr_rank_full": (ctypes.c_int, [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp, vp]),
    "ifr_pad_scores_min": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, vp]),
    "ifr_profile_enable": (ctypes.c_int, [vp, ctypes.c_int]),
    "ifr_profile_read": (ctypes.c_int, [vp, c_f64p, c_i64p, c_f64p]),
    "ifr_wide_overflows": (ctypes.c_int, [vp, c_i64p]),
    "ifr_wide_stats": (ctypes.c_int, [vp, c_i64p, c_f64p, c_f64p]),
    "ifr_debug_wide_counts": (ctypes.c_int, [vp, vp, ctypes.c_int64, c_i64p]),
    "ifr_debug_read_stamps": (ctypes.c_int, [vp, vp, ctypes.c_int64, c_i64p]),
    "ifr_factor_chain_order": (ctypes.c_int, [vp, c_i32p, ctypes.POINTER(ctypes.c_int)]),
}

_lib = None


def load():
    """Load libifr.so (after torch, so that both share one HIP runtime) and bind signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C recommenders_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; libifr then binds to the same runtime)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
