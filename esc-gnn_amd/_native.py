"""ctypes binding of libescgnn_hip.so (C ABI declared in include/escgnn_hip.h).

The library is the product: there is NO fallback.  If the shared object is missing or a symbol
cannot be resolved, importing/using the hot path raises immediately.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libescgnn_hip.so")
ABI_VERSION = 3

P, I64, I32, F32 = c_void_p, c_int64, c_int, c_float

class CollateArgs(ctypes.Structure):
    """mirror of `esc_collate_args` (include/escgnn_hip.h)"""
    _fields_ = ([(n, c_int64) for n in ("B", "x_dim", "y_dim", "n_cols")] +
                [(n, c_void_p) for n in (
                    "graph_ids", "offsets", "node_ptr", "edge_ptr", "nnz_ptr", "y_ptr", "x_all", "y_all",
                    "esrc_all", "edst_all", "pos_enc_all", "pos_index_all", "pos_batch_all",
                    "in_ptr_all", "in_edge_all", "out_ptr_all", "out_edge_all", "row_ptr_all", "c_perm_all",
                    "c_rank_all", "col_ptr", "col_prefix",
                    "x", "y", "edge_index", "batch", "pos_enc", "pos_index", "pos_batch",
                    "in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst",
                    "row_ptr", "bag_idx", "bag_val", "col_row", "col_val", "col_col", "edge_attr_all", "edge_attr")] +
                [("ea_words", c_int64), ("x_long", c_void_p), ("graph_ptr", c_void_p)])

class BnFuse(ctypes.Structure):
    """mirror of `esc_bn_fuse` (include/escgnn_hip.h)"""
    _fields_ = [("eps", c_float), ("momentum", c_float)] + [(n, c_void_p) for n in (
        "mean", "invstd", "running_mean", "running_var", "gamma", "beta", "scale", "shift")]


class BnFold(ctypes.Structure):
    """mirror of `esc_bn_fold` (include/escgnn_hip.h): a BatchNorm still in partial form, merged by its consumer"""
    _fields_ = [("partials", c_void_p), ("rows", c_int64), ("block_rows", c_int64), ("C", c_int64),
                ("eps", c_float), ("momentum", c_float)] + [(n, c_void_p) for n in (
        "gamma", "beta", "mean", "invstd", "scale", "shift", "running_mean", "running_var")]


class BnBwdFused(ctypes.Structure):
    """mirror of `esc_bn_bwd_fused`: a BatchNorm(+ReLU) backward applied to the dY operand of a Linear backward"""
    _fields_ = [("x", c_void_p), ("ld_x", c_int64)] + [(n, c_void_p) for n in ("mean", "invstd", "scale", "shift", "coef")] + \
               [("relu", ctypes.c_int32)]


class BnBwdNext(ctypes.Structure):
    """mirror of `esc_bn_bwd_next`: column sums of the NEXT BatchNorm backward from the dX tiles"""
    _fields_ = [("partial", c_void_p), ("x", c_void_p), ("ld_x", c_int64)] + \
               [(n, c_void_p) for n in ("mean", "invstd", "scale", "shift")] + [("relu", ctypes.c_int32)]


# name -> argtypes (every function returns int unless listed in _RET)
SIGNATURES = {
    "esc_abi_version": [],
    "esc_last_error": [],
    "esc_prof_enable": [I32, I32],
    "esc_prof_read": [I32, POINTER(c_int64), POINTER(c_double)],
    "esc_prof_reset": [I32],
    "esc_prof_read_all": [I32, POINTER(c_double), c_int64],
    "esc_prof_span_arm": [I32, c_int64, P],
    "esc_prof_span_read": [I32, POINTER(c_double), c_int64],
    "esc_bag_fwd": [P, I64, P, P, P, I64, P, I64, P],
    "esc_bag_fwd_rows": [P, I64, I64, P, P, P, I64, P, I64, I32, P, P],
    "esc_bag_fwd_stats_block_rows": [P, I64, I64, P, I64, I64],
    "esc_bag_bwd_scratch": [I64, I64],
    "esc_bag_bwd_table": [P, I64, I64, P, P, P, P, I64, I64, P, P, P],
    "esc_bag_bwd_classify": [P, I64, I64, I64, P, P],
    "esc_bag_bwd_table_rows": [P, I64, I64, P, P, P, P, I64, I64, I64, I32, P, P, P],
    "esc_gine_aggregate_fwd": [P, I64, P, I64, P, P, P, P, I64, I64, P, I64, P],
    "esc_gine_aggregate_bwd": [P, I64, P, I64, P, I64, P, P, P, P, I64, I64, P, I64, P, I64, I32, P, P],
    "esc_gine_aggregate_bwd_deps_slots": [I64],
    "esc_reduce_sum": [P, I64, P, P],
    "esc_reduce_sum_jobs": [P, I32, P],
    "esc_segment_pool_fwd": [P, I64, P, I64, I64, I32, P, I64, P],
    "esc_segment_pool_bwd": [P, I64, P, I64, I64, I32, P, I64, P],
    "esc_linear_fwd": [P, I64, P, I64, P, P, P, I64, I64, I64, P, I64, P, P],
    "esc_linear_bn_fwd": [P, I64, P, I64, P, P, P, I64, I64, I64, P, I64, P, POINTER(BnFuse), P],
    "esc_linear_fwd_l1": [P, I64, P, P, P, P, I64, I64, P, I64, F32, P, P, P],
    "esc_linear_fwd_l1_ok": [P, I64, P, I64, P, P],
    "esc_linear_fwd_from": [P, I64, P, I64, P, I64, P, P, P, I64, I64, I64, P, I64, P, P],
    "esc_linear_fwd_from_ok": [P, I64, P, I64, I64, I64, I64, I32],
    "esc_linear_stats_block_rows": [P, I64, P, I64, I64, I64, I64],
    "esc_linear_fold_available": [],
    "esc_engine_phase_times": [POINTER(c_double), I32],
    "esc_engine_set_collective": [P, P, I32, I32, P, P, I64],
    "esc_bn_sync_pack": [P, P, I64, F32, I64, I32, I32, P, P],
    "esc_bn_sync_finalize": [P, I32, I64, F32, F32, P, P, P, P, P, P, P, P, P, P],
    "esc_bn_sync_coef": [P, I64, P, P],
    "esc_linear_fwd_fold": [P, I64, P, I64, P, POINTER(BnFold), I64, I64, I64, P, I64, P, P],
    "esc_bn_stats_from_partials_rows": [P, I64, I64, I64, F32, F32, P, P, P, P, P, P, P, P, P],
    "esc_affine_act_fold": [P, I64, I64, I64, POINTER(BnFold), I32, P, I64, P],
    "esc_plan_csr_scratch": [I64, I64],
    "esc_plan_csr": [P, I64, I64, P, P, P, P, P],
    "esc_embed_plan_scratch": [I64, I64, I64],
    "esc_embed_plan": [P, I64, I64, P, P, P, P, P, P, P, P, P, P],
    "esc_tune_set": [I32, I32],
    "esc_debug_gemm_occupancy": [I32],
    "esc_linear_bwd_input": [P, I64, P, I64, I64, I64, I64, P, I64, I32, P],
    "esc_linear_bwd_weight_scratch": [I64, I64, I64],
    "esc_linear_bwd_weight": [P, I64, P, I64, P, P, I64, I64, I64, P, I64, P, P, P],
    "esc_linear_bwd_both": [P, I64, P, I64, P, P, P, I64, I64, I64, I64, P, I64, I32, P, I64, P, P, P],
    "esc_linear_bwd_both_deferred": [P, I64, P, I64, P, P, P, I64, I64, I64, I64, P, I64, I32, P, I64, P, P, P, P],
    "esc_slab_reduce_jobs": [P, I32, P],
    "esc_linear_bwd_both_bn_ok": [P, I64, POINTER(BnBwdFused), P, I64, P, I64, I64, I64, I64, P, I64, P, POINTER(BnBwdNext)],
    "esc_linear_bwd_bn_block_rows": [I64, I64, I64],
    "esc_linear_bwd_set_wgrad_stream": [P],
    "esc_linear_bwd_both_bn": [P, I64, POINTER(BnBwdFused), P, I64, P, P, P, I64, I64, I64, I64, P, I64, I32, P, I64, P, P, P,
                               POINTER(BnBwdNext), P],
    "esc_bn_bwd_coef": [P, I64, P, I64, P, I64, I64, I64, P, P, P, P, I32, P, P, P, P, P],
    "esc_bn_bwd_coef_from_partials": [P, I64, I64, I64, P, P, P, P],
    "esc_bn_scratch": [I64],
    "esc_bn_stats": [P, I64, I64, I64, F32, F32, P, P, P, P, P, P, P, P, P, P],
    "esc_bn_stats_from_partials": [P, I64, I64, F32, F32, P, P, P, P, P, P, P, P, P],
    "esc_bn_apply": [P, I64, I64, I64, P, P, P, P, I32, P, I64, P],
    "esc_bn_bwd": [P, I64, P, I64, P, I64, I64, I64, P, P, P, P, I32, P, I64, P, P, P, P],
    "esc_bn_bwd_sums": [P, I64, P, I64, P, I64, I64, I64, P, P, P, P, I32, P, P, P, P, P],
    "esc_bn_bwd_apply": [P, I64, P, I64, P, I64, I64, I64, P, P, P, P, I32, P, P, I64, P],
    "esc_affine_act": [P, I64, I64, I64, P, P, I32, P, I64, P],
    "esc_bn_eval_coef": [P, P, P, P, F32, I64, P, P, P],
    "esc_engine_set_side_stream": [I32],
    "esc_engine_set_materialise_edge_act": [I32],
    "esc_engine_set_gemm_stats": [I32],
    "esc_engine_workspace_floats": [P, I64, I64, I64],
    "esc_engine_train_step": [P, P, P, I64, P, P, P],
    "esc_engine_train_step_begin": [P, P, P, I64, P, P, P],
    "esc_engine_train_step_end": [],
    "esc_engine_forward_train": [P, P, P, P, P],
    "esc_engine_backward": [P, P, P, P, P],
    "esc_engine_predict": [P, P, P, P, P],
    "esc_l1_loss": [P, P, I64, I64, F32, P, P, P],
    "esc_bce_logits_loss": [P, P, I64, I64, P, P, P],
    "esc_adam_step": [P, P, P, P, I64, c_double, c_double, c_double, c_double, I64, P],
    "esc_adam_step_scaled": [P, P, P, P, I64, c_double, c_double, c_double, c_double, I64, P, P],
    "esc_collate_cols": [P, I64, P, I64, P, P, P, P],
    "esc_collate_fill": [POINTER(CollateArgs), P],
    "esc_engine_set_two_stream_min_edges": [I64],
    "esc_gine_aggregate_fwd_affine": [P, I64, P, P, P, I64, P, P, P, P, I64, I64, P, I64, P],
    "esc_gine_aggregate_bwd_affine": [P, I64, P, P, P, I64, P, I64, P, P, P, P, I64, I64, P, I64, P, I64, I32, P, P],
    "esc_gine_aggregate_bwd_stats_slots": [I64],
    "esc_gine_aggregate_bwd_affine_stats": [P, I64, P, P, P, P, P, I64, P, I64, P, P, P, P, I64, I64, P, I64, P, I64, I32, P, P, P],
    "esc_embed_fwd": [P, I64, I64, P, I64, P, I64, P, P],
    "esc_embed_bwd": [P, I64, P, I64, I64, I64, P, P],
    "esc_zinc_workspace_floats": [P, I64, I64, I64, I64],
    "esc_zinc_train_step": [P, P, P, I64, P, P, P],
    "esc_zinc_forward_train": [P, P, P, P, P],
    "esc_zinc_backward": [P, P, P, P, P],
    "esc_zinc_predict": [P, P, P, P, P],
    "esc_segment_broadcast_add": [P, I64, P, I64, P, I64, I64, I64, P, I64, P],
    "esc_dropout_fwd": [P, I64, I64, I64, ctypes.c_float, ctypes.c_uint64, P, I64, P, I64, P, P],
    "esc_bn_bwd_dropout_ok": [I64, I64, I64, I64],
    "esc_bn_bwd_dropout": [P, I64, P, I64, I64, I64, P, P, P, P, I32, P, ctypes.c_float, I32, P, I64, P, P, P, P],
    "esc_affine_act_dropout_fwd": [P, I64, I64, I64, P, P, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, P, I64, P, I64, P, P],
    "esc_dropout_bwd": [P, I64, I64, I64, ctypes.c_float, P, P, I64, P, I64, P],
    "esc_table_pack": [P, I64, P, P],
    "esc_table_unpack_grad": [P, I64, P, P],
    "esc_bag_fwd_acc": [P, I64, P, P, P, I64, P, I64, P],
    "esc_ogb_workspace_floats": [P, I64, I64, I64, I64, I64, I64],
    "esc_ogb_train_step": [P, P, P, I64, P, P, P],
    "esc_ogb_forward_train": [P, P, P, P, P],
    "esc_ogb_backward": [P, P, P, P, P],
    "esc_ogb_predict": [P, P, P, P, P],
    "esc_features_scratch_bytes": [I64, I64, I64, I64, I64, I32],
    "esc_features_count": [P, P, P, P, I64, I64, I64, I64, I64, I32, I32, I32, P, P, P, P, P],
    "esc_features_fill": [P, P, I64, I64, I64, I64, I64, I32, I32, I32, P, P, I64, P, P, P, P, P, P, P, P, P],
}
_RET = {"esc_last_error": ctypes.c_char_p, "esc_bag_bwd_scratch": c_int64, "esc_bag_fwd_stats_block_rows": c_int64, "esc_gine_aggregate_bwd_stats_slots": c_int64, "esc_linear_stats_block_rows": c_int64, "esc_plan_csr_scratch": c_int64, "esc_embed_plan_scratch": c_int64, "esc_prof_read_all": c_int64, "esc_prof_span_read": c_int64,
        "esc_linear_bwd_weight_scratch": c_int64, "esc_bn_scratch": c_int64, "esc_linear_bwd_bn_block_rows": c_int64,
        "esc_features_scratch_bytes": c_int64, "esc_engine_workspace_floats": c_int64,
        "esc_zinc_workspace_floats": c_int64, "esc_ogb_workspace_floats": c_int64}



KIND = {"agg_fwd": 0, "agg_bwd": 1, "bag_fwd": 2, "bag_bwd": 3, "linear": 4, "collate": 5,
        "features": 6, "norm": 7, "gemm_edge": 8}

_lib = None


class NativeLibraryError(ImportError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "esc_gnn_amd: %s not found. The HIP hot path has no fallback — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C esc-gnn_amd/csrc`." % LIB_PATH)
    h = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as exc:
            raise NativeLibraryError("esc_gnn_amd: %s lacks symbol %s (stale build?)" % (LIB_PATH, name)) from exc
        fn.argtypes = args
        fn.restype = _RET.get(name, c_int)
    if h.esc_abi_version() != ABI_VERSION:
        raise NativeLibraryError("esc_gnn_amd: ABI version mismatch (lib %d, python %d) — rebuild"
                                 % (h.esc_abi_version(), ABI_VERSION))
    _lib = h
    return h


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message on failure."""
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, h.esc_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def prof_enable(kind, on=True):
    call("esc_prof_enable", KIND[kind], int(on))


def prof_reset(kind):
    call("esc_prof_reset", KIND[kind])


def prof_read(kind):
    n, ms = c_int64(0), c_double(0.0)
    call("esc_prof_read", KIND[kind], ctypes.byref(n), ctypes.byref(ms))
    return n.value, ms.value


def prof_span_arm(kind, launches):
    """also stamp the in-kernel execution window of the next `launches` profiled launches (kernels with a span argument)"""
    call("esc_prof_span_arm", KIND[kind], int(launches), stream())


def prof_span_read(kind, cap=1 << 16):
    """execution windows (us) of the armed launches, in launch order — call after a device synchronise"""
    buf = (c_double * cap)()
    n = lib().esc_prof_span_read(KIND[kind], buf, cap)
    return [buf[i] for i in range(n)]


def prof_read_all(kind, cap=1 << 16):
    """per-launch durations (ms) of the recorded launches of a kernel family, in launch order"""
    buf = (c_double * cap)()
    n = lib().esc_prof_read_all(KIND[kind], buf, cap)
    return [buf[i] for i in range(n)]
