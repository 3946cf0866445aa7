"""ctypes binding of libt2p_hip.so (C ABI: include/t2p.h).

The product path has no CPU fallback: if the HIP library is missing or fails a call, this raises.
torch is imported first so that the library binds to the HIP runtime (libamdhip64.so.7) torch already loaded --
one runtime per process, which is what makes torch's device pointers and streams valid inside the kernels.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2P_LIB") or os.path.join(_HERE, "libt2p_hip.so")  # T2P_LIB: A/B builds of the same ABI
ABI_VERSION = 27

c_float_p = C.POINTER(C.c_float)
c_void = C.c_void_p


class CellWeights(C.Structure):
    _names = (["sa_w1", "sa_b1", "sa_w2", "sa_b2"], ["ga_w1", "ga_b1", "ga_w2", "ga_b2", "lin1_w", "lin1_b", "lin2_w",
              "lin2_b", "pn_w", "pn_b", "col_w1", "col_b1", "col_w2", "col_b2", "pos_w1", "pos_b1", "pos_w2", "pos_b2",
              "merge_w", "merge_b", "g_wp", "g_bp", "g_wq", "g_w2", "g_b2", "lin_w1", "lin_b1", "lin_w2", "lin_b2"])
    _fields_ = ([(n, c_void * 3) for n in _names[0]] + [(n, c_void) for n in _names[1]] +
                [("sa_w2_x3", c_void * 3), ("ga_w2_x3", c_void), ("sa_b2_x3", c_void * 3), ("sa_w2_scale", C.c_float * 3),
                 ("lin1_x3", c_void), ("lin2_x3", c_void), ("merge_x3", c_void), ("pn_x3", c_void), ("g_wp_x3", c_void),
                 ("g_wq_x3", c_void), ("lin1_scale", C.c_float), ("lin2_scale", C.c_float), ("merge_scale", C.c_float),
                 ("pn_scale", C.c_float), ("g_wp_scale", C.c_float), ("g_wq_scale", C.c_float),
                 ("sa_w1_x3", c_void * 3), ("ga_w1_x3", c_void),
                 ("class_embedding", c_void), ("color_embedding", c_void),
                 ("ga_w1_l1", C.c_float), ("ga_b1_absmax", C.c_float), ("sa_wp_l1", C.c_float * 3),
                 ("sa_a1_l1", C.c_float), ("sa_b1_absmax", C.c_float), ("g_w2_x3", c_void)])


class CellConfig(C.Structure):
    _fields_ = [("n_pts", C.c_int32), ("embed_dim", C.c_int32), ("pointnet_features", C.c_int32),
                ("use_class", C.c_int32), ("use_color", C.c_int32), ("use_position", C.c_int32),
                ("self_loops", C.c_int32), ("knn_k", C.c_int32), ("variation", C.c_int32),
                ("radius", C.c_float * 3), ("chunk_objects", C.c_int32), ("precision", C.c_int32),
                ("class_embed", C.c_int32), ("color_embed", C.c_int32), ("class_idx", c_void), ("color_idx", c_void),
                ("objects_only", C.c_int32), ("overflow_flag", c_void), ("tuning", C.c_int32)]


class CellTrace(C.Structure):
    _fields_ = [("fps_idx", c_void * 3), ("nbr", c_void * 3), ("cnt", c_void * 3), ("sa_out", c_void * 3),
                ("features0", c_void), ("features2", c_void), ("obj_emb", c_void), ("knn_idx", c_void),
                ("features1", c_void)]


class MatchWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("cross", c_void)] + \
               [(n, c_void) for n in ("wqkv", "bqkv", "wm", "bm", "w1", "b1", "w2", "b2", "wf", "bf")] + \
               [("bin_score", C.c_float)] + [(n, c_void) for n in ("wo1", "bo1", "wo2", "bo2")] + \
               [(n, c_void) for n in ("wqkv_x3", "wm_x3", "w1_x3", "w2_x3", "wf_x3")] + \
               [(n, C.c_float) for n in ("scale_qkv", "scale_m", "scale_1", "scale_2", "scale_f")]


class TextWeights(C.Structure):
    _fields_ = [("embedding", c_void), ("w_ih", c_void), ("w_hh", c_void), ("bias", c_void), ("w_hh_x3", c_void),
                ("w_hh_scale", C.c_float)]


# every symbol include/t2p.h declares: (restype, argtypes)
SYMBOLS = {
    "t2p_abi_version": (C.c_int, []),
    "t2p_last_error": (C.c_char_p, []),
    "t2p_encode_cells_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.POINTER(CellConfig)]),
    "t2p_encode_cells": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int64,
                                   C.POINTER(CellWeights), C.POINTER(CellConfig), c_void, C.POINTER(CellTrace), c_void,
                                   C.c_size_t, c_void]),
    "t2p_encode_text_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "t2p_encode_text": (C.c_int, [c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(TextWeights),
                                  c_void, c_void, c_void, C.c_size_t, c_void]),
    "t2p_sim_topk_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "t2p_sim_topk": (C.c_int, [c_void, c_void, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, c_void, c_void,
                               c_void, C.c_size_t, c_void]),
    "t2p_lstm_cell_forward": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        c_void, c_void, c_void, c_void, c_void, c_void]),
    "t2p_lstm_train_forward": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_void, c_void,
                                        c_void, c_void, c_void]),
    "t2p_lstm_train_backward": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, c_void, c_void,
                                         c_void]),
    "t2p_lstm_cell_backward": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int32,
                                         C.c_int32, c_void, c_void, c_void, c_void]),
    "t2p_bn_train_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "t2p_bn_relu_train_forward": (C.c_int, [c_void, c_void, C.c_int32, C.c_int64, C.c_int32, c_void, c_void, C.c_float,
                                            C.c_int32, c_void, c_void, c_void, c_void, c_void, C.c_size_t, c_void]),
    "t2p_bn_relu_train_backward": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int32, C.c_int64, C.c_int32, c_void, c_void,
                                             c_void, C.c_int32, c_void, c_void, c_void, c_void, C.c_size_t, c_void]),
    "t2p_edge_features_forward": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_edge_features_backward": (C.c_int, [c_void, c_void, C.c_int64, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_pair_features_forward": (C.c_int, [c_void, c_void, c_void, C.c_int64, C.c_int32, c_void, c_void]),
    "t2p_pair_features_backward": (C.c_int, [c_void, c_void, c_void, C.c_int64, C.c_int32, c_void, c_void]),
    "t2p_rownorm_backward": (C.c_int, [c_void, c_void, C.c_int64, C.c_int32, c_void, c_void]),
    "t2p_segment_mean_forward": (C.c_int, [c_void, c_void, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_segment_mean_backward": (C.c_int, [c_void, c_void, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_segment_max_forward": (C.c_int, [c_void, c_void, C.c_int32, C.c_int32, c_void, c_void, c_void]),
    "t2p_segment_max_backward": (C.c_int, [c_void, c_void, c_void, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_hardest_ranking": (C.c_int, [c_void, C.c_int32, C.c_float, c_void, c_void, c_void, c_void]),
    "t2p_pairwise_ranking": (C.c_int, [c_void, C.c_int32, C.c_float, c_void, c_void, c_void, c_void]),
    "t2p_pack_objects": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, c_void, c_void, c_void, c_void,
                                   c_void]),
    "t2p_pack_scene_objects": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, c_void, c_void,
                                         c_void, c_void, c_void, c_void]),
    "t2p_match_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "t2p_match": (C.c_int, [c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(MatchWeights),
                            C.c_int32, C.c_float, c_void, c_void, c_void, c_void, c_void, c_void, c_void, C.c_size_t,
                            c_void]),
    "t2p_profile_enable": (None, [C.c_int]),
    "t2p_profile_report": (C.c_int, [C.c_char_p, C.c_size_t]),
    "t2p_profile_repeat": (None, [C.c_char_p, C.c_int32]),
    "t2p_sample_group": (C.c_int, [c_void, C.c_int64, C.c_int32, c_float_p, C.POINTER(c_void), C.POINTER(c_void),
                                   C.POINTER(c_void), c_void]),
    "t2p_group_rows": (C.c_int, [c_void, C.c_int64, C.c_int32, c_float_p, C.c_int32, C.POINTER(c_void), C.POINTER(c_void),
                                 C.POINTER(c_void), c_void]),
    "t2p_edge_counts": (C.c_int, [c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_edge_expand": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_void, c_void, c_void]),
    "t2p_dedup_rows": (C.c_int, [c_void, c_void, C.c_int64, C.c_int32, c_void, c_void, c_void]),
    "t2p_knn": (C.c_int, [c_void, C.c_int32, c_void, C.c_int32, C.c_int32, C.c_int32, c_void, c_void]),
    "t2p_gemm": (C.c_int, [c_void, C.c_int32, c_void, c_void, c_void, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                           C.c_int32, C.c_int32, c_void]),
    "t2p_rownorm": (C.c_int, [c_void, C.c_int64, C.c_int32, c_void, c_void]),
    "t2p_linear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "t2p_linear_wgrad_f32": (C.c_int, [c_void, C.c_int32, c_void, C.c_int32, c_void, C.c_int32, c_void, C.c_int64, C.c_int32,
                                       C.c_int32, c_void, C.c_size_t, c_void]),
    "t2p_gemm_tn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "t2p_gemm_tn": (C.c_int, [c_void, C.c_int32, c_void, C.c_int32, c_void, C.c_int32, C.c_int64, C.c_int32, C.c_int32, c_void,
                              C.c_size_t, c_void]),
}

_lib = None


class T2PError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libt2p_hip.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise T2PError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python __graft_entry__.py build, or python text2pos-cvpr2022_amd/build.py)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the export is missing
            fn.restype, fn.argtypes = res, args
        if handle.t2p_abi_version() != ABI_VERSION:
            raise T2PError(f"libt2p_hip.so ABI {handle.t2p_abi_version()} != binding ABI {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().t2p_last_error().decode("utf-8", "replace")
        raise T2PError(f"{what} failed (rc={rc}): {msg}")
