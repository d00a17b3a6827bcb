"""ctypes binding of libjmac_hip.so (include/jmac_hip.h).

The library is the product: there is no CPU or PyTorch fallback anywhere in ``jmac_amd``.  If the
shared object is missing, ``lib()`` raises; if a tensor is not on a HIP device, the wrappers raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjmac_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jmac_hip.h")

_lib: Optional[C.CDLL] = None

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class JmacError(RuntimeError):
    pass


class View(C.Structure):
    """jmac_view_t"""
    _fields_ = [("ptr", vp), ("order", vp), ("items", vp), ("splits", vp), ("counts", vp),
                ("n_items_max", i64), ("n_splits_max", i64), ("n_parts_max", i64)]


# name -> (restype, argtypes); mirrors include/jmac_hip.h one to one
_SIGS = {
    "jmac_strerror": (C.c_char_p, [C.c_int]),
    "jmac_version": (C.c_int, []),
    "jmac_graph_workspace_bytes": (sz, [i64, i64]),
    "jmac_csr_build": (C.c_int, [vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
    "jmac_group_build": (C.c_int, [vp, i64, i64, vp, vp, vp, sz, vp]),
    "jmac_items_max": (i64, [i64, i64, i32]),
    "jmac_splits_max": (i64, [i64, i32]),
    "jmac_parts_max": (i64, [i64, i32]),
    "jmac_items_build": (C.c_int, [vp, i64, i32, vp, vp, vp, vp, sz, vp]),
    "jmac_rel_attn_fwd_workspace_bytes": (sz, [i64, i64]),
    "jmac_rel_attn_aggregate_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp,
                                                  i64, i64, i64, i64, i64, f32, i32, i64, f32, vp, i64, vp, vp,
                                                  vp, sz, vp]),
    "jmac_rel_attn_aggregate_fwd_bf16": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp,
                                                   i64, i64, i64, i64, i64, f32, i32, i64, f32, vp, i64, vp, vp,
                                                   vp, sz, vp]),
    "jmac_rel_attn_bwd_workspace_bytes": (sz, [i64, i64, i64, i64, i64, i64, i64, i32]),
    "jmac_rel_attn_aggregate_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp,
                                                  C.POINTER(View), C.POINTER(View), C.POINTER(View),
                                                  i64, i64, i64, i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, i64,
                                                  vp, i64, vp, i64, vp, i64, vp, i32, vp, sz, vp]),
    "jmac_bn_tanh_workspace_bytes": (sz, [i64, i64]),
    "jmac_bn_tanh_fwd_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, i32, f32, f32, vp, i64, vp, vp,
                                       vp, sz, vp]),
    "jmac_bn_tanh_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, i32, vp, i64, vp, vp,
                                       vp, sz, vp]),
    "jmac_col_moments_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, sz, vp]),
    "jmac_bn_tanh_apply_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, vp]),
    "jmac_bn_tanh_bwd_sums_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, sz, vp]),
    "jmac_bn_tanh_bwd_apply_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    "jmac_row_normalize_fwd_f32": (C.c_int, [vp, i64, i64, i64, f32, vp, i64, vp, vp]),
    "jmac_row_normalize_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, f32, vp, i64, vp]),
    "jmac_l1_score_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, vp]),
    "jmac_l1_score_bf16": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, vp]),
    "jmac_filtered_rank_f32": (C.c_int, [vp, i64, vp, vp, vp, i64, i64, i32, vp, vp]),
    "jmac_sim_matrix_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, vp]),
    "jmac_sim_topk_workspace_bytes": (sz, [i64, i64]),
    "jmac_sim_topk_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp]),
    "jmac_col_topk_workspace_bytes": (sz, [i64, i64, i32]),
    "jmac_col_topk_f32": (C.c_int, [vp, i64, i64, i64, i32, vp, vp, sz, vp]),
    "jmac_row_topk_f32": (C.c_int, [vp, i64, i64, i64, i32, vp, vp, vp]),
    "jmac_softmax_entropy_workspace_bytes": (sz, [i64, i64]),
    "jmac_softmax_entropy_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, f32, vp, vp, vp, sz, vp]),
    "jmac_masked_row_softmax_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, f32, f32, vp, i64, vp]),
    "jmac_csls_rank_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp]),
    "jmac_csls_apply_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, i64, vp]),
    "jmac_gemm_f32": (C.c_int, [vp, i64, i32, vp, i64, i32, i64, i64, i64, vp, i64, vp]),
    "jmac_triple_l1_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp]),
    "jmac_triple_l1_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, i64, vp, i64, vp]),
    "jmac_pair_cosine_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp]),
    "jmac_pair_cosine_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, i64, vp, i64, vp]),
    "jmac_scatter_sum_f32": (C.c_int, [vp, vp, i64, i64, i64, vp, vp]),
    "jmac_scatter_softmax_workspace_bytes": (sz, [i64, i64]),
    "jmac_scatter_softmax_f32": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, sz, vp]),
}


def header_symbols(path: str = HEADER_PATH):
    """Every function name declared in include/jmac_hip.h (used by the symbol-export test)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jmac_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JmacError(
                "jmac_amd: %s is missing. Build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(or `make -C jmac_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().jmac_strerror(rc)
        raise JmacError("%s failed: %s (rc=%d)" % (what or "jmac call", msg.decode() if msg else "?", rc))


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise JmacError("jmac_amd ops need tensors on a HIP device (got %s); there is no CPU path" % t.device)


def stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
