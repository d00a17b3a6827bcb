"""ctypes binding of libjmac_hip.so (include/jmac_hip.h).

The library is the product: there is no CPU or PyTorch fallback anywhere in ``jmac_amd``.  If the
shared object is missing, ``lib()`` raises; if a tensor is not on a HIP device, the wrappers raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import weakref
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JMAC_LIB_PATH") or os.path.join(_HERE, "libjmac_hip.so")   # env: A/B builds of the kernels
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jmac_hip.h")

_lib: Optional[C.CDLL] = None

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class JmacError(RuntimeError):
    pass


class View(C.Structure):
    """jmac_view_t"""
    _fields_ = [("ptr", vp), ("order", vp), ("items", vp), ("splits", vp), ("counts", vp),
                ("n_items_max", i64), ("n_splits_max", i64), ("n_parts_max", i64),
                ("item_edges", vp), ("n_empty", i64), ("n_coop", i64), ("entry_dst", vp)]


class LinkLayer(C.Structure):
    """jmac_link_layer_t"""
    _fields_ = [("ent", vp), ("ld_ent", i64), ("rel", vp), ("ld_rel", i64), ("table", vp), ("ld_table", i64)]


class AggFwdJob(C.Structure):
    """jmac_agg_fwd_job_t"""
    _fields_ = [("P", vp), ("ldp", i64), ("QZ", vp), ("ldqz", i64), ("RR", vp), ("ldrr", i64), ("a_att", vp), ("col", vp), ("etype", vp),
                ("by_dst", C.POINTER(View)), ("N", i64), ("d", i64), ("slope", f32), ("loop_rel", i32), ("self_off", i64),
                ("out_scale", f32), ("out", vp), ("ldo", i64), ("seg_max", vp), ("seg_den", vp), ("ws", vp), ("ws_bytes", sz)]


class AdamTask(C.Structure):
    """jmac_adam_task_t"""
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("vec4", i32)]


class GemmTask(C.Structure):
    """jmac_gemm_task_t"""
    _fields_ = [("A", vp), ("A2", vp), ("lda", i64), ("a_split", i64), ("transA", i32), ("transB", i32),
                ("B", vp), ("ldb", i64), ("C", vp), ("C2", vp), ("ldc", i64), ("c_split", i64),
                ("M", i64), ("N", i64), ("K", i64), ("act_src", vp), ("ld_act_src", i64),
                ("act", i32), ("accumulate", i32), ("slope", f32), ("pad_", i32)]


# name -> (restype, argtypes); mirrors include/jmac_hip.h one to one
_SIGS = {
    "jmac_strerror": (C.c_char_p, [C.c_int]),
    "jmac_version": (C.c_int, []),
    "jmac_graph_workspace_bytes": (sz, [i64, i64]),
    "jmac_index_check": (C.c_int, [vp, i32, i64, i64, i64, vp, vp]),
    "jmac_csr_build": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
    "jmac_group_build": (C.c_int, [vp, i64, i64, vp, vp, vp, sz, vp]),
    "jmac_items_max": (i64, [i64, i64, i32, i32]),
    "jmac_splits_max": (i64, [i64, i32, i32]),
    "jmac_parts_max": (i64, [i64, i32, i32]),
    "jmac_items_build": (C.c_int, [vp, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "jmac_rel_attn_fwd_workspace_bytes": (sz, [i64, i64]),
    "jmac_item_edges_build": (C.c_int, [vp, vp, i64, vp, vp, vp, vp]),
    "jmac_rel_attn_aggregate_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, C.POINTER(View),
                                                  i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, sz, vp]),
    "jmac_rel_attn_aggregate_fwd_bf16": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, C.POINTER(View),
                                                   i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, sz, vp]),
    "jmac_rel_attn_aggregate_fwd_bf16_padded": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, vp, vp, vp, C.POINTER(View),
                                                          i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, sz, vp]),
    "jmac_rel_attn_aggregate_fwd_jobs_f32": (C.c_int, [C.POINTER(AggFwdJob), i32, vp]),
    "jmac_softmax_parts_merge_f32": (C.c_int, [vp, i64, vp, vp, vp, i32, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, vp]),
    "jmac_rel_attn_bwd_workspace_bytes": (sz, [i64, i64, i64, i64, i64, i64, i64, i32]),
    "jmac_rel_attn_aggregate_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp,
                                                  C.POINTER(View), C.POINTER(View), C.POINTER(View),
                                                  i64, i64, i64, i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, i64,
                                                  vp, i64, vp, i64, vp, i64, vp, i32, vp, sz, vp]),
    "jmac_rel_attn_aggregate_bwd_phases_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp,
                                                         C.POINTER(View), C.POINTER(View), C.POINTER(View),
                                                         i64, i64, i64, i64, i64, f32, i32, i64, f32, vp, i64, vp, vp, vp, i64,
                                                         vp, i64, vp, i64, vp, i64, vp, i32, vp, sz, vp]),
    "jmac_bn_tanh_workspace_bytes": (sz, [i64, i64]),
    "jmac_bn_tanh_fwd_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, i32, f32, f32, vp, i64, vp, vp,
                                       vp, sz, vp]),
    "jmac_bn_tanh_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, i32, vp, i64, vp, vp,
                                       vp, sz, vp]),
    "jmac_bn_tanh_fwd2_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, i32, f32, f32, vp, i64, vp, i64, vp, vp,
                                        vp, sz, vp]),
    "jmac_bn_tanh_bwd2_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, i32, vp, i64, vp, vp,
                                        vp, sz, vp]),
    "jmac_bn_tanh_seg_workspace_bytes": (sz, [i32, i64]),
    "jmac_bn_tanh_seg_fwd2_f32": (C.c_int, [vp, i64, i64, i32, C.POINTER(i64), C.POINTER(i32), vp, vp, vp, vp, f32, f32, vp, i64,
                                            vp, i64, vp, vp, vp, sz, vp]),
    "jmac_bn_tanh_seg_bwd2_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, C.POINTER(i64), vp, vp, vp, vp, i64,
                                            vp, vp, vp, sz, vp]),
    "jmac_row_normalize_drop_fwd_f32": (C.c_int, [vp, i64, i64, i64, f32, vp, i64, f32, vp, i64, vp, vp]),
    "jmac_row_normalize_drop_bwd_f32": (C.c_int, [vp, i64, vp, vp, i64, f32, vp, i64, i64, i64, f32, vp, i64, i32, vp]),
    "jmac_row_normalize_dropseed_fwd_f32": (C.c_int, [vp, i64, i64, i64, f32, vp, f32, vp, i64, vp, vp]),
    "jmac_row_normalize_dropseed_bwd_f32": (C.c_int, [vp, i64, vp, vp, f32, vp, i64, i64, i64, f32, vp, i64, i32, vp]),
    "jmac_gemm_grouped_f32": (C.c_int, [C.POINTER(GemmTask), i32, vp]),
    "jmac_wcat_pack_f32": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, vp, i64, C.POINTER(vp), i32, vp]),
    "jmac_wcat_pack_seed_f32": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, vp, i64, C.POINTER(vp), i32, vp, vp, vp]),
    "jmac_adam_step_f32": (C.c_int, [C.POINTER(AdamTask), i32, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i32, i32,
                                     vp]),
    "jmac_wcat_unpack_f32": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, vp, i64, vp]),
    "jmac_rows_compact_f32": (C.c_int, [C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), i32, vp, i64, i64, vp]),
    "jmac_rows_expand_f32": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(i32), i32, vp, i64, i64, vp]),
    "jmac_col_moments_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, sz, vp]),
    "jmac_bn_tanh_apply_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, vp]),
    "jmac_bn_tanh_bwd_sums_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, sz, vp]),
    "jmac_bn_tanh_bwd_apply_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    "jmac_row_normalize_fwd_f32": (C.c_int, [vp, i64, i64, i64, f32, vp, i64, vp, vp]),
    "jmac_row_normalize_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, f32, vp, i64, vp]),
    "jmac_l1_score_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, vp]),
    "jmac_l1_score_bf16": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, vp]),
    "jmac_filtered_rank_f32": (C.c_int, [vp, i64, vp, vp, vp, i64, i64, i32, vp, vp]),
    "jmac_linkpred_rank_workspace_bytes": (sz, [i64, i64, i32]),
    "jmac_linkpred_rank_f32": (C.c_int, [vp, i32, vp, vp, i32, vp, vp, vp, i64, i64, i64, vp, vp, sz, vp]),
    "jmac_linkpred_rank_bf16": (C.c_int, [vp, i32, vp, vp, i32, vp, vp, vp, i64, i64, i64, vp, vp, sz, vp]),
    "jmac_sim_matrix_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, vp]),
    "jmac_sim_topk_workspace_bytes": (sz, [i64, i64, i32]),
    "jmac_sim_topk_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp]),
    "jmac_col_topk_workspace_bytes": (sz, [i64, i64, i32]),
    "jmac_col_topk_f32": (C.c_int, [vp, i64, i64, i64, i32, vp, vp, sz, vp]),
    "jmac_row_topk_f32": (C.c_int, [vp, i64, i64, i64, i32, vp, vp, vp]),
    "jmac_softmax_entropy_workspace_bytes": (sz, [i64, i64]),
    "jmac_softmax_entropy_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, f32, vp, vp, vp, sz, vp]),
    "jmac_masked_row_softmax_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, f32, f32, vp, i64, vp]),
    "jmac_row_softmax_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, f32, f32, vp, i64, vp, vp]),
    "jmac_col_softmax_workspace_bytes": (sz, [i64, i64]),
    "jmac_col_softmax_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, f32, f32, vp, i64, vp, vp, sz, vp]),
    "jmac_csls_rank_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp]),
    "jmac_csls_apply_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, i64, vp]),
    "jmac_gemm_f32": (C.c_int, [vp, i64, i32, vp, i64, i32, i64, i64, i64, vp, i64, vp]),
    "jmac_triple_l1_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp]),
    "jmac_triple_l1_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, i64, vp, i64, vp]),
    "jmac_triple_l1_margin_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    "jmac_triple_l1_margin_bwd_exact_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, i64, i64, vp, i64, i64, vp]),
    "jmac_pair_cosine_fwd_stats_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, vp]),
    "jmac_pair_cosine_bwd_sorted_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    "jmac_pair_cosine_fwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp]),
    "jmac_pair_cosine_bwd_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, i64, vp, i64, vp]),
    "jmac_pair_cosine_bwd_rows_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, vp, vp, f32, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp]),
    "jmac_triple_l1_margin_bwd_exact2_f32": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, vp, i64, i64, vp, vp,
                                                       vp, i64, i32, vp, i64, i32, vp]),
    "jmac_vec_mean_acc_f32": (C.c_int, [vp, i64, vp, vp, vp]),
    "jmac_margin_loss_fwd_acc_f32": (C.c_int, [vp, i64, i64, vp, vp, vp, vp]),
    "jmac_margin_loss_fwd_f32": (C.c_int, [vp, i64, i64, vp, vp, vp]),
    "jmac_margin_loss_bwd_f32": (C.c_int, [vp, i64, i64, vp, vp, vp, vp]),
    "jmac_scatter_sum_f32": (C.c_int, [vp, vp, i64, i64, i64, vp, vp]),
    "jmac_scatter_softmax_workspace_bytes": (sz, [i64, i64]),
    "jmac_scatter_softmax_f32": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, sz, vp]),
}


# entry points of libjmac_hip_testing.so only (include/jmac_hip_testing.h)
_TESTING_SIGS = {
    "jmac_gemm_nt_x3_f32": (C.c_int, [vp, i64, vp, i64, i64, i64, i64, vp, i64, vp]),
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
        # load order: torch bundles a HIP runtime with the same SONAME (libamdhip64.so.7) as /opt/rocm's, so whichever is
        # mapped first serves both.  This host side hands torch's device pointers and streams to the library: they must
        # belong to ONE runtime, torch's (loading this library first gave "no ROCm-capable device" on the first launch).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_tlib: Optional[C.CDLL] = None


def testing_lib() -> C.CDLL:
    """libjmac_hip_testing.so: the same library with the ATOMIC aggregation backward compiled in (mode 0) -- an independent
    second implementation for the parity tests.  Nothing on the product path calls this."""
    global _tlib
    if _tlib is None:
        path = os.path.join(_HERE, "libjmac_hip_testing.so")
        if not os.path.exists(path):
            raise JmacError("jmac_amd: %s is missing (the atomic backward lives in the testing build only: "
                            "`make -C jmac_amd/csrc`)" % path)
        import torch  # noqa: F401  (same load order as lib())
        l = C.CDLL(path)
        for name in ("jmac_rel_attn_aggregate_bwd_f32", "jmac_rel_attn_bwd_workspace_bytes", "jmac_strerror"):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = _SIGS[name]
        for name, (res, args) in _TESTING_SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _tlib = l
    return _tlib


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


# ---- index validation (the reference raises IndexError on an out-of-range id; the kernels trust theirs) -----------
_CHECKED: "dict" = {}
_CHECKED_CAP = 64


def check_index_range(idx, n: int, what: str = "index"):
    """Raise IndexError unless every entry of ``idx`` lies in [0, n).  Host data (lists, numpy, CPU tensors) is
    checked on the host; a device tensor is checked once per (tensor object, version) by jmac_index_check -- one host read
    the first time a tensor is seen, nothing afterwards, and nothing while a stream is being captured.  Hot loops should
    pass tensors that were checked on the host and marked before the upload (``mark_index_range``): a fresh device tensor
    per step costs one blocking read per step here.  A kernel outside torch that rewrites a validated tensor in place does
    not bump ``_version``: re-validate such a tensor yourself (``jmac_index_check``)."""
    import numpy as np
    import torch
    n = int(n)
    if not isinstance(idx, torch.Tensor):
        a = np.asarray(idx)
        if a.size and (a.min() < 0 or a.max() >= n):
            raise IndexError("%s out of range: values in [%s, %s], valid range [0, %d)" % (what, a.min(), a.max(), n))
        return
    if idx.numel() == 0:
        return
    if not idx.is_cuda:
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n:
            raise IndexError("%s out of range: values in [%d, %d], valid range [0, %d)" % (what, lo, hi, n))
        return
    if idx.dtype not in (torch.int32, torch.int64):
        raise TypeError("%s must be int32 or int64 (got %s)" % (what, idx.dtype))
    ok = getattr(idx, "_jmac_range_ok", None)             # set by a host-side check before the upload (mark_index_range)
    if ok is not None and ok <= n:
        return
    # validated tensors are remembered by IDENTITY through a weak reference (no strong reference: a [2,E] COO of a dropped
    # graph is not pinned by this cache), together with the version counter torch bumps on in-place writes
    key = id(idx)
    hit = _CHECKED.get(key)
    if hit is not None and hit[0]() is idx and hit[1] == idx._version and hit[2] >= 0 and hit[2] <= n:
        return
    if torch.cuda.is_current_stream_capturing():
        return                                             # cannot read back inside a capture; eager warm-up has checked
    t = idx if idx.is_contiguous() else idx.contiguous()
    bad = torch.zeros(1, dtype=torch.int32, device=idx.device)
    check(lib().jmac_index_check(ptr(t), t.element_size(), t.numel(), 0, n, ptr(bad), stream()), "jmac_index_check")
    nbad = int(bad.item())
    if nbad:
        raise IndexError("%s out of range: %d of %d entries outside [0, %d)" % (what, nbad, t.numel(), n))
    for k in [k for k, v in _CHECKED.items() if v[0]() is None]:
        del _CHECKED[k]                                    # tensors that have died since
    if len(_CHECKED) >= _CHECKED_CAP:
        _CHECKED.pop(next(iter(_CHECKED)))
    try:
        _CHECKED[key] = (weakref.ref(idx), idx._version, n)
    except TypeError:                                      # pragma: no cover  (not weak-referenceable: do not cache)
        pass


def mark_index_range(t, n: int):
    """Record on a (device) index tensor that its values are known to lie in [0, n) -- checked on the host before the
    upload -- so that the entry points do not check it again."""
    t._jmac_range_ok = int(n)
    return t


def stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
