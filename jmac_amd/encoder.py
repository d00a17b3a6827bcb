"""JMAC.forward_name / forward_no_name as ONE autograd node each (rows a9, a2-a8 of SURVEY.md section 8).

The reference's encoder (src/jmac_model.py:172-220) is a chain of ~60 small torch ops around three
RelationAwareLayer calls.  Run op by op on the device, a DBP-5L-size training step spends four fifths of
its time outside the aggregation kernels: ~40 launch-bound products on the ~10^3-row relation tables,
cat copies, the zero-fills / copies / adds autograd generates for slices and fan-outs, LeakyReLU passes.
Here the same function is evaluated by one ``torch.autograd.Function`` with a hand-written backward:

* every product on the relation side -- the layers' two relation transforms and the hoisted R''[Wb|Wg]
  (src/jmac_model.py:39-42; evaluated as act(.) (W2 [Wb|Wg]): see ``_Chain``), the two relation MLPs (:195-196),
  the folded name projection -- goes out in dependency LEVELS through ``jmac_gemm_grouped_f32``: 4 launches forward,
  4 backward, activations and their derivatives fused into the products, ``cat(rel_emb, loop_rel)`` read in place
  from two buffers;
* the operands of the three concatenations (:180, :192, :203) are WRITTEN into their cat buffers by the
  kernels that produce them (normalise+dropout, BatchNorm+tanh with two destinations, the library GEMM
  with a strided output), and their gradients are read from the adjoint buffers in place (BatchNorm's
  backward sums its two incoming gradients while it reads them);
* N-row products stay library GEMMs (torch.mm on hipBLASLt / rocBLAS), with ``out=`` views and
  ``addmm_`` accumulation instead of separate adds; the constant name embeddings get no input gradient.

Same function and the same fp32 arithmetic as the op-by-op path in ``jmac_amd.model`` up to the association of the
relation chain's products (that path stays as the second implementation and serves every configuration this node does
not cover); tests/test_gpu_encoder.py holds the two to each other.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops
from ._lib import AggFwdJob, GemmTask, check, lib, ptr, require_device, stream
from .graph import RelGraph

ACT_NONE, ACT_LEAKY, ACT_RELU, DACT_LEAKY, DACT_RELU = 0, 1, 2, 3, 4
MAX_TASKS = 24
# tests set this to a dict to receive {layer name: (ent_emb, rel_emb)} -- the inputs each RelationAwareLayer call of the
# encoder sees (what a forward pre-hook on the layer modules shows on the op-by-op path)
CAPTURE = None


# ---- grouped small GEMM -------------------------------------------------------------------------------------------------
def gemm_task(A, B, Cout, *, ta=False, tb=False, A2=None, C2=None, act=ACT_NONE, act_src=None, slope=0.0, accumulate=False,
              defer=False):
    """One product ``Cout (+)= epilogue(op(A) op(B))`` for jmac_gemm_grouped_f32.  ``A2`` / ``C2``: the rows of A (its
    MEMORY rows: the K rows when ``ta``) / of the output that follow A's / Cout's own rows live in a second buffer."""
    for t in (A, B, Cout, A2, C2, act_src):
        if t is not None and (t.dtype != torch.float32 or t.stride(-1) != 1):
            raise TypeError("grouped GEMM operands are fp32 with unit inner stride")
    a_rows = A.shape[0] + (A2.shape[0] if A2 is not None else 0)
    M, K = (A.shape[1], a_rows) if ta else (a_rows, A.shape[1])
    Kb, N = (B.shape[1], B.shape[0]) if tb else (B.shape[0], B.shape[1])
    c_rows = Cout.shape[0] + (C2.shape[0] if C2 is not None else 0)
    if K != Kb or c_rows != M or Cout.shape[1] != N:
        raise ValueError("grouped GEMM shapes: op(A) [%d,%d] op(B) [%d,%d] -> C [%d,%d]" % (M, K, Kb, N, c_rows, Cout.shape[1]))
    if A2 is not None and (A2.shape[1] != A.shape[1] or A2.stride(0) != A.stride(0)) and A2.shape[0] > 1:
        raise ValueError("A2 must share A's width and leading dimension")
    t = GemmTask()
    t.A, t.A2, t.lda, t.a_split = ptr(A), ptr(A2), A.stride(0), (A.shape[0] if A2 is not None else 0)
    t.transA, t.transB = int(ta), int(tb)
    t.B, t.ldb = ptr(B), B.stride(0)
    t.C, t.C2, t.ldc, t.c_split = ptr(Cout), ptr(C2), Cout.stride(0), (Cout.shape[0] if C2 is not None else 0)
    t.M, t.N, t.K = M, N, K
    t.act_src, t.ld_act_src = ptr(act_src), (act_src.stride(0) if act_src is not None else 0)
    t.act, t.accumulate, t.slope = int(act), int(bool(accumulate)), float(slope)
    t._keep = (A, B, Cout, A2, C2, act_src)          # the tensors outlive the (asynchronous) launch
    t._defer = bool(defer)                           # nothing later in the same batch of levels reads the output
    t._cost = ((M + 31) // 32) * ((N + 31) // 32) * ((K + 303) // 304)      # (tile, K chunk) units of the kernel
    return t


def grouped_gemm(tasks: Sequence[GemmTask]) -> None:
    """All ``tasks`` in one launch (they must not depend on each other's outputs)."""
    tasks = [t for t in tasks if t is not None]
    for i in range(0, len(tasks), MAX_TASKS):
        chunk = tasks[i:i + MAX_TASKS]
        arr = (GemmTask * len(chunk))(*chunk)
        check(lib().jmac_gemm_grouped_f32(arr, len(chunk), stream()), "jmac_gemm_grouped_f32")


def balance_levels(levels: List[List[GemmTask]]) -> List[List[GemmTask]]:
    """Move deferrable products (weight gradients: long-K products that nothing in the batch reads) from the crowded
    early levels to later, lighter ones.  A launch costs about as much as its rounds of resident blocks, and the levels of a
    backward are lopsided: the first carries ten products, the last two carry two each.  Order of execution per output is
    unchanged (a task only ever moves later; accumulating tasks keep their single predecessor), so results are too."""
    levels = [list(lv) for lv in levels]
    while levels and not levels[-1]:                     # an empty trailing level is not a launch to fill
        levels.pop()
    total = sum(t._cost for lv in levels for t in lv)
    if not total or len(levels) < 2:
        return levels
    target = total / len(levels)
    for i in range(len(levels) - 1):
        cost = sum(t._cost for t in levels[i])
        movable = sorted((t for t in levels[i] if t._defer), key=lambda t: -t._cost)
        for t in movable:
            if cost <= target:
                break
            levels[i].remove(t)
            levels[i + 1].append(t)
            cost -= t._cost
    return levels


def run_levels(levels: List[List[GemmTask]], balance: bool = False) -> None:
    for lv in (balance_levels(levels) if balance else levels):
        if lv:
            grouped_gemm(lv)


# ---- raw kernel wrappers (no autograd: the node below owns the backward) --------------------------------------------------
def _empty(dev, *shape):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _agg_fwd(PQZ, RR, a, graph: RelGraph, slope, out_scale=0.5, compact=False):
    """jmac_rel_attn_aggregate_fwd_{f32,bf16,bf16_padded} on the [P|Q|Z] table (fp32, or bf16 for the inference form: sums,
    softmax and the output stay fp32; bf16 tables may carry padded halves, ops.bf16_pad); the self loop is the last relation
    row."""
    L = lib()
    N, d3 = PQZ.shape
    dh, d = d3 // 3, int(a.numel())
    dev = PQZ.device
    bf16 = PQZ.dtype == torch.bfloat16
    out, smax, sden = _empty(dev, N, d), _empty(dev, max(N, 1)), _empty(dev, max(N, 1))
    s = graph.by_dst
    wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    etype = graph.etype_c if compact else graph.etype         # compact: RR holds the rows of graph.rel_used + the loop row
    sview = s.view_compact(graph.col, graph.etype_c) if compact else s.view()
    ev0 = ops._ev() if ops.PROFILE is not None else None
    if dh != d:
        if not bf16 or RR.shape[1] != 2 * dh:
            raise ValueError("padded table halves exist for bf16 tables only")
        check(L.jmac_rel_attn_aggregate_fwd_bf16_padded(
            ptr(PQZ), d3, PQZ.data_ptr() + dh * PQZ.element_size(), d3, ptr(RR), RR.stride(0), dh, ptr(a), ptr(graph.col),
            ptr(etype), C.byref(sview), N, d, float(slope), RR.shape[0] - 1, 0, float(out_scale), ptr(out), d, ptr(smax),
            ptr(sden), ptr(ws), wsb, stream()), "jmac_rel_attn_aggregate_fwd_bf16_padded")
    else:
        fwd = L.jmac_rel_attn_aggregate_fwd_bf16 if bf16 else L.jmac_rel_attn_aggregate_fwd_f32
        check(fwd(
            ptr(PQZ), d3, PQZ.data_ptr() + d * PQZ.element_size(), d3, ptr(RR), RR.stride(0), ptr(a), ptr(graph.col), ptr(etype),
            C.byref(sview), N, d, float(slope), RR.shape[0] - 1, 0, float(out_scale), ptr(out), d, ptr(smax), ptr(sden),
            ptr(ws), wsb, stream()), "jmac_rel_attn_aggregate_fwd_%s" % ("bf16" if bf16 else "f32"))
    if ev0 is not None:
        ops.PROFILE.append(("rel_attn_fwd_bf16" if bf16 else "rel_attn_fwd", ev0, ops._ev()))
    return out, smax, sden


PAIR_LAUNCHES = True      # tests / A-B: False runs the two independent first layers' aggregations as two launches each


def _agg_fwd_pair(specs, graph: RelGraph, out_scale=0.5, compact=False):
    """jmac_rel_attn_aggregate_fwd_jobs_f32: the forward aggregation of TWO independent layers on the same graph (fp32 tables
    [P|Q|Z], relation tables, attention vectors, slopes = ``specs``) as one launch -> [(out, seg_max, seg_den)] * 2."""
    L = lib()
    s = graph.by_dst
    etype = graph.etype_c if compact else graph.etype
    sview = s.view_compact(graph.col, graph.etype_c) if compact else s.view()
    jobs = (AggFwdJob * len(specs))()
    res, keep = [], []
    for k, (PQZ, RR, a, slope) in enumerate(specs):
        N, d3 = PQZ.shape
        d = d3 // 3
        dev = PQZ.device
        out, smax, sden = _empty(dev, N, d), _empty(dev, max(N, 1)), _empty(dev, max(N, 1))
        wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        j = jobs[k]
        j.P, j.ldp, j.QZ, j.ldqz, j.RR, j.ldrr, j.a_att = ptr(PQZ), d3, PQZ.data_ptr() + d * 4, d3, ptr(RR), RR.stride(0), ptr(a)
        j.col, j.etype, j.by_dst, j.N, j.d, j.slope = ptr(graph.col), ptr(etype), C.pointer(sview), N, d, float(slope)
        j.loop_rel, j.self_off, j.out_scale = RR.shape[0] - 1, 0, float(out_scale)
        j.out, j.ldo, j.seg_max, j.seg_den, j.ws, j.ws_bytes = ptr(out), d, ptr(smax), ptr(sden), ptr(ws), wsb
        res.append((out, smax, sden))
        keep.append(ws)
    ev0 = ops._ev() if ops.PROFILE is not None else None
    check(L.jmac_rel_attn_aggregate_fwd_jobs_f32(jobs, len(specs), stream()), "jmac_rel_attn_aggregate_fwd_jobs_f32")
    if ev0 is not None:
        ops.PROFILE.append(("rel_attn_fwd_pair", ev0, ops._ev()))
    return res


def _agg_bwd(PQZ, RR, a, graph: RelGraph, slope, out, smax, sden, G, out_scale=0.5, compact=False):
    """Deterministic backward (three launches): dPQZ [N,3d], dRR [nrel,2d], da [d]."""
    L = lib()
    N, d3 = PQZ.shape
    d = d3 // 3
    dev = PQZ.device
    nrel = RR.shape[0]
    if compact:
        graph.ensure_backward_views_compact()
    else:
        graph.ensure_backward_views()
    by_rel, etype = (graph.by_rel_c, graph.etype_c) if compact else (graph.by_rel, graph.etype)
    dPQZ, dRR, da = _empty(dev, N, d3), _empty(dev, nrel, 2 * d), _empty(dev, d)
    vd = graph.by_dst_bwd.view_compact(graph.col, graph.etype_c) if compact else graph.by_dst_bwd.view()
    vs, vr = graph.by_src.view(), by_rel.view()
    wsb = int(L.jmac_rel_attn_bwd_workspace_bytes(N, graph.E, nrel, d, graph.by_dst_bwd.n_parts_max, graph.by_src.n_parts_max,
                                                  by_rel.n_parts_max, 1))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ev0 = ops._ev() if ops.PROFILE is not None else None
    check(L.jmac_rel_attn_aggregate_bwd_f32(
        ptr(PQZ), d3, PQZ.data_ptr() + d * 4, d3, ptr(RR), RR.stride(0), ptr(a), ptr(graph.col), ptr(etype),
        ptr(graph.dst_of_slot), C.byref(vd), C.byref(vs), C.byref(vr), N, N, graph.E, nrel, d, float(slope), nrel - 1, 0,
        float(out_scale), ptr(out), d, ptr(smax), ptr(sden), ptr(G), G.stride(0), ptr(dPQZ), d3, dPQZ.data_ptr() + d * 4, d3,
        ptr(dRR), 2 * d, ptr(da), 1, ptr(ws), wsb, stream()), "jmac_rel_attn_aggregate_bwd_f32")
    if ev0 is not None:
        ops.PROFILE.append(("rel_attn_bwd", ev0, ops._ev()))
    return dPQZ, dRR, da


class RowBlocks:
    """Row blocks of a stacked launch set: the KGs that share one pass through the layer kernels (JMAC.forward_stacked).
    ``sizes``: rows per block in stack order; ``order``: stack position of the block the reference's loop would have
    encoded k-th (the order in which BatchNorm's running estimates see the blocks' batch statistics)."""

    def __init__(self, sizes, order=None):
        self.sizes = tuple(int(n) for n in sizes)
        self.nb = len(self.sizes)
        if not 1 <= self.nb <= 16 or min(self.sizes) <= 0:
            raise ValueError("RowBlocks: 1..16 non-empty blocks")
        off = [0]
        for n in self.sizes:
            off.append(off[-1] + n)
        self.offsets = tuple(off)
        self.order = tuple(int(o) for o in (order if order is not None else range(self.nb)))
        if sorted(self.order) != list(range(self.nb)):
            raise ValueError("RowBlocks: order must be a permutation of the blocks")
        self.c_ptr = (C.c_int64 * (self.nb + 1))(*off)
        self.c_order = (C.c_int32 * self.nb)(*self.order)


def _bn_fwd(x, bn, training, y, y2=None, seg=None):
    """tanh(BatchNorm1d(x)) into y (and y2), nn.BatchNorm1d bookkeeping included (src/jmac_model.py:52).  ``seg`` (RowBlocks,
    more than one block, batch statistics): every block of rows is normalised with ITS statistics and the running estimates
    move once per block, as one forward_base call per KG leaves them (src/jmac_model.py:325-326)."""
    L = lib()
    N, d = x.shape
    dev = x.device
    use_batch = bool(training or not bn.track_running_stats)
    nb = seg.nb if (seg is not None and use_batch) else 1
    if nb > 1 and seg.offsets[-1] != N:
        raise ValueError("RowBlocks cover %d rows, the layer has %d" % (seg.offsets[-1], N))
    if training and bn.track_running_stats:
        if _TRACKERS is not None:
            _TRACKERS.append(bn.num_batches_tracked)           # the encoder node bumps its layers' counters in ONE launch
        else:
            bn.num_batches_tracked.add_(nb)
    if nb > 1:
        mean, invstd = _empty(dev, nb, d), _empty(dev, nb, d)
        wsb = int(L.jmac_bn_tanh_seg_workspace_bytes(nb, d))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        check(L.jmac_bn_tanh_seg_fwd2_f32(ptr(x), x.stride(0), d, nb, seg.c_ptr, seg.c_order, ptr(bn.weight), ptr(bn.bias),
                                          ptr(bn.running_mean), ptr(bn.running_var), float(bn.momentum), float(bn.eps), ptr(y),
                                          y.stride(0), ptr(y2), y2.stride(0) if y2 is not None else 0, ptr(mean), ptr(invstd),
                                          ptr(ws), wsb, stream()), "jmac_bn_tanh_seg_fwd2_f32")
        return mean, invstd, use_batch
    mean, invstd = _empty(dev, d), _empty(dev, d)
    wsb = int(L.jmac_bn_tanh_workspace_bytes(N, d))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(L.jmac_bn_tanh_fwd2_f32(ptr(x), x.stride(0), N, d, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                  ptr(bn.running_var), 1 if use_batch else 0, float(bn.momentum), float(bn.eps), ptr(y),
                                  y.stride(0), ptr(y2), y2.stride(0) if y2 is not None else 0, ptr(mean), ptr(invstd), ptr(ws),
                                  wsb, stream()), "jmac_bn_tanh_fwd2_f32")
    return mean, invstd, use_batch


def _bn_bwd(x, y, gy, gy2, weight, mean, invstd, use_batch, seg=None):
    L = lib()
    N, d = x.shape
    dev = x.device
    gx, gbw = _empty(dev, N, d), _empty(dev, 2 * d)            # gbw = [grad bias | grad weight]
    if mean.dim() == 2:                                        # per-block statistics (segmented forward)
        nb = seg.nb
        wsb = int(L.jmac_bn_tanh_seg_workspace_bytes(nb, d))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        check(L.jmac_bn_tanh_seg_bwd2_f32(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(gy), gy.stride(0), ptr(gy2),
                                          gy2.stride(0) if gy2 is not None else 0, d, nb, seg.c_ptr, ptr(weight), ptr(mean),
                                          ptr(invstd), ptr(gx), d, gbw.data_ptr() + d * 4, ptr(gbw), ptr(ws), wsb, stream()),
              "jmac_bn_tanh_seg_bwd2_f32")
        return gx, gbw
    wsb = int(L.jmac_bn_tanh_workspace_bytes(N, d))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(L.jmac_bn_tanh_bwd2_f32(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(gy), gy.stride(0), ptr(gy2),
                                  gy2.stride(0) if gy2 is not None else 0, N, d, ptr(weight), ptr(mean), ptr(invstd),
                                  1 if use_batch else 0, ptr(gx), d, gbw.data_ptr() + d * 4, ptr(gbw), ptr(ws), wsb, stream()),
          "jmac_bn_tanh_bwd2_f32")
    return gx, gbw


def _norm_drop_fwd(x, p_drop, training, y, mask=None, seed=None):
    """completion_dropout(F.normalize(x)) into y (src/jmac_model.py:179,191): (inv, drop).  ``drop`` is what the backward
    needs to repeat the draws: None (no dropout), ("mask", tensor, scale) for a pre-drawn {0,1} tensor, or ("seed", tensor,
    p_drop) -- ``seed``: a one-element device int64; the kernels draw from it themselves, no mask tensor exists."""
    N, d = x.shape
    inv = _empty(x.device, max(N, 1))
    if not (training and p_drop > 0.0):
        mask = seed = None
    if seed is not None and mask is None:
        check(lib().jmac_row_normalize_dropseed_fwd_f32(ptr(x), x.stride(0), N, d, 1e-12, ptr(seed), float(p_drop), ptr(y),
                                                        y.stride(0), ptr(inv), stream()), "jmac_row_normalize_dropseed_fwd_f32")
        return inv, ("seed", seed, float(p_drop))
    scale = 1.0
    if training and p_drop > 0.0:
        if mask is None:
            mask = torch.empty((N, d), dtype=torch.float32, device=x.device).bernoulli_(1.0 - p_drop)
        scale = 1.0 / (1.0 - p_drop)
    check(lib().jmac_row_normalize_drop_fwd_f32(ptr(x), x.stride(0), N, d, 1e-12, ptr(mask), d, scale, ptr(y), y.stride(0),
                                                ptr(inv), stream()), "jmac_row_normalize_drop_fwd_f32")
    return inv, (("mask", mask, scale) if mask is not None else None)


def _norm_drop_bwd(x, inv, drop, g, gx, accumulate):
    N, d = x.shape
    if drop is not None and drop[0] == "seed":
        check(lib().jmac_row_normalize_dropseed_bwd_f32(ptr(x), x.stride(0), ptr(inv), ptr(drop[1]), drop[2], ptr(g), g.stride(0), N,
                                                        d, 1e-12, ptr(gx), gx.stride(0), 1 if accumulate else 0, stream()),
              "jmac_row_normalize_dropseed_bwd_f32")
        return
    mask, scale = (drop[1], drop[2]) if drop is not None else (None, 1.0)
    check(lib().jmac_row_normalize_drop_bwd_f32(ptr(x), x.stride(0), ptr(inv), ptr(mask), d, scale, ptr(g), g.stride(0), N, d,
                                                1e-12, ptr(gx), gx.stride(0), 1 if accumulate else 0, stream()),
          "jmac_row_normalize_drop_bwd_f32")


_TRACKERS = None      # list while an encoder node's forward runs: the num_batches_tracked buffers to bump at its end

# The encoder nodes hand their table inputs (comp_att, rel_comp) back as OUTPUTS (aliases: layer 0 of the completion layers, src/
# jmac_model.py:176,187), so that what the losses differentiate with respect to layer 0 arrives at the node's backward instead of
# at the leaf -- and the node adds its own input gradient onto that buffer in the kernels that produce it (addmm_ / accumulate
# epilogues) instead of autograd adding two [N, d] tensors afterwards.  Only a buffer one of this library's loss ops has just
# allocated and marked (jmac_amd.losses: ``_jmac_fresh_grad``) is written in place; any other incoming gradient (a sum the engine
# formed, a user's tensor) is left alone and returned beside the node's own, as before.
# Restriction: the buffer taken over is the gradient of the node's layer-0 OUTPUT.  Whoever else holds that very tensor
# -- ``torch.autograd.grad(loss, [comp0])``, ``comp0.retain_grad()``, a tensor hook on it -- sees loss gradient + the node's input
# gradient after this backward ran.  ``loss.backward()`` (the training loop, train.py:358) has no such observer; a caller that
# inspects the layer-0 output's gradient sets ``encoder.INPLACE_GRADS = False`` (the sum is then formed out of place).  A backward
# that is itself recorded (create_graph=True) never takes a buffer over.
INPLACE_GRADS = True
INPLACE_COUNT = 0         # tests: how many incoming gradient buffers a backward took over


def _take_grad(g, shape):
    """``g`` if this backward may accumulate onto it in place, else None."""
    global INPLACE_COUNT
    if (INPLACE_GRADS and not torch.is_grad_enabled() and g is not None and getattr(g, "_jmac_fresh_grad", False)
            and g.dtype == torch.float32 and g.is_contiguous()
            and tuple(g.shape) == tuple(shape)):
        g._jmac_fresh_grad = False
        INPLACE_COUNT += 1
        return g
    return None


def _plus(own, extra):
    """own + extra for a gradient the node could not take over (None-aware)."""
    if extra is None:
        return own
    return extra if own is None else own + extra


def _bump_trackers(trackers) -> None:
    if trackers:
        torch._foreach_add_(trackers, 1)


# ---- used-relation compaction ------------------------------------------------------------------------------------------
class _RelCompact:
    """The relation side of a node on the rows the graph's edges name (graph.rel_used + the loop row) instead of all nr rows.

    A DBP-5L KG touches 153-833 of its 961 relation rows (ja: 158), and a layer's relation transform and projection
    (src/jmac_model.py:39-42, the hoisted R''[Wb|Wg]) reach the output through the edges' gathers only -- rows no edge names
    contribute nothing forward and receive a zero gradient.  ``gather`` cuts the used rows of the relation tables out in one
    launch, the chains run on them (six times fewer rows on ja), ``scatter`` puts their gradients back into full tables
    (zeros elsewhere, or added onto a full-table gradient another branch wrote).  Off (tables pass through) when every row is
    used or none is."""

    def __init__(self, graph: RelGraph, nr: int):
        graph.ensure_rel_compact()
        self.graph, self.nr = graph, int(nr)
        self.on = 0 < graph.n_used < nr and graph.num_rel == nr + 1 and COMPACT_RELATIONS
        self.n = graph.n_used if self.on else self.nr

    def gather(self, tables):
        if not self.on:
            return list(tables)
        g, d = self.graph, tables[0].shape[1]
        tables = [t_ if t_.stride(-1) == 1 else t_.contiguous() for t_ in tables]
        out = [_empty(t_.device, g.n_used, d) for t_ in tables]
        ld = (C.c_int64 * len(tables))(*[t_.stride(0) for t_ in tables])
        check(lib().jmac_rows_compact_f32(_vp_array(tables), ld, _vp_array(out), len(tables), ptr(g.rel_used), g.n_used, d, stream()),
              "jmac_rows_compact_f32")
        return out

    def scatter(self, items):
        """items: [(compact gradient [n_used, d], full buffer or None, accumulate)] -> the full [nr, d] gradients (one launch)."""
        if not self.on:
            return [c for c, _, _ in items]
        g, d = self.graph, items[0][0].shape[1]
        full = [(f if f is not None else _empty(c.device, self.nr, d)) for c, f, _ in items]
        ld = (C.c_int64 * len(items))(*[f.stride(0) for f in full])
        acc = (C.c_int32 * len(items))(*[1 if (a_ and f0 is not None) else 0 for (_, f0, a_) in items])
        check(lib().jmac_rows_expand_f32(_vp_array([c for c, _, _ in items]), _vp_array(full), ld, acc, len(items), ptr(g.rel_pos),
                                         self.nr, d, stream()), "jmac_rows_expand_f32")
        return full

    def full_rows(self, compact_rows, fill=0.0):
        """tests / CAPTURE: a [n + 1, w] compact tensor (used rows + loop row) as [nr + 1, w], ``fill`` on unused rows."""
        if not self.on:
            return compact_rows
        out = torch.full((self.nr + 1, compact_rows.shape[1]), fill, dtype=compact_rows.dtype, device=compact_rows.device)
        out[self.graph.rel_used] = compact_rows[:-1]
        out[-1] = compact_rows[-1]
        return out


COMPACT_RELATIONS = True      # tests / A-B: False runs every relation-side product on all nr rows


# ---- active rows: the projections on the rows the graph needs ----------------------------------------------------------------
# P = X Wt is read for DESTINATIONS only, Q = X Wb for SOURCES only (src/jmac_model.py:75-76,85: x_i / x_j of the edges); only
# Z = X Wg is needed for every row (the self loop, :44-45).  On the real ja train graph that is 5 425 / 4 401 / 11 805 of 11 805
# rows: 61 % of the projection's flops, forward and in the input gradient.  The library's GEMMs take row RANGES, not row lists
# (profiles/r5_active_rows.txt: a row-list MFMA kernel of this library runs these shapes at half the library's rate), so the
# encoder node orders the entities BY CLASS inside itself -- [destination only | destination and source | source only | neither]
# -- which makes the destinations rows [0, nD) and the sources rows [s0, s1): three row-range products per projection instead of
# one full one.  The permutation never leaves the node: its inputs are gathered into class order on the way in, its outputs and
# input gradients gathered back on the way out (jmac_rows_{compact,expand}_f32: one launch each), the graph is re-indexed once
# per graph (cached on it).  P rows of non-destinations / Q rows of non-sources are never written and never read.
ACTIVE_ROWS = True            # tests / A-B: False keeps the entity order and the full products
ACTIVE_ROWS_MAX_FRACTION = 0.85   # taken only where (destinations + sources) / 2N is below this: the copies must pay for themselves
ACTIVE_ROWS_MIN_N = 4096
# Stacked KGs (forward_stacked: per-KG row blocks): the class order is per block, i.e. 2 range products per KG for P and Q instead
# of 2 in all -- measured SLOWER than the full products (real el + ja pair step 1.368 -> 1.43 ms, five-KG training step 7.16 ->
# 7.53 ms: launch-bound 2 000-6 000-row GEMMs and copies of the stacked tables).  Built, tested (tests run it), off.
ACTIVE_ROWS_STACKED = False


class _RowOrder:
    """Class order of one graph's entities + the graph re-indexed in it.  ``seg`` (RowBlocks of a stacked launch set): the order
    is by class INSIDE every block, so the blocks keep their row ranges (per-KG BatchNorm statistics, the losses' row windows)
    and destinations / sources are one row range per block (``dst_ranges`` / ``src_ranges``)."""

    def __init__(self, graph: RelGraph, seg=None):
        dev, N, E = graph.device, graph.N, graph.E
        deg = graph.degrees()
        is_dst = deg > 0
        is_src = torch.zeros(N, dtype=torch.bool, device=dev)
        col = graph.col[:E].long()
        is_src[col] = True
        cls = torch.where(is_dst & ~is_src, 0, torch.where(is_dst & is_src, 1, torch.where(is_src, 2, 3)))
        offsets = list(seg.offsets) if (seg is not None and seg.nb > 1) else [0, N]
        nb = len(offsets) - 1
        key = cls
        if nb > 1:
            bounds = torch.tensor(offsets[1:-1], dtype=torch.int64, device=dev)
            key = torch.bucketize(torch.arange(N, device=dev), bounds, right=True) * 4 + cls
        self.old_of_new = torch.sort(key, stable=True)[1].contiguous()          # int64 [N]: row of the caller's table per node row
        self.new_of_old = torch.empty_like(self.old_of_new)
        self.new_of_old[self.old_of_new] = torch.arange(N, device=dev)
        self.pos32 = self.new_of_old.to(torch.int32).contiguous()               # jmac_rows_expand_f32's position list
        cnt = torch.bincount(key, minlength=4 * nb).view(nb, 4).tolist()        # one host read per graph (build time)
        self.dst_ranges = [(offsets[k], offsets[k] + c[0] + c[1]) for k, c in enumerate(cnt)]
        self.src_ranges = [(offsets[k] + c[0], offsets[k] + c[0] + c[1] + c[2]) for k, c in enumerate(cnt)]
        self.nD, self.s0, self.s1 = self.dst_ranges[0][1], self.src_ranges[0][0], self.src_ranges[0][1]    # (one block: tests)
        self.N = N
        active = sum(b - a for a, b in self.dst_ranges) + sum(b - a for a, b in self.src_ranges)
        self.fraction = active / (2.0 * max(N, 1))
        dst = torch.repeat_interleave(torch.arange(N, device=dev), deg.long())
        ei = torch.stack((self.new_of_old[dst], self.new_of_old[col])).contiguous()
        et = graph.etype[:E].long().contiguous()
        from ._lib import mark_index_range
        mark_index_range(ei, N)
        mark_index_range(et, graph.num_rel)
        self.graph = RelGraph(ei, et, N, graph.num_rel, graph.chunk_arg)
        self.graph._key_refs = (ei, et)
        self._info = {}

    def info_rows(self, info):
        """The constant name embeddings in class order (kept: the cat buffer cache keys on the tensor)."""
        hit = self._info.get(id(info))
        if hit is None or hit[0] is not info or hit[1] != info._version:
            if len(self._info) > 8:
                self._info.clear()
            hit = (info, info._version, info.index_select(0, self.old_of_new).contiguous())
            self._info[id(info)] = hit
        return hit[2]


def _row_order(cfg, graph: RelGraph, N: int):
    """The class order for this call, or None: enough rows, fp32 tables, and few enough active rows."""
    if not ACTIVE_ROWS or N < ACTIVE_ROWS_MIN_N or graph.E == 0 or graph.num_src != graph.N or cfg.table_dtype != torch.float32:
        return None
    seg = getattr(cfg, "seg", None)
    if seg is not None and (seg.offsets[-1] != N or (seg.nb > 1 and not ACTIVE_ROWS_STACKED)):
        return None
    key = tuple(seg.offsets) if (seg is not None and seg.nb > 1) else None
    cache = graph.__dict__.setdefault("_row_orders", {})
    ro = cache.get(key)
    if ro is None:
        if torch.cuda.is_current_stream_capturing():     # built in an eager (warm-up) call only: it reads counts back to the host
            return None
        ro = cache[key] = _RowOrder(graph, seg if key is not None else None)
    return ro if ro.fraction <= ACTIVE_ROWS_MAX_FRACTION else None


def _gather_rows(tables, index):
    """[t[index] for t in tables] (fp32 [*, d] with unit inner stride; up to 4 per launch): jmac_rows_compact_f32."""
    n, d = int(index.numel()), tables[0].shape[1]
    tables = [t_ if t_.stride(-1) == 1 else t_.contiguous() for t_ in tables]
    out = [_empty(t_.device, n, d) for t_ in tables]
    ld = (C.c_int64 * len(tables))(*[t_.stride(0) for t_ in tables])
    check(lib().jmac_rows_compact_f32(_vp_array(tables), ld, _vp_array(out), len(tables), ptr(index), n, d, stream()),
          "jmac_rows_compact_f32")
    return out


def _scatter_rows(src, pos32, dst=None):
    """dst[r] (+)= src[pos32[r]]: jmac_rows_expand_f32 (``dst`` given: accumulate onto it; else a fresh table)."""
    n, d = int(pos32.numel()), src.shape[1]
    acc = dst is not None
    if dst is None:
        dst = _empty(src.device, n, d)
    ld = (C.c_int64 * 1)(dst.stride(0))
    check(lib().jmac_rows_expand_f32(_vp_array([src.contiguous()]), _vp_array([dst]), ld, (C.c_int32 * 1)(1 if acc else 0), 1, ptr(pos32), n,
                                     d, stream()), "jmac_rows_expand_f32")
    return dst


# ---- layer pieces -------------------------------------------------------------------------------------------------------
LAYER_PARAMS = ("rel_transform_weight1", "rel_transform_weight2", "loop_rel", "w_att", "a_att", "gcn_weight")


def _layer_inputs(lay):
    return [getattr(lay, n) for n in LAYER_PARAMS] + [lay.bn.weight, lay.bn.bias]


def _vp_array(tensors):
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def _extra_copy(copy):
    """(src pointer, dst pointer, floats) of an optional contiguous copy that rides along in a pack launch."""
    if copy is None:
        return None, None, 0
    src, dst = copy
    if not (src.is_contiguous() and dst.is_contiguous()) or src.numel() != dst.numel() or src.numel() % 4:
        raise ValueError("ride-along copy: contiguous blocks of equal size, a multiple of 4 floats")
    return ptr(src), ptr(dst), src.numel()


def _wcat_pack(w_atts, gcns, d, copy=None, counters=(), seed=None):
    """[Wt | Wb | Wgcn] [d, 3d] of each layer (w_att = [Wt; Wb] stacked by rows, src/jmac_model.py:24,75-76): ONE launch for
    all layers of the call (torch.cat: one per layer).  ``copy`` = (src, dst): one more block copied by the same launch;
    ``counters``: int64 device scalars it increments (the layers' num_batches_tracked); ``seed`` = (state, out): the persistent
    dropout seed words advanced by one and copied to ``out`` (the seeds of this step's draws)."""
    w_atts, gcns = [w.contiguous() for w in w_atts], [g.contiguous() for g in gcns]
    out = [_empty(w_atts[0].device, d, 3 * d) for _ in w_atts]
    counters = list(counters)
    check(lib().jmac_wcat_pack_seed_f32(_vp_array(w_atts), _vp_array(gcns), _vp_array(out), len(out), d, *_extra_copy(copy),
                                        _vp_array(counters) if counters else None, len(counters),
                                        ptr(seed[0]) if seed is not None else None, ptr(seed[1]) if seed is not None else None,
                                        stream()), "jmac_wcat_pack_seed_f32")
    return out


def _drop_seed_state(cache, dev):
    """The persistent dropout seed words of a model (device int64 [2]), or None.  Drawn from torch's generator whenever that
    generator has been touched since this function last drew (re-seeded, or consumed by anything else) -- so ``torch.manual_seed``
    reproduces a run exactly as before -- and otherwise left to the weight-pack launch, which advances it by one per step: a
    step then contains NO torch RNG op, and a hipGraph of it is replayed without the two generator-state fills (and the launch
    gap in front of them) torch adds to every replay of a graph that consumed random numbers.  Inside a stream capture the
    generator is never touched: the state of the eager warm-up steps is used (None if there was none: the caller draws per step)."""
    if cache is None:
        return None
    hit = cache.get("drop_seed")
    if torch.cuda.is_current_stream_capturing():
        return hit[0] if hit is not None and hit[0].device == dev else None
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    cur = (gen.initial_seed(), gen.get_offset())
    if hit is not None and hit[1] == cur and hit[0].device == dev:
        return hit[0]
    state = torch.empty(2, dtype=torch.int64, device=dev).random_()
    cache["drop_seed"] = (state, (gen.initial_seed(), gen.get_offset()))
    return state


def _wcat_unpack(dwcs, d, copy=None):
    """(d w_att [2d, d], d gcn_weight [d, d]) of each layer cut out of its d[Wt|Wb|Wg] [d, 3d]: one launch for all layers
    (instead of a cat and a strided clone per layer)."""
    dev = dwcs[0].device
    dw = [_empty(dev, 2 * d, d) for _ in dwcs]
    dg = [_empty(dev, d, d) for _ in dwcs]
    check(lib().jmac_wcat_unpack_f32(_vp_array(dwcs), _vp_array(dw), _vp_array(dg), len(dwcs), d, *_extra_copy(copy), stream()),
          "jmac_wcat_unpack_f32")
    return list(zip(dw, dg))


class _Chain:
    """The relation side of one layer: RR = R'' [Wb|Wg] with R'' = act(cat(R, loop) W1) W2 (src/jmac_model.py:39-42).

    R'' is used by nothing but that projection (the message is x_j - R''[type], and both of its consumers are linear:
    src/jmac_model.py:60-66,75-88), so the two weight matrices are multiplied first: W2g = W2 [Wb|Wg] (a [d,d] x [d,2d]
    product, no relation rows in it) and RR = act(.) W2g.  One dependency level and ~10 % of the relation-side flops less
    each way than R'' = T W2, RR = R'' [Wb|Wg]; the rounding differs from that order in the last bits only."""

    def __init__(self, lay, R, W1, W2, loop, wc, d):
        dev, nr = R.device, R.shape[0]
        self.R, self.W1, self.W2, self.loop, self.wc, self.d, self.nr = R, W1, W2, loop, wc, d, nr
        self.relu = lay.rel_activation == "relu"
        self.slope = float(lay.atv_mlp.negative_slope)
        self.T, self.W2g, self.RR = _empty(dev, nr + 1, d), _empty(dev, d, 2 * d), _empty(dev, nr + 1, 2 * d)

    def fwd_tasks(self):
        """(W2g task, T task, RR task): the first two are independent (same level), the third needs both."""
        return (gemm_task(self.W2, self.wc[:, self.d:], self.W2g),
                gemm_task(self.R, self.W1, self.T, A2=self.loop, act=ACT_RELU if self.relu else ACT_LEAKY, slope=self.slope),
                gemm_task(self.T, self.W2g, self.RR))

    def bwd_tasks(self, dRR, dwc, dR, dR_accumulate):
        """two levels; weight gradients ride along.  dwc[:, d:] accumulates (the node side wrote it first); dR receives
        d R (rows < nr).  Returns (levels, (dW1, dW2, dloop)) -- the gradients are NOT kept on self: a second reference
        makes autograd's AccumulateGrad clone every one of them instead of taking the buffer."""
        dev, d, nr = dRR.device, self.d, self.nr
        dT, dW2g = _empty(dev, nr + 1, d), _empty(dev, d, 2 * d)
        dW1, dW2, dloop = _empty(dev, d, d), _empty(dev, d, d), _empty(dev, 1, d)
        dact = DACT_RELU if self.relu else DACT_LEAKY
        wr = self.wc[:, d:]
        return [[gemm_task(dRR, self.W2g, dT, tb=True, act=dact, act_src=self.T, slope=self.slope),
                 gemm_task(self.T, dRR, dW2g, ta=True)],
                [gemm_task(dT, self.W1, dR, tb=True, C2=dloop, accumulate=dR_accumulate),
                 gemm_task(self.R, dT, dW1, ta=True, A2=self.loop, defer=True),
                 gemm_task(dW2g, wr, dW2, tb=True, defer=True),                         # d W2 = d W2g [Wb|Wg]^T
                 gemm_task(self.W2, dW2g, dwc[:, d:], ta=True, accumulate=True, defer=True)]], (dW1, dW2, dloop)


class _RelMLP:
    """leaky(R W1) W2 (src/jmac_model.py:195-196)."""

    def __init__(self, R, W1, W2, slope):
        dev, nr, d = R.device, R.shape[0], W1.shape[1]
        self.R, self.W1, self.W2, self.slope = R, W1, W2, float(slope)
        self.M, self.out = _empty(dev, nr, d), _empty(dev, nr, W2.shape[1])

    def fwd_tasks(self):
        return [gemm_task(self.R, self.W1, self.M, act=ACT_LEAKY, slope=self.slope), gemm_task(self.M, self.W2, self.out)]

    def bwd_tasks(self, g, dR, dR_accumulate):
        """two levels -> (levels, (dW1, dW2))"""
        dev = g.device
        dM = _empty(dev, *self.M.shape)
        dW1, dW2 = _empty(dev, *self.W1.shape), _empty(dev, *self.W2.shape)
        return [[gemm_task(g, self.W2, dM, tb=True, act=DACT_LEAKY, act_src=self.M, slope=self.slope),
                 gemm_task(self.M, g, dW2, ta=True, defer=True)],
                [gemm_task(dM, self.W1, dR, tb=True, accumulate=dR_accumulate),
                 gemm_task(self.R, dM, dW1, ta=True, defer=True)]], (dW1, dW2)


class _MlpChain:
    """conv2_alignment's relation side: its relation input is the output of a relation MLP that nothing else reads
    (rel_a_in = leaky(Ra L11u) L12u, src/jmac_model.py:196-197), so  cat(rel_a_in, loop) W1  is evaluated as
    cat(M (L12u W1), loop W1)  with M = leaky(Ra L11u): the [d,d] x [d,d] weight product replaces one nr-row product each way
    and the MLP's second level disappears from the dependency chain (3 levels instead of 4, forward and backward).
    RR = act(.) (W2 [Wb|Wg]) as in ``_Chain``."""

    def __init__(self, lay, Ra, L11u, L12u, mlp_slope, W1, W2, loop, wc, d):
        dev, nr = Ra.device, Ra.shape[0]
        self.Ra, self.L11u, self.L12u, self.W1, self.W2, self.loop, self.wc, self.d, self.nr = Ra, L11u, L12u, W1, W2, loop, wc, d, nr
        self.mlp_slope = float(mlp_slope)
        self.relu = lay.rel_activation == "relu"
        self.slope = float(lay.atv_mlp.negative_slope)
        self.M, self.Wp, self.W2g = _empty(dev, nr, L11u.shape[1]), _empty(dev, L12u.shape[0], d), _empty(dev, d, 2 * d)
        self.T, self.RR = _empty(dev, nr + 1, d), _empty(dev, nr + 1, 2 * d)

    def rel_in(self):
        """rel_a_in itself (tests only: the product path never forms it)."""
        return torch.mm(self.M, self.L12u)

    def fwd_tasks(self):
        act = ACT_RELU if self.relu else ACT_LEAKY
        return [[gemm_task(self.Ra, self.L11u, self.M, act=ACT_LEAKY, slope=self.mlp_slope),
                 gemm_task(self.L12u, self.W1, self.Wp), gemm_task(self.W2, self.wc[:, self.d:], self.W2g)],
                [gemm_task(self.M, self.Wp, self.T[:self.nr], act=act, slope=self.slope),
                 gemm_task(self.loop, self.W1, self.T[self.nr:], act=act, slope=self.slope)],
                [gemm_task(self.T, self.W2g, self.RR)]]

    def bwd_tasks(self, dRR, dwc, dRa, dRa_accumulate):
        """three levels -> (levels, (dW1, dW2, dloop), (dL11u, dL12u))"""
        dev, d, nr = dRR.device, self.d, self.nr
        dT, dW2g, dM, dWp = _empty(dev, nr + 1, d), _empty(dev, d, 2 * d), _empty(dev, *self.M.shape), _empty(dev, *self.Wp.shape)
        dW1, dW2, dloop = _empty(dev, d, d), _empty(dev, d, d), _empty(dev, 1, d)
        dL11u, dL12u = _empty(dev, *self.L11u.shape), _empty(dev, *self.L12u.shape)
        dact = DACT_RELU if self.relu else DACT_LEAKY
        return [[gemm_task(dRR, self.W2g, dT, tb=True, act=dact, act_src=self.T, slope=self.slope),
                 gemm_task(self.T, dRR, dW2g, ta=True)],
                [gemm_task(dT[:nr], self.Wp, dM, tb=True, act=DACT_LEAKY, act_src=self.M, slope=self.mlp_slope),
                 gemm_task(self.M, dT[:nr], dWp, ta=True),
                 gemm_task(dT[nr:], self.W1, dloop, tb=True),
                 gemm_task(self.loop, dT[nr:], dW1, ta=True),                          # the loop row's share of d W1 (K = 1)
                 gemm_task(dW2g, self.wc[:, d:], dW2, tb=True, defer=True),
                 gemm_task(self.W2, dW2g, dwc[:, d:], ta=True, accumulate=True, defer=True)],
                [gemm_task(dM, self.L11u, dRa, tb=True, accumulate=dRa_accumulate),
                 gemm_task(self.Ra, dM, dL11u, ta=True),
                 gemm_task(dWp, self.W1, dL12u, tb=True),
                 gemm_task(self.L12u, dWp, dW1, ta=True, accumulate=True)]], (dW1, dW2, dloop), (dL11u, dL12u)


_DEFER = object()


def _layer_fwd(lay, X, wc, RR, a, graph, training, y, y2=None, table_dtype=torch.float32, seg=None, compact=False, rows=None,
               agg=None):
    """Node side of one RelationAwareLayer (src/jmac_model.py:44-52) given its relation tables: state for the backward.
    ``table_dtype`` bf16 (inference form, no backward: BASELINE config 3): the [P|Q|Z] table comes out of a bf16 GEMM and the
    relation table is rounded to bf16; the aggregation gathers half the bytes, its arithmetic and everything after it is fp32."""
    if table_dtype == torch.bfloat16:
        # padded halves where the half-wave kernel has a form for them (d = 300 -> 304: 16-byte lane loads); the pad columns
        # of the weight are zero, so the GEMM writes zero pad columns
        d = wc.shape[0]
        dh = ops.bf16_pad(d)
        PQZ = torch.mm(X.to(torch.bfloat16), ops.pad_table_weight(wc, d, 3).to(torch.bfloat16))
        if dh != d:
            RRp = torch.zeros((RR.shape[0], 2 * dh), dtype=torch.bfloat16, device=RR.device)
            RRp[:, :d] = RR[:, :d]
            RRp[:, dh:dh + d] = RR[:, d:]
            RR = RRp
        else:
            RR = RR.to(torch.bfloat16)
    elif rows is not None:
        # class-ordered rows (_RowOrder): P for the destinations [0, nD), Q for the sources [s0, s1), Z for every row -- the other
        # P / Q rows are never read by the aggregation kernels (a destination without in-edges takes the self-loop path only)
        d = wc.shape[0]
        PQZ = _empty(X.device, X.shape[0], 3 * d)
        torch.mm(X, wc[:, 2 * d:], out=PQZ[:, 2 * d:])
        for r0, r1 in rows.dst_ranges:                                # (one range per KG of a stacked launch set)
            if r1 > r0:
                torch.mm(X[r0:r1], wc[:, :d], out=PQZ[r0:r1, :d])
        for r0, r1 in rows.src_ranges:
            if r1 > r0:
                torch.mm(X[r0:r1], wc[:, d:2 * d], out=PQZ[r0:r1, d:2 * d])
    else:
        PQZ = torch.mm(X, wc)                                         # [P|Q|Z]: one library GEMM
    slope = float(lay.atv_mlp.negative_slope)
    if agg is _DEFER:                                   # the caller aggregates (two layers in one launch) and finishes below
        return SimpleNamespace(X=X, wc=wc, RR=RR, a=a, PQZ=PQZ, slope=slope)
    pre, smax, sden = agg if agg is not None else _agg_fwd(PQZ, RR, a, graph, slope, compact=compact)
    mean, invstd, use_batch = _bn_fwd(pre, lay.bn, training, y, y2, seg)
    return SimpleNamespace(X=X, wc=wc, RR=RR, a=a, PQZ=PQZ, pre=pre, smax=smax, sden=sden, y=y, mean=mean, invstd=invstd,
                           use_batch=use_batch, slope=slope, bn_weight=lay.bn.weight, seg=seg, compact=compact, rows=rows)


def _layer_finish(lay, part, agg, training, y, y2=None, seg=None, compact=False, rows=None):
    """Second half of _layer_fwd for a layer whose tables were built with ``agg=_DEFER`` and aggregated by the caller."""
    pre, smax, sden = agg
    mean, invstd, use_batch = _bn_fwd(pre, lay.bn, training, y, y2, seg)
    return SimpleNamespace(X=part.X, wc=part.wc, RR=part.RR, a=part.a, PQZ=part.PQZ, pre=pre, smax=smax, sden=sden, y=y, mean=mean,
                           invstd=invstd, use_batch=use_batch, slope=part.slope, bn_weight=lay.bn.weight, seg=seg, compact=compact,
                           rows=rows)


def _layer_bwd(st, graph, gy, gy2, dX, dX_accumulate):
    """Backward of _layer_fwd.  dX: destination of the input gradient (None: not needed).  Returns dRR, dwc (node part),
    da, gbw."""
    gpre, gbw = _bn_bwd(st.pre, st.y, gy, gy2, st.bn_weight, st.mean, st.invstd, st.use_batch, st.seg)
    dPQZ, dRR, da = _agg_bwd(st.PQZ, st.RR, st.a, graph, st.slope, st.pre, st.smax, st.sden, gpre, compact=st.compact)
    rows = getattr(st, "rows", None)
    if dX is not None and rows is not None:
        # class-ordered rows: dP is zero outside the destinations, dQ outside the sources -- the three terms of the input gradient
        # on their row ranges (dZ: every row)
        d = st.wc.shape[0]
        if dX_accumulate:
            dX.addmm_(dPQZ[:, 2 * d:], st.wc[:, 2 * d:].t())
        else:
            torch.mm(dPQZ[:, 2 * d:], st.wc[:, 2 * d:].t(), out=dX)
        for r0, r1 in rows.dst_ranges:
            if r1 > r0:
                dX[r0:r1].addmm_(dPQZ[r0:r1, :d], st.wc[:, :d].t())
        for r0, r1 in rows.src_ranges:
            if r1 > r0:
                dX[r0:r1].addmm_(dPQZ[r0:r1, d:2 * d], st.wc[:, d:2 * d].t())
    elif dX is not None:
        if dX_accumulate:
            dX.addmm_(dPQZ, st.wc.t())
        else:
            torch.mm(dPQZ, st.wc.t(), out=dX)
    dwc = torch.mm(st.X.t(), dPQZ)                                    # [d, 3d]
    return dRR, dwc, da, gbw


def _layer_grads(chain_grads, dwatt_dgcn, da, gbw, d):
    """Gradients of one layer's parameters in LAYER_PARAMS + (bn.weight, bn.bias) order."""
    dW1, dW2, dloop = chain_grads
    d_watt, d_gcn = dwatt_dgcn                                        # _wcat_unpack of the layer's d[Wt|Wb|Wg]
    return [dW1, dW2, dloop, d_watt, da.view(d, 1), d_gcn, gbw[d:], gbw[:d]]


def _layer_ok(l, d) -> bool:
    """One layer's settings the nodes' correctness rests on: comp_op 'sub'; tanh after an affine BatchNorm with an ordinary
    momentum; d x d weights; attention / relation-transform slopes in [0, 1] (the backward reads act'(z) from the sign of
    act(z): wrong for a negative slope); the fused deterministic form asked for (``fused`` / ``bwd_mode`` of the layer are
    honoured, not overridden by the model-level node); no forward hooks (the node never calls the module)."""
    slope = float(l.atv_mlp.negative_slope)
    return (l.comp_op == "sub" and l.layer_act is torch.tanh and l.bn.momentum is not None and l.bn.affine
            and l.in_channels == d and l.out_channels == d and 0.0 <= slope <= 1.0
            and getattr(l, "fused", True) and getattr(l, "bwd_mode", ops.BWD_MODE_DETERMINISTIC) == ops.BWD_MODE_DETERMINISTIC
            and not l._forward_hooks and not l._forward_pre_hooks)


def supported(model, info_dim: Optional[int]) -> bool:
    """What the fused nodes cover: comp_op 'sub', fp32 tables (bf16 tables under no_grad: the inference form), d % 4 == 0
    (16-byte rows), tanh layers with an ordinary BatchNorm momentum, two GNN layers, LeakyReLU slopes in [0, 1], dropout
    p < 1, one chunk setting for the three layers (they share the graph).  Everything else runs op by op (jmac_amd.model)."""
    a = model.args
    d = model.entity_dim
    lays = (model.conv1_alignment, model.conv2_alignment, model.conv1_completion)
    return (getattr(a, "num_gcn_layer", 2) == 2 and d % 4 == 0 and (info_dim is None or info_dim % 4 == 0)
            and _tables_ok(getattr(model, "table_dtype", torch.float32))
            and 0.0 <= float(model.atv_mlp.negative_slope) <= 1.0 and float(model.completion_dropout.p) < 1.0
            and len({l.chunk for l in lays}) == 1
            and all(_layer_ok(l, d) and _tables_ok(l.table_dtype) for l in lays))


def _tables_ok(dtype) -> bool:
    """fp32 tables always; bf16 tables are the inference form (no backward exists for them): only without grad mode."""
    return dtype == torch.float32 or (dtype == torch.bfloat16 and not torch.is_grad_enabled())


# ---- one RelationAwareLayer ---------------------------------------------------------------------------------------------------
class _LayerNode(torch.autograd.Function):
    """RelationAwareLayer.forward (src/jmac_model.py:33-53; DBPv1: JMAC_DBPv1/models/jmac_model.py:43-63) as one node: the
    relation chain in three grouped launches, the projection GEMM, aggregation, BatchNorm + tanh -- and a hand-written
    backward (no slice / cat / add glue).  inputs: cfg, X, R, 8 layer tensors;  output: the layer's [N, d] output."""

    @staticmethod
    def forward(ctx, cfg, X, R, *pl):
        require_device(X, R)
        (lay,) = cfg.layers
        N, d = X.shape
        t = SimpleNamespace()
        (t.wc,) = _wcat_pack([pl[3]], [pl[5]], d)
        t.rc = _RelCompact(cfg.graph, R.shape[0])                 # relation side on the rows the edges name
        (Ru,) = t.rc.gather([R])
        t.ch = _Chain(lay, Ru, pl[0], pl[1], pl[2], t.wc, d)
        w2g, tt, rr = t.ch.fwd_tasks()
        run_levels([[w2g, tt], [rr]])
        y = _empty(X.device, N, d)
        t.st = _layer_fwd(lay, X, t.wc, t.ch.RR, pl[4].reshape(-1), cfg.graph, cfg.training, y, seg=getattr(cfg, "seg", None),
                          compact=t.rc.on)
        t.st.y = None                                            # the output reaches the backward through save_for_backward
        if CAPTURE is not None:                                  # tests: the very tables the kernel gathered + the relation
            CAPTURE["layer.tables"] = (t.st.PQZ, t.rc.full_rows(t.st.RR))   # transform's activation (its sign is its pre-activation's)
            CAPTURE["layer.rel_act"] = t.ch.T                    # compact rows: CAPTURE["layer.rel_used"] + the loop row
            CAPTURE["layer.rel_used"] = cfg.graph.rel_used if t.rc.on else None
        ctx.t, ctx.cfg, ctx.dims = t, cfg, (N, d)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        t, cfg = ctx.t, ctx.cfg
        N, d = ctx.dims
        (y,) = ctx.saved_tensors
        st = SimpleNamespace(**vars(t.st))
        st.y = y
        dev = y.device
        dX = _empty(dev, N, d) if ctx.needs_input_grad[1] else None
        dRR, dwc, da, gbw = _layer_bwd(st, cfg.graph, gy.contiguous(), None, dX, False)
        dRu = _empty(dev, *t.ch.R.shape)
        lv, cg = t.ch.bwd_tasks(dRR, dwc, dRu, False)
        run_levels(lv)
        (dR,) = t.rc.scatter([(dRu, None, False)])
        return (None, dX, dR, *_layer_grads(cg, _wcat_unpack([dwc], d)[0], da, gbw, d))


def layer_supported(lay, X, R) -> bool:
    d = lay.out_channels
    return (_layer_ok(lay, d) and lay.table_dtype == torch.float32 and d % 4 == 0 and X.is_cuda
            and X.dtype == torch.float32 and R.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1
            and X.stride(0) % 4 == 0 and R.is_contiguous() and R.shape[1] == d and R.shape[0] > 0)


def layer_forward(lay, X, R, graph: RelGraph):
    """The layer's output on the fused node (lay: jmac_amd.layer.RelationAwareLayer / RelationalAwareLayer)."""
    cfg = SimpleNamespace(layers=(lay,), graph=graph, training=lay.training)
    return _LayerNode.apply(cfg, X, R, *_layer_inputs(lay))


# ---- forward_name ---------------------------------------------------------------------------------------------------------
class _Cat0Slot:
    """``cat(comp0, info)`` (src/jmac_model.py:180) whose right block -- the constant name embeddings -- is written ONCE: the
    buffer lives with the model (one slot per ``info`` tensor, never replaced or freed while the model lives: a captured
    hipGraph that took it keeps a valid pointer), a forward takes it (rewriting only the left block) and its backward gives
    it back.  A forward that finds it taken (two forwards before a backward) gets a buffer of its own, as before.  ``lease``
    counts the takers: a backward that runs after a later forward has re-used the buffer (retain_graph + an interleaved
    forward) fails loudly instead of reading the other forward's rows.  The slot holds a strong reference to ``info``: its
    storage cannot be recycled for other rows while the slot's key still names it."""

    def __init__(self):
        self.buf, self.info, self.busy, self.lease = None, None, False, 0


class _Cat0Lease:
    def __init__(self, slot, buf, owns):
        self.slot, self.buf, self.owns, self.released = slot, buf, owns, False
        self.lease = slot.lease if owns else -1

    def check(self):
        if self.owns and self.slot.lease != self.lease:
            raise RuntimeError("jmac encoder: the cat(comp0, info) buffer of this forward was re-used by a later forward "
                               "(a second backward after retain_graph with a forward in between?)")

    def release(self):
        if self.owns and not self.released and self.slot.lease == self.lease:
            self.slot.busy = False
        self.released = True

    def __del__(self):
        self.release()


CAT0_MAX_SLOTS = 32      # distinct name-embedding tensors a model keeps a cat buffer for (KGs + KG pairs of DBP-5L: 15)


def _cat0_take(cache, info, N, d, dev, persistent=False) -> _Cat0Lease:
    """``persistent``: the caller vouches that ``info`` is a device tensor it keeps (jmac_amd.model caches the device copies
    of its name-embedding rows); only such tensors get a slot -- a temporary (a fresh ``.to(dev)`` per call) is never cached:
    its address can come back with other rows in it."""
    di = info.shape[1]
    slots = cache.setdefault("cat0", {}) if (cache is not None and persistent) else None
    key = (id(info), info._version, N, d, str(dev))
    slot = slots.get(key) if slots is not None else None
    if slot is not None and slot.info is info and not slot.busy:
        slot.busy, slot.lease = True, slot.lease + 1
        return _Cat0Lease(slot, slot.buf, True)
    buf = _empty(dev, N, d + di)
    buf[:, d:].copy_(info)
    # (never adopt a buffer allocated during a stream capture: it belongs to that graph's memory pool)
    if slots is not None and slot is None and len(slots) < CAT0_MAX_SLOTS and not torch.cuda.is_current_stream_capturing():
        slot = slots[key] = _Cat0Slot()
        slot.buf, slot.info, slot.busy, slot.lease = buf, info, True, 1
        return _Cat0Lease(slot, buf, True)
    return _Cat0Lease(_Cat0Slot(), buf, False)


class _EncoderName(torch.autograd.Function):
    """JMAC.forward_name (src/jmac_model.py:172-204), num_gcn_layer = 2.

    inputs : cfg, E (comp_att), Rc (rel_comp), Ra (rel_align), info, name_linear, uni_linear1_1, uni_linear2_1,
             all_linear_completion, rel_linear11, rel_linear12, rel_linear11_uni, rel_linear12_uni,
             8 tensors per layer (conv1_alignment, conv1_completion, conv2_alignment)
    outputs: align_out [N,d], c1 = completion layer 1 [N,d], rel_c1 [nr,d]"""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        global _TRACKERS
        _TRACKERS = []                            # the layers' num_batches_tracked: bumped by the weight-pack launch (below)
        try:
            out = _EncoderName._forward(ctx, cfg, *tensors)
            done = {id(c) for c in ctx.t.bumped}
            _bump_trackers([c for c in _TRACKERS if id(c) not in done])       # (none left in practice)
            seg = getattr(cfg, "seg", None)
            if seg is not None and seg.nb > 1 and ctx.t.bumped:               # one forward_base call per block in the reference:
                torch._foreach_add_(ctx.t.bumped, seg.nb - 1)                 # num_batches_tracked moves by the block count
            return out
        finally:
            _TRACKERS = None

    @staticmethod
    def _forward(ctx, cfg, E, Rc, Ra, info, NL, U11, U21, Wall, L11, L12, L11u, L12u, *lp):
        require_device(E, Rc, Ra, info)
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None (whole branches are skipped)
        la, lc, l2 = cfg.layers
        pa, pc, p2 = lp[0:8], lp[8:16], lp[16:24]
        graph, training, p_drop, mslope = cfg.graph, cfg.training, cfg.p_drop, cfg.mlp_slope
        dev = E.device
        N, d = E.shape
        di = info.shape[1]
        t = SimpleNamespace()
        # active rows: inside the node the entities are in CLASS order (_RowOrder) -- E_n / info_n are the inputs gathered into it,
        # the graph is the re-indexed one, the projections run on row ranges; outputs are gathered back at the end
        t.ro = ro = _row_order(cfg, graph, N)
        E_n, info_n = E, info
        if ro is not None:
            graph = ro.graph
            (E_n,) = _gather_rows([E], ro.old_of_new)
            info_n = ro.info_rows(info)
        # :177 + :180  cat(comp0, info @ name_linear) @ U11  ==  cat(comp0, info) @ [U11_top ; name_linear @ U11_bottom]
        t.w = _empty(dev, d + di, d)
        # weights: [Wt|Wb|Wg] per layer; the U11_top block of t.w is copied by the same launch
        u11 = U11.contiguous()
        # ... and so are the layers' num_batches_tracked counters (nn.BatchNorm1d bumps them in train mode)
        t.bumped = [lay.bn.num_batches_tracked for lay in (la, lc, l2) if training and lay.bn.track_running_stats]
        # ... and the step's dropout seeds: the model's persistent seed words advanced by one (no torch RNG op per step)
        seeds = None
        if training and p_drop > 0.0:
            state = _drop_seed_state(getattr(cfg, "cache", None), dev)
            seeds = (state, torch.empty(2, dtype=torch.int64, device=dev)) if state is not None else None
        t.wc = _wcat_pack([p[3] for p in (pa, pc, p2)], [p[5] for p in (pa, pc, p2)], d, copy=(u11[:d], t.w[:d]),
                          counters=t.bumped, seed=seeds)
        # ---- relation side: three dependency levels, one launch each; the layers' chains on the rows the edges name
        t.rc = _RelCompact(graph, Rc.shape[0])
        Ra_u, Rc_u = t.rc.gather([Ra, Rc])
        t.cha = _Chain(la, Ra_u, pa[0], pa[1], pa[2], t.wc[0], d)
        t.chc = _Chain(lc, Rc_u, pc[0], pc[1], pc[2], t.wc[1], d)
        t.mlc = _RelMLP(Rc, L11, L12, mslope)                          # rel_c1      (:195): an OUTPUT, all nr rows
        t.ch2 = _MlpChain(l2, Ra_u, L11u, L12u, mslope, p2[0], p2[1], p2[2], t.wc[2], d)   # rel_a_in (:196) + conv2's chain
        fa, fc, f2, mc = t.cha.fwd_tasks(), t.chc.fwd_tasks(), t.ch2.fwd_tasks(), t.mlc.fwd_tasks()
        run_levels([[fa[0], fa[1], fc[0], fc[1], *f2[0], mc[0], gemm_task(NL, U11[d:], t.w[d:])],
                    [fa[2], fc[2], *f2[1], mc[1]],
                    f2[2]])
        # ---- node side.  cat buffers: cat0 = [comp0 | info] (:180), cat1 = [c1n | a1] (:192), catA = [align0 | a1 | a2] (:203)
        t.cat0_lease = _cat0_take(getattr(cfg, "cache", None), info_n, N, d, dev,
                                   persistent=getattr(cfg, "info_persistent", False))     # right block = info, already in place
        t.cat0, t.cat1, t.catA = t.cat0_lease.buf, _empty(dev, N, 2 * d), _empty(dev, N, 3 * d)
        # dropout draws: two device-resident seed words per step (the model's persistent seed state, advanced by the weight-pack
        # launch above: fresh on every replay of a captured step); the normalise kernels draw from them, forward and backward -- no
        # [N, d] mask is written or read
        seeds = (seeds[1] if seeds is not None                     # (no persistent state, e.g. a first call inside a capture: drawn here)
                 else (torch.empty(2, dtype=torch.int64, device=dev).random_() if training and p_drop > 0.0 else None))
        sd = (lambda i: seeds[i:i + 1]) if seeds is not None else (lambda i: None)
        t.inv0, t.drop0 = _norm_drop_fwd(E_n, p_drop, training, t.cat0[:, :d], seed=sd(0))                # :179
        align0 = t.catA[:, :d]
        torch.mm(t.cat0, t.w, out=align0)                                                       # :180
        a_att = [p[4].reshape(-1) for p in (pa, pc, p2)]
        seg = getattr(cfg, "seg", None)
        c1 = _empty(dev, N, d)
        if PAIR_LAUNCHES and cfg.table_dtype == torch.float32:
            # conv1_alignment (:183) and conv1_completion (:190) do not depend on each other and share the graph: their tables first,
            # then BOTH aggregations as one launch (jmac_rel_attn_aggregate_fwd_jobs_f32), then each layer's BatchNorm + tanh
            pa_ = _layer_fwd(la, align0, t.wc[0], t.cha.RR, a_att[0], graph, training, None, rows=ro, agg=_DEFER)
            pc_ = _layer_fwd(lc, E_n, t.wc[1], t.chc.RR, a_att[1], graph, training, None, rows=ro, agg=_DEFER)
            ra, rc_ = _agg_fwd_pair([(pa_.PQZ, pa_.RR, pa_.a, pa_.slope), (pc_.PQZ, pc_.RR, pc_.a, pc_.slope)], graph, compact=t.rc.on)
            t.sa = _layer_finish(la, pa_, ra, training, t.catA[:, d:2 * d], t.cat1[:, d:], seg=seg, compact=t.rc.on, rows=ro)
            t.sc = _layer_finish(lc, pc_, rc_, training, c1, seg=seg, compact=t.rc.on, rows=ro)
        else:
            t.sa = _layer_fwd(la, align0, t.wc[0], t.cha.RR, a_att[0], graph, training, t.catA[:, d:2 * d], t.cat1[:, d:],
                              table_dtype=cfg.table_dtype, seg=seg, compact=t.rc.on, rows=ro)                           # :183
            t.sc = _layer_fwd(lc, E_n, t.wc[1], t.chc.RR, a_att[1], graph, training, c1, table_dtype=cfg.table_dtype, seg=seg,
                              compact=t.rc.on, rows=ro)                              # :190
        t.inv1, t.drop1 = _norm_drop_fwd(c1, p_drop, training, t.cat1[:, :d], seed=sd(1))                 # :191
        t.a_in = torch.mm(t.cat1, U21)                                                          # :192
        t.s2 = _layer_fwd(l2, t.a_in, t.wc[2], t.ch2.RR, a_att[2], graph, training, t.catA[:, 2 * d:],
                          table_dtype=cfg.table_dtype, seg=seg, compact=t.rc.on, rows=ro)                               # :197
        align_out = torch.mm(t.catA, Wall)                                                      # :203
        if ro is not None:                                   # back into the caller's entity order (one launch for both outputs)
            t.E_n, t.c1_n = E_n, c1                          # class-order tensors the backward reads (neither is an output)
            align_out, c1 = _gather_rows([align_out, c1], ro.new_of_old)
        if CAPTURE is not None:
            back = (lambda x: x[ro.new_of_old]) if ro is not None else (lambda x: x.clone())
            rel_a_in = torch.mm(F.leaky_relu(torch.mm(Ra.detach(), L11u.detach()), mslope), L12u.detach())   # (:196) on all rows
            CAPTURE.update(conv1_alignment=(back(align0), Ra.detach()), conv1_completion=(E.detach(), Rc.detach()),
                           conv2_alignment=(back(t.a_in), rel_a_in))
            for name, st, ch in (("conv1_alignment", t.sa, t.cha), ("conv1_completion", t.sc, t.chc), ("conv2_alignment", t.s2, t.ch2)):
                # the very tables the aggregation kernel gathered (class order: back in the caller's; P / Q rows no edge names hold
                # whatever the buffer held -- zero them for the tests' arithmetic)
                PQZ_c = st.PQZ
                if ro is not None:
                    PQZ_c = torch.zeros_like(st.PQZ)
                    PQZ_c[:, 2 * d:] = st.PQZ[:, 2 * d:]
                    for r0, r1 in ro.dst_ranges:
                        PQZ_c[r0:r1, :d] = st.PQZ[r0:r1, :d]
                    for r0, r1 in ro.src_ranges:
                        PQZ_c[r0:r1, d:2 * d] = st.PQZ[r0:r1, d:2 * d]
                    PQZ_c = PQZ_c[ro.new_of_old]
                CAPTURE[name + ".tables"] = (PQZ_c, t.rc.full_rows(st.RR))
                CAPTURE[name + ".rel_act"] = ch.T                        # the relation transform's activation (its sign = the kink side)
            # rows of the compact relation tables (graph.rel_used, then the loop row), None = all rows
            CAPTURE["rel_used"] = graph.rel_used if t.rc.on else None
            CAPTURE["rel_linear11.act"], CAPTURE["rel_linear11_uni.act"] = t.mlc.M, t.ch2.M      # (the second: compact rows)
        # c1 and rel_c1 are OUTPUTS: they reach the backward through save_for_backward / not at all (an attribute on ctx
        # would tie the output to its own grad_fn in a reference cycle)
        rel_c1, t.mlc.out, t.sc.y = t.mlc.out, None, None
        ctx.t, ctx.cfg, ctx.dims = t, cfg, (N, d, di)
        ctx.save_for_backward(E, Rc, Ra, NL, U11, U21, Wall, L11, L12, L11u, L12u, c1)
        return align_out, c1, rel_c1, E, Rc            # E, Rc: layer 0 of the completion layers, as aliases (see INPLACE_GRADS)

    @staticmethod
    def backward(ctx, g_align, g_c1, g_relc1, g_E0, g_Rc0):
        t, cfg = ctx.t, ctx.cfg
        N, d, di = ctx.dims
        E, Rc, Ra, NL, U11, U21, Wall, L11, L12, L11u, L12u, c1 = ctx.saved_tensors
        # c1 is an OUTPUT of this node (the unpacked tensor carries this node as grad_fn): it must not be stored on ctx.t --
        # that cycle keeps the whole graph (and the parameters' AccumulateGrad nodes, with the stream they were created on)
        # alive past the step, which breaks a later stream capture
        ro = t.ro
        E_x, c1_y = (t.E_n, t.c1_n) if ro is not None else (E, c1)       # class order inside the node (forward)
        sc = SimpleNamespace(**vars(t.sc))
        sc.y = c1_y
        graph = ro.graph if ro is not None else cfg.graph
        dev = E.device
        have_align = g_align is not None
        have_c = have_align or g_c1 is not None
        if ro is not None and have_c:                        # the incoming gradients into class order: one launch
            gin = [g for g in (g_align, g_c1) if g is not None]
            gout = _gather_rows(gin, ro.old_of_new)
            if g_align is not None:
                g_align = gout[0]
            if g_c1 is not None:
                g_c1 = gout[-1]
        dE = None
        dRa = dRc = None
        dWall = dU21 = dU11 = dNL = gL11 = gL12 = gL11u = gL12u = None
        ga = gc = g2 = [None] * 8
        levels: List[List[GemmTask]] = [[] for _ in range(5)]

        def add(first_level, tasks_and_grads):
            task_levels, grads = tasks_and_grads
            for i, lv in enumerate(task_levels):
                levels[first_level + i].extend(lv)
            return grads

        if have_align:
            g_align = g_align.contiguous()
            dcatA = torch.mm(g_align, Wall.t())                                  # [N,3d]: d align0 | d a1 | d a2
            dWall = torch.mm(t.catA.t(), g_align)
            # conv2_alignment
            d_ain = _empty(dev, N, d)
            dRR2, dwc2, da2, gbw2 = _layer_bwd(t.s2, graph, dcatA[:, 2 * d:], None, d_ain, False)
            dcat1 = torch.mm(d_ain, U21.t())                                     # [N,2d]: d c1n | d a1
            dU21 = torch.mm(t.cat1.t(), d_ain)
        # conv1_completion: gradient of c1 = normalise/dropout adjoint of d c1n (+ the loss' own gradient)
        if have_c:
            gy, gy2 = None, None
            if have_align:
                gy = _empty(dev, N, d)
                _norm_drop_bwd(c1_y, t.inv1, t.drop1, dcat1[:, :d], gy, False)
                gy2 = g_c1.contiguous() if g_c1 is not None else None
            else:
                gy = g_c1.contiguous()
            # the layer-0 loss gradient: this node's input gradient goes on top of it (class order: at the very end, below)
            taken = _take_grad(g_E0, (N, d)) if ro is None else None
            if taken is not None:
                dE, g_E0 = taken, None
            else:
                dE = _empty(dev, N, d)
            dRRc, dwcc, dac, gbwc = _layer_bwd(sc, graph, gy, gy2, dE, taken is not None)
        if have_align:
            # conv1_alignment: its output fed cat1 and catA -> two gradient sources; its input is align0 = catA[:, :d]
            d_align0 = dcatA[:, :d]
            dRRa, dwca, daa, gbwa = _layer_bwd(t.sa, graph, dcatA[:, d:2 * d], dcat1[:, d:], d_align0, True)
            # align0 = cat0 @ w: only comp0 needs an input gradient (the name embeddings are constants)
            d_comp0 = torch.mm(d_align0, t.w[:d].t())
            t.cat0_lease.check()
            dw = torch.mm(t.cat0.t(), d_align0)                                  # [d+di, d]
            t.cat0_lease.release()                                               # last reader of cat0
            _norm_drop_bwd(E_x, t.inv0, t.drop0, d_comp0, dE, True)
            dU11 = _empty(dev, 2 * d, d)                                         # [:d] <- dw[:d] by the unpack launch below
            dNL = _empty(dev, di, d)
            levels[0].extend([gemm_task(dw[d:], U11[d:], dNL, tb=True, defer=True), gemm_task(NL, dw[d:], dU11[d:], ta=True, defer=True)])
        # ---- relation side.  The chains' relation gradients land in COMPACT buffers (the rows the edges name) and are put back
        # into full tables by one expand launch; rel_c1's MLP (an output: all rows) writes the full d rel_comp directly
        wrote_c, dRc_full = False, None
        if have_align or g_relc1 is not None or have_c:
            rc = t.rc
            dRa_u = _empty(dev, rc.n, d) if have_align else None
            # d rel_comp: the layer-0 loss gradient's buffer where this backward may take it over (every writer below then adds)
            dRc_full = _take_grad(g_Rc0, Rc.shape)
            wrote_c = dRc_full is not None                                        # "the full buffer holds a contribution"
            if wrote_c:
                g_Rc0 = None
            elif g_relc1 is not None or (have_c and not rc.on):
                dRc_full = _empty(dev, *Rc.shape)
            dRc_u = (_empty(dev, rc.n, d) if rc.on else dRc_full) if have_c else None
            if have_align:
                cga = add(0, t.cha.bwd_tasks(dRRa, dwca, dRa_u, False))          # levels 0-1 -> d rel_align (first writer)
                lv2, cg2, (gL11u, gL12u) = t.ch2.bwd_tasks(dRR2, dwc2, dRa_u, True)     # levels 0-2; d rel_align += at level 2
                add(0, (lv2, None))
            if g_relc1 is not None:                                               # rel_c1 = MLP(rel_comp) needs only the loss' gradient:
                gL11, gL12 = add(0, t.mlc.bwd_tasks(g_relc1.contiguous(), dRc_full, wrote_c))  # levels 0-1 -> d rel_comp
                wrote_c = True
            if have_c:
                if rc.on:                            # own buffer: no ordering against the MLP's write
                    cgc = add(0, t.chc.bwd_tasks(dRRc, dwcc, dRc_u, False))
                else:                                # one buffer: the MLP writes it at level 1, the chain adds one level later
                    cgc = add(1 if wrote_c else 0, t.chc.bwd_tasks(dRRc, dwcc, dRc_full, wrote_c))
            run_levels(levels, balance=True)
            items = ([(dRa_u, None, False)] if have_align else []) + ([(dRc_u, dRc_full if wrote_c else None, True)] if (have_c and rc.on) else [])
            outs = rc.scatter(items) if items else []
            if have_align:
                dRa = outs[0]
            if have_c and rc.on:
                dRc = outs[-1]
            elif have_c or wrote_c:
                dRc = dRc_full
        dwcs = ([dwca, dwc2] if have_align else []) + ([dwcc] if have_c else [])
        cut = _wcat_unpack(dwcs, d, copy=(dw[:d], dU11[:d]) if have_align else None) if dwcs else []
        if have_align:
            ga = _layer_grads(cga, cut[0], daa, gbwa, d)
            g2 = _layer_grads(cg2, cut[1], da2, gbw2, d)
        if have_c:
            gc = _layer_grads(cgc, cut[-1], dac, gbwc, d)
        t.cat0_lease.release()
        if dRc is None and wrote_c:                    # nothing but the taken-over layer-0 gradient
            dRc = dRc_full
        if ro is not None and dE is not None:          # class order -> the caller's, onto the layer-0 loss gradient where it may
            taken = _take_grad(g_E0, (N, d))
            dE = _scatter_rows(dE, ro.pos32, dst=taken)
            if taken is not None:
                g_E0 = None
        dE, dRc = _plus(dE, g_E0), _plus(dRc, g_Rc0)   # layer-0 gradients this backward could not take over: added here
        return (None, dE, dRc, dRa, None, dNL, dU11, dU21, dWall, gL11, gL12, gL11u, gL12u, *ga, *gc, *g2)


# ---- forward_no_name ------------------------------------------------------------------------------------------------------
class _EncoderNoName(torch.autograd.Function):
    """JMAC.forward_no_name (src/jmac_model.py:207-220), num_gcn_layer = 2: c1 = conv1_completion(E, Rc), rel_c1 = MLP(Rc).

    inputs: cfg, E, Rc, rel_linear11, rel_linear12, 8 tensors of conv1_completion;  outputs: c1, rel_c1"""

    @staticmethod
    def forward(ctx, cfg, E, Rc, L11, L12, *pc):
        require_device(E, Rc)
        ctx.set_materialize_grads(False)
        (lc,) = cfg.layers
        N, d = E.shape
        t = SimpleNamespace()
        (t.wc,) = _wcat_pack([pc[3]], [pc[5]], d)
        t.rc = _RelCompact(cfg.graph, Rc.shape[0])
        (Rc_u,) = t.rc.gather([Rc])
        t.chc = _Chain(lc, Rc_u, pc[0], pc[1], pc[2], t.wc, d)
        t.mlc = _RelMLP(Rc, L11, L12, cfg.mlp_slope)
        fc, mc = t.chc.fwd_tasks(), t.mlc.fwd_tasks()
        run_levels([[fc[0], fc[1], mc[0]], [fc[2], mc[1]]])
        c1 = _empty(E.device, N, d)
        t.sc = _layer_fwd(lc, E, t.wc, t.chc.RR, pc[4].reshape(-1), cfg.graph, cfg.training, c1, table_dtype=cfg.table_dtype,
                          seg=getattr(cfg, "seg", None), compact=t.rc.on)
        if CAPTURE is not None:
            CAPTURE.update(conv1_completion=(E.detach(), Rc.detach()))
            CAPTURE["conv1_completion.tables"] = (t.sc.PQZ, t.rc.full_rows(t.sc.RR))
        rel_c1, t.mlc.out, t.sc.y = t.mlc.out, None, None
        ctx.t, ctx.cfg, ctx.dims = t, cfg, (N, d)
        ctx.save_for_backward(E, Rc, L11, L12, c1)
        return c1, rel_c1, E, Rc                         # E, Rc: layer 0 of the completion layers, as aliases (see INPLACE_GRADS)

    @staticmethod
    def backward(ctx, g_c1, g_relc1, g_E0, g_Rc0):
        t, cfg = ctx.t, ctx.cfg
        N, d = ctx.dims
        E, Rc, L11, L12, c1 = ctx.saved_tensors
        sc = SimpleNamespace(**vars(t.sc))                   # c1 is an output: never stored on ctx.t (reference cycle)
        sc.y = c1
        dev = E.device
        dE = dRc = None
        gc = [None] * 8
        levels: List[List[GemmTask]] = [[] for _ in range(5)]
        rc = t.rc
        # d rel_comp: the layer-0 loss gradient's buffer where this backward may take it over (every writer below then adds)
        dRc_full = _take_grad(g_Rc0, Rc.shape)
        took_rc = dRc_full is not None
        if took_rc:
            g_Rc0 = None
        elif g_relc1 is not None or not rc.on:
            dRc_full = _empty(dev, *Rc.shape)
        dRc_u = _empty(dev, rc.n, d) if rc.on else dRc_full
        wrote = False                                        # the chain wrote dRc_u
        if g_c1 is not None:
            taken = _take_grad(g_E0, (N, d))
            if taken is not None:
                dE, g_E0 = taken, None
            else:
                dE = _empty(dev, N, d)
            dRRc, dwcc, dac, gbwc = _layer_bwd(sc, cfg.graph, g_c1.contiguous(), None, dE, taken is not None)
            lv, cgc = t.chc.bwd_tasks(dRRc, dwcc, dRc_u, took_rc and not rc.on)
            for i, l in enumerate(lv):
                levels[i].extend(l)
            wrote = True
        gL11 = gL12 = None
        wrote_full = took_rc                                 # the full buffer holds a contribution
        if g_relc1 is not None:
            first = (rc.on or not wrote) and not took_rc     # compact: the MLP has the full buffer to itself
            lv, (gL11, gL12) = t.mlc.bwd_tasks(g_relc1.contiguous(), dRc_full, not first)
            for i, l in enumerate(lv):
                levels[(0 if (rc.on or not wrote) else 2) + i].extend(l)
            wrote_full = True
        run_levels(levels)
        if rc.on and wrote:
            (dRc_buf,) = rc.scatter([(dRc_u, dRc_full if wrote_full else None, True)])
        else:
            dRc_buf = dRc_full
        wrote = wrote or wrote_full
        if g_c1 is not None:
            gc = _layer_grads(cgc, _wcat_unpack([dwcc], d)[0], dac, gbwc, d)
        dRc = dRc_buf if wrote else None
        return (None, _plus(dE, g_E0), _plus(dRc, g_Rc0), gL11, gL12, *gc)


def _cfg(model, layers, graph):
    training = model.training
    return SimpleNamespace(layers=layers, graph=graph, training=training,
                           p_drop=float(model.completion_dropout.p) if model.completion_dropout.training else 0.0,
                           mlp_slope=float(model.atv_mlp.negative_slope),
                           table_dtype=getattr(model, "table_dtype", torch.float32))


def forward_name(model, comp_att, rel_comp, rel_align, info, graph: RelGraph, seg: Optional[RowBlocks] = None,
                 info_persistent: bool = False):
    """(align_out, c1, rel_c1, comp0, rel0) of JMAC.forward_name on the fused node.  ``seg``: the rows are a stack of KGs (``graph`` their
    block-diagonal union, the relation tables stacked likewise): BatchNorm statistics per block.  ``info_persistent``: the
    caller keeps ``info`` (a device tensor) alive and unchanged between calls -- its cat buffer may be cached."""
    la, lc, l2 = model.conv1_alignment, model.conv1_completion, model.conv2_alignment
    cfg = _cfg(model, (la, lc, l2), graph)
    cfg.training = la.training                                           # BatchNorm follows the layers' own mode
    cfg.cache = model.__dict__.setdefault("_encoder_cache", {})          # buffers that outlive a step (see _Cat0Slot)
    cfg.seg, cfg.info_persistent = seg, bool(info_persistent)
    # -> (align_out, c1, rel_c1, comp0, rel0): comp0 / rel0 alias comp_att / rel_comp -- hand THEM on as layer 0 of the completion
    # layers, so that the layer-0 loss gradient reaches this node's backward (see INPLACE_GRADS)
    return _EncoderName.apply(cfg, comp_att, rel_comp, rel_align, info, model.name_linear, model.uni_linear1_1,
                              model.uni_linear2_1, model.all_linear_completion, model.rel_linear11, model.rel_linear12,
                              model.rel_linear11_uni, model.rel_linear12_uni, *_layer_inputs(la), *_layer_inputs(lc),
                              *_layer_inputs(l2))


def forward_no_name(model, comp_att, rel_comp, graph: RelGraph, seg: Optional[RowBlocks] = None):
    """(c1, rel_c1, comp0, rel0) of JMAC.forward_no_name on the fused node (comp0 / rel0: aliases of the inputs, as above)."""
    lc = model.conv1_completion
    cfg = _cfg(model, (lc,), graph)
    cfg.training = lc.training
    cfg.seg = seg
    return _EncoderNoName.apply(cfg, comp_att, rel_comp, model.rel_linear11, model.rel_linear12, *_layer_inputs(lc))
