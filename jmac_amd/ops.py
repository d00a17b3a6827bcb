"""torch.autograd.Function wrappers over the C ABI (include/jmac_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic step of the hot path
runs in libjmac_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from ._lib import check, lib, ptr, require_device, stream, testing_lib
from .graph import RelGraph

BWD_MODE_ATOMIC = 0            # testing build only (libjmac_hip_testing.so): float atomics, a second implementation for the tests
BWD_MODE_DETERMINISTIC = 1

# bench.py sets this to a list to collect (name, start_event, end_event) around the aggregation launches;
# events are recorded on the stream the kernels are launched on (torch's current stream).
PROFILE = None


def _ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("jmac_amd ops compute in fp32 (got %s)" % t.dtype)
    return t


def _table(t: torch.Tensor) -> torch.Tensor:
    """Gathered tables: fp32, or bf16 for the inference form (arithmetic and outputs stay fp32)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("jmac_amd tables are fp32 or bf16 (got %s)" % t.dtype)
    return t


def bf16_pad(d: int) -> int:
    """Half pitch of the PADDED bf16 tables the half-wave aggregation kernel reads (16-byte aligned halves: a multiple of
    8 elements): d = 300 -> 304; d itself where no padded form exists."""
    return 304 if d == 300 else d


def pad_table_weight(w: torch.Tensor, d: int, parts: int) -> torch.Tensor:
    """[k, parts*d] projection weight -> [k, parts*bf16_pad(d)] with zero pad columns behind every d-wide part: the GEMM then
    writes [P|Q|Z] / [Rq|Rz] rows in the padded layout, pad columns zero."""
    dh = bf16_pad(d)
    if dh == d:
        return w
    out = w.new_zeros((w.shape[0], parts * dh))
    for p_ in range(parts):
        out[:, p_ * dh:p_ * dh + d] = w[:, p_ * d:(p_ + 1) * d]
    return out


def pad_table(t: torch.Tensor, d: int, parts: int) -> torch.Tensor:
    """[n, parts*d] table -> the padded layout [n, parts*bf16_pad(d)] (zero pad columns); benchmarks and tests that build
    their tables directly use it -- the layers get the layout from the GEMM (pad_table_weight)."""
    return pad_table_weight(t, d, parts)


class _RelAttnAggregate(torch.autograd.Function):
    """out = out_scale * ( sqrt(deg) * softmax-weighted sum over in-edges of (Z[j]-Rz[t]) + [Z[i]-Rz[loop]] ).

    PQZ: [N, 3d] = P | Q | Z (one GEMM output), RR: [nrel, 2d] = Rq | Rz, a: [d].
    """

    @staticmethod
    def forward(ctx, PQZ, RR, a, graph: RelGraph, slope: float, loop_rel: int, out_scale: float, bwd_mode: int):
        require_device(PQZ, RR, a)
        PQZ, RR, a = _table(PQZ).contiguous(), _table(RR).contiguous(), _f32c(a).contiguous()
        if PQZ.dtype != RR.dtype:
            raise TypeError("PQZ and RR must share a dtype")
        bf16 = PQZ.dtype == torch.bfloat16
        N, d3 = PQZ.shape
        dh = d3 // 3                       # half pitch of the table rows; > d for padded bf16 tables (bf16_pad)
        d = int(a.numel())
        if dh != d and not (bf16 and dh == bf16_pad(d) and RR.shape[1] == 2 * dh):
            raise ValueError("tables of half pitch %d do not go with a_att of %d elements" % (dh, d))
        if graph.N != N:
            raise ValueError("graph has %d nodes, tables have %d rows" % (graph.N, N))
        L = lib()
        dev = PQZ.device
        out = torch.empty((N, d), dtype=torch.float32, device=dev)
        seg_max = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        seg_den = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        s = graph.by_dst
        ws_bytes = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d))
        ws = _ws(ws_bytes, dev)
        esz = PQZ.element_size()
        ev0 = _ev() if PROFILE is not None else None
        if dh != d:
            check(L.jmac_rel_attn_aggregate_fwd_bf16_padded(
                ptr(PQZ), d3, PQZ.data_ptr() + dh * esz, d3, ptr(RR), RR.shape[1], dh, ptr(a),
                ptr(graph.col), ptr(graph.etype), C.byref(s.view()), N, d, float(slope), int(loop_rel), 0, float(out_scale),
                ptr(out), d, ptr(seg_max), ptr(seg_den), ptr(ws), ws_bytes, stream()), "jmac_rel_attn_aggregate_fwd_bf16_padded")
        else:
            fwd = L.jmac_rel_attn_aggregate_fwd_bf16 if bf16 else L.jmac_rel_attn_aggregate_fwd_f32
            check(fwd(
                ptr(PQZ), d3, PQZ.data_ptr() + d * esz, d3, ptr(RR), RR.shape[1], ptr(a),
                ptr(graph.col), ptr(graph.etype), C.byref(s.view()), N, d, float(slope), int(loop_rel), 0, float(out_scale),
                ptr(out), d, ptr(seg_max), ptr(seg_den), ptr(ws), ws_bytes, stream()),
                "jmac_rel_attn_aggregate_fwd_%s" % ("bf16" if bf16 else "f32"))
        if ev0 is not None:
            PROFILE.append(("rel_attn_fwd_bf16" if bf16 else "rel_attn_fwd", ev0, _ev()))
        if bf16:
            ctx.mark_non_differentiable(out)
            return out
        ctx.save_for_backward(PQZ, RR, a, out, seg_max, seg_den)
        ctx.graph, ctx.slope, ctx.loop_rel, ctx.out_scale, ctx.bwd_mode = graph, slope, loop_rel, out_scale, bwd_mode
        return out

    @staticmethod
    def backward(ctx, G):
        PQZ, RR, a, out, seg_max, seg_den = ctx.saved_tensors
        graph: RelGraph = ctx.graph
        mode = int(ctx.bwd_mode)
        L = lib() if mode == BWD_MODE_DETERMINISTIC else testing_lib()      # mode 0 exists in the testing build only
        dev = PQZ.device
        N, d3 = PQZ.shape
        d = d3 // 3
        nrel = RR.shape[0]
        G = _f32c(G).contiguous()
        if mode == BWD_MODE_DETERMINISTIC:
            graph.ensure_backward_views()
        dPQZ = torch.empty_like(PQZ)
        dRR = torch.empty_like(RR)
        da = torch.empty_like(a)
        vd = graph.by_dst_bwd.view()
        vs = graph.by_src.view() if mode else None
        vr = graph.by_rel.view() if mode else None
        ws_bytes = int(L.jmac_rel_attn_bwd_workspace_bytes(
            N, graph.E, nrel, d, graph.by_dst_bwd.n_parts_max,
            graph.by_src.n_parts_max if mode else 0, graph.by_rel.n_parts_max if mode else 0, mode))
        ws = _ws(ws_bytes, dev)
        esz = PQZ.element_size()
        ev0 = _ev() if PROFILE is not None else None
        check(L.jmac_rel_attn_aggregate_bwd_f32(
            ptr(PQZ), d3, PQZ.data_ptr() + d * esz, d3, ptr(RR), RR.shape[1], ptr(a),
            ptr(graph.col), ptr(graph.etype), ptr(graph.dst_of_slot) if mode else None,
            C.byref(vd), C.byref(vs) if mode else None, C.byref(vr) if mode else None,
            N, N, graph.E, nrel, d, float(ctx.slope), int(ctx.loop_rel), 0, float(ctx.out_scale),
            ptr(out), d, ptr(seg_max), ptr(seg_den), ptr(G), d,
            ptr(dPQZ), d3, dPQZ.data_ptr() + d * esz, d3, ptr(dRR), dRR.shape[1], ptr(da),
            mode, ptr(ws), ws_bytes, stream()), "jmac_rel_attn_aggregate_bwd_f32")
        if ev0 is not None:
            PROFILE.append(("rel_attn_bwd", ev0, _ev()))
        return dPQZ, dRR, da, None, None, None, None, None


def rel_attn_aggregate(PQZ: torch.Tensor, RR: torch.Tensor, a: torch.Tensor, graph: RelGraph, slope: float,
                       loop_rel: int = -1, out_scale: float = 1.0,
                       bwd_mode: int = BWD_MODE_DETERMINISTIC) -> torch.Tensor:
    if PQZ.dtype == torch.bfloat16 and torch.is_grad_enabled() and (PQZ.requires_grad or RR.requires_grad or a.requires_grad):
        raise RuntimeError("bf16 tables are the inference form of the op (no backward): run under torch.no_grad()")
    return _RelAttnAggregate.apply(PQZ, RR, a, graph, float(slope), int(loop_rel), float(out_scale), int(bwd_mode))


def rel_attn_split_fwd_raw(P, QZ, RR, a, graph: RelGraph, slope: float, out_scale: float, loop_rel: int, self_off: int):
    """jmac_rel_attn_aggregate_fwd_f32 with P [N_dst, d] and QZ [N_src, 2d] as separate tables (no autograd): out [N, d] and
    the per-destination softmax (max, denominator)."""
    N, d = P.shape
    if graph.N != N or graph.num_src != QZ.shape[0] or QZ.shape[1] != 2 * d:
        raise ValueError("graph / table shapes disagree")
    L = lib()
    dev = P.device
    out = torch.empty((N, d), dtype=torch.float32, device=dev)
    seg_max = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    seg_den = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    s = graph.by_dst
    ws_bytes = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d))
    ws = _ws(ws_bytes, dev)
    ev0 = _ev() if PROFILE is not None else None
    check(L.jmac_rel_attn_aggregate_fwd_f32(
        ptr(P), d, ptr(QZ), 2 * d, ptr(RR), RR.shape[1], ptr(a),
        ptr(graph.col), ptr(graph.etype), C.byref(s.view()), N, d, float(slope), int(loop_rel), int(self_off), float(out_scale),
        ptr(out), d, ptr(seg_max), ptr(seg_den), ptr(ws), ws_bytes, stream()), "jmac_rel_attn_aggregate_fwd_f32")
    if ev0 is not None:
        PROFILE.append(("rel_attn_fwd", ev0, _ev()))
    return out, seg_max, seg_den


def rel_attn_split_bwd_raw(P, QZ, RR, a, graph: RelGraph, slope: float, out_scale: float, loop_rel: int, self_off: int,
                           out, seg_max, seg_den, G):
    """jmac_rel_attn_aggregate_bwd_f32 (deterministic form) for the split tables: dP, dQZ, dRR, da."""
    L = lib()
    dev = P.device
    N, d = P.shape
    nsrc, nrel = QZ.shape[0], RR.shape[0]
    G = _f32c(G).contiguous()
    graph.ensure_backward_views()
    dP, dQZ, dRR, da = torch.empty_like(P), torch.empty_like(QZ), torch.empty_like(RR), torch.empty_like(a)
    vd, vs, vr = graph.by_dst_bwd.view(), graph.by_src.view(), graph.by_rel.view()
    ws_bytes = int(L.jmac_rel_attn_bwd_workspace_bytes(N, graph.E, nrel, d, graph.by_dst_bwd.n_parts_max,
                                                       graph.by_src.n_parts_max, graph.by_rel.n_parts_max, 1))
    ws = _ws(ws_bytes, dev)
    ev0 = _ev() if PROFILE is not None else None
    check(L.jmac_rel_attn_aggregate_bwd_f32(
        ptr(P), d, ptr(QZ), 2 * d, ptr(RR), RR.shape[1], ptr(a),
        ptr(graph.col), ptr(graph.etype), ptr(graph.dst_of_slot), C.byref(vd), C.byref(vs), C.byref(vr),
        N, nsrc, graph.E, nrel, d, float(slope), int(loop_rel), int(self_off), float(out_scale),
        ptr(out), d, ptr(seg_max), ptr(seg_den), ptr(G), d,
        ptr(dP), d, ptr(dQZ), 2 * d, ptr(dRR), dRR.shape[1], ptr(da), 1, ptr(ws), ws_bytes, stream()),
        "jmac_rel_attn_aggregate_bwd_f32")
    if ev0 is not None:
        PROFILE.append(("rel_attn_bwd", ev0, _ev()))
    return dP, dQZ, dRR, da


PHASE_A, PHASE_B, PHASE_C, PHASE_M, PHASE_ALL = 1, 2, 4, 8, 15


class SplitBackwardPhases:
    """jmac_rel_attn_aggregate_bwd_phases_f32 on the split tables: the deterministic backward as separately launched phases --
    pass A (+ pass C and the merges that hang on them) first, then pass B over SLABS of the source rows, each slab's d[Q|Z] rows
    complete when its call returns to the stream (the caller queues that slab's reduce-scatter and goes on with the next slab).
    ``slab_bounds``: ascending source-row bounds from 0 to the table's rows.  Results equal rel_attn_split_bwd_raw's bit for bit
    (the same kernels on the same items in the same order per output row)."""

    def __init__(self, P, QZ, RR, a, graph: RelGraph, slope: float, out_scale: float, loop_rel: int, self_off: int, out, seg_max,
                 seg_den, G, slab_bounds):
        self.L = lib()
        self.P, self.QZ, self.RR, self.a, self.graph = P, QZ, RR, a, graph
        self.slope, self.out_scale, self.loop_rel, self.self_off = float(slope), float(out_scale), int(loop_rel), int(self_off)
        self.out, self.seg_max, self.seg_den, self.G = out, seg_max, seg_den, _f32c(G).contiguous()
        graph.ensure_backward_views()
        self.bounds = [int(b) for b in slab_bounds]
        self.slabs = graph.src_slab_views(self.bounds)
        N, d = P.shape
        self.N, self.d, self.nsrc, self.nrel = N, d, QZ.shape[0], RR.shape[0]
        self.dP, self.dQZ, self.dRR, self.da = torch.empty_like(P), torch.empty_like(QZ), torch.empty_like(RR), torch.empty_like(a)
        ws_bytes = int(self.L.jmac_rel_attn_bwd_workspace_bytes(N, graph.E, self.nrel, d, graph.by_dst_bwd.n_parts_max,
                                                                graph.by_src.n_parts_max, graph.by_rel.n_parts_max, 1))
        self.ws, self.ws_bytes = _ws(ws_bytes, P.device), ws_bytes

    def _call(self, phases: int, src_view, row0: int, rows: int) -> None:
        g, d = self.graph, self.d
        vd, vr = g.by_dst_bwd.view(), g.by_rel.view()
        ev0 = _ev() if PROFILE is not None else None
        check(self.L.jmac_rel_attn_aggregate_bwd_phases_f32(
            ptr(self.P), d, ptr(self.QZ), 2 * d, ptr(self.RR), self.RR.shape[1], ptr(self.a),
            ptr(g.col), ptr(g.etype), ptr(g.dst_of_slot), C.byref(vd), C.byref(src_view), C.byref(vr),
            self.N, rows, g.E, self.nrel, d, self.slope, self.loop_rel, self.self_off - row0, self.out_scale,
            ptr(self.out), d, ptr(self.seg_max), ptr(self.seg_den), ptr(self.G), d,
            ptr(self.dP), d, self.dQZ.data_ptr() + row0 * 2 * d * 4, 2 * d, ptr(self.dRR), self.dRR.shape[1], ptr(self.da),
            int(phases), ptr(self.ws), self.ws_bytes, stream()), "jmac_rel_attn_aggregate_bwd_phases_f32")
        if ev0 is not None:
            PROFILE.append(("rel_attn_bwd_phase%d" % phases, ev0, _ev()))

    def begin(self) -> None:
        """Pass A, pass C and every merge but the by-source one: dP, dRR, da are final afterwards."""
        self._call(PHASE_A | PHASE_C | PHASE_M, self.graph.by_src.view(), 0, self.nsrc)

    def slab(self, c: int) -> torch.Tensor:
        """Pass B on source rows [bounds[c], bounds[c+1]): returns that slice of d[Q|Z] (final once the stream reaches here)."""
        r0, r1 = self.bounds[c], self.bounds[c + 1]
        if r1 > r0:
            self._call(PHASE_B, self.slabs[c].view(), r0, r1 - r0)
        return self.dQZ[r0:r1]


def softmax_parts_merge(parts, N: int, d: int, device, zself=None, rz_loop=None, out_scale: float = 1.0):
    """jmac_softmax_parts_merge_f32: parts = [(out_c [N,d], seg_max_c [N], seg_den_c [N], rowptr_c [N+1] int32), ...] of the
    same destinations over disjoint edge sets -> (out [N,d], seg_max [N], seg_den [N]) of their union; out = out_scale * (nb +
    zself[i] - rz_loop) with a self table (zself: [N, d] rows with any 4-aligned stride, rz_loop [d]), else out_scale * nb."""
    L = lib()
    n = len(parts)
    out = torch.empty((N, d), dtype=torch.float32, device=device)
    seg_max = torch.empty(max(N, 1), dtype=torch.float32, device=device)
    seg_den = torch.empty(max(N, 1), dtype=torch.float32, device=device)
    arr = lambda k: (C.c_void_p * max(n, 1))(*[ptr(p[k]) for p in parts])
    for o, m, l, rp in parts:
        require_device(o, m, l, rp)
        if o.shape != (N, d) or not o.is_contiguous() or rp.dtype != torch.int32 or rp.numel() != N + 1:
            raise ValueError("softmax_parts_merge: part shapes disagree")
    if zself is not None:
        require_device(zself, rz_loop)
        if zself.shape != (N, d) or zself.stride(1) != 1 or rz_loop.numel() != d or not rz_loop.is_contiguous():
            raise ValueError("softmax_parts_merge: self table shapes disagree")
    if n == 0:                       # no edges at all: the kernel is not given anything to read
        seg_max.fill_(float("-inf"))
        seg_den.zero_()
        if zself is None or N == 0:
            out.zero_()
            return out, seg_max, seg_den
    check(L.jmac_softmax_parts_merge_f32(arr(0), d, arr(1), arr(2), arr(3), n, N, d,
                                         ptr(zself) if zself is not None else None, zself.stride(0) if zself is not None else 0,
                                         ptr(rz_loop) if zself is not None else None, float(out_scale), ptr(out), d, ptr(seg_max),
                                         ptr(seg_den), stream()), "jmac_softmax_parts_merge_f32")
    return out, seg_max, seg_den


class _RelAttnAggregateSplit(torch.autograd.Function):
    """Same op with P [N_dst, d] and QZ [N_src, 2d] as separate tables (destination-sharded multi-GPU: P holds
    the rank's rows, QZ the all-gathered table).  The fused self term reads QZ[self_off + i] (loop_rel < 0: none)."""

    @staticmethod
    def forward(ctx, P, QZ, RR, a, graph: RelGraph, slope: float, out_scale: float, loop_rel: int, self_off: int):
        require_device(P, QZ, RR, a)
        P, QZ, RR, a = _f32c(P).contiguous(), _f32c(QZ).contiguous(), _f32c(RR).contiguous(), _f32c(a).contiguous()
        out, seg_max, seg_den = rel_attn_split_fwd_raw(P, QZ, RR, a, graph, slope, out_scale, loop_rel, self_off)
        ctx.save_for_backward(P, QZ, RR, a, out, seg_max, seg_den)
        ctx.graph, ctx.slope, ctx.out_scale, ctx.loop_rel, ctx.self_off = graph, slope, out_scale, int(loop_rel), int(self_off)
        return out

    @staticmethod
    def backward(ctx, G):
        P, QZ, RR, a, out, seg_max, seg_den = ctx.saved_tensors
        dP, dQZ, dRR, da = rel_attn_split_bwd_raw(P, QZ, RR, a, ctx.graph, ctx.slope, ctx.out_scale, ctx.loop_rel, ctx.self_off,
                                                  out, seg_max, seg_den, G)
        return dP, dQZ, dRR, da, None, None, None, None, None


def rel_attn_aggregate_split(P, QZ, RR, a, graph: RelGraph, slope: float, out_scale: float = 1.0, loop_rel: int = -1,
                             self_off: int = 0) -> torch.Tensor:
    return _RelAttnAggregateSplit.apply(P, QZ, RR, a, graph, float(slope), float(out_scale), int(loop_rel), int(self_off))


class _BnTanh(torch.autograd.Function):
    """tanh(BatchNorm1d(x)) with nn.BatchNorm1d semantics (src/jmac_model.py:52)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training: bool, momentum: float, eps: float):
        require_device(x, weight, bias)
        x = _f32c(x).contiguous()
        N, d = x.shape
        L = lib()
        dev = x.device
        y = torch.empty_like(x)
        save_mean = torch.empty(d, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(d, dtype=torch.float32, device=dev)
        ws_bytes = int(L.jmac_bn_tanh_workspace_bytes(N, d))
        ws = _ws(ws_bytes, dev)
        check(L.jmac_bn_tanh_fwd_f32(ptr(x), d, N, d, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
                                     1 if training else 0, float(momentum), float(eps), ptr(y), d, ptr(save_mean),
                                     ptr(save_invstd), ptr(ws), ws_bytes, stream()), "jmac_bn_tanh_fwd_f32")
        ctx.save_for_backward(x, y, weight, save_mean, save_invstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, save_mean, save_invstd = ctx.saved_tensors
        N, d = x.shape
        L = lib()
        dev = x.device
        gy = _f32c(gy).contiguous()
        gx = torch.empty_like(x)
        gbw = torch.empty(2 * d, dtype=torch.float32, device=dev)        # [grad bias | grad weight]: one reduction writes both
        gb, gw = gbw[:d], gbw[d:]
        ws_bytes = int(L.jmac_bn_tanh_workspace_bytes(N, d))
        ws = _ws(ws_bytes, dev)
        check(L.jmac_bn_tanh_bwd_f32(ptr(x), d, ptr(y), d, ptr(gy), d, N, d, ptr(weight), ptr(save_mean), ptr(save_invstd),
                                     1 if ctx.training else 0, ptr(gx), d, ptr(gw), ptr(gb), ptr(ws), ws_bytes, stream()),
              "jmac_bn_tanh_bwd_f32")
        return gx, gw, gb, None, None, None, None, None


def bn_tanh(x, weight, bias, running_mean, running_var, training: bool, momentum: float = 0.1, eps: float = 1e-5):
    if x.shape[1] % 4 != 0:
        raise ValueError("bn_tanh needs d % 4 == 0 (pad on the host)")
    return _BnTanh.apply(x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps))


# ---- small fp32 GEMM (relation-side projections) -----------------------------------------------------------------
def _gemm(A, ta, B, tb, M, N, K):
    C_ = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(lib().jmac_gemm_f32(ptr(A), A.stride(0), 1 if ta else 0, ptr(B), B.stride(0), 1 if tb else 0, M, N, K, ptr(C_), N,
                              stream()), "jmac_gemm_f32")
    return C_


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


class _SmallMM(torch.autograd.Function):
    """C = A @ B on jmac_gemm_f32 with its two backward forms (dA = G B^T, dB = A^T G)."""

    @staticmethod
    def forward(ctx, A, B):
        require_device(A, B)
        A, B = _rowmajor(_f32c(A)), _rowmajor(_f32c(B))
        if A.shape[1] != B.shape[0]:
            raise ValueError("small_mm: %s @ %s" % (tuple(A.shape), tuple(B.shape)))
        ctx.save_for_backward(A, B)
        return _gemm(A, False, B, False, A.shape[0], B.shape[1], A.shape[1])

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        G = _rowmajor(_f32c(G))
        M, K = A.shape
        N = B.shape[1]
        dA = _gemm(G, False, B, True, M, K, N) if ctx.needs_input_grad[0] else None      # [M,N] x [K,N]^T
        dB = _gemm(A, True, G, False, K, N, M) if ctx.needs_input_grad[1] else None      # [M,K]^T x [M,N]
        return dA, dB


def _gemm_any(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A @ B on jmac_gemm_f32 for operands that may be transposed views (``x.t()`` is passed as a transposed operand,
    not copied)."""
    def operand(t):
        if t.stride(1) == 1:
            return t, False
        if t.stride(0) == 1:
            return t.t(), True                                   # stored [cols, rows] row-major
        return t.contiguous(), False
    (a, ta), (b, tb) = operand(A), operand(B)
    return _gemm(a, ta, b, tb, A.shape[0], B.shape[1], A.shape[1])


# Row bound below which the layer routes its relation-side products (962 rows per DBP-5L KG) through small_mm.
# 0 = off (default): measured inside the training step the kernel takes 10.5-15 us per product against the library's
# 17 us untuned -- and 8-9 us once torch's TunableOp has picked the library kernel for the shape (bench.py turns it
# on), so the library keeps these.  The op stays available (deterministic summation order, any strides / transposes).
SMALL_MM_MAX_ROWS = 0


def small_mm(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``torch.mm`` for the relation-side products ([~10^3, d] x [d, d..2d]): one 32x32 tile per 4-wave block, K split
    over the waves, fragments straight from L2 -- ~4 us where the library GEMM takes ~17 us."""
    return _SmallMM.apply(A, B)


# ---- row L2 normalisation (F.normalize(x, 2, -1)) ------------------------------------------------------------------
class _RowNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        require_device(x)
        x = _rowmajor(_f32c(x))
        N, d = x.shape
        y = torch.empty((N, d), dtype=torch.float32, device=x.device)
        inv = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
        check(lib().jmac_row_normalize_fwd_f32(ptr(x), x.stride(0), N, d, float(eps), ptr(y), d, ptr(inv), stream()),
              "jmac_row_normalize_fwd_f32")
        ctx.save_for_backward(y, inv)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        g = _rowmajor(_f32c(g))
        N, d = y.shape
        gx = torch.empty((N, d), dtype=torch.float32, device=y.device)
        check(lib().jmac_row_normalize_bwd_f32(ptr(y), d, ptr(g), g.stride(0), ptr(inv), N, d, ctx.eps, ptr(gx), d, stream()),
              "jmac_row_normalize_bwd_f32")
        return gx, None


def row_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """``F.normalize(x, 2, -1)`` for a 2-D fp32 tensor: one kernel each way instead of norm / clamp / div and their
    six-kernel backward."""
    return _RowNormalize.apply(x, float(eps))


# ---- fp32 GEMM on the bf16 matrix cores (split-bf16, fp32-level accuracy) -------------------------------------------
def gemm_nt_x3(A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``A @ B.t()`` for fp32 row-major A [M,K], B [N,K] on jmac_gemm_nt_x3_f32 (three-term bf16 split, six products,
    fp32 accumulation).  K % 4 == 0; rows 16-byte aligned."""
    require_device(A, B)
    A, B = _rowmajor(_f32c(A)), _rowmajor(_f32c(B))
    M, K = A.shape
    N = B.shape[0]
    if B.shape[1] != K:
        raise ValueError("gemm_nt_x3: %s x %s^T" % (tuple(A.shape), tuple(B.shape)))
    C_ = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    # the testing library only (measured, rejected: DESIGN.md section 5): never the product library
    check(testing_lib().jmac_gemm_nt_x3_f32(ptr(A), A.stride(0), ptr(B), B.stride(0), M, N, K, ptr(C_), C_.stride(0), stream()),
          "jmac_gemm_nt_x3_f32")
    return C_


class _MMx3(torch.autograd.Function):
    """``A @ W`` for an N-row activation A [M,K] and a weight W [K,N]: forward and dA on the split-bf16 matrix-core GEMM
    (both are "NT" products with a k-contiguous weight operand: W^T for the forward, W itself for dA = G W^T); dW = A^T G
    contracts over the M rows (both operands k-strided) and stays on the library GEMM."""

    @staticmethod
    def forward(ctx, A, W):
        ctx.save_for_backward(A, W)
        return gemm_nt_x3(A, W.t().contiguous())

    @staticmethod
    def backward(ctx, G):
        A, W = ctx.saved_tensors
        G = _rowmajor(G)
        dA = gemm_nt_x3(G, W) if ctx.needs_input_grad[0] else None
        dW = torch.mm(A.t(), G) if ctx.needs_input_grad[1] else None
        return dA, dW


def mm_x3(A: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """``torch.mm(A, W)`` on the split-bf16 GEMM (with its backward).  EXPERIMENTAL, not used by the layer or the encoder:
    per product it is as accurate as an fp32 GEMM (3e-7 of sum |a||b|) and as fast as the tuned library kernel (~100
    TFLOP/s at DBP-5L size, 128 vs 101 at 10^6 rows), but v_mfma_f32_32x32x16_bf16 accumulates with a small NEGATIVE BIAS
    (mean signed error -1.3e-8 .. -2.2e-8 of the result on positive data against 2e-10 for the fp32 GEMM, growing with K):
    invisible per element, it adds up coherently in the column sums of the backward -- on the full-size DBP-5L step the
    worst parameter gradient moved from 6e-6 to 7e-4 of its scale, outside the 1e-4 bar."""
    return _MMx3.apply(A, W)
