"""Destination-sharded RelationAwareLayer over several GPUs (BASELINE config 4).

The per-destination softmax and sum (modules/helper/message_passing.py:24,28 both index on
``edge_index[0]``) make destination rows the natural shard: a rank that owns a contiguous row range owns
every incoming edge of those rows, needs no cross-GPU softmax and writes a disjoint output slab.

Per layer and rank:   x_loc [n_r, d]  --GEMM-->  [P|Q|Z]_loc  --all-gather [Q|Z] over xGMI-->  QZ [world*n_max, 2d]
                      aggregate own rows (HIP kernels)  -->  + self term  -->  BN with all-reduced [2,d] stats
Backward: reduce-scatter of d[Q|Z] (adjoint of the all-gather), all-reduce of the tiny parameter grads
(done by the caller / DDP-style helper ``allreduce_grads``).

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL).  The direct all-gather uses all seven
xGMI links of a GPU at once, unlike a ring all-reduce which is bound by one link.  Graphs of DBP-5L scale
are too small to shard profitably: run replicas instead.

The rank-local compute is pluggable (``local_aggregate``) only so that the CPU test-suite can exercise the
partitioning and the collectives under ``gloo`` with a stand-in; the default is the HIP op and nothing in
this module falls back to it silently.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

MAX_CHUNKS = 16          # JMAC_MERGE_MAX_PARTS of include/jmac_hip.h: parts one jmac_softmax_parts_merge_f32 call merges
OVERLAP_BACKWARD = True  # slab-pipelined exchange: queue slab c's reduce-scatter while pass B sums slab c + 1 (False: all after)
OVERLAP_COUNT = 0        # tests: backwards that took the overlapped form
HANDOFF_COUNT = 0        # tests: exchanges whose backward found its gradient already reduce-scattered


# ------------------------------------------------------------------------------------------------
# partitioning
# ------------------------------------------------------------------------------------------------
def partition_rows(in_degree: np.ndarray, world: int, node_weight: float = 1.0) -> np.ndarray:
    """Contiguous destination ranges balanced by work = edges + node_weight * rows.
    Returns boundaries b[0..world] with b[0] = 0, b[world] = N."""
    n = int(in_degree.shape[0])
    work = np.cumsum(in_degree.astype(np.float64) + node_weight)
    total = work[-1] if n else 0.0
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(work, total * r / world, side="left")) if n else 0)
    b.append(n)
    b = np.maximum.accumulate(np.asarray(b, dtype=np.int64))
    return b


class ShardedGraph:
    """Rank-local part of a typed edge list: edges whose destination lies in [lo, hi).

    Destinations are re-indexed locally (dst - lo); sources are re-indexed into the PADDED all-gather
    layout, position = owner_rank * n_max + (src - bounds[owner]), so the gathered [world*n_max, 2d] table
    is used as it arrives.

    ``chunks`` > 1 (slab-pipelined exchange): every rank's slab of n_max rows is cut into ``chunks`` row ranges
    [cb[c], cb[c+1]); the table is laid out CHUNK-major -- position = world * cb[c] + owner * rows_c + (local - cb[c]) --
    so that the all-gather of chunk c (every rank contributes its rows of that range: all xGMI links busy, like the
    one-piece all-gather) fills ONE contiguous slice of the table, and the rank's edges are split by the chunk of their
    SOURCE into ``chunks`` sub-graphs over the same destinations (``chunk_graph``).  The rank's OWN rows are not contiguous in
    that layout, so the table carries a copy of them behind the gathered part (rows [world*n_max, (world+1)*n_max):
    ``self_off``) for the fused self loop of the backward kernel."""

    def __init__(self, edge_index: np.ndarray, edge_type: np.ndarray, bounds: Sequence[int], rank: int,
                 already_local: bool = False, chunks: int = 1):
        bounds = np.asarray(bounds, dtype=np.int64)
        self.bounds, self.rank, self.world = bounds, rank, len(bounds) - 1
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.n_local = self.hi - self.lo
        self.n_max = int(np.max(bounds[1:] - bounds[:-1])) if self.world else 0
        self.n_global = int(bounds[-1])
        dst, src = np.asarray(edge_index[0]), np.asarray(edge_index[1])
        if not already_local:
            keep = (dst >= self.lo) & (dst < self.hi)
            dst, src, edge_type = dst[keep], src[keep], np.asarray(edge_type)[keep]
        owner = np.searchsorted(bounds, src, side="right") - 1
        local = src - bounds[owner]
        self.dst_local = (dst - self.lo).astype(np.int64)
        if int(chunks) > MAX_CHUNKS:       # fail here, not after every chunk's all-gather and partial pass has been queued
            raise ValueError("ShardedGraph: chunks=%d exceeds the %d parts jmac_softmax_parts_merge_f32 merges (JMAC_MERGE_MAX_PARTS)"
                             % (int(chunks), MAX_CHUNKS))
        self.chunks = max(1, min(int(chunks), max(self.n_max, 1)))
        cb = np.asarray([(k * self.n_max) // self.chunks for k in range(self.chunks + 1)], dtype=np.int64)
        self.chunk_bounds = cb
        if self.chunks == 1:
            self.src_chunk = np.zeros(local.shape[0], dtype=np.int64)
            self.src_padded = (owner * self.n_max + local).astype(np.int64)
        else:
            ch = np.searchsorted(cb, local, side="right") - 1
            self.src_chunk = ch.astype(np.int64)
            self.src_padded = (self.world * cb[ch] + owner * (cb[ch + 1] - cb[ch]) + (local - cb[ch])).astype(np.int64)
        self.edge_type = np.asarray(edge_type).astype(np.int64)
        self.E_local = int(self.dst_local.shape[0])
        self._chunk_edges = np.bincount(self.src_chunk, minlength=self.chunks)
        # rows of the table the kernels index / where the rank's own rows sit in it
        self.table_rows = self.world * self.n_max + (self.n_max if self.chunks > 1 else 0)
        self.self_off = self.world * self.n_max if self.chunks > 1 else self.rank * self.n_max
        self._rel_graph = None
        self._chunk_graphs: dict = {}

    def coo(self, device, chunk: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if chunk is None:
            d, s, t = self.dst_local, self.src_padded, self.edge_type
        else:
            sel = self.src_chunk == chunk
            d, s, t = self.dst_local[sel], self.src_padded[sel], self.edge_type[sel]
        return torch.from_numpy(np.stack([d, s])).to(device), torch.from_numpy(t).to(device)

    def chunk_edges(self, chunk: int) -> int:
        return int(self._chunk_edges[chunk])

    def rel_graph(self, device, num_rel: int):
        """CSR + schedules on the HIP device (cached)."""
        if self._rel_graph is None:
            from .graph import RelGraph
            ei, et = self.coo(device)
            self._rel_graph = RelGraph(ei, et, self.n_local, num_rel, None, num_src=self.table_rows)
        return self._rel_graph

    def chunk_graph(self, chunk: int, device, num_rel: int):
        """CSR + schedules of the edges whose source lies in ``chunk`` (same destinations, same table positions)."""
        if chunk not in self._chunk_graphs:
            from .graph import RelGraph
            ei, et = self.coo(device, chunk)
            self._chunk_graphs[chunk] = RelGraph(ei, et, self.n_local, num_rel, None, num_src=self.table_rows)
        return self._chunk_graphs[chunk]


# ------------------------------------------------------------------------------------------------
# differentiable collectives
# ------------------------------------------------------------------------------------------------
# tests set this to run the collectives even when the group has one rank (exercises the RCCL entry points on a
# single-GPU box); the product leaves it False and short-circuits world == 1
FORCE_COLLECTIVES = False

# bench.py sets this to a list to collect (name, start_event, end_event) around the data-path collectives; the events are
# recorded on the compute stream, which waits for the collective's completion (RCCL runs on its own stream)
COMM_PROFILE = None


def _cev(x: torch.Tensor):
    if COMM_PROFILE is None or not x.is_cuda:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _cdone(name: str, e0, x: torch.Tensor) -> None:
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        COMM_PROFILE.append((name, e0, e1, x.numel() * x.element_size()))


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _skip(group=None) -> bool:
    return _world(group) == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def _all_reduce(t: torch.Tensor, group=None) -> None:
    """In-place sum. RCCL for device tensors; under gloo (tests) device tensors are staged through the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


class _AllGatherRows(torch.autograd.Function):
    """[n_max, w] per rank -> [world*n_max, w]; backward = reduce-scatter (sum) of the gradient.

    ``defer``: start the collective and return at once -- RCCL runs it on its own stream; the caller enqueues independent
    work on the compute stream and calls ``pending_wait()`` before the first kernel that reads the gathered table."""

    pending = None          # (work handle, profile start event, gathered buffer, fp32 output or None) of a deferred all-gather

    @staticmethod
    def forward(ctx, x, group, defer=False, wire_dtype=None):
        ctx.group = group
        world = _world(group)
        if _skip(group):
            if wire_dtype is not None and wire_dtype != x.dtype:
                return x.to(wire_dtype).to(x.dtype)          # one rank: the same rounding as the wire would apply
            return x
        x = x.contiguous()
        final = None
        if wire_dtype is not None and wire_dtype != x.dtype:
            # reduced-precision WIRE format: the table travels as bf16 (half the xGMI bytes) and is widened again on
            # arrival; everything downstream (gathers, logits, sums) stays fp32 arithmetic on the rounded values
            final = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            x = x.to(wire_dtype)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        e0 = _cev(x)
        if x.is_cuda and dist.get_backend(group) == "gloo":           # tests only: stage through the host
            h = torch.empty(out.shape, dtype=x.dtype)
            dist.all_gather_into_tensor(h, x.cpu(), group=group)
            out.copy_(h)
        elif defer:
            work = dist.all_gather_into_tensor(out, x, group=group, async_op=True)
            _AllGatherRows.pending = (work, e0, out, final)
            return final if final is not None else out
        else:
            dist.all_gather_into_tensor(out, x, group=group)
        _cdone("all_gather_qz", e0, out)
        if final is not None:
            final.copy_(out)
            return final
        return out

    @staticmethod
    def pending_wait() -> None:
        """Make the compute stream wait for the deferred all-gather (no host block under RCCL); a reduced-precision wire
        buffer is widened into the fp32 table the caller already holds."""
        p, _AllGatherRows.pending = _AllGatherRows.pending, None
        if p is not None:
            work, e0, out, final = p
            work.wait()
            _cdone("all_gather_qz", e0, out)
            if final is not None:
                # outside forward() grad mode is on and ``final`` is the Function's OUTPUT: a recorded in-place copy would
                # put a CopyBackwards node between the table and its consumers and hand backward() an all-zero gradient
                with torch.no_grad():
                    final.copy_(out)

    @staticmethod
    def backward(ctx, g):
        world = _world(ctx.group)
        if _skip(ctx.group):
            return g, None, None, None
        g = g.contiguous()
        n = g.shape[0] // world
        if dist.get_backend(ctx.group) == "gloo":          # gloo has no reduce_scatter: all-reduce + slice
            _all_reduce(g, ctx.group)
            r = dist.get_rank(ctx.group)
            return g[r * n:(r + 1) * n].clone(), None, None, None
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        e0 = _cev(g)
        dist.reduce_scatter_tensor(out, g, group=ctx.group)
        _cdone("reduce_scatter_dqz", e0, g)
        return out, None, None, None


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        if _skip(group):
            return x
        y = x.clone()
        _all_reduce(y, group)
        return y

    @staticmethod
    def backward(ctx, g):
        if _skip(ctx.group):
            return g, None
        g = g.clone()
        _all_reduce(g, ctx.group)
        return g, None


def all_gather_rows(x: torch.Tensor, group=None, defer: bool = False, wire_dtype=None) -> torch.Tensor:
    return _AllGatherRows.apply(x, group, bool(defer), wire_dtype)


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    return _AllReduceSum.apply(x, group)


def allreduce_grads(params, group=None) -> None:
    """Sum the (replicated) parameters' gradients over ranks, one flat bucket."""
    if _skip(group):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    _all_reduce(flat, group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def _bn_momentum(bn: nn.BatchNorm1d) -> float:
    """nn.BatchNorm1d(momentum=None) keeps a cumulative moving average (factor 1/num_batches_tracked, a device-side
    count); the reference never builds such a module (src/jmac_model.py:27) and the sharded path does not carry it."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not supported on the sharded path")
    return float(bn.momentum)


def sync_batch_norm(x: torch.Tensor, n_global: int, bn: nn.BatchNorm1d, group=None) -> torch.Tensor:
    """BatchNorm1d(x) with batch statistics over the rows of ALL ranks (the bn of src/jmac_model.py:52):
    the [2,d] column sums are all-reduced; running statistics are updated like nn.BatchNorm1d."""
    if not (bn.training or not bn.track_running_stats):
        return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    sums = torch.stack([x.sum(0), (x * x).sum(0)])
    sums = all_reduce_sum(sums, group)
    mean = sums[0] / n_global
    var = (sums[1] / n_global - mean * mean).clamp_min(0)
    if bn.track_running_stats:
        with torch.no_grad():
            m = _bn_momentum(bn)
            bn.running_mean.mul_(1 - m).add_(mean.detach(), alpha=m)
            unb = var.detach() * (n_global / max(n_global - 1, 1))
            bn.running_var.mul_(1 - m).add_(unb, alpha=m)
            bn.num_batches_tracked.add_(1)
    return (x - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias


# ------------------------------------------------------------------------------------------------
# BatchNorm + tanh with batch statistics over the rows of ALL ranks, on the fused kernels
# ------------------------------------------------------------------------------------------------
class _HipBN:
    """Rank-local phases (libjmac_hip.so, include/jmac_hip.h "phased forms"); tests inject a torch stand-in under gloo."""

    @staticmethod
    def moments(x):
        from ._lib import check, lib, ptr, stream
        n, d = x.shape
        mean = torch.empty(d, dtype=torch.float32, device=x.device)
        m2 = torch.empty(d, dtype=torch.float32, device=x.device)
        wsb = int(lib().jmac_bn_tanh_workspace_bytes(n, d))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        check(lib().jmac_col_moments_f32(ptr(x), x.stride(0), n, d, ptr(mean), ptr(m2), ptr(ws), wsb, stream()), "jmac_col_moments_f32")
        return mean, m2

    @staticmethod
    def apply(x, weight, bias, mean, invstd):
        from ._lib import check, lib, ptr, stream
        n, d = x.shape
        y = torch.empty((n, d), dtype=torch.float32, device=x.device)
        check(lib().jmac_bn_tanh_apply_f32(ptr(x), x.stride(0), n, d, ptr(weight), ptr(bias), ptr(mean), ptr(invstd), ptr(y), d,
                                           stream()), "jmac_bn_tanh_apply_f32")
        return y

    @staticmethod
    def bwd_sums(x, y, gy, mean, invstd):
        from ._lib import check, lib, ptr, stream
        n, d = x.shape
        sums = torch.empty(2 * d, dtype=torch.float32, device=x.device)
        wsb = int(lib().jmac_bn_tanh_workspace_bytes(n, d))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        check(lib().jmac_bn_tanh_bwd_sums_f32(ptr(x), x.stride(0), ptr(y), d, ptr(gy), gy.stride(0), n, d, ptr(mean), ptr(invstd),
                                              ptr(sums), ptr(ws), wsb, stream()), "jmac_bn_tanh_bwd_sums_f32")
        return sums

    @staticmethod
    def bwd_apply(x, y, gy, weight, mean, invstd, sums, n_total):
        from ._lib import check, lib, ptr, stream
        n, d = x.shape
        gx = torch.empty((n, d), dtype=torch.float32, device=x.device)
        check(lib().jmac_bn_tanh_bwd_apply_f32(ptr(x), x.stride(0), ptr(y), d, ptr(gy), gy.stride(0), n, d, ptr(weight), ptr(mean),
                                               ptr(invstd), ptr(sums), int(n_total), ptr(gx), d, stream()), "jmac_bn_tanh_bwd_apply_f32")
        return gx


def _all_gather_small(t: torch.Tensor, group=None) -> torch.Tensor:
    """[k] per rank -> [world, k]."""
    return _all_gather_padded(t.view(1, -1), 1, group).view(_world(group) if not _skip(group) else 1, -1)


class _SyncBnTanh(torch.autograd.Function):
    """tanh(BatchNorm1d(x)) (src/jmac_model.py:52) with TRAIN-mode statistics over the rows of all ranks.

    forward : local (mean, M2) -> all-gather of [2d+1] per rank -> Chan's parallel combination -> fused apply
    backward: local sums of gz and gz*xhat -> all-reduce -> fused apply with the global row count.
    grad weight / grad bias are the RANK's contributions (allreduce_grads sums the replicated parameters' grads)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, n_global, group, kernels):
        x = x.contiguous()
        n_loc, d = x.shape
        if n_loc > 0:
            mean_l, m2_l = kernels.moments(x)
        else:
            mean_l, m2_l = x.new_zeros(d), x.new_zeros(d)
        packed = torch.cat([mean_l, m2_l, x.new_full((1,), float(n_loc))])
        allp = _all_gather_small(packed, group).double()                     # [world, 2d+1]
        cnt = allp[:, 2 * d]
        n = cnt.sum()
        mean = (allp[:, :d] * cnt.view(-1, 1)).sum(0) / n
        m2 = allp[:, d:2 * d].sum(0) + (cnt.view(-1, 1) * (allp[:, :d] - mean) ** 2).sum(0)
        var = m2 / n
        mean_f, invstd = mean.float(), torch.rsqrt(var + eps).float()
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean_f, alpha=momentum)
                unb = (var * (n / torch.clamp(n - 1, min=1))).float()
                running_var.mul_(1 - momentum).add_(unb, alpha=momentum)
                num_batches_tracked.add_(1)
        y = kernels.apply(x, weight, bias, mean_f, invstd)
        ctx.save_for_backward(x, y, weight, mean_f, invstd)
        ctx.group, ctx.kernels, ctx.n_total = group, kernels, int(n_global)   # host-side count: no device sync
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        gy = gy.contiguous()
        d = x.shape[1]
        sums_l = ctx.kernels.bwd_sums(x, y, gy, mean, invstd) if x.shape[0] > 0 else x.new_zeros(2 * d)
        sums = sums_l.clone()
        if not _skip(ctx.group):
            _all_reduce(sums, ctx.group)
        gx = ctx.kernels.bwd_apply(x, y, gy, weight, mean, invstd, sums, ctx.n_total) if x.shape[0] > 0 else torch.zeros_like(x)
        return gx, sums_l[d:].clone(), sums_l[:d].clone(), None, None, None, None, None, None, None, None


def sync_bn_tanh(x: torch.Tensor, bn: nn.BatchNorm1d, n_global: int, group=None, kernels=None) -> torch.Tensor:
    """tanh(bn(x)) with batch statistics over ALL ranks' rows (n_global of them); eval mode uses the running
    statistics (row-local)."""
    kernels = kernels or _HipBN
    if not (bn.training or not bn.track_running_stats):
        inv = torch.rsqrt(bn.running_var + bn.eps)
        return _EvalBnTanh.apply(x, bn.weight, bn.bias, bn.running_mean, inv, kernels)
    track = bn.track_running_stats
    return _SyncBnTanh.apply(x, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                             bn.num_batches_tracked if track else None, _bn_momentum(bn),
                             bn.eps, int(n_global), group, kernels)


class _EvalBnTanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mean, invstd, kernels):
        x = x.contiguous()
        y = kernels.apply(x, weight, bias, mean.contiguous(), invstd.contiguous())
        ctx.save_for_backward(x, y, weight, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        gz = gy * (1 - y * y)
        xhat = (x - mean) * invstd
        return gz * (weight * invstd), (gz * xhat).sum(0), gz.sum(0), None, None, None


# ------------------------------------------------------------------------------------------------
# slab-pipelined exchange: the all-gather in row chunks, chunk c aggregated while chunk c+1 is on the links
# ------------------------------------------------------------------------------------------------
class _HipChunked:
    """HIP kernels of the pipelined path (tests inject a torch stand-in with the same three methods under gloo on CPU)."""

    @staticmethod
    def partial(P, table, RR, a, sg: "ShardedGraph", chunk: int, slope: float):
        from . import ops
        g = sg.chunk_graph(chunk, P.device, RR.shape[0])
        out, m, l = ops.rel_attn_split_fwd_raw(P, table, RR, a, g, slope, 1.0, -1, 0)
        return out, m, l, g.rowptr

    @staticmethod
    def merge(parts, n: int, d: int, device, zself, rz_loop, out_scale: float):
        from . import ops
        return ops.softmax_parts_merge(parts, n, d, device, zself, rz_loop, out_scale)

    @staticmethod
    def backward(P, table, RR, a, sg: "ShardedGraph", slope: float, pre, seg_max, seg_den, G):
        from . import ops
        return ops.rel_attn_split_bwd_raw(P, table, RR, a, sg.rel_graph(P.device, RR.shape[0]), slope, 0.5, RR.shape[0] - 1,
                                          sg.self_off, pre, seg_max, seg_den, G)

    @staticmethod
    def backward_phased(P, table, RR, a, sg: "ShardedGraph", slope: float, pre, seg_max, seg_den, G, slab_bounds):
        """The same backward as separately launched phases (ops.SplitBackwardPhases): ``begin()`` = pass A, pass C and their
        merges; ``slab(c)`` = pass B on table rows [slab_bounds[c], slab_bounds[c+1]) -> that slice of d table, final in stream
        order -- the caller reduce-scatters it while the next slab is summed.  Attributes dP, dQZ, dRR, da."""
        from . import ops
        return ops.SplitBackwardPhases(P, table, RR, a, sg.rel_graph(P.device, RR.shape[0]), slope, 0.5, RR.shape[0] - 1, sg.self_off,
                                       pre, seg_max, seg_den, G, slab_bounds)


class _ChunkedAllGather(torch.autograd.Function):
    """[n_max, w] per rank -> the CHUNK-major [world*n_max, w] table, as ``sg.chunks`` all-gathers queued at once (RCCL runs
    them in order on its own stream, each one on all links).  Returns at once: ``pending[c]`` is chunk c's work handle, which
    ``_ChunkedAggregate`` waits for one by one.  Backward: one reduce-scatter per chunk slice."""

    _handoff = None         # forward -> chunked_all_gather(), within one call: the work handles of the table being returned

    @staticmethod
    def forward(ctx, x, sg: ShardedGraph, group):
        ctx.sg, ctx.group = sg, group
        world = _world(group)
        cb = sg.chunk_bounds
        x = x.contiguous()
        if x.shape[0] != sg.n_max:
            raise ValueError("the rank's table must be padded to n_max rows")
        works = [None] * sg.chunks
        table = torch.empty((sg.table_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _ChunkedAllGather._handoff = works                   # chunked_all_gather() hangs them on the returned table at once
        table[world * sg.n_max:].copy_(x)                    # the rank's own rows once more, contiguous (fused self loop)
        if _skip(group):
            table[:sg.n_max].copy_(x)                        # one rank: the chunk-major table is the input
            return table
        gloo = dist.get_backend(group) == "gloo"
        for c in range(sg.chunks):                           # every chunk's collective is queued now, in order
            dst, src = table[world * cb[c]:world * cb[c + 1]], x[cb[c]:cb[c + 1]]
            if src.shape[0] == 0:
                continue
            e0 = _cev(src)
            if gloo and src.is_cuda:                         # tests only: stage through the host
                h = torch.empty(dst.shape, dtype=src.dtype)
                dist.all_gather_into_tensor(h, src.cpu(), group=group)
                dst.copy_(h)
                _cdone("all_gather_qz_chunk", e0, dst)
            elif gloo:
                dist.all_gather_into_tensor(dst, src, group=group)
            else:
                works[c] = (dist.all_gather_into_tensor(dst, src, group=group, async_op=True), e0, dst)
        return table

    @staticmethod
    def wait(table, chunk: int) -> None:
        """Make the compute stream wait for chunk ``chunk`` of ``table`` (no host block under RCCL).  The work handles ride on
        the table tensor itself (``_jmac_pending``): a forward that raises between the exchange and the aggregation leaves
        nothing behind for the next forward to overwrite."""
        p = getattr(table, "_jmac_pending", None)
        if p is not None and p[chunk] is not None:
            work, e0, dst = p[chunk]
            p[chunk] = None
            work.wait()
            _cdone("all_gather_qz_chunk", e0, dst)

    @staticmethod
    def backward(ctx, g):
        sg, group = ctx.sg, ctx.group
        world = _world(group)
        # _ChunkedAggregate.backward may already have reduce-scattered this very gradient, slab by slab.  The hand-off is valid only
        # while ``g`` is still that tensor, untouched: a second differentiable consumer of the table makes autograd accumulate
        # its gradient onto ``g`` IN PLACE (the attribute survives, the version counter moves) -- then the full reduce below is the
        # right thing, because ``g`` is the complete sum.
        hand = getattr(g, "_jmac_reduced", None)
        if hand is not None:
            g._jmac_reduced = None
            done, version, ptr = hand
            if g._version == version and g.data_ptr() == ptr:
                global HANDOFF_COUNT
                HANDOFF_COUNT += 1
                return done, None, None
        own = g[world * sg.n_max:]                           # gradient of the own-rows copy (the self loop's dZ)
        if _skip(group):
            return g[:sg.n_max] + own, None, None
        cb = sg.chunk_bounds
        g = g.contiguous()
        out = torch.empty((sg.n_max,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        r = dist.get_rank(group)
        gloo = dist.get_backend(group) == "gloo"
        for c in range(sg.chunks):                           # adjoint of each chunk's all-gather
            rows = int(cb[c + 1] - cb[c])
            if rows == 0:
                continue
            gc = g[world * cb[c]:world * cb[c + 1]]
            if gloo:                                         # gloo has no reduce_scatter: all-reduce + slice
                gc = gc.clone()
                _all_reduce(gc, group)
                out[cb[c]:cb[c + 1]] = gc[r * rows:(r + 1) * rows]
            else:
                e0 = _cev(gc)
                dist.reduce_scatter_tensor(out[cb[c]:cb[c + 1]], gc, group=group)
                _cdone("reduce_scatter_dqz_chunk", e0, gc)
        out += own
        return out, None, None


class _ChunkedAggregate(torch.autograd.Function):
    """pre[i] = (sqrt(deg_i) * sum_e alpha_e (Z[j] - Rz[t]) + Z[i] - Rz[loop]) / 2 over the rank's rows from a table whose chunks
    are still arriving: the compute stream waits for chunk c only, aggregates the sub-graph of sources in chunk c into a partial
    (out_c, max_c, den_c) and goes on to c+1 while that chunk is on the links; one merge pass combines the partials per
    destination, adds the self loop and halves (jmac_softmax_parts_merge_f32), and yields the whole graph's (max, den) -- so the
    BACKWARD is the ordinary fused one on the whole graph and the whole table.  Price: ``chunks`` partial [n, d] outputs written
    and read once more (the merge)."""

    @staticmethod
    def forward(ctx, P, table, RR, a, sg: ShardedGraph, slope: float, kernels):
        P, RR, a = P.contiguous(), RR.contiguous(), a.contiguous()
        n, d = sg.n_local, P.shape[1]
        parts = []
        try:
            for c in range(sg.chunks):
                _ChunkedAllGather.wait(table, c)
                if n > 0 and sg.chunk_edges(c) > 0:
                    parts.append(kernels.partial(P, table, RR, a, sg, c, slope))
        finally:                                             # whatever happened: no un-waited collective outlives this call
            for c in range(sg.chunks):
                _ChunkedAllGather.wait(table, c)
            table._jmac_pending = None
        zself = table[sg.self_off:sg.self_off + n, d:]
        pre, seg_max, seg_den = kernels.merge(parts, n, d, P.device, zself, RR[-1, d:].contiguous(), 0.5)
        ctx.save_for_backward(P, table, RR, a, pre, seg_max, seg_den)
        ctx.sg, ctx.slope, ctx.kernels = sg, slope, kernels
        ctx.group = getattr(table, "_jmac_group", None)      # (group,) when the table came from chunked_all_gather
        return pre

    @staticmethod
    def backward(ctx, G):
        global OVERLAP_COUNT
        P, table, RR, a, pre, seg_max, seg_den = ctx.saved_tensors
        sg, kernels = ctx.sg, ctx.kernels
        # the phased form exists to hide d table's reduce-scatters: without a gradient for the table there is nothing to exchange
        if not (OVERLAP_BACKWARD and ctx.group is not None and ctx.needs_input_grad[1] and hasattr(kernels, "backward_phased")):
            dP, dT, dRR, da = kernels.backward(P, table, RR, a, sg, ctx.slope, pre, seg_max, seg_den, G.contiguous())
            return dP, dT, dRR, da, None, None, None
        # Overlapped form (adjoint of the slab-pipelined exchange): pass B -- the by-source sums that produce d table -- runs slab by
        # slab in the table's chunk-major row order, and slab c's reduce-scatter is queued (RCCL: on its own stream, behind the
        # compute stream's work so far) while pass B goes on with slab c + 1.  The last slab is the rank's own-rows copy (the fused
        # self loop's dZ): it stays local.
        (group,) = ctx.group
        world = _world(group)
        cb = sg.chunk_bounds
        bounds = [world * int(b) for b in cb] + [int(table.shape[0])]
        bp = kernels.backward_phased(P, table, RR, a, sg, ctx.slope, pre, seg_max, seg_den, G.contiguous(), bounds)
        bp.begin()
        out = torch.empty((sg.n_max,) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
        skip = _skip(group)
        gloo = (not skip) and dist.get_backend(group) == "gloo"
        r = 0 if skip else dist.get_rank(group)
        works = []
        for c in range(sg.chunks):
            gc = bp.slab(c)                                  # rows [world cb[c], world cb[c+1]) of d table: final in stream order
            rows = int(cb[c + 1] - cb[c])
            if rows == 0:
                continue
            dst = out[cb[c]:cb[c + 1]]
            if skip:
                dst.copy_(gc)
            elif gloo:                                       # gloo has no reduce_scatter (tests): all-reduce + slice, blocking
                t = gc.cpu() if gc.is_cuda else gc.clone()
                _all_reduce(t, group)
                dst.copy_(t[r * rows:(r + 1) * rows])
            else:
                e0 = _cev(gc)
                works.append((dist.reduce_scatter_tensor(dst, gc, group=group, async_op=True), e0, gc))
        own = bp.slab(sg.chunks)                             # the own-rows copy's gradient (self loop)
        for w, e0, gc in works:
            w.wait()                                         # the compute stream waits; no host block
            _cdone("reduce_scatter_dqz_chunk", e0, gc)
        out += own
        dT = bp.dQZ
        # _ChunkedAllGather.backward hands ``out`` on instead of reducing again -- if dT reaches it as it is now (version, storage)
        dT._jmac_reduced = (out, dT._version, dT.data_ptr())
        OVERLAP_COUNT += 1
        return bp.dP, dT, bp.dRR, bp.da, None, None, None


def chunked_all_gather(x: torch.Tensor, sg: ShardedGraph, group=None) -> torch.Tensor:
    table = _ChunkedAllGather.apply(x, sg, group)
    table._jmac_pending, _ChunkedAllGather._handoff = _ChunkedAllGather._handoff, None
    table._jmac_group = (group,)                       # the aggregation's backward queues the reduce-scatters itself (overlap)
    return table


def chunked_aggregate(P, table, RR, a, sg: ShardedGraph, slope: float, kernels=None) -> torch.Tensor:
    return _ChunkedAggregate.apply(P, table, RR, a, sg, float(slope), kernels if kernels is not None else _HipChunked)


# ------------------------------------------------------------------------------------------------
# the sharded layer
# ------------------------------------------------------------------------------------------------
def hip_local_aggregate(P, QZ, RR, a, sg: ShardedGraph, slope: float) -> torch.Tensor:
    """nb[i] = sqrt(deg_i) * sum_e alpha_e (Z[j]-Rz[t]) for the rank's rows, on the HIP kernels."""
    from . import ops
    return ops.rel_attn_aggregate_split(P, QZ, RR, a, sg.rel_graph(P.device, RR.shape[0]), slope, 1.0)


def hip_local_layer(P, QZ, RR, a, sg: ShardedGraph, slope: float) -> torch.Tensor:
    """(nb[i] + Z[i] - Rz[loop]) / 2 for the rank's rows in ONE kernel: the fused self term reads the rank's own rows of
    the gathered table at offset rank * n_max."""
    from . import ops
    return ops.rel_attn_aggregate_split(P, QZ, RR, a, sg.rel_graph(P.device, RR.shape[0]), slope, 0.5,
                                        loop_rel=RR.shape[0] - 1, self_off=sg.rank * sg.n_max)


class ShardedRelationAwareLayer(nn.Module):
    """RelationAwareLayer (src/jmac_model.py:10-53) on a destination-sharded graph.

    ``forward(x_local [n_r, d], rel_emb [nr, d], sg)`` returns the rank's [n_r, d] output slab; parameters
    are replicated (same names as the reference layer), their gradients summed with ``allreduce_grads``.

    Product path (default): the aggregation, the self loop and the /2 are one kernel (``hip_local_layer``), BatchNorm +
    tanh run on the fused phased kernels with the statistics combined across ranks (``sync_bn_tanh``).  With an injected
    ``local_aggregate`` (CPU test double) the torch formulation of the same steps is used."""

    def __init__(self, layer: nn.Module, group=None, local_aggregate: Optional[Callable] = None, bn_kernels=None,
                 wire_dtype=None, chunk_kernels=None):
        super().__init__()
        self.layer = layer                       # a jmac_amd.layer.RelationAwareLayer (holds the parameters)
        self.group = group
        self.local_aggregate = local_aggregate
        self.bn_kernels = bn_kernels
        self.chunk_kernels = chunk_kernels       # None: the HIP kernels; a CPU stand-in in the gloo tests (sg.chunks > 1)
        # None / torch.float32: the [Q|Z] table crosses xGMI in fp32 (default: results equal the one-GPU layer's).
        # torch.bfloat16: it crosses as bf16 -- half the bytes of the exchange that bounds the 8-GPU step (16.8 GB
        # received per GPU and layer at config 4 x 8) -- and is widened on arrival; the gathered Q / Z values then
        # carry bf16 rounding (relative 2^-9 per element; measured on the layer output: tests/test_dist_gloo.py), the
        # rank's own P, the relation tables, logits, softmax, sums and BN stay fp32.  The backward exchanges fp32.
        self.wire_dtype = wire_dtype

    def forward(self, x_local: torch.Tensor, rel_emb: torch.Tensor, sg: ShardedGraph) -> torch.Tensor:
        L = self.layer
        if L.comp_op != "sub":
            raise NotImplementedError("the sharded path covers comp_op='sub' (the factorised form)")
        d = L.out_channels
        if d % 4:
            raise NotImplementedError("sharded path needs out_channels % 4 == 0")
        # The exchange goes first: [Q|Z] of the rank's own rows (1/world of that GEMM), then its all-gather over xGMI is
        # STARTED, and everything that does not read the gathered table -- the P projection, the relation transforms and
        # the [Rq|Rz] projection -- is enqueued behind it on the compute stream and runs while the links are busy.
        d_in = L.in_channels
        wqz = torch.cat([L.w_att[d_in:], L.gcn_weight], dim=1)       # [Wb | Wg]   (w_att = [Wt; Wb], src/jmac_model.py:24)
        QZ_loc = torch.mm(x_local, wqz)                              # [n_r, 2d]
        Z_loc = QZ_loc[:, d:]
        if sg.n_local < sg.n_max:                                    # pad to the common slab height
            QZ_loc = F.pad(QZ_loc, (0, 0, 0, sg.n_max - sg.n_local))
        slope = L.atv_mlp.negative_slope
        if sg.chunks > 1:
            # slab-pipelined exchange: the table crosses in sg.chunks row chunks, chunk c is aggregated while c+1 is on the
            # links (_ChunkedAllGather + _ChunkedAggregate)
            if self.wire_dtype is not None and self.wire_dtype != QZ_loc.dtype:
                raise NotImplementedError("the pipelined exchange carries the fp32 wire only")
            table = chunked_all_gather(QZ_loc, sg, self.group)       # all chunks queued; nothing waits yet
            P = torch.mm(x_local, L.w_att[:d_in])
            rel = L.transform_relations(rel_emb)
            RR = L._rel_mm(rel, wqz)
            a = L.a_att.reshape(-1).float()
            pre = chunked_aggregate(P, table, RR, a, sg, slope, self.chunk_kernels)     # self loop and /2 fused in the merge
            if L.layer_act is torch.tanh and (self.chunk_kernels is None or self.bn_kernels is not None):
                return sync_bn_tanh(pre, L.bn, sg.n_global, self.group, self.bn_kernels)
            return L.layer_act(sync_batch_norm(pre, sg.n_global, L.bn, self.group))
        QZ = all_gather_rows(QZ_loc.contiguous(), self.group, defer=True, wire_dtype=self.wire_dtype)   # [world*n_max, 2d], in flight
        P = torch.mm(x_local, L.w_att[:d_in])
        rel = L.transform_relations(rel_emb)
        RR = L._rel_mm(rel, wqz)
        a = L.a_att.reshape(-1).float()
        _AllGatherRows.pending_wait()
        if self.local_aggregate is None:
            pre = hip_local_layer(P.contiguous(), QZ, RR, a, sg, slope)
        else:
            nb = self.local_aggregate(P.contiguous(), QZ, RR, a, sg, slope)
            pre = (nb + Z_loc - RR[-1, d:]) * 0.5                    # self loop: softmax over a singleton
        if L.layer_act is torch.tanh and (self.local_aggregate is None or self.bn_kernels is not None):
            return sync_bn_tanh(pre, L.bn, sg.n_global, self.group, self.bn_kernels)
        return L.layer_act(sync_batch_norm(pre, sg.n_global, L.bn, self.group))


# ------------------------------------------------------------------------------------------------
# query-sharded scoring (SURVEY.md section 8e "Scoring"; BASELINE config 5: OpenEA 15K alignment on 2 GPUs)
# ------------------------------------------------------------------------------------------------
# Rows of the score matrix (queries) are split over ranks, the candidate table is replicated: top-k and ranks are
# row-local, so get_neg needs only a gather of the [L_r, k] index slabs.  CSLS is the one real exchange: its column
# term r2[j] = mean of the k largest entries of COLUMN j runs over all ranks' rows, so each rank's per-column top-k
# candidates ([N2, k] values) are all-gathered and merged before the row-local rescoring and rank count.
class _HipScoring:
    """The product's local kernels (jmac_amd.scoring); tests inject a torch stand-in under gloo."""

    @staticmethod
    def sim_topk(a, b, k):
        from . import scoring
        return scoring.sim_topk(a, b, k)

    @staticmethod
    def sim_matrix(a, b):
        from . import scoring
        return scoring.sim_matrix(a, b)

    @staticmethod
    def row_topk_values(s, k):
        from . import scoring
        return scoring.row_topk(s, k)[0]

    @staticmethod
    def col_topk_values(s, k):
        """[n2, k]: the k largest entries of every column of the rank's row block (no transpose)."""
        from . import scoring
        return scoring.col_topk_values(s, k)

    @staticmethod
    def rank_of_gold(s, gold):
        """1-based rank of column gold[i] in row i of s, descending, ties -> lower index first."""
        from . import scoring
        return scoring.filtered_rank(s, gold.to(torch.int32), descending=True).to(torch.int64)


def shard_rows(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous even split of n query rows: ranks [0, n % world) get one extra row."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _all_gather_padded(x: torch.Tensor, n_max: int, group=None) -> torch.Tensor:
    """x [n_r, ...] -> [world, n_max, ...] (rows past n_r are zero)."""
    world = _world(group)
    if x.shape[0] < n_max:
        x = torch.cat([x, x.new_zeros((n_max - x.shape[0],) + tuple(x.shape[1:]))], 0)
    x = x.contiguous()
    if _skip(group):
        return x.unsqueeze(0)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if x.is_cuda and dist.get_backend(group) == "gloo":               # tests only: stage through the host
        h = torch.empty(out.shape, dtype=x.dtype)
        dist.all_gather_into_tensor(h, x.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, x, group=group)
    return out.view((world,) + tuple(x.shape))


def sharded_get_neg(ILL, emb_src: torch.Tensor, emb_dst: torch.Tensor, k: int, group=None, kernels=None) -> torch.Tensor:
    """get_neg (modules/utils/util.py:31-54; DBPv1: JMAC_DBPv1/modules/utils/util.py:35-58) with the seed rows split
    over ranks: each rank scores its slice of ILL against ALL of emb_dst (MFMA sim + top-k), the [L_r, k] index slabs
    are gathered.  Every rank returns the full flattened [L*k] int64 result, identical to the one-rank call."""
    kn = kernels or _HipScoring
    world, rank = _world(group), _rank(group)
    ill = torch.as_tensor(ILL, dtype=torch.long, device=emb_src.device).reshape(-1)
    L = int(ill.numel())
    lo, hi = shard_rows(L, world, rank)
    idx = kn.sim_topk(emb_src.index_select(0, ill[lo:hi]), emb_dst, k) if hi > lo else \
        torch.zeros((0, k), dtype=torch.int64, device=emb_src.device)
    n_max = shard_rows(L, world, 0)[1]
    slabs = _all_gather_padded(idx.to(torch.int64), n_max, group)     # [world, n_max, k]
    parts = [slabs[r, : shard_rows(L, world, r)[1] - shard_rows(L, world, r)[0]] for r in range(world)]
    return torch.cat(parts, 0).reshape(-1)


def sharded_alignment_test(embeds1: torch.Tensor, embeds2: torch.Tensor, top_k=(1, 5, 10), csls_k: int = 10, group=None,
                           kernels=None):
    """modules/finding/evaluation.py:20-28 -> alignment.py:10-112 (metric='cosine', accurate=True, CSLS csls_k) with
    the rows of the similarity matrix split over ranks.  Row i of embeds1 is aligned with row i of embeds2; both tables
    are replicated.  Returns (top_k, hits [%], mr, mrr), identical on every rank and to scoring.alignment_test."""
    kn = kernels or _HipScoring
    world, rank = _world(group), _rank(group)
    n1, n2 = embeds1.shape[0], embeds2.shape[0]
    lo, hi = shard_rows(n1, world, rank)
    e1 = F.normalize(embeds1[lo:hi], 2, -1)
    e2 = F.normalize(embeds2, 2, -1)
    s = kn.sim_matrix(e1, e2)                                         # [n_r, n2] row block of the score matrix
    if csls_k > 0:
        r1 = kn.row_topk_values(s, csls_k).mean(1)                    # row term: local
        kk = min(csls_k, max(hi - lo, 1))
        if hi > lo:
            colv = kn.col_topk_values(s, kk)                          # [n2, kk]: this rank's candidates per column
        else:
            colv = s.new_full((n2, kk), float("-inf"))
        if kk < csls_k:
            colv = torch.cat([colv, colv.new_full((n2, csls_k - kk), float("-inf"))], 1)
        allv = _all_gather_padded(colv, n2, group)                    # [world, n2, csls_k]
        merged = allv.permute(1, 0, 2).reshape(n2, -1)
        r2 = merged.topk(csls_k, dim=1).values.mean(1)                # k largest over all ranks' rows
        s = 2 * s - r1.view(-1, 1) - r2.view(1, -1)
    gold = torch.arange(lo, hi, device=s.device)
    rk = kn.rank_of_gold(s, gold).to(torch.float64) if hi > lo else torch.zeros(0, dtype=torch.float64, device=s.device)
    stats = torch.stack([(rk <= k).double().sum() for k in top_k] + [rk.sum(), (1.0 / rk).sum()])
    if not _skip(group):
        _all_reduce(stats, group)
    stats = stats / n1
    hits = [round(float(h) * 100.0, 3) for h in stats[: len(top_k)]]
    return list(top_k), hits, float(stats[-2]), float(stats[-1])
