"""Device-resident relation graph: CSR by aggregation destination + work schedules.

The reference hands every layer call a COO ``edge_index [2,E]`` / ``edge_type [E]`` pair of int64
tensors (train.py:116-135, src/utils.py:112-149) and recomputes degrees per call
(src/jmac_model.py:99-109).  Here the COO pair is converted once by ``jmac_csr_build`` and the result
is cached per tensor identity, so the drop-in layer can keep the reference's call signature.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import View, check, check_index_range, lib, ptr, require_device, stream

DEFAULT_CHUNK = None          # None -> auto_chunk()
# schedules of up to this many items carry the first two entries of each item inline (the forward kernel gives every
# item its own wave there -- aggregate.hip fwd_grid -- and is bound by dependent round trips, not by bandwidth)
INLINE_EDGES_MAX_ITEMS = 65536          # (tools/_knobs.py sets these module attributes from the environment for the probes)
# the by-source / by-relation views (backward passes B / C) switch to their small-graph form (short items, inline entries) on
# their own threshold: measured on the 56 589-entity union the forward gains from the small form (136 -> 117 us fp32, 122 -> 86 us
# bf16) while the backward loses (306 -> 346 us)
SMALL_BWD_MAX_ITEMS = 16384
# order of a destination's edges inside its CSR row: by relation type, then input order (False: input order)
SORT_ROWS_BY_TYPE = True


def auto_chunk(n_entries: int, cap: int = 512) -> int:
    """Entries per work item: aim for >= ~16k items (2 per wave slot of 256 CUs x 32 waves) so that small
    graphs are not latency-bound on a handful of long segments, capped at 512 (the chunk size the hardware
    guide measured for skewed per-destination sums) and floored at 32 to bound the partial-row traffic."""
    c = max(1, int(n_entries) // 16384)
    p = 1
    while p < c:
        p *= 2
    return int(min(cap, max(32, p)))


# The by-destination schedule (forward kernel, backward pass A) takes shorter items than the by-source / by-relation
# views: measured on config 4 (interleaved repetitions) the forward runs 11.46 / 11.11 / 11.04 / 11.44 ms at 512 / 256 /
# 128 / 64 entries per item, while the backward's partial-row passes prefer 512 (20.6 ms against 23.3 ms at 128).
DST_CHUNK_CAP = 256
SMALL_BWD_CHUNK = 32


# cooperative splits (forward kernel, small graphs): rows longer than COOP_MIN and up to COOP_MAX entries are processed by
# the four waves of one workgroup and merged in LDS.  Measured on the DBP-5L ja shape: the kernel's duration was set by
# its longest row (28 entries = 14 dependent gather rounds in one wave); capping the rows at 8 entries took 19.4 -> 15.8 us.
COOP_MIN, COOP_MAX = 8, 256
# ... and on graphs past COOP_SIZE_SPLIT items, which have the waves to hide a longer row behind, only rows longer than this
# (56 589-entity union, bf16 tables: 87.3 / 79.1 / 76.5 us at 8 / 12 / 16, fp32 tables 116 +- 1 us at any of them)
COOP_MIN_LARGE, COOP_SIZE_SPLIT = 16, 16384


def coop_min_for(n_seg: int, n_entries: int, chunk: int) -> int:
    return COOP_MIN if n_seg + n_entries // max(chunk, 1) + 1 <= COOP_SIZE_SPLIT else COOP_MIN_LARGE


class _Schedule:
    """items / splits / counts for one grouping of the CSR slots (jmac_items_build)."""

    def __init__(self, seg_ptr: torch.Tensor, n_seg: int, n_entries: int, chunk: int, order: Optional[torch.Tensor],
                 coop: bool = False):
        L = lib()
        dev = seg_ptr.device
        self.ptr = seg_ptr
        self.order = order
        cmin, cmax = (coop_min_for(n_seg, n_entries, chunk), COOP_MAX) if coop else (0, 0)
        self.n_items_max = int(L.jmac_items_max(n_seg, n_entries, chunk, cmin))
        self.n_splits_max = int(L.jmac_splits_max(n_entries, chunk, cmin))
        self.n_parts_max = int(L.jmac_parts_max(n_entries, chunk, cmin))
        self.items = torch.empty((self.n_items_max, 4), dtype=torch.int32, device=dev)
        self.splits = torch.empty((self.n_splits_max, 4), dtype=torch.int32, device=dev)
        self.counts = torch.zeros(8, dtype=torch.int32, device=dev)
        ws_bytes = int(L.jmac_graph_workspace_bytes(n_entries, n_seg))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(L.jmac_items_build(ptr(seg_ptr), n_seg, chunk, cmin, cmax, ptr(self.items), ptr(self.splits), ptr(self.counts),
                                 ptr(ws), ws_bytes, stream()), "jmac_items_build")
        # one host read at build time (never on the hot path): exact launch bounds, and the combine
        # kernels are skipped altogether when no segment was split
        n_items, n_splits, n_parts, n_empty, n_coop = self.counts.tolist()[:5]
        self.n_items_max, self.n_splits_max, self.n_parts_max = int(n_items), int(n_splits), int(n_parts)
        self.n_empty = int(n_empty)            # empty segments = the last n_empty items of the schedule
        self.n_coop = int(n_coop)              # cooperative segments = the first 4 * n_coop items / first n_coop splits
        self.item_edges = None
        self.entry_dst = None                  # by-source / by-relation views: destination of every entry, grouped order
        self._view = None

    def view(self) -> View:
        if self._view is None:
            self._view = View(ptr(self.ptr), ptr(self.order), ptr(self.items), ptr(self.splits), ptr(self.counts),
                              self.n_items_max, self.n_splits_max, self.n_parts_max, ptr(self.item_edges), self.n_empty,
                              self.n_coop, ptr(self.entry_dst))
        return self._view

    def view_compact(self, col: torch.Tensor, etype_c: torch.Tensor) -> View:
        """The same schedule for kernels that read COMPACT relation numbering (RelGraph.ensure_rel_compact): the inline
        {col, type} pairs of a by-destination schedule carry relation ids, so they exist a second time with ``etype_c``."""
        if self.item_edges is None:
            return self.view()
        if getattr(self, "_view_c", None) is None:
            self.item_edges_c = torch.empty((self.n_items_max, 4), dtype=torch.int32, device=col.device)
            check(lib().jmac_item_edges_build(ptr(self.items), ptr(self.counts), self.n_items_max, ptr(col), ptr(etype_c),
                                              ptr(self.item_edges_c), stream()), "jmac_item_edges_build")
            self._view_c = View(ptr(self.ptr), ptr(self.order), ptr(self.items), ptr(self.splits), ptr(self.counts),
                                self.n_items_max, self.n_splits_max, self.n_parts_max, ptr(self.item_edges_c), self.n_empty,
                                self.n_coop, ptr(self.entry_dst))
        return self._view_c

    def build_item_edges(self, col: torch.Tensor, etype: torch.Tensor) -> None:
        """{col, type} of the first two entries of every item, inline with the schedule (small graphs: one dependent
        round trip fewer per wave).  By-source / by-relation schedules pass (order, entry_dst)."""
        if self.item_edges is None and self.n_items_max > 0:
            self.item_edges = torch.empty((self.n_items_max, 4), dtype=torch.int32, device=col.device)
            check(lib().jmac_item_edges_build(ptr(self.items), ptr(self.counts), self.n_items_max, ptr(col), ptr(etype),
                                              ptr(self.item_edges), stream()), "jmac_item_edges_build")
            self._view = None


class RelGraph:
    """CSR (by destination) of a typed edge list, on one HIP device.

    edge_index[0] = aggregation destination, edge_index[1] = message source
    (modules/helper/message_passing.py:75,79,87).
    """

    def __init__(self, edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int, num_rel: int,
                 chunk: Optional[int] = DEFAULT_CHUNK, num_src: Optional[int] = None):
        require_device(edge_index, edge_type)
        self.chunk_arg = chunk                 # as asked for (None = automatic): what a re-indexed copy of this graph is built with
        chunk_dst = chunk
        if chunk is None:
            chunk = auto_chunk(int(edge_index.shape[1]))
            chunk_dst = auto_chunk(int(edge_index.shape[1]), DST_CHUNK_CAP)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        if edge_type.shape[0] != edge_index.shape[1]:
            raise ValueError("edge_type must be [E]")
        L = lib()
        dev = edge_index.device
        self.N, self.E, self.num_rel, self.chunk = int(num_nodes), int(edge_index.shape[1]), int(num_rel), int(chunk)
        # sources may live in another index space than destinations (destination-sharded multi-GPU)
        self.num_src = int(num_src) if num_src is not None else self.N
        self.device = dev
        ei = edge_index.contiguous().to(torch.int64)
        et = edge_type.contiguous().to(torch.int64)
        E, N = self.E, self.N
        # the reference's torch indexing raises IndexError on a bad id; here an unchecked id would corrupt device memory
        # (rowptr writes, table gathers, gradient scatters): validate once per graph (build time, never on the hot path)
        if E > 0:
            check_index_range(ei[0], N, "edge_index[0] (aggregation destination)")
            check_index_range(ei[1], self.num_src, "edge_index[1] (message source)")
            check_index_range(et, self.num_rel, "edge_type")
        self.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.etype = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        ws_bytes = int(L.jmac_graph_workspace_bytes(E, N))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(L.jmac_csr_build(ptr(ei), ptr(et), E, N, self.num_rel if SORT_ROWS_BY_TYPE else 0, ptr(self.rowptr), ptr(self.col),
                               ptr(self.etype), ptr(self.perm), ptr(ws), ws_bytes, stream()), "jmac_csr_build")
        # small graphs: one wave per item, long rows shared by a workgroup, first entries inline with the item headers
        small = E > 0 and N + E // max(chunk_dst, 1) + 1 <= INLINE_EDGES_MAX_ITEMS
        self.by_dst = _Schedule(self.rowptr, N, E, chunk_dst, None, coop=small)
        if small:
            self.by_dst.build_item_edges(self.col, self.etype)
        # pass A of the backward walks a by-destination schedule too.  Past the backward's own small-graph threshold it gets a
        # plain one (no cooperative quarters: their partial dP rows cost more than the long rows they split; measured on the
        # 56 589-entity union: 306 us with the plain schedule, 346 us with the forward's)
        self.by_dst_bwd = self.by_dst
        self._dst_bwd_plain = small and not (N + E // max(chunk, 1) + 1 <= SMALL_BWD_MAX_ITEMS)
        self._chunk_dst = chunk_dst
        self._ei = ei
        self._bwd_ready = False
        self.dst_of_slot = None
        self.by_src = None
        self.by_rel = None
        self._relc = False             # used-relation compaction (ensure_rel_compact): built on first use
        self.rel_used = self.rel_pos = self.etype_c = self.by_rel_c = None
        self.n_used = 0

    def ensure_rel_compact(self) -> None:
        """Used-relation compaction: a DBP-5L KG names 153-833 of its 961 relation rows in its edges (ja: 158), and the
        layer's relation transform / projection (src/jmac_model.py:39-42) matter for named rows only.  ``rel_used`` (int64,
        ascending) = the non-loop relation ids some edge carries, ``n_used`` their number, ``rel_pos`` (int32 [num_rel]) the
        compact row of every relation (-1: unused; the loop relation num_rel - 1 -> n_used), ``etype_c`` the CSR's edge types in
        compact numbering.  One host read at build time (torch.unique), never on the hot path."""
        if self._relc:
            return
        dev, E, loop = self.device, self.E, self.num_rel - 1
        et = self.etype[:E].long() if E > 0 else torch.zeros(0, dtype=torch.int64, device=dev)
        used = torch.unique(et)
        self.rel_used = used[used < loop].contiguous()
        self.n_used = int(self.rel_used.numel())
        pos = torch.full((self.num_rel,), -1, dtype=torch.int32, device=dev)
        pos[self.rel_used] = torch.arange(self.n_used, dtype=torch.int32, device=dev)
        pos[loop] = self.n_used
        self.rel_pos = pos
        self.etype_c = (pos[et] if E > 0 else torch.zeros(1, dtype=torch.int32, device=dev)).contiguous()
        self._relc = True

    # the backward needs the by-source and by-relation groupings of the CSR slots; built on first use
    def ensure_backward_views(self) -> None:
        if self._bwd_ready:
            return
        L = lib()
        dev, E, N = self.device, self.E, self.N
        if self._dst_bwd_plain:
            self.by_dst_bwd = _Schedule(self.rowptr, N, E, self._chunk_dst, None, coop=False)
        if E > 0:
            self.dst_of_slot = self._ei[0].index_select(0, self.perm[:E].long()).to(torch.int32)
        else:
            self.dst_of_slot = torch.zeros(1, dtype=torch.int32, device=dev)

        # small graphs: shorter items (a 32-entry item is 8 dependent gather rounds in one wave; the merge pass keeps
        # four partial rows in flight per lane, so the longer partial lists cost less than the rounds they save)
        small = self.by_dst.item_edges is not None and N + E // max(self.chunk, 1) + 1 <= SMALL_BWD_MAX_ITEMS
        chunk_bwd = min(self.chunk, SMALL_BWD_CHUNK) if small else self.chunk

        def group(keys: torch.Tensor, n_seg: int) -> _Schedule:
            seg_ptr = torch.empty(n_seg + 1, dtype=torch.int32, device=dev)
            order = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            ws_bytes = int(L.jmac_graph_workspace_bytes(E, n_seg))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            check(L.jmac_group_build(ptr(keys), E, n_seg, ptr(seg_ptr), ptr(order), ptr(ws), ws_bytes, stream()),
                  "jmac_group_build")
            sch = _Schedule(seg_ptr, n_seg, E, chunk_bwd, order)
            # destination of every entry in grouped order: passes B / C read it beside `order` instead of behind it
            sch.entry_dst = (self.dst_of_slot.index_select(0, order[:E].long()) if E > 0
                             else torch.zeros(1, dtype=torch.int32, device=dev))
            if small:
                sch.build_item_edges(order, sch.entry_dst)
            return sch

        self.by_src = group(self.col, self.num_src)
        self.by_rel = group(self.etype, self.num_rel)
        if self._relc:
            self.by_rel_c = group(self.etype_c, self.n_used + 1)
        self._group = group
        self._ei = None
        self._bwd_ready = True

    def src_slab_views(self, bounds) -> list:
        """By-source schedules of the SLABS [bounds[c], bounds[c+1]) of the source rows (bounds: ascending ints from 0 to
        num_src): pass B of the backward can then run slab by slab (jmac_rel_attn_aggregate_bwd_phases_f32, phase 2), e.g. so
        that a destination-sharded layer reduce-scatters slab c of d[Q|Z] while slab c+1 is being summed.  Every view is built on
        the slab's slice of the by-source segment pointer: its segments are numbered from the slab's first row, ``order`` /
        ``entry_dst`` are the whole graph's.  Cached per bounds tuple."""
        self.ensure_backward_views()
        key = tuple(int(b) for b in bounds)
        if key[0] != 0 or key[-1] != self.num_src or any(key[i] > key[i + 1] for i in range(len(key) - 1)):
            raise ValueError("src_slab_views: bounds must ascend from 0 to num_src")
        hit = getattr(self, "_src_slabs", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        base = self.by_src
        ptr_host = base.ptr.cpu()                        # one host read at build time (never on the hot path)
        chunk = min(self.chunk, SMALL_BWD_CHUNK) if base.item_edges is not None else self.chunk
        views = []
        for c in range(len(key) - 1):
            s0, s1 = key[c], key[c + 1]
            if s1 == s0:
                views.append(None)
                continue
            n_ent = int(ptr_host[s1]) - int(ptr_host[s0])
            sch = _Schedule(base.ptr[s0:s1 + 1], s1 - s0, n_ent, chunk, base.order)
            sch.entry_dst = base.entry_dst
            if base.item_edges is not None:
                sch.build_item_edges(base.order, base.entry_dst)
            views.append(sch)
        self._src_slabs = (key, views)
        return views

    def ensure_backward_views_compact(self) -> None:
        """by_rel_c: the by-relation grouping of the CSR slots in compact relation numbering (pass C of the backward on compact
        relation tables)."""
        self.ensure_rel_compact()
        self.ensure_backward_views()
        if self.by_rel_c is None:
            self.by_rel_c = self._group(self.etype_c, self.n_used + 1)

    def degrees(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1])


class GraphCache:
    """Small LRU keyed on the identity of the COO tensors the reference passes to every layer call."""

    def __init__(self, capacity: int = 16):
        self.capacity = capacity
        self._d: "OrderedDict[Tuple, RelGraph]" = OrderedDict()

    def get(self, edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int, num_rel: int,
            chunk: Optional[int] = DEFAULT_CHUNK) -> RelGraph:
        key = (edge_index.data_ptr(), edge_type.data_ptr(), tuple(edge_index.shape), edge_index._version,
               edge_type._version, int(num_nodes), int(num_rel), -1 if chunk is None else int(chunk),
               str(edge_index.device))
        g = self._d.get(key)
        if g is None:
            g = RelGraph(edge_index, edge_type, num_nodes, num_rel, chunk)
            # keep the COO tensors alive so a recycled data_ptr can never alias a stale entry
            g._key_refs = (edge_index, edge_type)
            self._d[key] = g
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)
        else:
            self._d.move_to_end(key)
        return g

    def clear(self) -> None:
        self._d.clear()


graph_cache = GraphCache()


class UnionGraphCache:
    """Block-diagonal union of several typed edge lists (one RelGraph over the stacked id spaces), keyed on the identity of
    the component COO tensors.  The reference's training step encodes two KGs per batch with one set of layer weights
    (src/jmac_model.py:325-326, 263-264); their union -- entity ids offset by the rows in front of the block, relation ids
    by the relation rows in front of it, ONE loop relation behind all of them -- goes through the layer kernels as one
    launch set (jmac_amd.model.JMAC.forward_stacked)."""

    def __init__(self, capacity: int = 32):
        self.capacity = capacity
        self._d: "OrderedDict[Tuple, RelGraph]" = OrderedDict()

    def get(self, parts, chunk: Optional[int] = DEFAULT_CHUNK) -> RelGraph:
        """``parts``: sequence of (edge_index [2,E_k], edge_type [E_k], num_nodes_k, num_rel_k) in stack order."""
        key = tuple((ei.data_ptr(), et.data_ptr(), tuple(ei.shape), ei._version, et._version, int(n), int(nr), str(ei.device))
                    for ei, et, n, nr in parts) + (-1 if chunk is None else int(chunk),)
        g = self._d.get(key)
        if g is not None:
            self._d.move_to_end(key)
            return g
        eis, ets = [], []
        row0 = rel0 = 0
        for ei, et, n, nr in parts:
            require_device(ei, et)
            if ei.shape[1] > 0:                     # per block: an id past its own block must not land in a neighbour's rows
                check_index_range(ei, int(n), "edge_index (block of %d entities)" % int(n))
                # like the single-KG path (RelGraph with num_rel = nr + 1) a block may name its loop relation, id nr_k
                check_index_range(et, int(nr) + 1, "edge_type (block of %d relations + the loop relation)" % int(nr))
            eis.append(ei.to(torch.int64) + row0)
            et64 = et.to(torch.int64)
            ets.append(torch.where(et64 == int(nr), torch.full_like(et64, -1), et64 + rel0))   # -1: the union's ONE loop row (below)
            row0 += int(n)
            rel0 += int(nr)
        ei_u, et_u = torch.cat(eis, dim=1).contiguous(), torch.cat(ets).contiguous()
        et_u = torch.where(et_u < 0, torch.full_like(et_u, rel0), et_u)
        _lib.mark_index_range(ei_u, row0)
        _lib.mark_index_range(et_u, rel0 + 1)
        g = RelGraph(ei_u, et_u, row0, rel0 + 1, chunk)
        # the component tensors stay alive with the entry, so that a recycled data_ptr can never alias it
        g._key_refs = tuple((ei, et) for ei, et, _, _ in parts)
        self._d[key] = g
        while len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def clear(self) -> None:
        self._d.clear()


union_cache = UnionGraphCache()
