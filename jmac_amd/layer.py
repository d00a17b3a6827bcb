"""Drop-in RelationAwareLayer on libjmac_hip.so.

Same constructor, ``forward`` signature, parameter names and ``state_dict`` keys as the reference layer
(src/jmac_model.py:10-53; DBPv1 variant JMAC_DBPv1/models/jmac_model.py:20-63), so that
``JMAC.__init__`` (src/jmac_model.py:147-149) can instantiate it unchanged and reference checkpoints load.

The per-edge GEMMs of the reference are linear in the gathered rows for comp_op='sub', so they are
hoisted to per-node / per-relation GEMMs (one [N,d]x[d,3d] and one [nr+1,d]x[d,2d], rocBLAS through
torch.mm); everything indexed by edges runs in the fused HIP kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import xavier_normal_

from . import ops, scatter as jscatter
from ._lib import require_device
from .graph import DEFAULT_CHUNK, graph_cache


def get_param(shape):
    """xavier_normal parameter, as modules/helper/helper.py:68-74."""
    p = nn.Parameter(torch.empty(*shape))
    xavier_normal_(p.data)
    return p


def _identity(x):
    return x


def _wcat(w_att, gcn_weight, d_in, d, dp):
    """[Wt | Wb | Wgcn] as one [d_in, 3dp] matrix (w_att = [Wt; Wb] stacked by rows, src/jmac_model.py:24,75-76)."""
    wt, wb, wg = w_att[:d_in], w_att[d_in:], gcn_weight
    if dp != d:
        wt, wb, wg = (F.pad(w, (0, dp - d)) for w in (wt, wb, wg))
    return torch.cat([wt, wb, wg], dim=1)


class _ProjectTables(torch.autograd.Function):
    """[P|Q|Z] = X [Wt|Wb|Wg],  [Rq|Rz] = R'' [Wb|Wg]  with one hand-written backward.

    Left to autograd, the slices of w_att / of the concatenated weight each cost a zero-fill, a copy and an add in
    the backward (12 launch-bound kernels per layer at DBP-5L size); here d[Wt|Wb|Wg] is produced by one GEMM, the
    relation term is accumulated into its column slice by the second GEMM (addmm), and the two parameter gradients
    are cut from it."""

    @staticmethod
    def forward(ctx, ent_emb, rel, w_att, gcn_weight, dp):
        d_in, d = gcn_weight.shape
        wcat = _wcat(w_att, gcn_weight, d_in, d, dp)
        ctx.save_for_backward(ent_emb, rel, wcat)
        ctx.dims = (d_in, d, dp)
        return torch.mm(ent_emb, wcat), RelationAwareLayer._rel_mm_nograd(rel, wcat[:, dp:])

    @staticmethod
    def backward(ctx, dPQZ, dRR):
        ent_emb, rel, wcat = ctx.saved_tensors
        d_in, d, dp = ctx.dims
        need = ctx.needs_input_grad
        d_ent = torch.mm(dPQZ, wcat.t()) if need[0] else None
        d_rel = RelationAwareLayer._rel_mm_nograd(dRR, wcat[:, dp:].t()) if need[1] else None
        d_watt = d_gcn = None
        if need[2] or need[3]:
            dw = torch.mm(ent_emb.t(), dPQZ)                           # [d_in, 3dp]
            dw[:, dp:].addmm_(rel.t(), dRR)                            # relation rows see [Wb|Wg] only
            d_watt = torch.cat([dw[:, :d], dw[:, dp:dp + d]], dim=0)   # back to the [2 d_in, d] stacking
            d_gcn = dw[:, 2 * dp:2 * dp + d]
        return d_ent, d_rel, d_watt, d_gcn, None


class RelationAwareLayer(nn.Module):
    """ctor ``(in_channels, out_channels, rel_dim, act, args)`` reading ``args.leaky_relu_w`` and
    ``args.comp_op`` (src/jmac_model.py:14-30)."""

    rel_activation = "leaky_relu"      # src/jmac_model.py:41 ; the DBPv1 subclass uses ReLU

    def __init__(self, in_channels, out_channels, rel_dim, act=_identity, args=None):
        super().__init__()
        self.layer_act = act
        self.args = args
        self.rel_transform_weight1 = get_param((rel_dim, in_channels))
        self.rel_transform_weight2 = get_param((in_channels, in_channels))
        self.gcn_weight = get_param((in_channels, out_channels))
        self.loop_rel = get_param((1, rel_dim))
        self.w_att = get_param((2 * in_channels, out_channels))
        self.a_att = get_param((out_channels, 1))
        self.atv_mlp = nn.LeakyReLU(args.leaky_relu_w)
        self.bn = nn.BatchNorm1d(out_channels)
        self.comp_op = getattr(args, "comp_op", None) or getattr(args, "opn", "sub")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bwd_mode = ops.BWD_MODE_DETERMINISTIC
        self.chunk = DEFAULT_CHUNK
        # torch.bfloat16: inference form -- the [P|Q|Z] / [Rq|Rz] tables are produced by bf16 GEMMs and gathered as
        # bf16 (half the bytes of the HBM-bound kernel); logits, softmax, sums, BN stay fp32.  Needs no_grad.
        self.table_dtype = torch.float32
        # True: forward() runs as ONE autograd node (jmac_amd.encoder._LayerNode: grouped relation-side products, hand-written
        # backward) wherever that node covers the configuration; False: op by op (the second implementation)
        self.fused = True

    # -- pieces ---------------------------------------------------------------------------------
    @staticmethod
    def _rel_mm_nograd(a, b):
        """_rel_mm inside a hand-written backward (no autograd graph wanted)."""
        if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[0] <= ops.SMALL_MM_MAX_ROWS:
            return ops._gemm_any(a, b)
        return torch.mm(a, b)

    @staticmethod
    def _rel_mm(a, b):
        """Products on the relation table (~10^3 rows): the library GEMM is launch/occupancy bound there."""
        if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[0] <= ops.SMALL_MM_MAX_ROWS:
            return ops.small_mm(a, b)
        return torch.mm(a, b)

    def transform_relations(self, rel_emb):
        rel = torch.cat([rel_emb, self.loop_rel], dim=0)                  # jmac_model.py:39
        rel = self._rel_mm(rel, self.rel_transform_weight1)
        rel = self.atv_mlp(rel) if self.rel_activation == "leaky_relu" else F.relu(rel)
        return self._rel_mm(rel, self.rel_transform_weight2)              # :42

    def _padded_a(self):
        d = self.out_channels
        dp = (d + 3) // 4 * 4
        a = self.a_att.reshape(-1)
        if dp != d:
            a = F.pad(a, (0, dp - d))
        return a.float(), dp

    def _tables(self, ent_emb, rel):
        """P|Q|Z and Rq|Rz, zero-padded to a multiple of 4 columns for 16-byte rows."""
        d_in, d = self.in_channels, self.out_channels
        dp = (d + 3) // 4 * 4
        a = self.a_att.reshape(-1)
        if dp != d:
            a = F.pad(a, (0, dp - d))
        if self.table_dtype == torch.bfloat16:
            if torch.is_grad_enabled() and (ent_emb.requires_grad or self.w_att.requires_grad):
                raise RuntimeError("table_dtype=bfloat16 is the inference form of the layer: call it under torch.no_grad()")
            dh = ops.bf16_pad(dp)                                         # padded halves (300 -> 304): 16-byte lane loads
            wcat = ops.pad_table_weight(_wcat(self.w_att, self.gcn_weight, d_in, d, dp), dp, 3).to(torch.bfloat16)
            PQZ = torch.mm(ent_emb.to(torch.bfloat16), wcat)              # [N, 3dh]
            RR = torch.mm(rel.to(torch.bfloat16), wcat[:, dh:])           # [nr+1, 2dh]
            return PQZ, RR, a.float(), dp
        PQZ, RR = _ProjectTables.apply(ent_emb, rel, self.w_att, self.gcn_weight, dp)
        return PQZ, RR, a.float(), dp

    def pre_bn(self, ent_emb, rel_emb, edge_index, edge_type):
        """(message_neighbors + message_self) / 2 of src/jmac_model.py:49-52."""
        require_device(ent_emb, rel_emb, edge_index, edge_type)
        n = ent_emb.size(0)
        rel = self.transform_relations(rel_emb)
        if self.comp_op == "sub":
            graph = graph_cache.get(edge_index, edge_type, n, rel.size(0), self.chunk)
            PQZ, RR, a, dp = self._tables(ent_emb, rel)
            pre = ops.rel_attn_aggregate(PQZ, RR, a, graph, self.atv_mlp.negative_slope,
                                         loop_rel=rel.size(0) - 1, out_scale=0.5, bwd_mode=self.bwd_mode)
            return pre if dp == self.out_channels else pre[:, : self.out_channels]
        if self.comp_op == "mult":
            return self._pre_bn_mult(ent_emb, rel, edge_index, edge_type)
        raise NotImplementedError(self.comp_op)

    def _pre_bn_mult(self, x, rel, edge_index, edge_type):
        """comp_op='mult' (src/jmac_model.py:61-64): the message (x_j * r_t) W does not factor into per-node and
        per-relation tables, but it still factors into ONE [E+N, d] x [d, 2d] GEMM over per-edge rows: row e of the
        table is (x_src(e) * r_type(e)) [Wb|Wg], rows E.. are the self loops (x_i * r_loop) [Wb|Wg].  The fused
        aggregation kernel then runs on the graph whose "source" of slot e is row e of that table and whose only
        relation is a zero row -- same kernel, same deterministic backward, no [E,d] scatter passes."""
        n, E = x.size(0), int(edge_index.shape[1])
        d_in, d = self.in_channels, self.out_channels
        a, dp = self._padded_a()
        graph = self._edge_row_graph(edge_index, edge_type, n)
        wt, wb, wg = self.w_att[:d_in], self.w_att[d_in:], self.gcn_weight
        if dp != d:
            wt, wb, wg = (F.pad(w, (0, dp - d)) for w in (wt, wb, wg))
        rows = torch.cat((x.index_select(0, edge_index[1]) * rel.index_select(0, edge_type), x * rel[-1:]), dim=0)
        QZ = torch.mm(rows, torch.cat((wb, wg), dim=1))                  # [E + N, 2dp]
        P = torch.mm(x, wt)                                              # [N, dp]
        zero_rel = torch.zeros((1, 2 * dp), dtype=torch.float32, device=x.device)
        pre = ops.rel_attn_aggregate_split(P, QZ, zero_rel, a, graph, self.atv_mlp.negative_slope, out_scale=0.5,
                                           loop_rel=0, self_off=E)
        return pre if dp == d else pre[:, :d]

    def _edge_row_graph(self, edge_index, edge_type, n):
        """CSR of (destination <- edge id): cached per COO tensor like graph_cache does for the node graph."""
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, int(n), str(edge_index.device))
        hit = getattr(self, "_erg", None)
        if hit is None or hit[0] != key:
            E = int(edge_index.shape[1])
            ei = torch.stack((edge_index[0].to(torch.int64), torch.arange(E, dtype=torch.int64, device=edge_index.device)))
            et = torch.zeros(E, dtype=torch.int64, device=edge_index.device)
            from .graph import RelGraph
            hit = (key, RelGraph(ei, et, n, 1, self.chunk, num_src=E + n), edge_index)   # edge_index kept alive: no aliasing
            self._erg = hit
        return hit[1]

    def _pre_bn_unfactorised(self, x, rel, edge_index, edge_type):
        """comp_op='mult' in the reference's own formulation -- per-edge torch GEMMs + this library's
        torch_scatter-compatible kernels (jmac_scatter_*): an independent second implementation of _pre_bn_mult
        (tests), and what boundary B2 (the torch_scatter trio) looks like in use."""
        n = x.size(0)
        slope = self.atv_mlp.negative_slope

        def propagate(ei, et, use_norm):
            dst, src = ei[0], ei[1]
            m = x[src] * rel.index_select(0, et)
            s = torch.mm(F.leaky_relu(torch.mm(torch.cat((x[dst], m), dim=1), self.w_att), slope), self.a_att)
            o = torch.mm(m, self.gcn_weight)
            alpha = jscatter.scatter_softmax(s, dst, dim=0, dim_size=n)
            if use_norm:
                deg = jscatter.scatter_add(torch.ones_like(dst, dtype=torch.float32), dst, dim=0, dim_size=n)
                alpha = alpha * deg.sqrt()[dst].view(-1, 1)
            return jscatter.scatter(o * alpha, dst, dim=0, dim_size=n, reduce="sum")

        loop = torch.arange(n, device=x.device)
        nb = propagate(edge_index, edge_type, True)
        sl = propagate(torch.stack([loop, loop]), torch.full((n,), rel.size(0) - 1, dtype=torch.long, device=x.device), False)
        return (nb + sl) / 2

    # -- reference signature ----------------------------------------------------------------------
    def forward(self, ent_emb, rel_emb, edge_index, edge_type):
        if self.fused and self.bwd_mode == ops.BWD_MODE_DETERMINISTIC:    # the whole layer as one autograd node
            from . import encoder
            if encoder.layer_supported(self, ent_emb, rel_emb):
                require_device(ent_emb, rel_emb, edge_index, edge_type)
                graph = graph_cache.get(edge_index, edge_type, ent_emb.size(0), rel_emb.size(0) + 1, self.chunk)
                return encoder.layer_forward(self, ent_emb, rel_emb.contiguous(), graph)
        pre = self.pre_bn(ent_emb, rel_emb, edge_index, edge_type)
        bn = self.bn
        d = self.out_channels
        # momentum=None means a cumulative moving average (factor 1/num_batches_tracked) in nn.BatchNorm1d; the reference
        # never sets it (src/jmac_model.py:27): such a module takes torch's own BatchNorm below
        if self.layer_act is torch.tanh and d % 4 == 0 and bn.momentum is not None:
            if bn.training and bn.track_running_stats:
                bn.num_batches_tracked.add_(1)
            use_batch = bn.training or not bn.track_running_stats
            return ops.bn_tanh(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, bn.momentum, bn.eps)
        return self.layer_act(bn(pre))                                    # jmac_model.py:52


class RelationalAwareLayer(RelationAwareLayer):
    """DBPv1 variant: ctor ``(in_channels, out_channels, num_rels, rel_dim, act, args)``, ReLU between the
    two relation transforms, ``args.opn`` (JMAC_DBPv1/models/jmac_model.py:20-41,51)."""

    rel_activation = "relu"

    def __init__(self, in_channels, out_channels, num_rels, rel_dim, act=_identity, args=None):
        super().__init__(in_channels, out_channels, rel_dim, act=act, args=args)
        self.num_rels = num_rels
        self.margin = 5.0
