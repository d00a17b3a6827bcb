"""JMAC encoder + scoring call sites on the HIP layer (rows a9-a11, a15, a16 of SURVEY.md section 8).

A host-side mirror of ``class JMAC`` (src/jmac_model.py:125-380): same constructor arguments, same
parameter / buffer names (reference ``state_dict``s load with ``strict=True``), same method names and
argument meaning.  The three GNN layers are ``jmac_amd.layer.RelationAwareLayer``; link-prediction
distances go through ``jmac_amd.scoring.l1_scores``; the dense mixes and the margin losses stay in
torch exactly as SURVEY.md scopes them.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import encoder, losses, ops, scoring
from .graph import graph_cache
from ._lib import check_index_range, mark_index_range
from .layer import RelationAwareLayer, get_param


def _idx(x, device, n=None):
    """Index lists arrive as python lists, int64 tensors or float64 numpy arrays (train.py:190-193).  With ``n`` host
    data is range-checked against [0, n) before the upload (IndexError, as torch indexing in the reference) and the
    device tensor is marked so that the kernels' entry points do not check it again."""
    if isinstance(x, torch.Tensor):
        return x.reshape(-1).to(device=device, dtype=torch.long)
    a = np.asarray(x).reshape(-1).astype(np.int64)
    if n is None:
        return torch.as_tensor(a, device=device)
    check_index_range(a, n, "index")
    return mark_index_range(torch.as_tensor(a, device=device), n)


def _link_columns(links, device, n1, n2):
    """[L,2] seed links -> two int64 device index vectors (host data checked before the upload)."""
    if isinstance(links, torch.Tensor):
        lk = links.to(device=device, dtype=torch.long)
        return lk[:, 0], lk[:, 1]
    a = np.asarray(links).astype(np.int64).reshape(-1, 2)
    return _idx(a[:, 0], device, n1), _idx(a[:, 1], device, n2)


def _rows(table, lo, hi):
    """table[lo:hi] (src/jmac_model.py:173-176) -- the table itself when the range is all of it: the backward of a slice is
    a zero-fill of the whole table plus a copy, per step and per table, for nothing when a model holds one KG."""
    if lo == 0 and hi == table.shape[0]:
        return table
    return table[lo:hi]


class JMAC(nn.Module):
    def __init__(self, args, entity_name_emb, num_relations, num_entities):
        super().__init__()
        self.args = args
        self.act = torch.tanh
        self.ent_info_att = torch.as_tensor(np.asarray(entity_name_emb), dtype=torch.float32)   # :133 (plain attribute)
        assert self.ent_info_att.shape[0] == num_entities
        d = args.dim
        self.entity_dim = self.relation_dim = d
        self.device = args.device
        self.ent_init_att_completion = get_param((num_entities, d))
        self.rel_init_att_completion = get_param((num_relations, d))
        self.rel_init_att_alignment = get_param((num_relations, d))
        self.completion_dropout = nn.Dropout(args.dropout)
        self.atv_mlp = nn.LeakyReLU(args.leaky_relu_w)
        mk = lambda: RelationAwareLayer(d, d, rel_dim=d, act=self.act, args=args)
        self.conv1_alignment, self.conv2_alignment, self.conv1_completion = mk(), mk(), mk()
        self.name_linear = get_param((self.ent_info_att.shape[1], d))
        self.margin_align = args.margin_align
        self.k = args.num_negative
        self.margin_completion = nn.Parameter(torch.tensor([float(args.margin_completion)]), requires_grad=False)
        L = args.num_gcn_layer
        self.uni_linear1_1, self.uni_linear1_2 = get_param((d * L, d)), get_param((d, d))
        self.uni_linear2_1, self.uni_linear2_2 = get_param((d * L, d)), get_param((d, d))
        self.rel_linear11, self.rel_linear12 = get_param((d, d)), get_param((d, d))
        self.rel_linear21, self.rel_linear22 = get_param((d, d)), get_param((d, d))
        self.rel_linear11_uni, self.rel_linear12_uni = get_param((d, d)), get_param((d, d))
        self.all_linear_completion = get_param((d * (L + 1), d))
        self.forward_base = self.forward_no_name if args.no_name_info else self.forward_name   # :166-169
        # True: forward_name / forward_no_name run as ONE autograd node each (jmac_amd.encoder: grouped relation-side
        # products, cat operands written in place, hand-written backward) wherever that node covers the configuration;
        # False: always op by op (the second implementation the tests hold the node to)
        self.fused_encoder = True

    def set_table_dtype(self, dtype) -> None:
        """torch.bfloat16: inference form (BASELINE config 3) -- the three layers gather bf16 [P|Q|Z] / [Rq|Rz]
        tables and forward_linkpred scores bf16 entity tables; arithmetic stays fp32.  torch.float32: default."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("table dtype must be float32 or bfloat16")
        self.table_dtype = dtype
        for lay in (self.conv1_alignment, self.conv2_alignment, self.conv1_completion):
            lay.table_dtype = dtype

    # ---- encoders -------------------------------------------------------------------------------
    def _rel_mlp(self, r, w1, w2):
        mm = RelationAwareLayer._rel_mm                                   # src/jmac_model.py:195-196
        return mm(self.atv_mlp(mm(r, w1)), w2)

    def _fused(self, info_dim):
        return (self.fused_encoder and self.ent_init_att_completion.is_cuda
                and self.ent_init_att_completion.dtype == torch.float32 and encoder.supported(self, info_dim))

    def _graph(self, edge_index, edge_type, n, nr):
        lay = self.conv1_completion
        return graph_cache.get(edge_index, edge_type, n, nr + 1, lay.chunk)

    def forward_name(self, edge_index, edge_type, ent_bases, rel_bases):
        """src/jmac_model.py:172-204."""
        e0, e1 = ent_bases
        r0, r1 = rel_bases
        dev = self.ent_init_att_completion.device
        comp_att = _rows(self.ent_init_att_completion, e0, e1)
        rel_comp = _rows(self.rel_init_att_completion, r0, r1)
        rel_align = _rows(self.rel_init_att_alignment, r0, r1)
        if self._fused(self.ent_info_att.shape[1]):
            info = _rows(self.ent_info_att, e0, e1).to(dev)
            graph = self._graph(edge_index, edge_type, comp_att.shape[0], rel_comp.shape[0])
            align_out, c1, rel_c1 = encoder.forward_name(self, comp_att, rel_comp, rel_align, info, graph)
            return align_out, [comp_att, c1], [rel_comp, rel_c1]
        comp0 = self.completion_dropout(ops.row_normalize(comp_att))
        # :177 + :180  cat(comp0, info @ name_linear) @ W  ==  cat(comp0, info) @ [W_top ; name_linear @ W_bottom]:
        # the [N,300]x[300,300] product of the constant name embeddings (and its [N,300]x[300,300] adjoint) becomes a
        # [300,300]x[300,300] product of the two parameters -- same function, two N-row GEMMs fewer per step
        d = self.entity_dim
        w = torch.cat((self.uni_linear1_1[:d], torch.mm(self.name_linear, self.uni_linear1_1[d:])), dim=0)
        align0 = torch.mm(torch.cat((comp0, _rows(self.ent_info_att, e0, e1).to(dev)), dim=1), w)
        a1 = self.conv1_alignment(align0, rel_align, edge_index, edge_type)
        align_layers, comp_layers, comp_rel_layers = [align0, a1], [comp_att], [rel_comp]
        if self.args.num_gcn_layer == 2:
            c1 = self.conv1_completion(comp_att, rel_comp, edge_index, edge_type)
            c1n = self.completion_dropout(ops.row_normalize(c1))
            a_in = torch.mm(torch.cat((c1n, a1), dim=1), self.uni_linear2_1)
            rel_c1 = self._rel_mlp(rel_comp, self.rel_linear11, self.rel_linear12)
            rel_a_in = self._rel_mlp(rel_align, self.rel_linear11_uni, self.rel_linear12_uni)
            a2 = self.conv2_alignment(a_in, rel_a_in, edge_index, edge_type)
            align_layers.append(a2)
            comp_layers.append(c1)
            comp_rel_layers.append(rel_c1)
        align_out = torch.mm(torch.cat(align_layers, dim=1), self.all_linear_completion)
        return align_out, comp_layers, comp_rel_layers

    def forward_no_name(self, edge_index, edge_type, ent_bases, rel_bases):
        """src/jmac_model.py:207-220."""
        e0, e1 = ent_bases
        r0, r1 = rel_bases
        comp_att = _rows(self.ent_init_att_completion, e0, e1)
        rel_comp = _rows(self.rel_init_att_completion, r0, r1)
        if self._fused(None):
            graph = self._graph(edge_index, edge_type, comp_att.shape[0], rel_comp.shape[0])
            c1, rel_c1 = encoder.forward_no_name(self, comp_att, rel_comp, graph)
            return c1, [comp_att, c1], [rel_comp, rel_c1]
        comp_layers, comp_rel_layers = [comp_att], [rel_comp]
        if self.args.num_gcn_layer == 2:
            comp_layers.append(self.conv1_completion(comp_att, rel_comp, edge_index, edge_type))
            comp_rel_layers.append(self._rel_mlp(rel_comp, self.rel_linear11, self.rel_linear12))
        return comp_layers[-1], comp_layers, comp_rel_layers

    def get_emb(self, edge_index, edge_type, ent_bases, rel_bases, pyt=False):
        """src/jmac_model.py:223-234."""
        align_out, comp_layers, _ = self.forward_base(edge_index, edge_type, ent_bases, rel_bases)
        a = ops.row_normalize(align_out).detach()
        c = ops.row_normalize(comp_layers[-1]).detach()
        if pyt:
            return a.cpu(), c.cpu()
        return a.cpu().numpy(), c.cpu().numpy()

    # ---- scoring ----------------------------------------------------------------------------------
    def forward_linkpred(self, e_index, r_index, edge_index, edge_type, all_index, ent_bases, rel_bases,
                         pred_head=False, cached=None):
        """src/jmac_model.py:295-313.  ``cached`` = a previous forward_base result: the encoder output
        is constant across evaluation batches (src/validate.py:46-50 recomputes it per batch)."""
        if cached is None:
            cached = self.forward_base(edge_index, edge_type, ent_bases, rel_bases)
        _, comp_layers, comp_rel_layers = cached
        n = comp_layers[0].shape[0]
        # the reference always gathers ent[all_index] (:312); its only caller passes list(range(N)) (src/validate.py:43-44).
        # The gather is skipped only for an all_index that IS the identity: None, range(n), or host data (list / numpy)
        # compared element by element on the host -- no device sync, and a permuted list is never mistaken for it.
        # A device tensor is always gathered.
        if all_index is None or (isinstance(all_index, range) and all_index == range(n)):
            identity = True
        elif isinstance(all_index, torch.Tensor):
            identity = False
        else:
            ai_host = np.asarray(all_index)
            identity = ai_host.shape == (n,) and bool((ai_host == np.arange(n)).all())
        if not identity:
            ai = _idx(all_index, comp_layers[0].device, n)
            comp_layers = [c.index_select(0, ai) for c in comp_layers]
        layers = range(self.args.num_gcn_layer)
        return scoring.linkpred_dist([comp_layers[l] for l in layers], [comp_rel_layers[l] for l in layers],
                                     e_index, r_index, pred_head, table_dtype=getattr(self, "table_dtype", torch.float32))

    def linkpred_ranks(self, e_index, r_index, gold, edge_index, edge_type, ent_bases, rel_bases, filt_ptr=None,
                       filt_idx=None, pred_head=False, cached=None):
        """Filtered ranks of ``gold`` among ALL entities of the KG: forward_linkpred (:295-313) followed by the ranking loop
        of CompletionEvaluator.test (src/validate.py:50-64) in one fused op that never writes the [B, N] distance matrix
        (scoring.linkpred_ranks).  ``filt_ptr`` / ``filt_idx``: scoring.build_filter_csr of the batch, or None (raw)."""
        if cached is None:
            cached = self.forward_base(edge_index, edge_type, ent_bases, rel_bases)
        _, comp_layers, comp_rel_layers = cached
        layers = range(self.args.num_gcn_layer)
        return scoring.linkpred_ranks([comp_layers[l] for l in layers], [comp_rel_layers[l] for l in layers], e_index, r_index,
                                      gold, filt_ptr, filt_idx, pred_head,
                                      table_dtype=getattr(self, "table_dtype", torch.float32))

    # ---- losses (src/jmac_model.py:237-292, :316-380): gathers + L1 / cosine fused in HIP (jmac_amd.losses),
    # the margin arithmetic on the resulting [T] / [L] vectors stays in torch ----------------------------
    @staticmethod
    def _cos_dist(e1, i1, e2, i2):
        return losses.pair_cosine_distance(e1, i1, e2, i2)

    def alignment_loss_simple(self, links, ent_embeddings1, ent_embeddings2):
        if not len(links):
            return 0
        dev = ent_embeddings1.device
        l0, l1 = _link_columns(links, dev, ent_embeddings1.shape[0], ent_embeddings2.shape[0])
        return self._cos_dist(ent_embeddings1, l0, ent_embeddings2, l1).mean()

    def alignment_loss(self, feeddict, edge_index1, edge_type1, edge_index2, edge_type2):
        links = feeddict["links"]
        if not len(links):
            return 0
        e1, _, _ = self.forward_base(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"])
        e2, _, _ = self.forward_base(edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])
        dev = e1.device
        l0, l1 = _link_columns(links, dev, e1.shape[0], e2.shape[0])
        n = int(l0.numel())
        d = (self._cos_dist(e1, l0, e2, l1) + self.margin_align).view(n, 1)
        total = 0
        for left, right in (("neg_left", "neg_right"), ("neg2_left", "neg2_right")):
            b = self._cos_dist(e1, _idx(feeddict[left], dev, e1.shape[0]), e2, _idx(feeddict[right], dev, e2.shape[0]))
            total = total + F.relu(d - b.view(n, -1)).sum()
        return total / (2 * self.k * n)

    def completion_loss(self, data, edge_index1, edge_type1, edge_index2, edge_type2, feeddict, source=True):
        _, comp1, rel1 = self.forward_base(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"])
        _, comp2, rel2 = self.forward_base(edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])
        h, t, r = data["batch_h"], data["batch_t"], data["batch_r"]
        bs = self.args.batch_size
        loss = 0
        for layer in range(self.args.num_gcn_layer):
            ent, rel = (comp1[layer], rel1[layer]) if source else (comp2[layer], rel2[layer])
            # the L1 scores (src/jmac_model.py:345-350) and pos / neg views + max + mean (:351-378) as one node; the reference
            # consumes the b-major negative block as n-major (view(-1, B).permute): kept as is inside the fused op
            loss_res = losses.triple_l1_margin_loss(ent, rel, h, r, t, bs, self.margin_completion)
            loss = loss + loss_res + self.alignment_loss_simple(feeddict["links"], comp1[layer], comp2[layer])
        return loss
