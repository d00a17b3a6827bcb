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
from .graph import graph_cache, union_cache
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
    """[L,2] seed links -> two contiguous int64 device index vectors.  Host data is range-checked before the upload; the
    columns of a device tensor are cut (and checked) once per tensor version and kept on it -- a training loop hands the same
    ``feeddict["links"]`` to every batch of an epoch (train.py:347-352), and a fresh strided view per step would cost a copy,
    a check and a host read each."""
    if isinstance(links, torch.Tensor):
        hit = getattr(links, "_jmac_cols", None)
        if hit is not None and hit[0] == (links._version, str(device), int(n1), int(n2)):
            return hit[1], hit[2]
        lk = links.to(device=device, dtype=torch.long)
        c0, c1 = lk[:, 0].contiguous(), lk[:, 1].contiguous()
        if not torch.cuda.is_current_stream_capturing():
            check_index_range(c0, n1, "links[:, 0]")
            check_index_range(c1, n2, "links[:, 1]")
            mark_index_range(c0, n1), mark_index_range(c1, n2)
            try:
                links._jmac_cols = ((links._version, str(device), int(n1), int(n2)), c0, c1)
            except AttributeError:                       # pragma: no cover
                pass
        return c0, c1
    a = np.asarray(links).astype(np.int64).reshape(-1, 2)
    return _idx(a[:, 0], device, n1), _idx(a[:, 1], device, n2)


def _rows(table, lo, hi):
    """table[lo:hi] (src/jmac_model.py:173-176) -- the table itself when the range is all of it: the backward of a slice is
    a zero-fill of the whole table plus a copy, per step and per table, for nothing when a model holds one KG."""
    if lo == 0 and hi == table.shape[0]:
        return table
    return table[lo:hi]


def _stack_rows(table, ranges):
    """Rows ``table[lo:hi]`` of every (lo, hi) in ``ranges``, stacked: ONE slice (or the table itself) when the ranges are
    adjacent in the table -- the KGs of a pair sorted by id base usually are -- a cat otherwise."""
    if all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)):
        return _rows(table, ranges[0][0], ranges[-1][1])
    return torch.cat([table[lo:hi] for lo, hi in ranges], dim=0)


class Stacked:
    """Encoder outputs of several KGs that went through the layer kernels as ONE launch set (JMAC.forward_stacked).

    ``align_out`` [N, d], ``comp`` = [layer-0 rows, completion layer 1] and ``rel`` likewise are the STACKED tensors (blocks
    in table order); ``ent_win[k]`` / ``rel_win[k]`` = (first row, rows) of the k-th block IN CALL ORDER.  ``block(k)`` cuts
    the k-th forward_base result out as views (src/jmac_model.py:204 return convention); the loss paths pass the stacked
    tensors with a window instead, which keeps slice adjoints (zero-fill + copy + add per view) out of the backward."""

    def __init__(self, align_out, comp, rel, ent_win, rel_win):
        self.align_out, self.comp, self.rel, self.ent_win, self.rel_win = align_out, comp, rel, ent_win, rel_win

    def block(self, k):
        (e0, n), (r0, m) = self.ent_win[k], self.rel_win[k]
        return (self.align_out[e0:e0 + n] if self.align_out is not None else None,
                [c[e0:e0 + n] for c in self.comp], [r[r0:r0 + m] for r in self.rel])


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
        # True: completion_loss / alignment_loss encode their two KGs as ONE launch set on the block-diagonal union of the two
        # graphs (forward_stacked: stacked tables, per-KG BatchNorm statistics) instead of two forward_base calls
        self.batched_pairs = True

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
            info = self._info_rows(((int(e0), int(e1)),), dev)
            graph = self._graph(edge_index, edge_type, comp_att.shape[0], rel_comp.shape[0])
            align_out, c1, rel_c1, comp0, rel0 = encoder.forward_name(self, comp_att, rel_comp, rel_align, info, graph,
                                                                      info_persistent=True)
            return align_out, [comp0, c1], [rel0, rel_c1]        # comp0 / rel0: comp_att / rel_comp through the node (aliases)
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
            c1, rel_c1, comp0, rel0 = encoder.forward_no_name(self, comp_att, rel_comp, graph)
            return c1, [comp0, c1], [rel0, rel_c1]
        comp_layers, comp_rel_layers = [comp_att], [rel_comp]
        if self.args.num_gcn_layer == 2:
            comp_layers.append(self.conv1_completion(comp_att, rel_comp, edge_index, edge_type))
            comp_rel_layers.append(self._rel_mlp(rel_comp, self.rel_linear11, self.rel_linear12))
        return comp_layers[-1], comp_layers, comp_rel_layers

    def _info_rows(self, ranges, dev):
        """Name-embedding rows of the given id ranges as ONE persistent device tensor (the reference keeps ent_info_att on
        the host and uploads a slice per forward_base call, src/jmac_model.py:133,176): cached per range tuple, so that the
        encoder node's cat buffer can be cached against it and no upload runs per step.  Never evicted -- a captured
        hipGraph may hold the pointer; a model sees its KGs and KG pairs only (DBP-5L: 5 + 10)."""
        cache = self.__dict__.setdefault("_info_cache", {})
        src = self.ent_info_att
        key = (tuple(ranges), str(dev), src.data_ptr(), src._version)
        hit = cache.get(key)
        if hit is None:
            if len(ranges) == 1 and tuple(ranges[0]) == (0, src.shape[0]) and src.device == dev:
                rows = src
            else:
                rows = _stack_rows(src, list(ranges)).to(dev).contiguous()
            hit = cache[key] = (rows, src)      # src stays referenced: its address cannot come back with other contents
        return hit[0]

    def forward_stacked(self, blocks):
        """``blocks``: [(edge_index, edge_type, ent_bases, rel_bases), ...] -- the arguments of the forward_base calls the
        reference makes one after the other with the SAME layer weights (completion_loss / alignment_loss: two KGs,
        src/jmac_model.py:325-326, 263-264; the five KGs of a union) -- encoded as one launch set: the entity / relation
        rows stacked in table order, the graphs joined block-diagonally (graph.union_cache), every kernel and every library
        GEMM once over the stack, BatchNorm with per-block batch statistics and the running estimates moved once per block
        in the callers' order.  Returns a ``Stacked``; ``None`` when the configuration is not covered (callers then make the
        separate calls)."""
        name = not self.args.no_name_info
        if not (self.batched_pairs and len(blocks) >= 2 and self._fused(self.ent_info_att.shape[1] if name else None)):
            return None
        nb = len(blocks)
        if nb > 16:
            return None
        pos = sorted(range(nb), key=lambda k: (blocks[k][2][0], k))       # stack order: by entity id base
        eranges = [(int(blocks[k][2][0]), int(blocks[k][2][1])) for k in pos]
        rranges = [(int(blocks[k][3][0]), int(blocks[k][3][1])) for k in pos]
        if min(hi - lo for lo, hi in eranges) <= 0 or min(hi - lo for lo, hi in rranges) <= 0:
            return None
        dev = self.ent_init_att_completion.device
        comp_att = _stack_rows(self.ent_init_att_completion, eranges)
        rel_comp = _stack_rows(self.rel_init_att_completion, rranges)
        sizes = [hi - lo for lo, hi in eranges]
        rsizes = [hi - lo for lo, hi in rranges]
        lay = self.conv1_completion
        graph = union_cache.get([(blocks[k][0], blocks[k][1], n, m) for k, n, m in zip(pos, sizes, rsizes)], lay.chunk)
        stackpos = {k: i for i, k in enumerate(pos)}
        seg = encoder.RowBlocks(sizes, order=[stackpos[k] for k in range(nb)])
        eoff, roff = [0], [0]
        for n, m in zip(sizes, rsizes):
            eoff.append(eoff[-1] + n)
            roff.append(roff[-1] + m)
        ent_win = [(eoff[stackpos[k]], sizes[stackpos[k]]) for k in range(nb)]
        rel_win = [(roff[stackpos[k]], rsizes[stackpos[k]]) for k in range(nb)]
        if name:
            rel_align = _stack_rows(self.rel_init_att_alignment, rranges)
            info = self._info_rows(tuple(eranges), dev)
            align_out, c1, rel_c1, comp0, rel0 = encoder.forward_name(self, comp_att, rel_comp, rel_align, info, graph, seg=seg,
                                                                      info_persistent=True)
            return Stacked(align_out, [comp0, c1], [rel0, rel_c1], ent_win, rel_win)
        c1, rel_c1, comp0, rel0 = encoder.forward_no_name(self, comp_att, rel_comp, graph, seg=seg)
        return Stacked(c1, [comp0, c1], [rel0, rel_c1], ent_win, rel_win)

    def forward_blocks(self, blocks):
        """[forward_base(*b) for b in blocks] -- through ONE launch set where forward_stacked covers the configuration."""
        st = self.forward_stacked(blocks)
        if st is None:
            return [self.forward_base(*b) for b in blocks]
        return [st.block(k) for k in range(len(blocks))]

    def get_emb_blocks(self, blocks, pyt=False):
        """[get_emb(*b) for b in blocks] with one encoder pass (train.py:450-451 calls get_emb once per KG of the pair)."""
        outs = []
        for align_out, comp_layers, _ in self.forward_blocks(blocks):
            a = ops.row_normalize(align_out).detach().cpu()
            c = ops.row_normalize(comp_layers[-1]).detach().cpu()
            outs.append((a, c) if pyt else (a.numpy(), c.numpy()))
        return outs

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
        st = self.forward_stacked([(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"]),
                                   (edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])])
        if st is not None:                      # both KGs in one launch set: the pairs index two windows of ONE table
            e1 = e2 = st.align_out
            w1, w2 = st.ent_win
        else:
            e1, _, _ = self.forward_base(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"])
            e2, _, _ = self.forward_base(edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])
            w1, w2 = (0, e1.shape[0]), (0, e2.shape[0])
        dev = e1.device
        l0, l1 = _link_columns(links, dev, w1[1], w2[1])
        n = int(l0.numel())
        d = (losses.pair_cosine_distance(e1, l0, e2, l1, w1, w2) + self.margin_align).view(n, 1)
        total = 0
        for left, right in (("neg_left", "neg_right"), ("neg2_left", "neg2_right")):
            b = losses.pair_cosine_distance(e1, _idx(feeddict[left], dev, w1[1]), e2, _idx(feeddict[right], dev, w2[1]), w1, w2)
            total = total + F.relu(d - b.view(n, -1)).sum()
        return total / (2 * self.k * n)

    def completion_loss(self, data, edge_index1, edge_type1, edge_index2, edge_type2, feeddict, source=True):
        h, t, r = data["batch_h"], data["batch_t"], data["batch_r"]
        bs = self.args.batch_size
        links = feeddict["links"]
        st = self.forward_stacked([(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"]),
                                   (edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])])
        loss = None
        if st is not None:                      # both KGs in one launch set; every gather addresses a window of a stacked table
            k = 0 if source else 1
            dev = st.comp[0].device
            cols = _link_columns(links, dev, st.ent_win[0][1], st.ent_win[1][1]) if len(links) else None
            lk = (cols[0], cols[1], st.ent_win[0], st.ent_win[1]) if cols is not None else None
            for layer in range(self.args.num_gcn_layer):
                # one node per layer: the L1 scores (src/jmac_model.py:345-350), the margin ranking term (:351-378, the reference's
                # n-major consumption of the b-major negative block kept), alignment_loss_simple on the links (:237-249) and the
                # sum with the previous layer's term -- one gradient per table, no element-wise glue (losses.completion_layer_loss)
                loss = losses.completion_layer_loss(st.comp[layer], st.rel[layer], h, r, t, bs, self.margin_completion, st.ent_win[k],
                                                    st.rel_win[k], links=lk, add_to=loss)
            return loss if loss is not None else 0
        _, comp1, rel1 = self.forward_base(edge_index1, edge_type1, feeddict["ent_bases1"], feeddict["rel_bases1"])
        _, comp2, rel2 = self.forward_base(edge_index2, edge_type2, feeddict["ent_bases2"], feeddict["rel_bases2"])
        cols = (_link_columns(links, comp1[0].device, comp1[0].shape[0], comp2[0].shape[0]) if len(links) else None)
        for layer in range(self.args.num_gcn_layer):
            ent, rel = (comp1[layer], rel1[layer]) if source else (comp2[layer], rel2[layer])
            loss = losses.completion_layer_loss(ent, rel, h, r, t, bs, self.margin_completion, add_to=loss)
            if cols is not None:                                                 # alignment_loss_simple (:237-249), two tables
                loss = losses.pair_cosine_mean(comp1[layer], cols[0], comp2[layer], cols[1], add_to=loss)
        return loss if loss is not None else 0
