"""DBPv1 variant of the encoder + losses on the HIP layer (row a17 of SURVEY.md section 8).

Host-side mirror of ``class JMAC_MODEL`` (JMAC_DBPv1/models/jmac_model.py:116-277): same constructor arguments,
parameter names (reference ``state_dict``s load with ``strict=True``) and methods.  Differences from the root model
(SURVEY.md a17): ONE merged graph of both KGs with inverse edges (relation tables have ``2 * num_rel`` rows,
jmac_trainer.py:93-96), no id-base slicing, ReLU between the layer's two relation transforms
(``RelationalAwareLayer``), ``completion_loss`` L2-normalises h, t, r before the L1 score (:245-247), no
``forward_linkpred``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import losses, ops
from .layer import RelationalAwareLayer, get_param
from .model import _idx, _link_columns


class JMAC_MODEL(nn.Module):
    def __init__(self, num_ent, num_rel, ent_info_att, args=None):
        super().__init__()
        self.args = args
        self.act = torch.tanh                                           # BaseModel, :10-14
        self.dim = d = args.emb_dim
        self.completion_dropout = nn.Dropout(args.completion_dropout_rate)
        self.atv_mlp = nn.LeakyReLU(args.leaky_relu_w)
        num_rel = num_rel * 2                                           # :125 inverse relations
        self.ent_info_att = torch.as_tensor(np.asarray(ent_info_att), dtype=torch.float32)   # plain attribute (:127)
        self.ent_completion_att = get_param((num_ent, d))
        self.rel_completion_att = get_param((num_rel, d))
        self.rel_info_att = get_param((num_rel, d))
        self.margin_align = args.margin_align
        self.k = args.num_negative
        mk = lambda: RelationalAwareLayer(d, d, num_rel, rel_dim=d, act=self.act, args=args)
        self.conv1_align, self.conv2_align, self.conv1_completion = mk(), mk(), mk()
        self.margin_completion = nn.Parameter(torch.tensor([float(args.margin_completion)]), requires_grad=False)
        self.align_linear1_1, self.align_linear1_2 = get_param((d * 2, d)), get_param((d, d))
        self.align_linear2_1, self.align_linear2_2 = get_param((d * 2, d)), get_param((d, d))
        self.rel_linear11, self.rel_linear12 = get_param((d, d)), get_param((d, d))
        self.rel_linear21, self.rel_linear22 = get_param((d, d)), get_param((d, d))
        self.rel_linear11_align, self.rel_linear12_align = get_param((d, d)), get_param((d, d))
        self.all_linear_comp = get_param((d * (args.num_gcn_layer + 1), d))

    def _rel_mlp(self, r, w1, w2):
        return torch.mm(self.atv_mlp(torch.mm(r, w1)), w2)

    def forward_base(self, edge_index, edge_type):
        """JMAC_DBPv1/models/jmac_model.py:151-178."""
        dev = self.ent_completion_att.device
        init_comp = self.completion_dropout(ops.row_normalize(self.ent_completion_att))
        a0 = torch.mm(torch.cat((init_comp, self.ent_info_att.to(dev)), dim=1), self.align_linear1_1)
        a1 = self.conv1_align(a0, self.rel_info_att, edge_index, edge_type)
        align_layers, comp_layers, comp_rel_layers = [a0, a1], [self.ent_completion_att], [self.rel_completion_att]
        if self.args.num_gcn_layer == 2:
            c1 = self.conv1_completion(self.ent_completion_att, self.rel_completion_att, edge_index, edge_type)
            c1n = self.completion_dropout(ops.row_normalize(c1))
            a_in = torch.mm(torch.cat((c1n, a1), dim=1), self.align_linear2_1)
            rel_c1 = self._rel_mlp(self.rel_completion_att, self.rel_linear11, self.rel_linear12)
            rel_a_in = self._rel_mlp(self.rel_info_att, self.rel_linear11_align, self.rel_linear12_align)
            a2 = self.conv2_align(a_in, rel_a_in, edge_index, edge_type)
            align_layers.append(a2)
            comp_layers.append(c1)
            comp_rel_layers.append(rel_c1)
        return torch.mm(torch.cat(align_layers, dim=1), self.all_linear_comp), comp_layers, comp_rel_layers

    def get_emb(self, edge_index, edge_type, pyt=False):
        """:181-189."""
        a, comp, _ = self.forward_base(edge_index, edge_type)
        a, c = ops.row_normalize(a).detach().cpu(), ops.row_normalize(comp[-1]).detach().cpu()
        return (a, c) if pyt else (a.numpy(), c.numpy())

    def alignment_loss_simple(self, links, ent_embeddings):
        """:192-199 (both sides index ONE table)."""
        n = ent_embeddings.shape[0]
        l0, l1 = _link_columns(links, ent_embeddings.device, n, n)
        return losses.pair_cosine_distance(ent_embeddings, l0, ent_embeddings, l1).mean()

    def alignment_loss(self, feeddict, edge_index, edge_type):
        """:202-232."""
        e, _, _ = self.forward_base(edge_index, edge_type)
        dev = e.device
        ne = e.shape[0]
        l0, l1 = _link_columns(feeddict["links"], dev, ne, ne)
        n = int(l0.numel())
        d = (losses.pair_cosine_distance(e, l0, e, l1) + self.margin_align).view(n, 1)
        total = 0
        for left, right in (("neg_left", "neg_right"), ("neg2_left", "neg2_right")):
            b = losses.pair_cosine_distance(e, _idx(feeddict[left], dev, ne), e, _idx(feeddict[right], dev, ne))
            total = total + F.relu(d - b.view(n, -1)).sum()
        return total / (2 * self.k * n)

    def completion_loss(self, data, edge_index, edge_type, feeddict):
        """:235-277.  normalize(E[idx]) == normalize(E)[idx]: the tables are normalised once, then the fused L1 gather."""
        _, comp, rel = self.forward_base(edge_index, edge_type)
        h, t, r = data["batch_h"], data["batch_t"], data["batch_r"]
        bs = self.args.completion_batch_size
        loss = 0
        for layer in range(self.args.num_gcn_layer):
            score = losses.triple_l1_score(ops.row_normalize(comp[layer]), ops.row_normalize(rel[layer]), h, r, t, period=bs)
            pos, neg = score[:bs], score[bs:]
            pos = pos.view(-1, min(bs, len(pos))).permute(1, 0)
            neg = neg.view(-1, min(bs, len(neg))).permute(1, 0)
            loss_res = torch.max(pos - neg, -self.margin_completion).mean() + self.margin_completion
            loss = loss + loss_res + self.alignment_loss_simple(feeddict["links"], comp[layer])
        return loss
