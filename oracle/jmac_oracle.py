"""CPU oracle for the JMAC hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement of the reference's algorithm for the relation-aware GNN
layer and the triple / entity-pair scoring.  It deliberately keeps the
reference's *un-factorised* formulation (per-edge gather -> cat -> mm -> scatter) so that it is an
independent check of the factorised HIP path in ``jmac_amd``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``jmac_amd/`` imports it: the product path fails loudly without the HIP extension.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the reference itself (from
/root/reference, in the build container) and stores its outputs as fixtures under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function below against those fixtures.  The one
third-party dependency of the path, ``torch_scatter`` (unpinned: JMAC_DBPv1/requirements.txt:7; call
sites src/jmac_model.py:7,105 and modules/helper/message_passing.py:2,24,28), is absent from the
reference tree; its published composite algorithm is restated in ``scatter_sum`` /
``scatter_softmax`` below, and that restatement is what the reference was run with when the
fixtures were captured.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# torch_scatter semantics (third party; rusty1s/pytorch_scatter, scatter/composite softmax)
# --------------------------------------------------------------------------------------------
def _bcast_index(index: Tensor, src: Tensor) -> Tensor:
    """torch_scatter.utils.broadcast for dim=0: index [E] -> shape of src."""
    if index.dim() == 1 and src.dim() > 1:
        index = index.view(-1, *([1] * (src.dim() - 1)))
    return index.expand_as(src)


def scatter_sum(src: Tensor, index: Tensor, dim_size: Optional[int] = None) -> Tensor:
    """torch_scatter.scatter_add / scatter(reduce='sum') along dim 0.

    Call sites: src/jmac_model.py:105 (degree), modules/helper/message_passing.py:28 (aggregate).
    Rows that receive nothing stay 0.
    """
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    if src.numel() == 0:
        return out
    return out.scatter_add(0, _bcast_index(index, src), src)


def scatter_max(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    out = torch.full((dim_size,) + tuple(src.shape[1:]), -math.inf, dtype=src.dtype)
    if src.numel() == 0:
        return out
    return out.scatter_reduce(0, _bcast_index(index, src), src, reduce="amax", include_self=True)


def scatter_softmax(src: Tensor, index: Tensor, dim_size: Optional[int] = None) -> Tensor:
    """torch_scatter.composite.scatter_softmax along dim 0 (message_passing.py:24).

    Published algorithm: per-index max, subtract, exp, per-index sum, divide.
    """
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    idx = _bcast_index(index, src)
    mx = scatter_max(src, index, dim_size).gather(0, idx)
    ex = (src - mx).exp()
    den = scatter_sum(ex, index, dim_size).gather(0, idx)
    return ex / den


# --------------------------------------------------------------------------------------------
# RelationAwareLayer (src/jmac_model.py:10-109, JMAC_DBPv1/models/jmac_model.py:20-113)
# --------------------------------------------------------------------------------------------
LAYER_PARAM_NAMES = (
    "rel_transform_weight1", "rel_transform_weight2", "gcn_weight", "loop_rel", "w_att", "a_att",
    "bn.weight", "bn.bias",
)


def edge_norm(edge_index: Tensor, num_ent: int) -> Tensor:
    """compute_norm, src/jmac_model.py:99-109: deg(dst)^-1/2 looked up per edge (dst = row 0)."""
    row = edge_index[0]
    deg = scatter_sum(torch.ones(row.shape[0], dtype=torch.float32), row, num_ent)
    dinv = deg.pow(-0.5)
    dinv[dinv == float("inf")] = 0
    return dinv[row]


def _message_and_aggregate(x: Tensor, rel: Tensor, edge_index: Tensor, edge_type: Tensor,
                           norm: Optional[Tensor], p: Dict[str, Tensor], slope: float,
                           comp_op: str, kink_mask: Optional[Tensor] = None) -> Tensor:
    """propagate + message + scatter_ (message_passing.py:55-90, :4-29; jmac_model.py:56-89).

    Aggregation destination is edge_index[0]; the message source is edge_index[1].

    ``kink_mask`` ([E, d] bool, tests only): evaluate the attention LeakyReLU (:76) as ``where(mask, h, slope*h)``
    instead of by the sign of this evaluation's own h.  The forward is continuous at h = 0, its derivative is not
    (1 vs slope), so two correct evaluations in different precisions (fp32 on the GPU, float64 here) legitimately
    disagree in the gradient wherever a pre-activation of magnitude ~1e-7 rounds to the other side of zero.
    Passing the other evaluation's side of the kink makes the two gradients comparable at full tolerance.
    """
    n = x.shape[0]
    dst, src = edge_index[0], edge_index[1]
    x_i, x_j = x[dst], x[src]                                   # message_passing.py:75,79
    r = rel.index_select(0, edge_type)                          # jmac_model.py:85
    if comp_op == "sub":                                        # :61-64
        m = x_j - r
    elif comp_op == "mult":
        m = x_j * r
    else:
        raise NotImplementedError(comp_op)
    pre_att = torch.cat((x_i, m), dim=1) @ p["w_att"]                       # :75
    if kink_mask is None:
        hidden = F.leaky_relu(pre_att, slope)                               # :76
    else:
        hidden = torch.where(kink_mask, pre_att, pre_att * slope)
    score = hidden @ p["a_att"]                                 # [E,1]
    o = m @ p["gcn_weight"]                                     # :88
    alpha = scatter_softmax(score, dst, n)                      # message_passing.py:24
    if norm is not None:
        alpha = alpha / norm.view(-1, 1)                        # :26  (== alpha * sqrt(deg))
    return scatter_sum(o * alpha, dst, n)                       # :27-28


def _leaky(x: Tensor, slope: float, mask: Optional[Tensor] = None) -> Tensor:
    """LeakyReLU; ``mask`` (tests only, same shape as x): evaluate it as ``where(mask, x, slope*x)`` -- the other
    evaluation's side of every kink (see ``_message_and_aggregate``)."""
    return F.leaky_relu(x, slope) if mask is None else torch.where(mask, x, x * slope)


def transform_relations(p: Dict[str, Tensor], rel_emb: Tensor, slope: float, rel_act: str,
                        kink_mask: Optional[Tensor] = None) -> Tensor:
    """jmac_model.py:39-42 (LeakyReLU) / DBPv1 jmac_model.py:48-52 (ReLU)."""
    rel = torch.cat([rel_emb, p["loop_rel"]], dim=0)
    rel = rel @ p["rel_transform_weight1"]
    rel = _leaky(rel, slope, kink_mask) if rel_act == "leaky_relu" else F.relu(rel)
    return rel @ p["rel_transform_weight2"]


def layer_pre_bn(p: Dict[str, Tensor], ent_emb: Tensor, rel_emb: Tensor, edge_index: Tensor,
                 edge_type: Tensor, slope: float = 0.05, comp_op: str = "sub",
                 rel_act: str = "leaky_relu", kink_mask: Optional[Tensor] = None,
                 rel_kink_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """(message_neighbors, message_self, (nb+self)/2) of jmac_model.py:39-52, before BN/tanh."""
    n = ent_emb.shape[0]
    rel = transform_relations(p, rel_emb, slope, rel_act, rel_kink_mask)
    loop = torch.arange(n)
    loop_index = torch.stack([loop, loop])                      # :44
    loop_type = torch.full((n,), rel.shape[0] - 1, dtype=torch.long)   # :45
    norm = edge_norm(edge_index, n)                             # :47
    nb = _message_and_aggregate(ent_emb, rel, edge_index, edge_type, norm, p, slope, comp_op, kink_mask)   # :49
    sl = _message_and_aggregate(ent_emb, rel, loop_index, loop_type, None, p, slope, comp_op)   # :50
    return nb, sl, (nb + sl) / 2


def layer_forward(p: Dict[str, Tensor], ent_emb: Tensor, rel_emb: Tensor, edge_index: Tensor,
                  edge_type: Tensor, slope: float = 0.05, comp_op: str = "sub",
                  rel_act: str = "leaky_relu", training: bool = True,
                  running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None,
                  momentum: float = 0.1, eps: float = 1e-5, act=torch.tanh,
                  kink_mask: Optional[Tensor] = None, rel_kink_mask: Optional[Tensor] = None) -> Tensor:
    """RelationAwareLayer.forward, jmac_model.py:33-53: act(BatchNorm1d((nb+self)/2)).

    ``running_mean/var`` are updated in place in training mode exactly like nn.BatchNorm1d.
    """
    _, _, pre = layer_pre_bn(p, ent_emb, rel_emb, edge_index, edge_type, slope, comp_op, rel_act, kink_mask, rel_kink_mask)
    d = pre.shape[1]
    if running_mean is None:
        running_mean = torch.zeros(d, dtype=pre.dtype)
    if running_var is None:
        running_var = torch.ones(d, dtype=pre.dtype)
    y = F.batch_norm(pre, running_mean, running_var, p["bn.weight"], p["bn.bias"], training,
                     momentum, eps)
    return act(y)


# --------------------------------------------------------------------------------------------
# JMAC encoder (src/jmac_model.py:172-234)
# --------------------------------------------------------------------------------------------
def _sub(params: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    k = len(prefix) + 1
    return {n[k:]: v for n, v in params.items() if n.startswith(prefix + ".")}


def forward_name(params: Dict[str, Tensor], name_emb: Tensor, edge_index: Tensor, edge_type: Tensor,
                 ent_bases: Sequence[int], rel_bases: Sequence[int], num_gcn_layer: int = 2,
                 slope: float = 0.05, comp_op: str = "sub", training: bool = False,
                 bn_state: Optional[Dict[str, Tensor]] = None,
                 kink_masks: Optional[Dict[str, Tensor]] = None):
    """JMAC.forward_name, jmac_model.py:172-204, with dropout p=0 (eval / deterministic parity).

    ``params`` uses the reference's state_dict names. ``bn_state`` maps e.g.
    'conv1_alignment.bn.running_mean' -> tensor (defaults to fresh BN statistics).
    ``kink_masks`` (tests only) maps a layer name to the ``kink_mask`` of ``_message_and_aggregate``, ``<layer>.rel`` to the
    mask of that layer's relation transform (:41), ``rel_linear11`` / ``rel_linear11_uni`` to the masks of the two relation
    MLPs (:195-196).
    """
    bn_state = bn_state if bn_state is not None else {}
    kink_masks = kink_masks if kink_masks is not None else {}

    def conv(name: str, x: Tensor, r: Tensor) -> Tensor:
        return layer_forward(_sub(params, name), x, r, edge_index, edge_type, slope, comp_op,
                             "leaky_relu", training,
                             bn_state.get(name + ".bn.running_mean"),
                             bn_state.get(name + ".bn.running_var"), kink_mask=kink_masks.get(name),
                             rel_kink_mask=kink_masks.get(name + ".rel"))

    e0, e1 = ent_bases
    r0, r1 = rel_bases
    comp_att = params["ent_init_att_completion"][e0:e1]                       # :176
    name_att = name_emb[e0:e1] @ params["name_linear"]                        # :177
    comp0 = F.normalize(comp_att)                                             # :179 (dropout off)
    align0 = torch.cat((comp0, name_att), dim=1) @ params["uni_linear1_1"]    # :180
    rel_align = params["rel_init_att_alignment"][r0:r1]
    rel_comp = params["rel_init_att_completion"][r0:r1]
    a1 = conv("conv1_alignment", align0, rel_align)                           # :183
    align_layers = [align0, a1]
    comp_layers = [comp_att]
    comp_rel_layers = [rel_comp]
    if num_gcn_layer == 2:
        c1 = conv("conv1_completion", comp_att, rel_comp)                     # :190
        c1n = F.normalize(c1)                                                 # :191
        a_in = torch.cat((c1n, a1), dim=1) @ params["uni_linear2_1"]          # :192
        rel_c1 = _leaky(rel_comp @ params["rel_linear11"], slope, kink_masks.get("rel_linear11")) @ params["rel_linear12"]   # :195
        rel_a_in = (_leaky(rel_align @ params["rel_linear11_uni"], slope, kink_masks.get("rel_linear11_uni"))
                    @ params["rel_linear12_uni"])                                                       # :196
        a2 = conv("conv2_alignment", a_in, rel_a_in)                          # :197
        align_layers.append(a2)
        comp_layers.append(c1)
        comp_rel_layers.append(rel_c1)
    align_out = torch.cat(align_layers, dim=1) @ params["all_linear_completion"]   # :203
    return align_out, comp_layers, comp_rel_layers


def forward_no_name(params: Dict[str, Tensor], edge_index: Tensor, edge_type: Tensor,
                    ent_bases: Sequence[int], rel_bases: Sequence[int], num_gcn_layer: int = 2,
                    slope: float = 0.05, comp_op: str = "sub", training: bool = False,
                    bn_state: Optional[Dict[str, Tensor]] = None):
    """JMAC.forward_no_name, jmac_model.py:207-220."""
    bn_state = bn_state if bn_state is not None else {}
    e0, e1 = ent_bases
    r0, r1 = rel_bases
    comp_att = params["ent_init_att_completion"][e0:e1]
    rel_comp = params["rel_init_att_completion"][r0:r1]
    comp_layers, comp_rel_layers = [comp_att], [rel_comp]
    if num_gcn_layer == 2:
        c1 = layer_forward(_sub(params, "conv1_completion"), comp_att, rel_comp, edge_index,
                           edge_type, slope, comp_op, "leaky_relu", training,
                           bn_state.get("conv1_completion.bn.running_mean"),
                           bn_state.get("conv1_completion.bn.running_var"))
        rel_c1 = F.leaky_relu(rel_comp @ params["rel_linear11"], slope) @ params["rel_linear12"]
        comp_layers.append(c1)
        comp_rel_layers.append(rel_c1)
    return comp_layers[-1], comp_layers, comp_rel_layers


def get_emb(align_out: Tensor, comp_layers: List[Tensor]) -> Tuple[Tensor, Tensor]:
    """JMAC.get_emb, jmac_model.py:223-234: row L2-normalise align output and last completion layer."""
    return F.normalize(align_out, 2, -1), F.normalize(comp_layers[-1], 2, -1)


# --------------------------------------------------------------------------------------------
# Completion scoring + filtered ranking (jmac_model.py:295-313, src/validate.py:22-80)
# --------------------------------------------------------------------------------------------
def l1_scores(er: Tensor, table: Tensor) -> Tensor:
    """torch.cdist(er, table, p=1) (jmac_model.py:312) written as the plain definition."""
    out = torch.empty(er.shape[0], table.shape[0], dtype=er.dtype)
    step = max(1, (1 << 24) // max(1, table.shape[0] * table.shape[1]))
    for s in range(0, er.shape[0], step):
        out[s:s + step] = (er[s:s + step, None, :] - table[None, :, :]).abs().sum(-1)
    return out


def linkpred_dist(comp_layers: List[Tensor], comp_rel_layers: List[Tensor], e_index: Sequence[int],
                  r_index: Sequence[int], pred_head: bool = False) -> Tensor:
    """JMAC.forward_linkpred after forward_base, jmac_model.py:301-313 (all_index = range(N))."""
    dist = 0
    e_index = torch.as_tensor(e_index, dtype=torch.long)
    r_index = torch.as_tensor(r_index, dtype=torch.long)
    for ent, rel in zip(comp_layers, comp_rel_layers):
        e, r = ent[e_index], rel[r_index]
        er = e - r if pred_head else e + r
        dist = dist + torch.cdist(er, ent, p=1)
    return dist


def build_filter_csr(heads: Sequence[int], rels: Sequence[int], true_tail: Dict) -> Tuple[np.ndarray, np.ndarray]:
    """Filter lists of validate.py:53 (er_vocab[(h, r)]) packed as CSR over the batch."""
    ptr, idx = [0], []
    for h, r in zip(heads, rels):
        tails = list(true_tail.get((int(h), int(r)), []))
        idx.extend(int(t) for t in tails)
        ptr.append(len(idx))
    return np.asarray(ptr, dtype=np.int32), np.asarray(idx, dtype=np.int32)


def filtered_ranks(dist: Tensor, gold: Sequence[int], filt_ptr=None, filt_idx=None) -> np.ndarray:
    """CompletionEvaluator.test inner loop, validate.py:50-64.

    predictions = -dist; filtered entries (except the gold) are pushed to -1e6; rank = 1 + position
    of the gold in the descending order.  Tie policy (the reference's torch.sort gives none):
    entries equal to the gold's score count as ranked before it iff their index is lower.
    """
    pred = -dist.clone()
    b = pred.shape[0]
    gold_t = torch.as_tensor(gold, dtype=torch.long)
    if filt_ptr is not None:
        for j in range(b):
            f = torch.as_tensor(filt_idx[filt_ptr[j]:filt_ptr[j + 1]], dtype=torch.long)
            keep = pred[j, gold_t[j]].item()
            pred[j, f] = -1e6                                   # validate.py:56
            pred[j, gold_t[j]] = keep                           # :57
    g = pred.gather(1, gold_t.view(-1, 1))
    ar = torch.arange(pred.shape[1]).view(1, -1)
    before = (pred > g) | ((pred == g) & (ar < gold_t.view(-1, 1)))
    return (before.sum(1) + 1).numpy().astype(np.int32)


def ranking_metrics(ranks: np.ndarray) -> Tuple[float, float, float]:
    """Hits@1, Hits@10, MRR as validate.py:66-74."""
    ranks = np.asarray(ranks, dtype=np.float64)
    return float((ranks <= 1).mean()), float((ranks <= 10).mean()), float((1.0 / ranks).mean())


# --------------------------------------------------------------------------------------------
# Alignment scoring (modules/utils/util.py:31-54, train.py:231-259)
# --------------------------------------------------------------------------------------------
def get_neg(ill: Sequence[int], emb_src: Tensor, emb_dst: Tensor, k: int) -> Tensor:
    """get_neg, util.py:31-54: top-k most similar dst rows per seed; flattened [t*k]."""
    sim = emb_src[torch.as_tensor(ill, dtype=torch.long)] @ emb_dst.t()
    return sim.topk(k, dim=1)[1].reshape(-1)


def topk_lowest_index(sim: Tensor, k: int) -> Tensor:
    """Deterministic top-k (descending value, ties -> lowest index first): the stated tie policy."""
    order = torch.sort(sim, dim=1, descending=True, stable=True)[1]
    return order[:, :k]


def alignment_entropy(e1: Tensor, e2: Tensor, scale: float = 20.0) -> Tuple[Tensor, Tensor, Tensor]:
    """First half of compute_alignment_quality, train.py:235-248, on already-selected rows.

    Returns (entropy, per-row entropies [n1], per-column entropies [n2]).
    """
    simi = e1 @ e2.t()
    p1 = torch.softmax(simi * scale, dim=1)
    h1 = (-torch.log(p1) * p1).sum(1)
    p2 = torch.softmax(simi.t() * scale, dim=1)
    h2 = (-torch.log(p2) * p2).sum(1)
    return h1.mean() + h2.mean(), h1, h2


def alignment_quality(emb1: Tensor, emb2: Tensor, list1: Sequence[int], list2: Sequence[int],
                      scale: float = 20.0):
    """compute_alignment_quality, train.py:231-259 (O(N*T) python scans replaced by masks)."""
    l1 = torch.as_tensor(list1, dtype=torch.long)
    l2 = torch.as_tensor(list2, dtype=torch.long)
    entropy, _, _ = alignment_entropy(emb1[l1], emb2[l2], scale)
    simi = emb1 @ emb2.t()
    m1 = torch.ones(emb1.shape[0], dtype=torch.bool)
    m1[l1] = False
    m2 = torch.ones(emb2.shape[0], dtype=torch.bool)
    m2[l2] = False
    simi[m1] = -1                                               # train.py:254
    simi[:, m2] = -1                                            # :255
    return entropy, torch.softmax(simi * scale, dim=1), torch.softmax(simi.t() * scale, dim=1)


# --------------------------------------------------------------------------------------------
# Losses (jmac_model.py:237-292, :316-380) -- stay in torch in the product too; restated for the
# harness-level parity tests.
# --------------------------------------------------------------------------------------------
def triple_l1_score(ent: Tensor, rel: Tensor, h: Tensor, r: Tensor, t: Tensor) -> Tensor:
    """jmac_model.py:345-350: h = E[batch_h]; t = E[batch_t]; r = R[batch_r]; norm((h + r) - t, 1, -1)."""
    return torch.norm((ent[h] + rel[r]) - ent[t], 1, -1).flatten()


def pair_cosine_distance(e1: Tensor, i1: Tensor, e2: Tensor, i2: Tensor) -> Tensor:
    """jmac_model.py:245-247 / :271-273 / :276-279: 1 - sum(normalize(E1[i]) * normalize(E2[j]), dim=1)."""
    return 1 - torch.sum(F.normalize(e1[i1], 2, -1) * F.normalize(e2[i2], 2, -1), dim=1)


def alignment_loss_simple(links, emb1: Tensor, emb2: Tensor):
    """jmac_model.py:237-249."""
    if not len(links):
        return 0
    links = torch.as_tensor(np.asarray(links), dtype=torch.long)
    a = F.normalize(emb1[links[:, 0]], 2, -1)
    b = F.normalize(emb2[links[:, 1]], 2, -1)
    return (1 - (a * b).sum(1)).mean()


def completion_loss(comp1, rel1, comp2, rel2, batch_h: Tensor, batch_r: Tensor, batch_t: Tensor,
                    links, batch_size: int, margin: float, source: bool = True):
    """jmac_model.py:328-380 given both KGs' forward_base outputs.

    Keeps the reference's layout quirk: negatives are consumed as view(-1, B).permute(1, 0).
    """
    loss = 0
    for layer in range(len(comp1)):
        ent = comp1[layer] if source else comp2[layer]
        rel = rel1[layer] if source else rel2[layer]
        score = torch.norm(ent[batch_h] + rel[batch_r] - ent[batch_t], 1, -1).flatten()
        pos = score[:batch_size]
        pos = pos.view(-1, min(batch_size, len(pos))).permute(1, 0)
        neg = score[batch_size:]
        neg = neg.view(-1, min(batch_size, len(neg))).permute(1, 0)
        m = torch.tensor([margin], dtype=score.dtype)
        loss_res = torch.max(pos - neg, -m).mean() + m
        loss = loss + loss_res + alignment_loss_simple(links, comp1[layer], comp2[layer])
    return loss


def alignment_loss(emb1: Tensor, emb2: Tensor, links, neg_left, neg_right, neg2_left, neg2_right,
                   k: int, margin_align: float):
    """jmac_model.py:265-292 given both KGs' align outputs."""
    links = torch.as_tensor(np.asarray(links), dtype=torch.long)
    n = len(links)

    def cosd(i1, i2):
        a = F.normalize(emb1[torch.as_tensor(np.asarray(i1), dtype=torch.long).reshape(-1)], 2, -1)
        b = F.normalize(emb2[torch.as_tensor(np.asarray(i2), dtype=torch.long).reshape(-1)], 2, -1)
        return 1 - (a * b).sum(1)

    a = cosd(links[:, 0], links[:, 1])
    d = (a + margin_align).view(n, 1)
    l1 = F.relu(-cosd(neg_left, neg_right).view(n, -1) + d)
    l2 = F.relu(-cosd(neg2_left, neg2_right).view(n, -1) + d)
    return (l1.sum() + l2.sum()) / (2 * k * n)


# --------------------------------------------------------------------------------------------
# Factorised identity used by the HIP path -- here only so that the CPU tests
# can check the algebra the kernels rely on against the un-factorised restatement above.
# --------------------------------------------------------------------------------------------
def factorised_pre_bn(p: Dict[str, Tensor], ent_emb: Tensor, rel_emb: Tensor, edge_index: Tensor,
                      edge_type: Tensor, slope: float = 0.05, rel_act: str = "leaky_relu") -> Tensor:
    n, d = ent_emb.shape
    rel = transform_relations(p, rel_emb, slope, rel_act)
    wt, wb = p["w_att"][:d], p["w_att"][d:]
    P, Q, Z = ent_emb @ wt, ent_emb @ wb, ent_emb @ p["gcn_weight"]
    Rq, Rz = rel @ wb, rel @ p["gcn_weight"]
    dst, src = edge_index[0], edge_index[1]
    h = P[dst] + Q[src] - Rq[edge_type]
    s = F.leaky_relu(h, slope) @ p["a_att"]
    alpha = scatter_softmax(s, dst, n)
    deg = scatter_sum(torch.ones(dst.shape[0]), dst, n)
    nb = scatter_sum(alpha * (Z[src] - Rz[edge_type]), dst, n) * deg.sqrt().view(-1, 1)
    return (nb + Z - Rz[-1]) / 2


def aggregate_from_tables(PQZ: Tensor, RR: Tensor, a: Tensor, edge_index: Tensor, edge_type: Tensor, slope: float,
                          loop_rel: int = -1, out_scale: float = 1.0, kink_mask: Optional[Tensor] = None) -> Tensor:
    """The factorised aggregation (factorised_pre_bn above; message_passing.py:4-29 + jmac_model.py:56-89 after
    hoisting the per-edge GEMMs) on GIVEN tables PQZ [N,3d] = P|Q|Z and RR [nr,2d] = Rq|Rz, in their dtype.
    Used to check the bf16-table kernel: pass the bf16-rounded tables widened to float64.  ``kink_mask`` ([E, d] bool,
    tests only): the side of the attention LeakyReLU's kink to take per element (see ``_message_and_aggregate``)."""
    n, d = PQZ.shape[0], PQZ.shape[1] // 3
    P, Q, Z = PQZ[:, :d], PQZ[:, d:2 * d], PQZ[:, 2 * d:]
    Rq, Rz = RR[:, :d], RR[:, d:]
    dst, src = edge_index[0], edge_index[1]
    h = P[dst] + Q[src] - Rq[edge_type]
    s = _leaky(h, slope, kink_mask) @ a.reshape(-1, 1).to(h.dtype)
    alpha = scatter_softmax(s, dst, n)
    deg = scatter_sum(torch.ones(dst.shape[0], dtype=h.dtype), dst, n)
    out = scatter_sum(alpha * (Z[src] - Rz[edge_type]), dst, n) * deg.sqrt().view(-1, 1)
    if loop_rel >= 0:
        out = out + Z - Rz[loop_rel]
    return out * out_scale


def _dst_slices(dst: np.ndarray, nslices: int) -> List[np.ndarray]:
    """Edge subsets by destination range, about equal edge counts, every destination's edges in ONE subset."""
    order = np.argsort(dst, kind="stable")
    ds = dst[order]
    e = order.size
    cuts = [0]
    for k in range(1, nslices):
        c = max(k * e // nslices, cuts[-1])
        while 0 < c < e and ds[c] == ds[c - 1]:
            c += 1
        cuts.append(c)
    cuts.append(e)
    return [order[cuts[k]:cuts[k + 1]] for k in range(nslices) if cuts[k + 1] > cuts[k]]


def aggregate_from_tables_sliced(PQZ: Tensor, RR: Tensor, a: Tensor, edge_index: Tensor, edge_type: Tensor, slope: float,
                                 loop_rel: int = -1, out_scale: float = 1.0, dtype=torch.float64, G: Optional[Tensor] = None,
                                 kink_mask: Optional[Tensor] = None, nslices: int = 8):
    """``aggregate_from_tables`` in ``dtype`` on graphs of ~10^6 edges x d = 300: the softmax and the sum are per destination
    (message_passing.py:24,28 both index on edge_index[0]), so the edge list is cut by destination RANGE and every slice is an
    independent call -- float64 then needs a few GB instead of tens.  With ``G`` [N,d] also returns the gradients
    (dPQZ, dRR, da) of sum(out * G), accumulated slice by slice.  ``kink_mask`` [E, d] bool in the order of edge_index's
    columns.  Tests / bench parity legs only."""
    n, d = PQZ.shape[0], PQZ.shape[1] // 3
    grad = G is not None
    P_, R_, a_ = (t.detach().to(dtype).clone().requires_grad_(grad) for t in (PQZ, RR, a))
    Gd = G.to(dtype) if grad else None
    out = torch.zeros((n, d), dtype=dtype)
    with torch.set_grad_enabled(grad):
        for sl in _dst_slices(edge_index[0].numpy(), nslices):
            idx = torch.from_numpy(sl)
            km = kink_mask[idx] if kink_mask is not None else None
            part = aggregate_from_tables(P_, R_, a_, edge_index[:, idx], edge_type[idx], slope, -1, 1.0, kink_mask=km)
            if grad:
                (part * Gd).sum().mul(out_scale).backward()
            out += part.detach()
            del part
        if loop_rel >= 0:                                        # the self-loop propagate, jmac_model.py:44-45,50
            self_term = P_[:, 2 * d:] - R_[loop_rel, d:]
            if grad:
                (self_term * Gd).sum().mul(out_scale).backward()
            out += self_term.detach()
    out *= out_scale
    if grad:
        return out, (P_.grad, R_.grad, a_.grad)
    return out


# --------------------------------------------------------------------------------------------
# DBPv1 model (row a17): JMAC_DBPv1/models/jmac_model.py:151-277
# --------------------------------------------------------------------------------------------
def dbpv1_forward_base(params: Dict[str, Tensor], info: Tensor, edge_index: Tensor, edge_type: Tensor, num_gcn_layer: int = 2,
                       slope: float = 0.05, training: bool = False, bn_state: Optional[Dict[str, Tensor]] = None):
    """JMAC_MODEL.forward_base, :151-178, dropout off; the layers use ReLU between the relation transforms (:51)."""
    bn_state = bn_state if bn_state is not None else {}

    def conv(name, x, r):
        return layer_forward(_sub(params, name), x, r, edge_index, edge_type, slope, "sub", "relu", training,
                             bn_state.get(name + ".bn.running_mean"), bn_state.get(name + ".bn.running_var"))
    ent, relc, reli = params["ent_completion_att"], params["rel_completion_att"], params["rel_info_att"]
    a0 = torch.cat((F.normalize(ent, p=2, dim=-1), info), dim=1) @ params["align_linear1_1"]
    a1 = conv("conv1_align", a0, reli)
    align_layers, comp_layers, rel_layers = [a0, a1], [ent], [relc]
    if num_gcn_layer == 2:
        c1 = conv("conv1_completion", ent, relc)
        a_in = torch.cat((F.normalize(c1), a1), dim=1) @ params["align_linear2_1"]
        rel_c1 = F.leaky_relu(relc @ params["rel_linear11"], slope) @ params["rel_linear12"]
        rel_a = F.leaky_relu(reli @ params["rel_linear11_align"], slope) @ params["rel_linear12_align"]
        a2 = conv("conv2_align", a_in, rel_a)
        align_layers.append(a2)
        comp_layers.append(c1)
        rel_layers.append(rel_c1)
    return torch.cat(align_layers, dim=1) @ params["all_linear_comp"], comp_layers, rel_layers


def dbpv1_completion_loss(comp, rel, batch_h, batch_r, batch_t, links, batch_size: int, margin: float):
    """:235-277: rows L2-normalised before the L1 score; alignment_loss_simple on one table."""
    links = torch.as_tensor(np.asarray(links), dtype=torch.long)
    loss = 0
    for ent, rl in zip(comp, rel):
        h, t, r = F.normalize(ent[batch_h], 2, -1), F.normalize(ent[batch_t], 2, -1), F.normalize(rl[batch_r], 2, -1)
        score = torch.norm((h + r) - t, 1, -1).flatten()
        pos = score[:batch_size].view(-1, min(batch_size, len(score[:batch_size]))).permute(1, 0)
        neg = score[batch_size:].view(-1, min(batch_size, len(score[batch_size:]))).permute(1, 0)
        m = torch.tensor([margin], dtype=score.dtype)
        loss = loss + torch.max(pos - neg, -m).mean() + m + pair_cosine_distance(ent, links[:, 0], ent, links[:, 1]).mean()
    return loss


# --------------------------------------------------------------------------------------------
# EnTr bookkeeping (next row f4): train.py:297-325
# --------------------------------------------------------------------------------------------
def dbpv1_get_neg(ill: Sequence[int], output_layer: Tensor, k: int) -> Tensor:
    """get_neg(ILL, output_layer, k), JMAC_DBPv1/modules/utils/util.py:35-58: ONE table on both sides, so the top-k of a
    seed runs over all entities of both KGs, the seed itself included (:53-56); flattened [t*k]."""
    ill = torch.as_tensor(np.asarray(ill), dtype=torch.long)
    sim = output_layer[ill] @ output_layer.t()                               # :54-56
    return sim.topk(k, dim=1)[1].reshape(-1)                                 # :57


def dbpv1_alignment_quality(embedding: Tensor, list1: Sequence[int], list2: Sequence[int], scale: float = 20.0):
    """Trainer.compute_alignment_quality, JMAC_DBPv1/trainer/jmac_trainer.py:281-300: (entropy, softmax(simi*20, 1),
    softmax(simi.t()*20, 1)) on the [T1, T2] block of the one embedding table."""
    e1 = embedding[torch.as_tensor(np.asarray(list1), dtype=torch.long)]     # :286-287
    e2 = embedding[torch.as_tensor(np.asarray(list2), dtype=torch.long)]
    simi = e1 @ e2.t()                                                       # :289
    p1 = torch.softmax(simi * scale, dim=1)                                  # :291
    h1 = (-torch.log(p1) * p1).sum(1).mean()                                 # :292-293
    p2 = torch.softmax(simi.t() * scale, dim=1)                              # :295
    h2 = (-torch.log(p2) * p2).sum(1).mean()                                 # :296-297
    return h1 + h2, p1, p2                                                   # :299-300


def transfer_knowledge(triples_src, triples_dst, pairs, keys1: set, keys2: set):
    """transfer_knowledge, train.py:297-325, with the dict of train.py:202 built from the [L,2] pair list.
    Loop restatement (small cases only).  keys*: sets of (h, r, t) tuples, updated in place like the reference's
    string sets.  `links.get(x)` is tested for truthiness: a missing entry and a mapped id of 0 both fail."""
    links = {int(a): int(b) for a, b in pairs}
    inverse = {v: k for k, v in links.items()}
    add_src, add_dst = [], []
    for h, r, t in triples_src:
        if links.get(int(h)) and links.get(int(t)):
            key = (links[int(h)], int(r), links[int(t)])
            if key not in keys2:
                keys2.add(key)
                add_dst.append(key)
    for h, r, t in triples_dst:
        if inverse.get(int(h)) and inverse.get(int(t)):
            key = (inverse[int(h)], int(r), inverse[int(t)])
            if key not in keys1:
                keys1.add(key)
                add_src.append(key)
    return [tuple(int(v) for v in x) for x in triples_src] + add_src, [tuple(int(v) for v in x) for x in triples_dst] + add_dst


# --------------------------------------------------------------------------------------------
# Alignment evaluation (next row f1): modules/finding/similarity.py:13-84, alignment.py:10-112
# --------------------------------------------------------------------------------------------
def csls_sim(sim: Tensor, k: int) -> Tensor:
    """csls_sim, similarity.py:58-78.  calculate_nearest_k (:81-84) takes np.partition(-sim, k+1)[:, :k], i.e.
    *some* k of the k+1 largest entries; this restatement uses the exact k largest (stated policy)."""
    r1 = sim.topk(k, dim=1).values.mean(1)
    r2 = sim.t().topk(k, dim=1).values.mean(1)
    return 2 * sim - r1.view(-1, 1) - r2.view(1, -1)


def alignment_test(e1: Tensor, e2: Tensor, top_k=(1, 5, 10), csls_k: int = 10):
    """test -> greedy_alignment -> calculate_rank(accurate=True) with metric='cosine', normalize=False
    (train.py:105-113): gold of row i is column i; rank = position in the descending similarity order
    (ties: lower index first)."""
    s = F.normalize(e1.double(), 2, -1) @ F.normalize(e2.double(), 2, -1).t()   # scipy cdist path is float64
    s = s.float()
    if csls_k > 0:
        s = csls_sim(s, csls_k)
    n = s.shape[0]
    g = s.diag().view(-1, 1)
    ar = torch.arange(n).view(1, -1)
    rank = ((s > g) | ((s == g) & (ar < torch.arange(n).view(-1, 1)))).sum(1) + 1
    rank = rank.double()
    hits = [round(float((rank <= k).double().mean() * 100), 3) for k in top_k]
    return list(top_k), hits, float(rank.mean()), float((1.0 / rank).mean()), s
