"""Completion / alignment scoring on libjmac_hip.so (boundary B3).

    l1_scores            == torch.cdist(er, table, p=1)                      src/jmac_model.py:312
    linkpred_dist        == JMAC.forward_linkpred after forward_base          src/jmac_model.py:301-313
    filtered_rank        == filter + sort + np.where of CompletionEvaluator   src/validate.py:50-64
    sim_topk / get_neg   == mm + topk                                         modules/utils/util.py:31-54
    align_entropy        == first half of compute_alignment_quality           train.py:235-248
    alignment_quality    == compute_alignment_quality                         train.py:231-259
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import check, check_index_range, lib, ptr, require_device, stream


def _rows16(t: torch.Tensor, allow_bf16: bool = False) -> torch.Tensor:
    """fp32 (or bf16 where the kernel has a bf16-table form), contiguous, row length padded to a multiple of 4
    elements with zeros (16-byte fp32 rows / 8-byte bf16 rows)."""
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise TypeError("fp32%s only (got %s)" % (" / bf16" if allow_bf16 else "", t.dtype))
    d = t.shape[1]
    if d % 4:
        t = torch.nn.functional.pad(t, (0, 4 - d % 4))
    return t.contiguous()


def l1_scores(er: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None,
              accumulate: bool = False) -> torch.Tensor:
    """``torch.cdist(er, table, p=1)`` -> fp32 [B, N].  A bf16 ``table`` selects the bf16-operand kernel
    (``er`` is rounded to bf16 to match; the accumulation stays fp32) -- BASELINE config 3."""
    require_device(er, table)
    bf16 = table.dtype == torch.bfloat16
    if bf16 and er.dtype != torch.bfloat16:
        er = er.to(torch.bfloat16)
    er, table = _rows16(er, bf16), _rows16(table, bf16)       # zero padding adds |0-0| = 0
    B, d = er.shape
    N = table.shape[0]
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=er.device)
        accumulate = False
    fn, name = (lib().jmac_l1_score_bf16, "jmac_l1_score_bf16") if bf16 else (lib().jmac_l1_score_f32, "jmac_l1_score_f32")
    check(fn(ptr(er), er.shape[1], ptr(table), table.shape[1], B, N, d, ptr(out), out.stride(0),
             1 if accumulate else 0, stream()), name)
    return out


def linkpred_dist(comp_layers: Sequence[torch.Tensor], comp_rel_layers: Sequence[torch.Tensor], e_index, r_index,
                  pred_head: bool = False, table_dtype=torch.float32) -> torch.Tensor:
    """sum over layers of cdist(E_l[h] +/- R_l[r], E_l, p=1) (src/jmac_model.py:302-313).  table_dtype=bfloat16
    rounds the query rows and the candidate table to bf16 (fp32 accumulation) -- BASELINE config 3."""
    dev = comp_layers[0].device
    e_index = torch.as_tensor(e_index, dtype=torch.long, device=dev)
    r_index = torch.as_tensor(r_index, dtype=torch.long, device=dev)
    dist = None
    for ent, rel in zip(comp_layers, comp_rel_layers):
        e, r = ent[e_index], rel[r_index]
        er = e - r if pred_head else e + r                               # jmac_model.py:308-311
        if table_dtype == torch.bfloat16:
            er, ent = er.to(torch.bfloat16), ent.to(torch.bfloat16)
        dist = l1_scores(er, ent, out=dist, accumulate=dist is not None)
    return dist


def linkpred_ranks(comp_layers: Sequence[torch.Tensor], comp_rel_layers: Sequence[torch.Tensor], e_index, r_index, gold,
                   filt_ptr: Optional[torch.Tensor] = None, filt_idx: Optional[torch.Tensor] = None,
                   pred_head: bool = False, table_dtype=torch.float32) -> torch.Tensor:
    """Filtered ranks of the gold tails -- ``filtered_rank(linkpred_dist(...), gold, filt_ptr, filt_idx)`` without the
    [B, N] distance matrix (forward_linkpred src/jmac_model.py:302-313 + the ranking loop of src/validate.py:50-64, i.e. what
    CompletionEvaluator.test needs).  The distance of a candidate is one running fp32 sum over (layer, k) where the
    materialised path rounds once more per layer: same rank unless the gold is tied with a neighbour at fp32 rounding."""
    from ._lib import LinkLayer
    require_device(*comp_layers, *comp_rel_layers)
    nl = len(comp_layers)
    if nl != len(comp_rel_layers) or not 1 <= nl <= 4:
        raise ValueError("1..4 layers of (entity table, relation table)")
    dev = comp_layers[0].device
    N, d = comp_layers[0].shape
    bf16 = table_dtype == torch.bfloat16
    keep, arr = [], (LinkLayer * nl)()
    for l, (ent, rel) in enumerate(zip(comp_layers, comp_rel_layers)):
        if ent.shape != (N, d) or rel.shape[1] != d:
            raise ValueError("layer tables disagree in shape")
        ent, rel = ent.detach().float().contiguous(), rel.detach().float().contiguous()
        tab = _rows16(ent.to(torch.bfloat16), True) if bf16 else ent
        keep += [ent, rel, tab]
        arr[l] = LinkLayer(ptr(ent), ent.stride(0), ptr(rel), rel.stride(0), ptr(tab), tab.stride(0))
    check_index_range(e_index, N, "e_index")
    check_index_range(r_index, comp_rel_layers[0].shape[0], "r_index")
    check_index_range(gold, N, "gold")
    h = torch.as_tensor(e_index, device=dev).to(torch.int32).contiguous()
    r = torch.as_tensor(r_index, device=dev).to(torch.int32).contiguous()
    g = torch.as_tensor(gold, device=dev).to(torch.int32).contiguous()
    B = h.numel()
    if r.numel() != B or g.numel() != B:
        raise ValueError("e_index, r_index and gold must have one entry per query")
    rank = torch.empty(B, dtype=torch.int32, device=dev)
    L = lib()
    ws_bytes = int(L.jmac_linkpred_rank_workspace_bytes(B, d, nl))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    fn, name = (L.jmac_linkpred_rank_bf16, "jmac_linkpred_rank_bf16") if bf16 else (L.jmac_linkpred_rank_f32, "jmac_linkpred_rank_f32")
    check(fn(arr, nl, ptr(h), ptr(r), 1 if pred_head else 0, ptr(g), ptr(filt_ptr), ptr(filt_idx), B, N, d, ptr(rank), ptr(ws),
             ws_bytes, stream()), name)
    return rank


def build_filter_csr(heads, rels, true_tail: Dict, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pack er_vocab[(h, r)] lists (src/validate.py:53; knowledgegraph.py:62-86) as CSR over the batch."""
    ptr_l, idx = [0], []
    for h, r in zip(heads, rels):
        tails = np.unique(np.asarray(true_tail.get((int(h), int(r)), []), dtype=np.int64))
        idx.extend(tails.tolist())
        ptr_l.append(len(idx))
    return (torch.tensor(ptr_l, dtype=torch.int32, device=device),
            torch.tensor(idx if idx else [0], dtype=torch.int32, device=device))


def filtered_rank(dist: torch.Tensor, gold, filt_ptr: Optional[torch.Tensor] = None,
                  filt_idx: Optional[torch.Tensor] = None, descending: bool = False) -> torch.Tensor:
    """1-based rank of the gold tail under ascending distance (``descending``: under descending similarity); filtered
    entries are skipped.  Ties: an equal score counts as ranked before the gold iff its index is lower."""
    require_device(dist)
    if dist.dtype != torch.float32 or dist.stride(1) != 1:
        raise TypeError("dist must be fp32 with unit column stride")
    B, N = dist.shape
    check_index_range(gold, N, "gold")
    gold = torch.as_tensor(gold, device=dist.device).to(torch.int32).contiguous()
    rank = torch.empty(B, dtype=torch.int32, device=dist.device)
    check(lib().jmac_filtered_rank_f32(ptr(dist), dist.stride(0), ptr(gold), ptr(filt_ptr), ptr(filt_idx), B, N,
                                       1 if descending else 0, ptr(rank), stream()), "jmac_filtered_rank_f32")
    return rank


def sim_matrix(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a @ b.T on the fp32-input MFMA (exact fp32 products).  ``out``: optional preallocated [M, N] fp32 result."""
    require_device(a, b)
    a, b = _rows16(a), _rows16(b)
    M, d = a.shape
    N = b.shape[0]
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if c.shape != (M, N) or c.dtype != torch.float32 or not c.is_contiguous():
        raise ValueError("sim_matrix: out must be a contiguous fp32 [%d, %d] tensor" % (M, N))
    check(lib().jmac_sim_matrix_f32(ptr(a), d, ptr(b), d, M, N, d, ptr(c), N, stream()), "jmac_sim_matrix_f32")
    return c


def sim_topk(a: torch.Tensor, b: torch.Tensor, k: int, return_values: bool = False):
    """Indices [L,k] (int64) of the k most similar rows of b for every row of a (descending; ties -> lower index)."""
    require_device(a, b)
    a, b = _rows16(a), _rows16(b)
    L_, d = a.shape
    N = b.shape[0]
    L = lib()
    idx = torch.empty((L_, k), dtype=torch.int32, device=a.device)
    val = torch.empty((L_, k), dtype=torch.float32, device=a.device) if return_values else None
    ws_bytes = int(L.jmac_sim_topk_workspace_bytes(L_, N, int(k)))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=a.device)
    check(L.jmac_sim_topk_f32(ptr(a), d, ptr(b), d, L_, N, d, int(k), ptr(val), ptr(idx), ptr(ws), ws_bytes, stream()),
          "jmac_sim_topk_f32")
    idx = idx.to(torch.int64)
    return (idx, val) if return_values else idx


def row_topk(s: torch.Tensor, k: int):
    require_device(s)
    s = s.contiguous()
    L_, N = s.shape
    idx = torch.empty((L_, k), dtype=torch.int32, device=s.device)
    val = torch.empty((L_, k), dtype=torch.float32, device=s.device)
    check(lib().jmac_row_topk_f32(ptr(s), N, L_, N, int(k), ptr(val), ptr(idx), stream()), "jmac_row_topk_f32")
    return val, idx.to(torch.int64)


COL_TOPK_KMAX = 16


def col_topk_values(s: torch.Tensor, k: int) -> torch.Tensor:
    """The k largest values of every column of s, [n2, k] descending == ``row_topk(s.t().contiguous(), k)[0]`` without
    the transpose (k <= 16; larger k takes the transposing path)."""
    require_device(s)
    s = s.contiguous()
    n1, n2 = s.shape
    if k > COL_TOPK_KMAX:
        return row_topk(s.t().contiguous(), k)[0]
    val = torch.empty((n2, k), dtype=torch.float32, device=s.device)
    L = lib()
    wsb = int(L.jmac_col_topk_workspace_bytes(n1, n2, int(k)))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=s.device)
    check(L.jmac_col_topk_f32(ptr(s), n2, n1, n2, int(k), ptr(val), ptr(ws), wsb, stream()), "jmac_col_topk_f32")
    return val


def get_neg(ILL, emb_src: torch.Tensor, emb_dst: torch.Tensor, k: int) -> torch.Tensor:
    """Same contract as modules/utils/util.py:31-54: flattened [len(ILL)*k] int64 indices into emb_dst."""
    ill = torch.as_tensor(ILL, dtype=torch.long, device=emb_src.device)
    return sim_topk(emb_src.index_select(0, ill), emb_dst, k).reshape(-1)


def align_entropy(e1: torch.Tensor, e2: torch.Tensor, scale: float = 20.0):
    """(entropy, row entropies [n1], column entropies [n2]) of softmax(scale * e1 e2^T)."""
    require_device(e1, e2)
    e1, e2 = _rows16(e1), _rows16(e2)
    n1, d = e1.shape
    n2 = e2.shape[0]
    L = lib()
    hr = torch.empty(n1, dtype=torch.float32, device=e1.device)
    hc = torch.empty(n2, dtype=torch.float32, device=e1.device)
    ws_bytes = int(L.jmac_softmax_entropy_workspace_bytes(n1, n2))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=e1.device)
    check(L.jmac_softmax_entropy_f32(ptr(e1), d, ptr(e2), d, n1, n2, d, float(scale), ptr(hr), ptr(hc), ptr(ws), ws_bytes,
                                     stream()), "jmac_softmax_entropy_f32")
    return hr.mean() + hc.mean(), hr, hc


def masked_row_softmax(s: torch.Tensor, row_mask: Optional[torch.Tensor], col_mask: Optional[torch.Tensor],
                       fill: float = -1.0, scale: float = 20.0) -> torch.Tensor:
    require_device(s)
    s = s.contiguous()
    n1, n2 = s.shape
    out = torch.empty_like(s)
    rm = row_mask.to(torch.uint8).contiguous() if row_mask is not None else None
    cm = col_mask.to(torch.uint8).contiguous() if col_mask is not None else None
    check(lib().jmac_masked_row_softmax_f32(ptr(s), n2, n1, n2, ptr(rm), ptr(cm), float(fill), float(scale), ptr(out), n2,
                                            stream()), "jmac_masked_row_softmax_f32")
    return out


def _mask8(m: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return m.to(torch.uint8).contiguous() if m is not None else None


def row_softmax(s: torch.Tensor, row_mask=None, col_mask=None, fill: float = -1.0, scale: float = 20.0,
                want_out: bool = True, want_entropy: bool = False):
    """softmax over each row of ``where(keep, s, fill) * scale`` (keep = row_mask[i] & col_mask[j]; None = all):
    (probabilities [n1,n2] or None, row entropies [n1] or None)."""
    require_device(s)
    s = s.contiguous()
    n1, n2 = s.shape
    out = torch.empty_like(s) if want_out else None
    ent = torch.empty(n1, dtype=torch.float32, device=s.device) if want_entropy else None
    rm, cm = _mask8(row_mask), _mask8(col_mask)
    check(lib().jmac_row_softmax_f32(ptr(s), n2, n1, n2, ptr(rm), ptr(cm), float(fill), float(scale), ptr(out), n2, ptr(ent),
                                     stream()), "jmac_row_softmax_f32")
    return out, ent


def col_softmax(s: torch.Tensor, row_mask=None, col_mask=None, fill: float = -1.0, scale: float = 20.0,
                want_out: bool = True, want_entropy: bool = False):
    """softmax over each COLUMN of the same masked, scaled matrix, returned transposed:
    ``torch.softmax(where(keep, s, fill).t() * scale, dim=1)`` [n2,n1] (or None) and the column entropies [n2] (or None)
    -- from the row-major ``s`` itself, i.e. without the second, transposed similarity GEMM."""
    require_device(s)
    s = s.contiguous()
    n1, n2 = s.shape
    out_t = torch.empty((n2, n1), dtype=torch.float32, device=s.device) if want_out else None
    ent = torch.empty(n2, dtype=torch.float32, device=s.device) if want_entropy else None
    rm, cm = _mask8(row_mask), _mask8(col_mask)
    L = lib()
    wsb = int(L.jmac_col_softmax_workspace_bytes(n1, n2))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=s.device)
    check(L.jmac_col_softmax_f32(ptr(s), n2, n1, n2, ptr(rm), ptr(cm), float(fill), float(scale), ptr(out_t), n1, ptr(ent),
                                 ptr(ws), wsb, stream()), "jmac_col_softmax_f32")
    return out_t, ent


def alignment_quality(emb1: torch.Tensor, emb2: torch.Tensor, list1, list2, scale: float = 20.0):
    """compute_alignment_quality, train.py:231-259: (entropy, softmax rows [N1,N2], softmax cols [N2,N1]).
    The O(N*T) python membership scans of :252-253 become boolean masks; each of the two similarity matrices
    (:239 and :250) is ONE GEMM, its transposed softmax (:245, :257) is a column pass over the same matrix."""
    dev = emb1.device
    l1 = torch.as_tensor(list1, dtype=torch.long, device=dev)
    l2 = torch.as_tensor(list2, dtype=torch.long, device=dev)
    entropy, _, _ = align_entropy(emb1.index_select(0, l1), emb2.index_select(0, l2), scale)
    m1 = torch.zeros(emb1.shape[0], dtype=torch.bool, device=dev)
    m1[l1] = True
    m2 = torch.zeros(emb2.shape[0], dtype=torch.bool, device=dev)
    m2[l2] = True
    simi = sim_matrix(emb1, emb2)
    return entropy, row_softmax(simi, m1, m2, -1.0, scale)[0], col_softmax(simi, m1, m2, -1.0, scale)[0]


# ---- DBPv1 call sites (row a18): ONE embedding table on both sides ------------------------------------------------
def get_neg_dbpv1(ILL, output_layer: torch.Tensor, k: int) -> torch.Tensor:
    """get_neg(ILL, output_layer, k), JMAC_DBPv1/modules/utils/util.py:35-58: the k most similar rows of the WHOLE
    table (own KG and the entity itself included) for every entity of ILL, flattened [len(ILL)*k] int64."""
    return get_neg(ILL, output_layer, output_layer, k)


def alignment_quality_dbpv1(embedding: torch.Tensor, list1, list2, scale: float = 20.0):
    """Trainer.compute_alignment_quality(embedding, list1, list2), JMAC_DBPv1/trainer/jmac_trainer.py:281-300:
    (entropy, softmax(simi*20, dim=1) [T1,T2], softmax(simi.t()*20, dim=1) [T2,T1]) with simi = E[list1] E[list2]^T.
    One similarity GEMM; the row pass yields softmax + row entropies, the column pass the transposed softmax +
    column entropies."""
    dev = embedding.device
    l1 = torch.as_tensor(list1, dtype=torch.long, device=dev)
    l2 = torch.as_tensor(list2, dtype=torch.long, device=dev)
    simi = sim_matrix(embedding.index_select(0, l1), embedding.index_select(0, l2))
    p_rows, h_rows = row_softmax(simi, None, None, 0.0, scale, True, True)
    p_cols, h_cols = col_softmax(simi, None, None, 0.0, scale, True, True)
    return h_rows.mean() + h_cols.mean(), p_rows, p_cols


# ---- alignment evaluation (next row f1: modules/finding/similarity.py:13-84, alignment.py:10-112) -------------
def csls_sim(sim: torch.Tensor, k: int) -> torch.Tensor:
    """csls_sim, similarity.py:58-78, with calculate_nearest_k defined as the exact mean of the k largest entries
    (the reference's np.partition(-sim, k+1)[:, :k] returns *some* k of the top k+1)."""
    require_device(sim)
    sim = sim.contiguous()
    n1, n2 = sim.shape
    r1 = row_topk(sim, k)[0].mean(1)
    r2 = col_topk_values(sim, k).mean(1)
    out = torch.empty_like(sim)
    check(lib().jmac_csls_apply_f32(ptr(sim), n2, n1, n2, ptr(r1), ptr(r2), ptr(out), n2, stream()), "jmac_csls_apply_f32")
    return out


def csls_rank(sim: torch.Tensor, k: int, gold) -> torch.Tensor:
    """1-based rank of column gold[i] in row i of ``csls_sim(sim, k)`` (descending, ties -> lower index first) without
    materialising the rescored matrix: the same r1 / r2 and the same ``2 s - r1 - r2`` arithmetic, consumed by the count."""
    require_device(sim)
    sim = sim.contiguous()
    n1, n2 = sim.shape
    r1 = row_topk(sim, k)[0].mean(1)
    r2 = col_topk_values(sim, k).mean(1)
    check_index_range(gold, n2, "gold")
    gold = torch.as_tensor(gold, device=sim.device).to(torch.int32).contiguous()
    rank = torch.empty(n1, dtype=torch.int32, device=sim.device)
    check(lib().jmac_csls_rank_f32(ptr(sim), n2, n1, n2, ptr(r1), ptr(r2), ptr(gold), ptr(rank), stream()), "jmac_csls_rank_f32")
    return rank


def alignment_sim(embed1: torch.Tensor, embed2: torch.Tensor, metric: str = "cosine", normalize: bool = False,
                  csls_k: int = 0) -> torch.Tensor:
    """sim(), similarity.py:13-55, for the metrics train.py uses ('cosine', 'inner')."""
    if normalize or metric == "cosine":
        from .ops import row_normalize
        embed1, embed2 = row_normalize(embed1.detach()), row_normalize(embed2.detach())
    elif metric != "inner":
        raise NotImplementedError("metric %r (train.py:105-113 uses 'cosine')" % metric)
    s = sim_matrix(embed1, embed2)
    return csls_sim(s, csls_k) if csls_k > 0 else s


def alignment_test(embeds1: torch.Tensor, embeds2: torch.Tensor, top_k=(1, 5, 10), metric: str = "cosine",
                   normalize: bool = False, csls_k: int = 10):
    """test() / greedy_alignment() / calculate_rank(accurate=True), evaluation.py:20-28, alignment.py:10-112:
    row i of embeds1 is aligned with row i of embeds2.  Returns (top_k, hits [%], mr, mrr)."""
    s = alignment_sim(embeds1, embeds2, metric, normalize, 0)
    n = s.shape[0]
    gold = torch.arange(n, device=s.device, dtype=torch.int32)
    if csls_k > 0:                                                     # CSLS rescoring fused into the rank count
        rank = csls_rank(s, csls_k, gold).to(torch.float64)
    else:
        rank = filtered_rank(s, gold, descending=True).to(torch.float64)   # 1-based position in the descending order
    hits = [float((rank <= k).double().mean().item() * 100.0) for k in top_k]
    return list(top_k), [round(h, 3) for h in hits], float(rank.mean().item()), float((1.0 / rank).mean().item())
