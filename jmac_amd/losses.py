"""Fused gathers of the completion / alignment losses.

``triple_l1_score`` is ``torch.norm(E[h] + R[r] - E[t], 1, -1)`` of ``JMAC.completion_loss``
(src/jmac_model.py:345-350); ``pair_cosine_distance`` is ``1 - sum(normalize(E1[i]) * normalize(E2[j]), 1)``
of ``alignment_loss`` / ``alignment_loss_simple`` (src/jmac_model.py:245-247, 271-291).  One kernel each
way instead of three (two) gathers, the elementwise chain and three (two) index_add passes.
"""
from __future__ import annotations

import torch

from ._lib import check, check_index_range, lib, ptr, require_device, stream


def _rows(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("jmac_amd losses compute in fp32 (got %s)" % t.dtype)
    return t if t.stride(-1) == 1 and t.dim() == 2 else t.contiguous()


def _index(i: torch.Tensor, n: int, device, what: str = "index") -> torch.Tensor:
    """int64, contiguous, on ``device`` and range-checked against the n rows it addresses (the backward scatters into
    those rows with float atomics: the reference raises IndexError on a bad id, so does this)."""
    check_index_range(i, n, what)                      # host tensors on the host; device tensors once per tensor
    if i.dim() != 1:                                   # (a 1-D tensor stays THE SAME OBJECT: per-tensor caches key on identity)
        i = i.reshape(-1)
    if i.dtype != torch.int64 or i.device != device:
        i = i.to(device=device, dtype=torch.int64)
    return i.contiguous()


class _TripleL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent, rel, h, r, t, period):
        require_device(ent, rel, h, r, t)
        ent, rel = _rows(ent), _rows(rel)
        T, d = h.numel(), ent.shape[1]
        if rel.shape[1] != d or r.numel() != T or t.numel() != T:
            raise ValueError("triple_l1_score: shapes disagree")
        score = torch.empty(T, dtype=torch.float32, device=ent.device)
        check(lib().jmac_triple_l1_fwd_f32(ptr(ent), ent.stride(0), ptr(rel), rel.stride(0), ptr(h), ptr(r), ptr(t), T,
                                           int(period), d, ptr(score), stream()), "jmac_triple_l1_fwd_f32")
        ctx.save_for_backward(ent, rel, h, r, t)
        ctx.period = int(period)
        return score

    @staticmethod
    def backward(ctx, g):
        ent, rel, h, r, t = ctx.saved_tensors
        T, d = h.numel(), ent.shape[1]
        g = g.contiguous()
        dent = torch.zeros((ent.shape[0], d), dtype=torch.float32, device=ent.device)
        drel = torch.zeros((rel.shape[0], d), dtype=torch.float32, device=ent.device)
        check(lib().jmac_triple_l1_bwd_f32(ptr(ent), ent.stride(0), ptr(rel), rel.stride(0), ptr(h), ptr(r), ptr(t), T,
                                           ctx.period, d, ptr(g), ptr(dent), d, ptr(drel), d, stream()), "jmac_triple_l1_bwd_f32")
        return dent, drel, None, None, None, None


def triple_l1_score(ent: torch.Tensor, rel: torch.Tensor, h: torch.Tensor, r: torch.Tensor, t: torch.Tensor,
                    period: int = 0) -> torch.Tensor:
    """``torch.norm(ent[h] + rel[r] - ent[t], 1, -1)`` -> [T] (src/jmac_model.py:345-350).

    ``period``: hint that triples x, x+period, ... share (h, r), as in the reference's batches
    (train.py:347-352, period = batch size); results do not depend on it."""
    dev = ent.device
    return _TripleL1.apply(ent, rel, _index(h, ent.shape[0], dev, "batch_h"), _index(r, rel.shape[0], dev, "batch_r"),
                           _index(t, ent.shape[0], dev, "batch_t"), int(period))


_PAIR_INDEX: dict = {}
_PAIR_INDEX_CAP = 64


def _pair_index(i1: torch.Tensor, i2: torch.Tensor, off1: int, off2: int, same: bool):
    """rec int32 [2L, 4] of jmac_pair_cosine_bwd_sorted_f32: the (pair, side) incidences sorted by the gradient row they touch,
    built ONCE per pair of index tensors (identity + version, weak references) -- the seed links of a KG pair are the same
    tensors for every batch of an epoch (train.py:347-352), the mined negatives until the next refresh -- like the CSR of a
    graph.  ``off1`` / ``off2``: first row of the windows the ids are local to; ``same``: both sides share one gradient table.
    Inside a stream capture an unseen pair is sorted inside the capture (and not remembered: those tensors belong to the
    graph's pool)."""
    import weakref
    key = (id(i1), id(i2), int(off1), int(off2), bool(same))
    hit = _PAIR_INDEX.get(key)
    if hit is not None and hit[0]() is i1 and hit[1]() is i2 and hit[2] == (i1._version, i2._version):
        return hit[3]
    L = i1.numel()
    big = 1 << 40                                               # side-1 rows of a second table sort behind every side-0 row
    keys = torch.cat((i1 + off1, i2 + (off2 if same else off2 + big)))
    skey, order = torch.sort(keys, stable=True)
    side = order >= L
    x = torch.where(side, order - L, order)
    own = torch.where(side, i2[x], i1[x])
    partner = torch.where(side, i1[x], i2[x])
    head = torch.ones_like(skey, dtype=torch.bool)
    head[1:] = skey[1:] != skey[:-1]
    row = torch.where(side & (not same), skey - big, skey) if not same else skey
    flags = (head.to(torch.int64) << 31) | (side.to(torch.int64) << 30) | ((side & (not same)).to(torch.int64) << 29) | row
    rec = torch.stack((x, own, partner, flags), dim=1)
    rec = (rec & 0xFFFFFFFF).to(torch.int64)
    rec = torch.where(rec >= (1 << 31), rec - (1 << 32), rec).to(torch.int32).contiguous()      # two's-complement int32 words
    if not torch.cuda.is_current_stream_capturing():
        if int(row.max()) >= (1 << 29):
            raise ValueError("pair cosine backward: gradient rows beyond 2^29")
        for k in [k for k, v in _PAIR_INDEX.items() if v[0]() is None or v[1]() is None]:
            del _PAIR_INDEX[k]
        if len(_PAIR_INDEX) >= _PAIR_INDEX_CAP:
            _PAIR_INDEX.pop(next(iter(_PAIR_INDEX)))
        try:
            _PAIR_INDEX[key] = (weakref.ref(i1), weakref.ref(i2), (i1._version, i2._version), rec)
        except TypeError:                                      # pragma: no cover
            pass
    return rec


class _PairCosine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e1, e2, i1, i2, off1, off2):
        require_device(e1, e2, i1, i2)
        ctx.same_table = e1 is e2                     # pairs inside one table (the caller passed the same tensor twice)
        e1, e2 = _rows(e1), _rows(e2)
        L, d = i1.numel(), e1.shape[1]
        if e2.shape[1] != d or i2.numel() != L:
            raise ValueError("pair_cosine_distance: shapes disagree")
        dist = torch.empty(L, dtype=torch.float32, device=e1.device)
        needs_grad = (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and d <= 512
        stats = torch.empty((max(L, 1), 4), dtype=torch.float32, device=e1.device) if needs_grad else None
        if stats is not None:
            check(lib().jmac_pair_cosine_fwd_stats_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), ptr(i1), ptr(i2), L,
                                                       d, ptr(dist), ptr(stats), stream()), "jmac_pair_cosine_fwd_stats_f32")
        else:
            check(lib().jmac_pair_cosine_fwd_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), ptr(i1), ptr(i2), L, d,
                                                 ptr(dist), stream()), "jmac_pair_cosine_fwd_f32")
        ctx.save_for_backward(e1, e2, i1, i2)
        ctx.offs, ctx.stats = (off1, off2), stats
        return dist

    @staticmethod
    def backward(ctx, g):
        e1, e2, i1, i2 = ctx.saved_tensors
        off1, off2 = ctx.offs
        L, d = i1.numel(), e1.shape[1]
        g = g.contiguous()
        de1 = torch.zeros((e1.shape[0], d), dtype=torch.float32, device=e1.device)
        # both sides gathered from ONE table (pairs inside a KG, or two blocks of one stacked table): one gradient buffer --
        # no second zero-fill, no add of the two halves afterwards
        same = ctx.same_table and e1.data_ptr() == e2.data_ptr()
        de2 = de1 if same else torch.zeros((e2.shape[0], d), dtype=torch.float32, device=e1.device)
        if ctx.stats is not None and L > 0:
            # deterministic: incidences sorted by gradient row (once per index tensor pair), every touched row written once
            rec = _pair_index(i1, i2, off1, off2, same)
            check(lib().jmac_pair_cosine_bwd_sorted_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), L, d, ptr(g),
                                                        ptr(ctx.stats), ptr(rec), ptr(de1), d, ptr(de2), d, stream()),
                  "jmac_pair_cosine_bwd_sorted_f32")
        else:                                          # d > 512: the atomic form (order-dependent sums)
            check(lib().jmac_pair_cosine_bwd_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), ptr(i1), ptr(i2), L, d,
                                                 ptr(g), _wptr(de1, off1), d, _wptr(de2, off2), d, stream()), "jmac_pair_cosine_bwd_f32")
        return de1, (None if same else de2), None, None, None, None


def pair_cosine_distance(e1: torch.Tensor, i1: torch.Tensor, e2: torch.Tensor, i2: torch.Tensor, win1=None,
                         win2=None) -> torch.Tensor:
    """``1 - sum(F.normalize(e1[i1], 2, -1) * F.normalize(e2[i2], 2, -1), 1)`` -> [L].  ``win1`` / ``win2`` = (first row,
    rows): the ids are local to that window of the table."""
    dev = e1.device
    off1, n1 = _win(e1, win1)
    off2, n2 = _win(e2, win2)
    return _PairCosine.apply(e1, e2, _index(i1, n1, dev), _index(i2, n2, dev), off1, off2)


class _MarginLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, margin, B, K):
        require_device(score, margin)
        score = score.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=score.device)
        check(lib().jmac_margin_loss_fwd_f32(ptr(score), B, K, ptr(margin), ptr(loss), stream()), "jmac_margin_loss_fwd_f32")
        ctx.save_for_backward(score, margin)
        ctx.bk = (B, K)
        return loss

    @staticmethod
    def backward(ctx, g):
        score, margin = ctx.saved_tensors
        B, K = ctx.bk
        g = g.contiguous()
        dscore = torch.empty_like(score)
        check(lib().jmac_margin_loss_bwd_f32(ptr(score), B, K, ptr(margin), ptr(g), ptr(dscore), stream()), "jmac_margin_loss_bwd_f32")
        return dscore, None, None, None


def _win(table: torch.Tensor, win):
    """(row offset, rows) of the window of ``table`` the indices address: all of it, or the block of a stacked launch set
    (JMAC.forward_stacked) the caller names -- the kernels then see the block's first row as row 0, so the reference's
    block-local ids are used as they are and no sliced view (with its zero-fill + copy + add in the backward) exists."""
    if win is None:
        return 0, int(table.shape[0])
    off, n = int(win[0]), int(win[1])
    if off < 0 or n < 0 or off + n > table.shape[0]:
        raise ValueError("row window [%d, %d) outside a table of %d rows" % (off, off + n, table.shape[0]))
    return off, n


def _wptr(t: torch.Tensor, off: int) -> int:
    return t.data_ptr() + off * t.stride(0) * t.element_size()


class _TripleL1Margin(torch.autograd.Function):
    """score = triple L1 distances, loss = margin ranking loss of the score vector, as ONE node: the backward derives the score
    gradient inside the L1 adjoint kernel (no dscore vector, no launch for it).  ``eoff`` / ``roff``: first row of the entity /
    relation window the indices are local to."""

    @staticmethod
    def forward(ctx, ent, rel, h, r, t, margin, B, K, eoff, roff, en, rn):
        require_device(ent, rel, h, r, t, margin)
        ent, rel = _rows(ent), _rows(rel)
        T, d = h.numel(), ent.shape[1]
        if rel.shape[1] != d or r.numel() != T or t.numel() != T or T != B * (K + 1):
            raise ValueError("triple_l1_margin_loss: shapes disagree")
        score = torch.empty(T, dtype=torch.float32, device=ent.device)
        loss = torch.empty(1, dtype=torch.float32, device=ent.device)
        check(lib().jmac_triple_l1_fwd_f32(_wptr(ent, eoff), ent.stride(0), _wptr(rel, roff), rel.stride(0), ptr(h), ptr(r), ptr(t),
                                           T, B, d, ptr(score), stream()), "jmac_triple_l1_fwd_f32")
        check(lib().jmac_margin_loss_fwd_f32(ptr(score), B, K, ptr(margin), ptr(loss), stream()), "jmac_margin_loss_fwd_f32")
        ctx.save_for_backward(ent, rel, h, r, t, score, margin)
        ctx.bk = (B, K, eoff, roff, en, rn)
        return loss

    @staticmethod
    def backward(ctx, g):
        ent, rel, h, r, t, score, margin = ctx.saved_tensors
        B, K, eoff, roff, en, rn = ctx.bk
        d = ent.shape[1]
        g = g.contiguous()
        dent = torch.zeros((ent.shape[0], d), dtype=torch.float32, device=ent.device)
        drel = torch.zeros((rel.shape[0], d), dtype=torch.float32, device=ent.device)
        # bitwise reproducible: the atomics add exact integers (the loss' gradient is gloss / (2 B K) times an integer matrix),
        # a second pass scales the two windows
        check(lib().jmac_triple_l1_margin_bwd_exact_f32(_wptr(ent, eoff), ent.stride(0), _wptr(rel, roff), rel.stride(0), ptr(h), ptr(r),
                                                        ptr(t), B, K, d, ptr(score), ptr(margin), ptr(g), _wptr(dent, eoff), d, en,
                                                        _wptr(drel, roff), d, rn, stream()),
              "jmac_triple_l1_margin_bwd_exact_f32")
        return dent, drel, None, None, None, None, None, None, None, None, None, None


def triple_l1_margin_loss(ent: torch.Tensor, rel: torch.Tensor, h: torch.Tensor, r: torch.Tensor, t: torch.Tensor,
                          batch_size: int, margin: torch.Tensor, ent_win=None, rel_win=None) -> torch.Tensor:
    """``margin_loss(triple_l1_score(ent, rel, h, r, t, period=batch_size), batch_size, margin)`` (src/jmac_model.py:345-378)
    as one autograd node; batches that are not ``B (K + 1)`` triples long, or a margin that wants a gradient, take the two
    ops.  ``ent_win`` / ``rel_win`` = (first row, rows): the ids are local to that window of the table (a KG's block of a
    stacked encoder output)."""
    dev = ent.device
    T, B = int(h.numel()), int(batch_size)
    eoff, en = _win(ent, ent_win)
    roff, rn = _win(rel, rel_win)
    if B <= 0 or T <= B or (T - B) % B != 0 or margin.numel() != 1 or margin.requires_grad:
        return margin_loss(triple_l1_score(ent[eoff:eoff + en], rel[roff:roff + rn], h, r, t, period=B), B, margin)
    return _TripleL1Margin.apply(ent, rel, _index(h, en, dev, "batch_h"), _index(r, rn, dev, "batch_r"),
                                 _index(t, en, dev, "batch_t"), margin.reshape(1).to(torch.float32), B, (T - B) // B, eoff, roff, en, rn)


def margin_loss(score: torch.Tensor, batch_size: int, margin: torch.Tensor) -> torch.Tensor:
    """``torch.max(pos - neg, -margin).mean() + margin`` of completion_loss (src/jmac_model.py:351-378) on one batch's score
    vector [B + B*K]: ``pos = score[:B].view(-1, B).permute(1, 0)``, ``neg = score[B:].view(-1, B).permute(1, 0)`` -- the
    reference consumes its b-major negative block as n-major; kept as is.  ``margin``: the model's fixed one-element
    parameter (no gradient).  One launch each way instead of ~10; shape [1] like the reference's expression."""
    T, B = int(score.numel()), int(batch_size)
    if score.dim() != 1 or score.dtype != torch.float32 or margin.numel() != 1 or margin.requires_grad \
            or T <= B or (T - B) % B != 0:
        pos, neg = score[:B], score[B:]                           # ragged / exotic inputs: the reference's own expression
        pos = pos.view(-1, min(B, len(pos))).permute(1, 0)
        neg = neg.view(-1, min(B, len(neg))).permute(1, 0)
        return torch.max(pos - neg, -margin).mean() + margin
    return _MarginLoss.apply(score, margin.reshape(1).to(torch.float32), B, (T - B) // B)
