"""Fused gathers of the completion / alignment losses.

``triple_l1_score`` is ``torch.norm(E[h] + R[r] - E[t], 1, -1)`` of ``JMAC.completion_loss``
(src/jmac_model.py:345-350); ``pair_cosine_distance`` is ``1 - sum(normalize(E1[i]) * normalize(E2[j]), 1)``
of ``alignment_loss`` / ``alignment_loss_simple`` (src/jmac_model.py:245-247, 271-291).  One kernel each
way instead of three (two) gathers, the elementwise chain and three (two) index_add passes.
"""
from __future__ import annotations

import torch

from ._lib import check, check_index_range, lib, ptr, require_device, stream


def _fresh(*grads) -> None:
    """Mark gradient buffers this op has just allocated and hands to autograd: the encoder nodes (jmac_amd.encoder._take_grad) may
    add their own contribution to such a buffer in place instead of letting autograd add two [N, d] tensors."""
    for g in grads:
        if g is not None:
            g._jmac_fresh_grad = True


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
_PAIR_PINNED: dict = {}        # entries a stream capture used (strong references; never evicted)


def _pair_index(i1: torch.Tensor, i2: torch.Tensor, off1: int, off2: int, same: bool, n1: int = 0, n2: int = 0):
    """(rec, rowptr).  rec int32 [2L, 4] of jmac_pair_cosine_bwd_{sorted,rows}_f32: the (pair, side) incidences sorted by the gradient row they touch,
    built ONCE per pair of index tensors (identity + version, weak references) -- the seed links of a KG pair are the same
    tensors for every batch of an epoch (train.py:347-352), the mined negatives until the next refresh -- like the CSR of a
    graph.  ``off1`` / ``off2``: first row of the windows the ids are local to; ``same``: both sides share one gradient table.
    Inside a stream capture an unseen pair is sorted inside the capture (and not remembered: those tensors belong to the
    graph's pool).  rowptr int32 [n1 + n2 + 1] (the rows form: n1 rows of the first gradient table, n2 of the second, 0 when both
    sides share one): first sorted position of every gradient row's run."""
    import weakref
    key = (id(i1), id(i2), int(off1), int(off2), bool(same), int(n1), int(n2))
    hit = _PAIR_INDEX.get(key)
    if hit is not None and hit[0]() is i1 and hit[1]() is i2 and hit[2] == (i1._version, i2._version):
        if torch.cuda.is_current_stream_capturing():           # a captured graph bakes these pointers in: the entry (and the index
            _PAIR_PINNED[key] = (i1, i2, hit[3])               # tensors its key names) must outlive every eviction
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
    rowptr = None
    if n1 + n2 > 0:                                            # sorted positions are ordered by (table, row): a CSR pointer over them
        tkey = row if same else torch.where(side, row + n1, row)
        # tkey is non-decreasing (the sort key): the pointer is a search, with no host read (torch.bincount reads its maximum
        # back, which would abort a stream capture that meets an unseen pair -- ADVICE r5)
        rowptr = torch.searchsorted(tkey, torch.arange(n1 + n2 + 1, dtype=tkey.dtype, device=rec.device)).to(torch.int32)
    rec = (rec, rowptr)
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
        # both sides gathered from ONE table (pairs inside a KG, or two blocks of one stacked table): one gradient buffer --
        # no add of the two halves afterwards
        same = ctx.same_table and e1.data_ptr() == e2.data_ptr()
        if ctx.stats is not None and L > 0:
            # deterministic: incidences sorted by gradient row (once per index tensor pair); one wave per gradient ROW writes it
            # once (zeros where no pair touches it): no zero fill of the tables
            de1, de2 = _pair_cosine_rows_bwd(e1, e2, i1, i2, off1, off2, same, ctx.stats, g, None, 0.0)
            return de1, (None if same else de2), None, None, None, None
        de1 = torch.zeros((e1.shape[0], d), dtype=torch.float32, device=e1.device)
        de2 = de1 if same else torch.zeros((e2.shape[0], d), dtype=torch.float32, device=e1.device)
        if L > 0:                                      # d > 512: the atomic form (order-dependent sums)
            check(lib().jmac_pair_cosine_bwd_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), ptr(i1), ptr(i2), L, d,
                                                 ptr(g), _wptr(de1, off1), d, _wptr(de2, off2), d, stream()), "jmac_pair_cosine_bwd_f32")
        return de1, (None if same else de2), None, None, None, None


def _pair_cosine_rows_bwd(e1, e2, i1, i2, off1, off2, same, stats, gvec, gscalar, gscale, de1=None):
    """jmac_pair_cosine_bwd_rows_f32 -> (de1, de2): every row of the gradient table(s) written exactly once."""
    L, d = i1.numel(), e1.shape[1]
    n1, n2 = int(e1.shape[0]), (0 if same else int(e2.shape[0]))
    rec, rowptr = _pair_index(i1, i2, off1, off2, same, n1, n2)
    if de1 is None:
        de1 = torch.empty((n1, d), dtype=torch.float32, device=e1.device)
    de2 = de1 if same else torch.empty((n2, d), dtype=torch.float32, device=e1.device)
    check(lib().jmac_pair_cosine_bwd_rows_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), L, d, ptr(gvec), ptr(gscalar),
                                              float(gscale), ptr(stats), ptr(rec), ptr(rowptr), n1, n2, ptr(de1), d,
                                              ptr(de2) if not same else None, d, stream()), "jmac_pair_cosine_bwd_rows_f32")
    return de1, de2


class _PairCosineMean(torch.autograd.Function):
    """mean_x (1 - cos(e1[i1[x]], e2[i2[x]])) (+ add_to) -> [1]: alignment_loss_simple (src/jmac_model.py:237-249) with the mean
    and the sum with the step's running loss inside the op (no mean / add launch forward, no expand / div backward)."""

    @staticmethod
    def forward(ctx, e1, e2, i1, i2, off1, off2, add_to):
        require_device(e1, e2, i1, i2)
        ctx.same_table = e1 is e2
        e1, e2 = _rows(e1), _rows(e2)
        L, d = i1.numel(), e1.shape[1]
        if e2.shape[1] != d or i2.numel() != L or L == 0 or d > 512:
            raise ValueError("pair_cosine_mean: shapes disagree (or L == 0, d > 512)")
        dev = e1.device
        dist = torch.empty(L, dtype=torch.float32, device=dev)
        stats = torch.empty((L, 4), dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib().jmac_pair_cosine_fwd_stats_f32(_wptr(e1, off1), e1.stride(0), _wptr(e2, off2), e2.stride(0), ptr(i1), ptr(i2), L,
                                                   d, ptr(dist), ptr(stats), stream()), "jmac_pair_cosine_fwd_stats_f32")
        check(lib().jmac_vec_mean_acc_f32(ptr(dist), L, ptr(add_to), ptr(out), stream()), "jmac_vec_mean_acc_f32")
        ctx.save_for_backward(e1, e2, i1, i2, stats)
        ctx.offs, ctx.has_add = (off1, off2), add_to is not None
        return out

    @staticmethod
    def backward(ctx, g):
        e1, e2, i1, i2, stats = ctx.saved_tensors
        off1, off2 = ctx.offs
        g = g.contiguous()
        same = ctx.same_table and e1.data_ptr() == e2.data_ptr()
        need = ctx.needs_input_grad
        de1 = de2 = None
        if need[0] or need[1]:
            de1, de2 = _pair_cosine_rows_bwd(e1, e2, i1, i2, off1, off2, same, stats, None, g, float(i1.numel()))
            _fresh(de1, None if same else de2)
        return de1, (None if same else de2), None, None, None, None, (g if ctx.has_add else None)


def pair_cosine_mean(e1: torch.Tensor, i1: torch.Tensor, e2: torch.Tensor, i2: torch.Tensor, win1=None, win2=None,
                     add_to: torch.Tensor = None) -> torch.Tensor:
    """``pair_cosine_distance(e1, i1, e2, i2, win1, win2).mean() (+ add_to)`` -> [1] as one node (``add_to``: a one-element
    fp32 device tensor, the step's running loss)."""
    dev = e1.device
    off1, n1 = _win(e1, win1)
    off2, n2 = _win(e2, win2)
    i1, i2 = _index(i1, n1, dev), _index(i2, n2, dev)
    if i1.numel() == 0 or e1.shape[1] > 512:
        out = _PairCosine.apply(e1, e2, i1, i2, off1, off2).mean()
        return out if add_to is None else out + add_to
    if add_to is not None:
        add_to = add_to.reshape(1)
    return _PairCosineMean.apply(e1, e2, i1, i2, off1, off2, add_to)


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


_CNT: dict = {}               # (device, rows_ent, rows_rel, d) -> persistent count tables of the exact margin adjoint (always zero between calls)


def _count_tables(dev, rows_e: int, rows_r: int, d: int):
    """The zero-at-rest integer tables jmac_triple_l1_margin_bwd_exact2_f32 accumulates into (and clears again): one pair per
    table shape, shared by every loss op of that shape on the device's stream.  Inside a stream capture an unseen shape gets
    tables of its own (zero-filled inside the capture, not remembered: they belong to the graph's pool)."""
    key = (str(dev), int(rows_e), int(rows_r), int(d))
    hit = _CNT.get(key)
    if hit is None:
        hit = (torch.zeros((rows_e, d), dtype=torch.float32, device=dev), torch.zeros((rows_r, d), dtype=torch.float32, device=dev))
        if not torch.cuda.is_current_stream_capturing():
            _CNT[key] = hit
    return hit


class _LayerLoss(torch.autograd.Function):
    """One layer's term of completion_loss (src/jmac_model.py:331-380) as ONE node:
        loss = add_to + margin_loss(||ent[h] + rel[r] - ent[t]||_1) [+ mean_x (1 - cos(ent[c0[x]], ent[c1[x]]))]
    (the bracket: alignment_loss_simple on the seed links, both sides windows of the same stacked table).  The backward writes
    ONE gradient per table, each row exactly once: the cosine adjoint row by row (first writer, zeros where no link touches a
    row), then the exact-integer L1 adjoint from its persistent count tables on top -- no zero fill, no add of two gradient
    contributions, no scalar arithmetic launches."""

    @staticmethod
    def forward(ctx, ent, rel, h, r, t, margin, add_to, c0, c1, B, K, eoff, roff, coff0, coff1):
        require_device(ent, rel, h, r, t, margin)
        ent, rel = _rows(ent), _rows(rel)
        T, d = h.numel(), ent.shape[1]
        dev = ent.device
        score = torch.empty(T, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib().jmac_triple_l1_fwd_f32(_wptr(ent, eoff), ent.stride(0), _wptr(rel, roff), rel.stride(0), ptr(h), ptr(r), ptr(t),
                                           T, B, d, ptr(score), stream()), "jmac_triple_l1_fwd_f32")
        check(lib().jmac_margin_loss_fwd_acc_f32(ptr(score), B, K, ptr(margin), ptr(add_to), ptr(loss), stream()),
              "jmac_margin_loss_fwd_acc_f32")
        stats = None
        if c0 is not None:
            L = c0.numel()
            dist = torch.empty(L, dtype=torch.float32, device=dev)
            stats = torch.empty((L, 4), dtype=torch.float32, device=dev)
            total = torch.empty(1, dtype=torch.float32, device=dev)
            check(lib().jmac_pair_cosine_fwd_stats_f32(_wptr(ent, coff0), ent.stride(0), _wptr(ent, coff1), ent.stride(0), ptr(c0), ptr(c1),
                                                       L, d, ptr(dist), ptr(stats), stream()), "jmac_pair_cosine_fwd_stats_f32")
            check(lib().jmac_vec_mean_acc_f32(ptr(dist), L, ptr(loss), ptr(total), stream()), "jmac_vec_mean_acc_f32")
            loss = total
        ctx.save_for_backward(ent, rel, h, r, t, score, margin, c0, c1, stats)
        ctx.cfg = (B, K, eoff, roff, coff0, coff1, add_to is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        ent, rel, h, r, t, score, margin, c0, c1, stats = ctx.saved_tensors
        B, K, eoff, roff, coff0, coff1, has_add = ctx.cfg
        d = ent.shape[1]
        dev = ent.device
        g = g.contiguous()
        rows_e, rows_r = int(ent.shape[0]), int(rel.shape[0])
        dent = torch.empty((rows_e, d), dtype=torch.float32, device=dev)
        drel = torch.empty((rows_r, d), dtype=torch.float32, device=dev)
        if c0 is not None:                             # first writer of dent: every row once, zeros where no link touches it
            _pair_cosine_rows_bwd(ent, ent, c0, c1, coff0, coff1, True, stats, None, g, float(c0.numel()), de1=dent)
        cnt_e, cnt_r = _count_tables(dev, rows_e, rows_r, d)
        check(lib().jmac_triple_l1_margin_bwd_exact2_f32(_wptr(ent, eoff), ent.stride(0), _wptr(rel, roff), rel.stride(0), ptr(h), ptr(r),
                                                         ptr(t), B, K, d, ptr(score), ptr(margin), ptr(g), eoff, roff, ptr(cnt_e),
                                                         ptr(cnt_r), ptr(dent), rows_e, 1 if c0 is not None else 0, ptr(drel), rows_r, 0,
                                                         stream()), "jmac_triple_l1_margin_bwd_exact2_f32")
        _fresh(dent, drel)
        return (dent, drel, None, None, None, None, (g if has_add else None), None, None, None, None, None, None, None, None)


def completion_layer_loss(ent: torch.Tensor, rel: torch.Tensor, h: torch.Tensor, r: torch.Tensor, t: torch.Tensor, batch_size: int,
                          margin: torch.Tensor, ent_win=None, rel_win=None, links=None, add_to: torch.Tensor = None) -> torch.Tensor:
    """One layer's term of JMAC.completion_loss (src/jmac_model.py:331-380) -> [1]:
    ``add_to + triple_l1_margin_loss(ent, rel, h, r, t, B, margin, ent_win, rel_win) [+ pair_cosine_distance(ent, c0, ent, c1,
    win0, win1).mean()]`` with ``links = (c0, c1, win0, win1)`` (alignment_loss_simple on the seed links: both sides windows of
    ``ent``), as ONE autograd node where the fused form covers the inputs (dense fp32 tables, d % 4 == 0, d <= 512, a batch of
    B (K + 1) triples, a margin without gradient); the separate ops otherwise."""
    dev = ent.device
    T, B = int(h.numel()), int(batch_size)
    eoff, en = _win(ent, ent_win)
    roff, rn = _win(rel, rel_win)
    d = ent.shape[1]
    fused = (B > 0 and T > B and (T - B) % B == 0 and margin.numel() == 1 and not margin.requires_grad and d % 4 == 0 and d <= 512
             and 4 * B * ((T - B) // B) < (1 << 24) and ent.dtype == torch.float32 and rel.dtype == torch.float32
             and (links is None or links[0].numel() > 0))
    if not fused:
        out = triple_l1_margin_loss(ent, rel, h, r, t, B, margin, ent_win, rel_win)
        if links is not None and links[0].numel() > 0:
            out = out + pair_cosine_distance(ent, links[0], ent, links[1], links[2], links[3]).mean()
        return out if add_to is None else out + add_to
    c0 = c1 = None
    coff0 = coff1 = 0
    if links is not None:
        coff0, n0 = _win(ent, links[2])
        coff1, n1 = _win(ent, links[3])
        c0, c1 = _index(links[0], n0, dev), _index(links[1], n1, dev)
    if add_to is not None:
        add_to = add_to.reshape(1)
    return _LayerLoss.apply(ent, rel, _index(h, en, dev, "batch_h"), _index(r, rn, dev, "batch_r"), _index(t, en, dev, "batch_t"),
                            margin.reshape(1).to(torch.float32), add_to, c0, c1, B, (T - B) // B, eoff, roff, coff0, coff1)


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
