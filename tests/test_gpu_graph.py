"""GPU: graph ingest (jmac_csr_build / jmac_group_build / jmac_items_build) -- bit-exact index work."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import random_graph


def _csr_ref(ei, et, n):
    order = np.lexsort((np.arange(ei.shape[1]), et, ei[0]))          # by destination, then relation type, then input order
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, ei[0] + 1, 1)
    return np.cumsum(rowptr), ei[1][order], et[order], order


@pytest.mark.parametrize("n,nr,e,chunk", [(50, 5, 400, 16), (1000, 40, 20000, 64), (7, 2, 0, 8), (3000, 961, 9000, 256),
                                          (20000, 30, 60000, 32)])
def test_csr_and_schedules(n, nr, e, chunk):
    from jmac_amd.graph import RelGraph
    rng = np.random.default_rng(n + e)
    ei, et = random_graph(rng, n, nr, e, hub=min(e, 5 * chunk + 3) if e else None) if e else (np.zeros((2, 0), np.int64), np.zeros(0, np.int64))
    g = RelGraph(torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda(), n, nr + 1, chunk)
    rowptr, col, typ, perm = _csr_ref(ei, et, n)
    assert (g.rowptr.cpu().numpy() == rowptr).all()
    if e:
        assert (g.perm[:e].cpu().numpy() == perm).all()          # deterministic order
        assert (g.col[:e].cpu().numpy() == col).all()
        assert (g.etype[:e].cpu().numpy() == typ).all()
    # schedule: items tile every row exactly; plain rows <= chunk entries and finalised by their wave (pslot < 0), long rows
    # split into <= chunk-entry items with consecutive partial slots; on small graphs rows of COOP_MIN < deg <= COOP_MAX
    # entries are cooperative: exactly four items of ceil(deg / 4) entries at the HEAD of the list (block aligned), and
    # the empty rows are the TAIL of the list
    from jmac_amd import graph as jgraph
    sch = g.by_dst
    cnt = sch.counts.cpu().numpy()
    items = sch.items.cpu().numpy()[: cnt[0]]
    splits = sch.splits.cpu().numpy()[: cnt[1]]
    assert cnt[0] == sch.n_items_max and cnt[1] == sch.n_splits_max and cnt[2] == sch.n_parts_max
    assert cnt[3] == sch.n_empty and cnt[4] == sch.n_coop
    small = e > 0 and n + e // chunk + 1 <= jgraph.INLINE_EDGES_MAX_ITEMS
    degs = np.diff(rowptr)
    is_coop = (degs > jgraph.coop_min_for(n, e, chunk)) & (degs <= jgraph.COOP_MAX) if small else np.zeros(n, bool)
    assert sch.n_coop == int(is_coop.sum()) and sch.n_empty == int((degs == 0).sum())
    cover = np.zeros(max(e, 1), dtype=np.int64)
    seen_rows = np.zeros(n, dtype=np.int64)
    for pos, (seg, b, en, ps) in enumerate(items):
        assert rowptr[seg] <= b <= en <= rowptr[seg + 1]
        cover[b:en] += 1
        seen_rows[seg] += 1
        deg = degs[seg]
        if is_coop[seg]:
            assert pos < 4 * sch.n_coop and ps == pos and en - b <= -(-deg // 4)
        else:
            assert pos >= 4 * sch.n_coop and en - b <= chunk and (ps < 0) == (deg <= chunk)
        assert (deg == 0) == (pos >= cnt[0] - sch.n_empty)
    assert (cover[:e] == 1).all() and (seen_rows >= 1).all() and (seen_rows[is_coop] == 4).all()
    for k, (seg, p0, nch, flag) in enumerate(splits):
        mine = items[items[:, 0] == seg]
        assert len(mine) == nch and (np.sort(mine[:, 3]) == np.arange(p0, p0 + nch)).all()
        assert (flag == 1) == bool(is_coop[seg]) == (k < sch.n_coop)
    if sch.item_edges is not None:
        ie = sch.item_edges.cpu().numpy()[: cnt[0]]
        for (seg, b, en, ps), row in zip(items, ie):
            want = [col[b] if en > b else -1, typ[b] if en > b else -1, col[b + 1] if en > b + 1 else -1, typ[b + 1] if en > b + 1 else -1]
            assert list(row) == want
    # backward views
    g.ensure_backward_views()
    if e:
        assert (g.dst_of_slot.cpu().numpy() == ei[0][perm]).all()
        so = g.by_src.order[:e].cpu().numpy()
        assert (so == np.argsort(col, kind="stable")).all()
        sp = g.by_src.ptr.cpu().numpy()
        assert (np.diff(sp) == np.bincount(col, minlength=n)).all()
        to = g.by_rel.order[:e].cpu().numpy()
        assert (to == np.argsort(typ, kind="stable")).all()
        assert (np.diff(g.by_rel.ptr.cpu().numpy()) == np.bincount(typ, minlength=nr + 1)).all()
        # destination of every entry in grouped order, and (small graphs) the {slot, destination} pairs of the first two
        # entries of every item inline with the schedule
        dos = ei[0][perm]
        # the backward's views take their small-graph form on their own (lower) threshold; past it, pass A walks a plain
        # by-destination schedule (no cooperative quarters) beside the forward's
        small_bwd = small and n + e // g.chunk + 1 <= jgraph.SMALL_BWD_MAX_ITEMS
        assert (g.by_dst_bwd is g.by_dst) == (small_bwd or not small)
        if g.by_dst_bwd is not g.by_dst:
            c2 = g.by_dst_bwd.counts.cpu().numpy()
            its = g.by_dst_bwd.items.cpu().numpy()[: c2[0]]
            assert g.by_dst_bwd.n_coop == 0 and g.by_dst_bwd.item_edges is None and c2[4] == 0
            cov = np.zeros(e, dtype=np.int64)
            for seg, b, en, ps in its:
                assert rowptr[seg] <= b <= en <= rowptr[seg + 1] and en - b <= chunk and (ps < 0) == (degs[seg] <= chunk)
                cov[b:en] += 1
            assert (cov == 1).all() and len(set(its[:, 0])) == n
        for view, order in ((g.by_src, so), (g.by_rel, to)):
            ed = view.entry_dst[:e].cpu().numpy()
            assert (ed == dos[order]).all()
            assert (view.item_edges is not None) == small_bwd
            if small_bwd:
                c2 = view.counts.cpu().numpy()
                its = view.items.cpu().numpy()[: c2[0]]
                ie = view.item_edges.cpu().numpy()[: c2[0]]
                for (seg, b, en, ps), row in zip(its, ie):
                    want = [order[b] if en > b else -1, ed[b] if en > b else -1,
                            order[b + 1] if en > b + 1 else -1, ed[b + 1] if en > b + 1 else -1]
                    assert list(row) == want


def test_out_of_range_ids_raise_index_error():
    """The reference's torch indexing raises IndexError on a bad entity / relation id; unchecked, the kernels would write
    outside rowptr or scatter gradients outside the tables.  Graph build, loss gathers and the ranking entry point
    validate their indices (device tensors: one jmac_index_check per tensor, cached)."""
    from jmac_amd import losses, scoring
    from jmac_amd.graph import RelGraph
    ei = torch.tensor([[0, 5, 2], [1, 2, 3]]).cuda()
    et = torch.tensor([0, 1, 1]).cuda()
    RelGraph(ei, et, 6, 2)                                            # fine
    with pytest.raises(IndexError):
        RelGraph(ei, et, 5, 2)                                        # destination 5 >= N
    with pytest.raises(IndexError):
        RelGraph(ei, torch.tensor([0, 2, 1]).cuda(), 6, 2)            # relation 2 >= nrel
    with pytest.raises(IndexError):
        RelGraph(torch.tensor([[0, 1, 2], [1, -1, 3]]).cuda(), et, 6, 2)
    ent, rel = torch.randn(10, 8).cuda(), torch.randn(3, 8).cuda()
    h, r, t_ = torch.tensor([1, 2]).cuda(), torch.tensor([0, 2]).cuda(), torch.tensor([9, 3]).cuda()
    losses.triple_l1_score(ent, rel, h, r, t_)
    with pytest.raises(IndexError):
        losses.triple_l1_score(ent, rel, h, torch.tensor([0, 3]).cuda(), t_)
    with pytest.raises(IndexError):
        losses.pair_cosine_distance(ent, torch.tensor([10]).cuda(), ent, torch.tensor([0]).cuda())
    with pytest.raises(IndexError):
        scoring.filtered_rank(torch.randn(2, 10).cuda(), [3, 10])
    # an in-place edit invalidates the cached verdict
    h2 = torch.tensor([1, 2]).cuda()
    losses.triple_l1_score(ent, rel, h2, r, t_)
    h2[0] = 11
    with pytest.raises(IndexError):
        losses.triple_l1_score(ent, rel, h2, r, t_)


def test_union_block_may_name_its_loop_relation():
    """A KG's edge list may carry the loop relation's id (nr_k): the single-KG path builds RelGraph with num_rel = nr_k + 1 and
    accepts it; the block-diagonal union (JMAC.forward_stacked) maps every block's loop id onto the union's ONE loop row
    (it used to raise IndexError there)."""
    from jmac_amd.graph import UnionGraphCache
    dev = torch.device("cuda")
    ei1 = torch.tensor([[0, 1, 2], [1, 2, 0]], device=dev)
    et1 = torch.tensor([0, 4, 2], device=dev)                         # 4 = block 1's loop id (nr_1 = 4)
    ei2 = torch.tensor([[0, 1], [1, 0]], device=dev)
    et2 = torch.tensor([3, 1], device=dev)                            # 3 = block 2's loop id (nr_2 = 3)
    g = UnionGraphCache().get([(ei1, et1, 3, 4), (ei2, et2, 2, 3)])
    assert g.N == 5 and g.num_rel == 8                                # 4 + 3 relation rows + the one loop row (id 7)
    perm = g.perm[:5].long()
    want = torch.tensor([0, 7, 2, 7, 4 + 1], device=dev)[perm]
    assert torch.equal(g.etype[:5].long(), want)
    with pytest.raises((IndexError, RuntimeError, ValueError)):
        UnionGraphCache().get([(ei1, torch.tensor([0, 5, 2], device=dev), 3, 4)])     # past the loop id: still refused
