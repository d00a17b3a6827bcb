"""GPU parity: scoring kernels (L1 score, filtered rank, MFMA similarity, top-k, entropy, masked softmax,
torch_scatter trio) against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.jmac_oracle as orc
from util import assert_close, load_golden, t


@pytest.mark.parametrize("B,N,d", [(37, 301, 48), (128, 1000, 300), (5, 64, 7), (1000, 2111, 256)])
def test_l1_scores(B, N, d):
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(B + N)
    er, tab = torch.randn(B, d, generator=gen), torch.randn(N, d, generator=gen)
    ref = torch.cdist(er.double(), tab.double(), p=1)
    out = scoring.l1_scores(er.cuda(), tab.cuda())
    assert_close(out, ref, 1e-5)
    out2 = scoring.l1_scores(er.cuda(), tab.cuda(), out=out.clone(), accumulate=True)
    assert_close(out2, 2 * ref, 1e-5)


def test_linkpred_and_ranks_match_reference_golden():
    from jmac_amd import scoring
    g = load_golden("model_small")
    comp = [t(g["comp1_l0"], "cuda"), t(g["comp1_l1"], "cuda")]
    rel = [t(g["rel1_l0"], "cuda"), t(g["rel1_l1"], "cuda")]
    dist = scoring.linkpred_dist(comp, rel, g["lp_h"].tolist(), g["lp_r"].tolist())
    assert_close(dist, g["lp_dist"], 1e-5)
    gold = g["lp_t"]
    fp, fi = t(g["filt_ptr"], "cuda"), t(g["filt_idx"], "cuda")
    # bit-exact index work: the kernel's ranks equal the oracle's rank function on the SAME distances
    for filt in (False, True):
        got = scoring.filtered_rank(dist, gold, fp if filt else None, fi if filt else None).cpu().numpy()
        want = orc.filtered_ranks(dist.cpu(), gold.tolist(), g["filt_ptr"] if filt else None, g["filt_idx"] if filt else None)
        assert (got == want).all()
        # and the reference's own ranks wherever the gold's margin to its neighbours exceeds rounding
        ref = g["ranks_filt%d" % int(filt)]
        d_ref = g["lp_dist"]
        gd = d_ref[np.arange(len(gold)), gold][:, None]
        gap = np.abs(d_ref - gd)
        gap[np.arange(len(gold)), gold] = np.inf
        safe = gap.min(1) > 1e-4 * np.abs(gd[:, 0])
        assert safe.mean() > 0.5 and (got[safe] == ref[safe]).all()
        assert np.allclose(orc.ranking_metrics(got), g["eval_filt%d" % int(filt)], atol=0.02)


def test_filtered_rank_ties_and_random():
    from jmac_amd import scoring
    rng = np.random.default_rng(0)
    B, N = 64, 777
    dist = torch.from_numpy(rng.integers(0, 40, (B, N)).astype(np.float32))     # many exact ties
    gold = rng.integers(0, N, B)
    ptr = [0]
    idx = []
    for b in range(B):
        f = rng.choice(N, rng.integers(0, 30), replace=False)
        idx.extend(f.tolist())
        ptr.append(len(idx))
    ptr, idx = np.array(ptr, np.int32), np.array(idx + [0], np.int32)
    got = scoring.filtered_rank(dist.cuda(), gold, t(ptr, "cuda"), t(idx, "cuda")).cpu().numpy()
    assert (got == orc.filtered_ranks(dist, gold.tolist(), ptr, idx)).all()


@pytest.mark.parametrize("M,N,d", [(33, 70, 8), (300, 1000, 300), (129, 257, 48), (2264, 3000, 256)])
def test_sim_matrix_mfma(M, N, d):
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(M)
    a, b = torch.randn(M, d, generator=gen), torch.randn(N, d, generator=gen)
    # exact-integer check of the MFMA operand / accumulator maps with an asymmetric B
    ai = torch.randint(-3, 4, (M, d), generator=gen).float()
    bi = (torch.arange(N).view(-1, 1) % 5 + torch.arange(d).view(1, -1) % 3).float()
    assert torch.equal(scoring.sim_matrix(ai.cuda(), bi.cuda()).cpu(), ai @ bi.t())
    assert_close(scoring.sim_matrix(a.cuda(), b.cuda()), a.double() @ b.double().t(), 1e-5)


def test_get_neg_matches_reference_golden():
    from jmac_amd import scoring
    g = load_golden("model_small")
    e1, e2 = t(g["emb1_align"], "cuda"), t(g["emb2_align"], "cuda")
    links = g["links"]
    k = g["neg_right"].shape[0] // len(links)
    assert (scoring.get_neg(links[:, 0].tolist(), e1, e2, k).cpu().numpy() == g["neg_right"]).all()
    assert (scoring.get_neg(links[:, 1].tolist(), e2, e1, k).cpu().numpy() == g["neg2_left"]).all()


@pytest.mark.parametrize("L,N,k", [(50, 1000, 25), (7, 30000, 25), (3, 40, 40)])
def test_topk_bit_exact_with_ties(L, N, k):
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(N)
    s = torch.randint(0, 50, (L, N), generator=gen).float()          # heavy ties -> exercises lowest-index rule
    val, idx = scoring.row_topk(s.cuda(), k)
    want = orc.topk_lowest_index(s, k)
    assert (idx.cpu() == want).all()
    assert torch.equal(val.cpu(), s.gather(1, want))
    s2 = torch.randn(L, N, generator=gen)
    _, idx2 = scoring.row_topk(s2.cuda(), k)
    assert (idx2.cpu() == s2.topk(k, dim=1)[1]).all()


@pytest.mark.parametrize("L,N,k", [(40, 30000, 25), (33, 10501, 10), (7, 56589, 25), (16, 1027, 25)])
def test_topk_large_rows_two_pass_select(L, N, k):
    """Config-5 shapes: the histogram/select path (random similarities), odd row lengths (unaligned rows take the
    scalar loads), signed zeros (equal under float comparison: lower index first) and a constant row (every entry
    ties: the candidate list overflows and the in-place arg-max rounds run)."""
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(N + k)
    s = torch.randn(L, N, generator=gen) * 0.06
    s[1, ::3] = 0.0
    s[1, 1::3] = -0.0
    s[1, 2::3] = -1.0
    s[2] = 0.25
    s[3, : N // 2] = s[3, N // 2: 2 * (N // 2)]                        # every value twice
    val, idx = scoring.row_topk(s.cuda(), k)
    want = orc.topk_lowest_index(s, k)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(val.cpu(), s.gather(1, want))


def test_alignment_quality_matches_reference_golden():
    from jmac_amd import scoring
    g = load_golden("model_small")
    e1, e2 = t(g["emb1_align"], "cuda"), t(g["emb2_align"], "cuda")
    ent, sm1, sm2 = scoring.alignment_quality(e1, e2, g["aq_list1"].tolist(), g["aq_list2"].tolist())
    assert abs(ent.item() - float(g["aq_entropy"])) < 1e-4 * abs(float(g["aq_entropy"]))
    assert_close(sm1, g["aq_softmax_rows"], 1e-4)
    assert_close(sm2, g["aq_softmax_cols"], 1e-4)
    gen = torch.Generator().manual_seed(4)
    a = torch.nn.functional.normalize(torch.randn(500, 300, generator=gen))
    b = torch.nn.functional.normalize(torch.randn(700, 300, generator=gen))
    e, hr, hc = scoring.align_entropy(a.cuda(), b.cuda())
    eo, hro, hco = orc.alignment_entropy(a, b)
    assert_close(hr, hro, 1e-4) and assert_close(hc, hco, 1e-4)
    assert abs(e.item() - eo.item()) < 1e-4 * eo.item()


def test_scatter_trio_matches_oracle():
    from jmac_amd import scatter as js
    gen = torch.Generator().manual_seed(8)
    E, d, N = 5000, 24, 300
    src = torch.randn(E, d, generator=gen)
    idx = torch.randint(0, N - 20, (E,), generator=gen)
    assert_close(js.scatter_add(src.cuda(), idx.cuda(), dim=0, dim_size=N), orc.scatter_sum(src, idx, N), 1e-5)
    assert_close(js.scatter(src.cuda(), idx.cuda(), dim=0, out=None, dim_size=N, reduce="sum"), orc.scatter_sum(src, idx, N), 1e-5)
    sc = src[:, :1].contiguous()
    sg = sc.cuda().requires_grad_(True)
    so = sc.clone().requires_grad_(True)
    y = js.scatter_softmax(sg, idx.cuda(), dim=0)
    yo = orc.scatter_softmax(so, idx, N)
    assert_close(y, yo, 1e-5)
    w = torch.randn(E, 1, generator=gen)
    (y * w.cuda()).sum().backward()
    (yo * w).sum().backward()
    assert_close(sg.grad, so.grad, 1e-4, 1e-7)
    deg = js.scatter_add(torch.ones(E).cuda(), idx.cuda(), dim=0, dim_size=N)
    assert torch.equal(deg.cpu(), torch.bincount(idx, minlength=N).float())


def test_alignment_eval_matches_reference_golden():
    from jmac_amd import scoring
    g = load_golden("align_eval")
    e1, e2 = t(g["e1"], "cuda"), t(g["e2"], "cuda")
    for k in (0, 10):
        top_k, hits, mr, mrr = scoring.alignment_test(e1, e2, (1, 5, 10), "cosine", False, k)
        assert np.allclose(hits, g["hits_csls%d" % k], atol=1e-9)
        assert abs(mr - float(g["mr_csls%d" % k])) < 1e-9 and abs(mrr - float(g["mrr_csls%d" % k])) < 1e-9
        s = scoring.alignment_sim(e1, e2, "cosine", False, k)
        _, _, _, _, so = orc.alignment_test(e1.cpu(), e2.cpu(), (1, 5, 10), k)
        assert_close(s, so, 1e-5, 1e-6)
    gen = torch.Generator().manual_seed(12)
    a = torch.randn(2600, 300, generator=gen)
    b = a + 0.8 * torch.randn(2600, 300, generator=gen)
    got = scoring.alignment_test(a.cuda(), b.cuda(), (1, 5, 10), "cosine", False, 10)
    want = orc.alignment_test(a, b, (1, 5, 10), 10)
    assert got[1] == want[1] and abs(got[2] - want[2]) < 1e-9 and abs(got[3] - want[3]) < 1e-12


@pytest.mark.parametrize("n1,n2,k", [(500, 333, 10), (10500, 1000, 10), (64, 5000, 16), (37, 70, 1), (300, 129, 25)])
def test_col_topk_values_equal_transposed_row_topk(n1, n2, k):
    """CSLS column term without the transpose (k > 16 takes the transposing path): same values, bit for bit; ties and
    -inf padding (fewer rows than one block's waves) included."""
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(n1 + n2)
    s = torch.randn(n1, n2, generator=gen)
    s[:, 3] = 0.5                                          # a constant column
    s[: n1 // 2, 5] = s[n1 // 2: 2 * (n1 // 2), 5]          # every value twice
    got = scoring.col_topk_values(s.cuda(), k)
    want = s.t().topk(k, dim=1).values
    assert torch.equal(got.cpu(), want)
    assert torch.equal(got, scoring.row_topk(s.cuda().t().contiguous(), k)[0])


def test_csls_rank_equals_csls_sim_then_rank():
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(9)
    a = torch.nn.functional.normalize(torch.randn(700, 64, generator=gen))
    b = torch.nn.functional.normalize(a + 0.5 * torch.randn(700, 64, generator=gen))
    s = scoring.sim_matrix(a.cuda(), b.cuda())
    s[5] = 0.25                                                          # a constant row: every entry ties after rescoring? no: r2 differs
    gold = torch.randint(0, 700, (700,), generator=gen)
    want = scoring.filtered_rank(scoring.csls_sim(s, 10), gold, descending=True)
    assert torch.equal(scoring.csls_rank(s, 10, gold), want)


def test_dbpv1_get_neg_and_alignment_quality_match_reference_golden():
    """Row a18 on the HIP kernels: get_neg(ILL, output_layer, k) (JMAC_DBPv1/modules/utils/util.py:35-58) bit-exact,
    Trainer.compute_alignment_quality (trainer/jmac_trainer.py:281-300) at 1e-4 -- fixture from the reference's own
    functions; entropy + both softmax orientations come from ONE similarity GEMM."""
    from jmac_amd import scoring
    g = load_golden("scoring_dbpv1")
    emb, k, links = t(g["emb"], "cuda"), int(g["k"]), g["links"]
    assert (scoring.get_neg_dbpv1(links[:, 1].tolist(), emb, k).cpu().numpy() == g["neg2_left"]).all()
    assert (scoring.get_neg_dbpv1(links[:, 0].tolist(), emb, k).cpu().numpy() == g["neg_right"]).all()
    ent, p1, p2 = scoring.alignment_quality_dbpv1(emb, g["list1"].tolist(), g["list2"].tolist())
    assert abs(ent.item() - float(g["entropy"])) < 1e-4 * abs(float(g["entropy"]))
    assert_close(p1, g["softmax_simi"], 1e-4)
    assert_close(p2, g["softmax_simi2"], 1e-4)


@pytest.mark.parametrize("n1,n2", [(90, 90), (257, 1000), (2265, 2265), (64, 4097), (3000, 65)])
def test_row_and_col_softmax_against_torch(n1, n2):
    """jmac_row_softmax_f32 / jmac_col_softmax_f32 (masked, scaled; probabilities + entropies) against float64 torch:
    the column form must equal softmax of the TRANSPOSED matrix (train.py:245,257) without a transposed GEMM."""
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(n1 + n2)
    s = torch.randn(n1, n2, generator=gen) * 0.3
    m1 = torch.rand(n1, generator=gen) < 0.7
    m2 = torch.rand(n2, generator=gen) < 0.6
    m1[0], m2[0] = True, True
    for rm, cm, fill in ((None, None, 0.0), (m1, m2, -1.0)):
        x = s.double()
        if rm is not None:
            keep = rm.view(-1, 1) & cm.view(1, -1)
            x = torch.where(keep, x, torch.full_like(x, fill))
        x = x * 20.0
        pr = torch.softmax(x, dim=1)
        pc = torch.softmax(x.t(), dim=1)
        hr = -(torch.log(pr.clamp_min(1e-300)) * pr).sum(1)
        hc = -(torch.log(pc.clamp_min(1e-300)) * pc).sum(1)
        rmg = rm.cuda() if rm is not None else None
        cmg = cm.cuda() if cm is not None else None
        o, e = scoring.row_softmax(s.cuda(), rmg, cmg, fill, 20.0, True, True)
        assert_close(o, pr, 1e-5, 1e-9, "row softmax")
        assert_close(e, hr, 1e-4, 1e-6, "row entropy")
        o, e = scoring.col_softmax(s.cuda(), rmg, cmg, fill, 20.0, True, True)
        assert o.shape == (n2, n1)
        assert_close(o, pc, 1e-5, 1e-9, "col softmax")
        assert_close(e, hc, 1e-4, 1e-6, "col entropy")
        assert scoring.col_softmax(s.cuda(), rmg, cmg, fill, 20.0, False, True)[0] is None


def _undecided(dist, gold, rel=2e-6):
    """Queries whose gold distance is within fp32 rounding of a neighbour's: the fused and the materialised paths may order
    those differently (one more rounding per layer in the materialised sum)."""
    d = dist.double().cpu().numpy()
    gd = d[np.arange(len(gold)), gold][:, None]
    gap = np.abs(d - gd)
    gap[np.arange(len(gold)), gold] = np.inf
    return gap.min(1) <= rel * np.abs(gd[:, 0])


@pytest.mark.parametrize("B,N,d,nl,bf16,pred_head", [(37, 301, 48, 2, False, False), (130, 1000, 300, 2, False, False),
                                                      (64, 517, 30, 1, False, True), (100, 777, 256, 2, True, False),
                                                      (9, 70, 7, 3, False, False), (1000, 4000, 300, 2, False, False)])
def test_fused_linkpred_ranks_equal_materialised_path(B, N, d, nl, bf16, pred_head):
    """jmac_linkpred_rank_*: forward_linkpred + the filtered ranking loop without the [B, N] matrix (src/jmac_model.py:302-313,
    src/validate.py:50-64) must give the ranks of linkpred_dist -> filtered_rank on the same inputs -- filtered and raw, one to
    three layers, e - r, bf16 tables (config 3), rows that are no multiple of 16 bytes."""
    from jmac_amd import scoring
    rng = np.random.default_rng(B * N + d)
    gen = torch.Generator().manual_seed(d + nl)
    nrel = 11
    comp = [torch.randn(N, d, generator=gen).cuda() for _ in range(nl)]
    rel = [torch.randn(nrel, d, generator=gen).cuda() for _ in range(nl)]
    h, r, gold = rng.integers(0, N, B), rng.integers(0, nrel, B), rng.integers(0, N, B)
    ptr_l, idx = [0], []
    for b in range(B):
        f = rng.choice(N, rng.integers(0, 40), replace=False)
        if b % 3 == 0:
            f = np.unique(np.append(f, gold[b]))                      # the gold itself is listed, as in er_vocab
        idx.extend(f.tolist())
        ptr_l.append(len(idx))
    fp = torch.tensor(ptr_l, dtype=torch.int32).cuda()
    fi = torch.tensor(idx if idx else [0], dtype=torch.int32).cuda()
    dt = torch.bfloat16 if bf16 else torch.float32
    dist = scoring.linkpred_dist(comp, rel, h, r, pred_head=pred_head, table_dtype=dt)
    und = _undecided(dist, gold)
    assert und.mean() < 0.25          # random tables: candidates crowd around the gold (real embeddings separate far better)
    for filt in (True, False):
        want = scoring.filtered_rank(dist, gold, fp if filt else None, fi if filt else None).cpu().numpy()
        got = scoring.linkpred_ranks(comp, rel, h, r, gold, fp if filt else None, fi if filt else None, pred_head=pred_head,
                                     table_dtype=dt).cpu().numpy()
        assert (got[~und] == want[~und]).all(), (int((got != want).sum()), np.abs(got - want).max())
        assert np.abs(got - want).max() <= 2                           # a tie moves a rank by one place
    # and the oracle's rank function on float64 distances of the same (rounded) tables
    if not bf16:
        sign = -1.0 if pred_head else 1.0
        d64 = sum(torch.cdist((c[h].double() + sign * rl[r].double()), c.double(), p=1) for c, rl in zip(comp, rel)).cpu()
        want64 = orc.filtered_ranks(d64, gold.tolist(), np.asarray(ptr_l, dtype=np.int32), np.asarray(idx if idx else [0], dtype=np.int32))
        got = scoring.linkpred_ranks(comp, rel, h, r, gold, fp, fi, pred_head=pred_head).cpu().numpy()
        und64 = _undecided(d64.float(), gold, rel=1e-5)
        assert (got[~und64] == want64[~und64]).all()


def test_fused_linkpred_ranks_match_reference_golden():
    from jmac_amd import scoring
    g = load_golden("model_small")
    comp = [t(g["comp1_l0"], "cuda"), t(g["comp1_l1"], "cuda")]
    rel = [t(g["rel1_l0"], "cuda"), t(g["rel1_l1"], "cuda")]
    gold = g["lp_t"]
    fp, fi = t(g["filt_ptr"], "cuda"), t(g["filt_idx"], "cuda")
    d_ref = g["lp_dist"]
    gd = d_ref[np.arange(len(gold)), gold][:, None]
    gap = np.abs(d_ref - gd)
    gap[np.arange(len(gold)), gold] = np.inf
    safe = gap.min(1) > 1e-4 * np.abs(gd[:, 0])
    for filt in (False, True):
        got = scoring.linkpred_ranks(comp, rel, g["lp_h"].tolist(), g["lp_r"].tolist(), gold, fp if filt else None,
                                     fi if filt else None).cpu().numpy()
        ref = g["ranks_filt%d" % int(filt)]                          # the reference's CompletionEvaluator.test
        assert safe.mean() > 0.5 and (got[safe] == ref[safe]).all()
        assert np.allclose(orc.ranking_metrics(got), g["eval_filt%d" % int(filt)], atol=0.02)


def _two_step_topk(a, b, k):
    from jmac_amd import scoring
    return scoring.row_topk(scoring.sim_matrix(a, b), k)


@pytest.mark.parametrize("L,N,d,k", [(700, 20000, 300, 25), (130, 9000, 64, 10), (33, 30000, 300, 64)])
def test_sim_topk_fused_running_topk_is_bit_identical_to_two_step_form(L, N, d, k):
    """SURVEY K8 / modules/utils/util.py:52-53: at N >= 8192 jmac_sim_topk_f32 never writes the L x N matrix (threshold from a
    column sample, candidates from the product's epilogue, selection from the lists).  Same scores bit for bit, so indices AND
    values must equal sim_matrix + row_topk exactly; the workspace is a fraction of the matrix."""
    from jmac_amd import scoring
    from jmac_amd._lib import lib
    gen = torch.Generator(device="cuda").manual_seed(L + N)
    b = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=gen))
    a = b[torch.randperm(N, device="cuda", generator=gen)[:L]] + 0.05 * torch.randn(L, d, device="cuda", generator=gen)
    idx, val = scoring.sim_topk(a, b, k, return_values=True)
    val2, idx2 = _two_step_topk(a.contiguous(), b, k)
    assert torch.equal(idx, idx2)
    assert torch.equal(val, val2)
    assert int(lib().jmac_sim_topk_workspace_bytes(L, N, k)) < (0.3 if N >= 20000 else 0.6) * L * N * 4


def test_sim_gemm_both_tiles_give_the_same_bits():
    """Round 6: launch_sim picks the 128 x 256 tile (a wave owns 64 x 128) when its quantised makespan is no larger than the
    128 x 128 tile's, else the narrow one; the contraction order per output element is the same, so WHICH one ran must not be
    visible.  A 4 059 x 8 091 product takes the wide tile (32 x 32 = 1 024 wide tiles = 2 rounds of the 512 resident blocks
    against 3 rounds of 768 for the 2 048 narrow ones, on the 256 CUs of an MI355X); the same product asked for in 128-row blocks
    has 32 wide tiles per call -- fewer than the resident blocks -- and takes the narrow one.  Ragged edges on both sides, d = 300 (a row's last 16-k slab is zero-filled past k = 300), and the fused
    top-k (FILTER epilogue of both tiles) against the two-step form on the same shape."""
    from jmac_amd import scoring
    gen = torch.Generator(device="cuda").manual_seed(77)
    M, N, d = 4096 - 37, 8192 - 101, 300
    a = torch.nn.functional.normalize(torch.randn(M, d, device="cuda", generator=gen))
    b = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=gen))
    whole = scoring.sim_matrix(a, b)
    blocks = torch.cat([scoring.sim_matrix(a[i:i + 128].contiguous(), b) for i in range(0, M, 128)])
    assert torch.equal(whole, blocks)
    ref = a[:64].double() @ b.double().t()
    assert float((whole[:64].double() - ref).abs().max()) < 2e-6
    # the fused top-k: jmac_sim_topk_f32 samples the first 2 048 columns and runs the FILTER product over the other 8 142 --
    # 32 x 32 wide tiles again (against 32 x 64 narrow ones): the wide FILTER tile; 130 query rows: the narrow one
    b2 = torch.nn.functional.normalize(torch.randn(10240 - 50, d, device="cuda", generator=gen))
    idx, val = scoring.sim_topk(a, b2, 25, return_values=True)
    val2, idx2 = _two_step_topk(a, b2, 25)
    assert torch.equal(idx, idx2) and torch.equal(val, val2)
    idx3, val3 = scoring.sim_topk(a[:130].contiguous(), b2, 25, return_values=True)
    assert torch.equal(idx3, idx2[:130]) and torch.equal(val3, val2[:130])


@pytest.mark.parametrize("d", [4, 20, 304])
def test_sim_gemm_wide_tile_operand_and_accumulator_maps(d):
    """The 128 x 256 tile's LDS planes, fragment reads and C/D map on exact integers (asymmetric B, ragged edges on both sides);
    d = 4: one 16-k slab, mostly zero-filled; d = 20: two slabs, the second with one live float4; d = 304: 19 full slabs."""
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(d)
    # 32 x 32 = 1 024 wide tiles = 2 rounds of the 512 resident blocks (2 x 512 x 2 = 2 048 units) against 32 x 64 = 2 048 narrow
    # tiles = 3 rounds of 768 (2 304): launch_sim takes WJ = 4 (256 CUs: MI355X); both edges ragged
    M, N = 4096 - 5, 8192 - 3
    ai = torch.randint(-3, 4, (M, d), generator=gen).float()
    bi = (torch.arange(N).view(-1, 1) % 5 + torch.arange(d).view(1, -1) % 3).float()
    got = scoring.sim_matrix(ai.cuda(), bi.cuda())
    assert torch.equal(got.cpu(), ai @ bi.t())
    blocks = torch.cat([scoring.sim_matrix(ai[i:i + 256].cuda(), bi.cuda()) for i in range(0, M, 256)])     # narrow tile
    assert torch.equal(got, blocks)


def test_sim_topk_fused_overflowing_rows_recompute_exactly():
    """Mass ties push more than the list capacity over a row's threshold: those rows recompute their scores with the product's
    own MFMA sequence and must still return the two-step answer (ties -> lower index first)."""
    from jmac_amd import scoring
    gen = torch.Generator(device="cuda").manual_seed(4)
    N, d, k = 12000, 96, 25
    base = torch.nn.functional.normalize(torch.randn(40, d, device="cuda", generator=gen))
    b = base[torch.arange(N, device="cuda") % 40].contiguous()            # every row of b occurs 300 times: 300-way exact ties
    a = torch.cat((base[:5] + 0.01 * torch.randn(5, d, device="cuda", generator=gen),
                   torch.zeros(3, d, device="cuda"),                         # constant rows: every score ties
                   torch.nn.functional.normalize(torch.randn(6, d, device="cuda", generator=gen))))
    idx, val = scoring.sim_topk(a, b, k, return_values=True)
    val2, idx2 = _two_step_topk(a, b, k)
    assert torch.equal(idx, idx2)
    assert torch.equal(val, val2)
    assert idx[5].tolist() == list(range(k))                                # all-equal scores: the k lowest indices
    # and a mix in one call: unique rows (normal lists) beside overflowing ones
    b2 = torch.cat((b[:6000], torch.nn.functional.normalize(torch.randn(6000, d, device="cuda", generator=gen))))
    idx, val = scoring.sim_topk(a, b2, k, return_values=True)
    val2, idx2 = _two_step_topk(a, b2, k)
    assert torch.equal(idx, idx2) and torch.equal(val, val2)


def test_get_neg_at_config5_size_against_oracle_mm_topk():
    """BASELINE config 5's hard-negative mining at full size -- get_neg on [3 000 seeds, 30 000 entities], d = 300, k = 25
    (JMAC_DBPv1/modules/utils/util.py:35-58; root: modules/utils/util.py:31-54), i.e. the FUSED similarity + running top-k path --
    directly against the oracle's mm + topk in float64 on the same normalised embeddings (not through the two-step HIP form):
    every row whose 25th and 26th largest similarities are at least 1e-6 apart (fp32 rounding of a 300-term dot product of
    unit rows is ~1e-7) must return the oracle's index SET, and where all of its top-26 gaps are decided, the oracle's ORDER."""
    import oracle.jmac_oracle as orc
    from jmac_amd import scoring
    gen = torch.Generator().manual_seed(5)
    N, L, d, k = 30000, 3000, 300, 25
    emb = torch.nn.functional.normalize(torch.randn(N, d, generator=gen) + 0.3 * torch.randn(1, d, generator=gen))
    ill = torch.randperm(N, generator=gen)[:L].tolist()
    got = scoring.get_neg_dbpv1(ill, emb.cuda(), k).cpu().view(L, k)
    sim = emb[ill].double() @ emb.double().t()
    top = torch.topk(sim, k + 1, dim=1)
    ref = top.indices[:, :k]
    assert torch.equal(orc.dbpv1_get_neg(ill, emb.double(), k).view(L, k), ref)         # the oracle IS mm + topk
    gaps = top.values[:, :-1] - top.values[:, 1:]                                        # [L, k]: gap below each of the top k
    set_decided = gaps[:, k - 1] >= 1e-6
    order_decided = (gaps >= 1e-6).all(1)
    assert set_decided.float().mean() > 0.95 and order_decided.float().mean() > 0.5
    same_set = (got.sort(1).values == ref.sort(1).values).all(1)
    assert bool(same_set[set_decided].all()), int((~same_set[set_decided]).sum())
    assert bool((got == ref)[order_decided].all()), int((got != ref)[order_decided].any(1).sum())
    # undecided rows differ only inside their near-tie: by at most the entries whose similarity is within 1e-6 of the k-th
    kth = top.values[:, k - 1:k]
    for r in torch.nonzero(~same_set).flatten().tolist():
        wrong = set(got[r].tolist()) ^ set(ref[r].tolist())
        assert all(abs(float(sim[r, j] - kth[r])) < 1e-6 for j in wrong), r
