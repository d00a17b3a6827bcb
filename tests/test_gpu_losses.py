"""GPU parity: fused loss gathers (row f3) against the oracle's restatement of src/jmac_model.py:245-247,
271-291, 345-350 evaluated in float64, forward and backward, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.jmac_oracle as orc
from util import assert_close, load_golden, t


@pytest.mark.parametrize("period", [0, 100, 7])
@pytest.mark.parametrize("N,nr,d,T", [(50, 7, 300, 400), (301, 20, 256, 2600), (64, 5, 7, 33), (11, 3, 30, 1), (40, 4, 512, 999)])
def test_triple_l1_score_fwd_bwd(N, nr, d, T, period):
    from jmac_amd import losses
    gen = torch.Generator().manual_seed(N + T)
    ent = torch.randn(N, d, generator=gen)
    rel = torch.randn(nr, d, generator=gen)
    h = torch.randint(0, N, (T,), generator=gen)
    r = torch.randint(0, nr, (T,), generator=gen)
    tl = torch.randint(0, N, (T,), generator=gen)
    if T > 4:
        tl[:3] = h[:3]                                # h == t: the row receives +g s and -g s
    if period == 100 and T > 100:                     # the reference's batch layout: (h, r) repeat with the period ...
        h, r = h[:100].repeat(T // 100 + 1)[:T].clone(), r[:100].repeat(T // 100 + 1)[:T].clone()
        h[150], r[250] = (h[150] + 1) % N, (r[250] + 1) % nr      # ... except where the hint is wrong
    w = torch.randn(T, generator=gen)
    e64, r64 = ent.double().requires_grad_(True), rel.double().requires_grad_(True)
    ref = orc.triple_l1_score(e64, r64, h, r, tl)
    (ref * w.double()).sum().backward()
    eg, rg = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
    out = losses.triple_l1_score(eg, rg, h.cuda(), r.cuda(), tl.cuda(), period=period)
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, 1e-5, what="score")
    assert_close(eg.grad, e64.grad, 1e-5, what="d ent")     # atomics: order free, fp32 rounding only
    assert_close(rg.grad, r64.grad, 1e-5, what="d rel")


def test_triple_l1_sign_of_zero_and_strided_rows():
    from jmac_amd import losses
    ent = torch.zeros(4, 8)
    rel = torch.zeros(2, 8)
    ent[1, :4] = 1.0
    big = torch.randn(6, 24)                            # column slice: row stride 24, 16-byte aligned
    idx = torch.tensor([0, 1, 2])
    eg = ent.cuda().requires_grad_(True)
    rg = rel.cuda().requires_grad_(True)
    out = losses.triple_l1_score(eg, rg, torch.tensor([0, 1]).cuda(), torch.tensor([0, 1]).cuda(), torch.tensor([2, 3]).cuda())
    out.sum().backward()
    assert out.cpu().tolist() == [0.0, 4.0]
    assert eg.grad.cpu()[0].abs().sum().item() == 0.0   # sign(0) = 0, as torch.norm's backward
    assert eg.grad.cpu()[1].tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    sl = big.cuda()[:, 8:16]
    ref = orc.triple_l1_score(big[:, 8:16].double(), big[:2, 8:16].double(), idx, torch.tensor([0, 1, 0]), idx.flip(0))
    got = losses.triple_l1_score(sl, sl[:2], idx.cuda(), torch.tensor([0, 1, 0]).cuda(), idx.flip(0).cuda())
    assert_close(got, ref, 1e-6)


@pytest.mark.parametrize("N1,N2,d,L", [(50, 60, 300, 500), (300, 200, 256, 5000), (9, 9, 5, 17), (30, 30, 64, 1)])
def test_pair_cosine_distance_fwd_bwd(N1, N2, d, L):
    from jmac_amd import losses
    gen = torch.Generator().manual_seed(N1 + L)
    e1, e2 = torch.randn(N1, d, generator=gen), torch.randn(N2, d, generator=gen) * 3
    i1 = torch.randint(0, N1, (L,), generator=gen)
    i2 = torch.randint(0, N2, (L,), generator=gen)
    w = torch.randn(L, generator=gen)
    a64, b64 = e1.double().requires_grad_(True), e2.double().requires_grad_(True)
    ref = orc.pair_cosine_distance(a64, i1, b64, i2)
    (ref * w.double()).sum().backward()
    ag, bg = e1.cuda().requires_grad_(True), e2.cuda().requires_grad_(True)
    out = losses.pair_cosine_distance(ag, i1.cuda(), bg, i2.cuda())
    (out * w.cuda()).sum().backward()
    assert_close(out, ref, 1e-5, atol=2e-6, what="dist")
    assert_close(ag.grad, a64.grad, 2e-5, what="d e1")
    assert_close(bg.grad, b64.grad, 2e-5, what="d e2")


def test_pair_cosine_same_table_and_zero_row():
    from jmac_amd import losses
    gen = torch.Generator().manual_seed(3)
    e = torch.randn(20, 16, generator=gen)
    e[5] = 0                                             # F.normalize clamps the norm at 1e-12: distance 1, finite grads
    i1, i2 = torch.tensor([0, 5, 7, 7]), torch.tensor([1, 2, 7, 5])
    e64 = e.double().requires_grad_(True)
    ref = orc.pair_cosine_distance(e64, i1, e64, i2)
    ref.sum().backward()
    eg = e.cuda().requires_grad_(True)
    out = losses.pair_cosine_distance(eg, i1.cuda(), eg, i2.cuda())
    out.sum().backward()
    assert_close(out, ref, 1e-5, atol=2e-6)
    assert torch.isfinite(eg.grad).all()
    keep = torch.ones(20, dtype=torch.bool)
    keep[5] = False                                      # the zero row's gradient is 1/eps-scaled in both; compare the rest
    assert_close(eg.grad.cpu()[keep], e64.grad[keep], 2e-5, atol=1e-6)


def test_losses_on_reference_golden_model():
    """completion_loss / alignment_loss pieces on the reference's captured embeddings (model_small.npz)."""
    from jmac_amd import losses
    g = load_golden("model_small")
    comp, rel = t(g["comp1_l1"]), t(g["rel1_l1"])
    h, r = t(g["lp_h"]).long(), t(g["lp_r"]).long()
    tl = t(g["lp_t"]).long()
    ref = orc.triple_l1_score(comp.double(), rel.double(), h, r, tl)
    got = losses.triple_l1_score(comp.cuda(), rel.cuda(), h.cuda(), r.cuda(), tl.cuda())
    assert_close(got, ref, 1e-5)
    # the same numbers are the gold column of the reference's link-prediction distances, layer by layer summed
    d0 = orc.triple_l1_score(t(g["comp1_l0"]).double(), t(g["rel1_l0"]).double(), h, r, tl)
    lp = torch.from_numpy(g["lp_dist"])[torch.arange(len(h)), tl]
    assert_close(ref + d0, lp, 1e-5)


@pytest.mark.parametrize("N,d", [(300, 300), (1000, 256), (17, 7), (1, 1), (64, 513)])
def test_row_normalize_fwd_bwd(N, d):
    """F.normalize(x, 2, -1) (src/jmac_model.py:179,191,227-228) incl. a zero row (norm clamped at eps) and a strided input."""
    from jmac_amd import ops
    gen = torch.Generator().manual_seed(N + d)
    big = torch.randn(N, d + 4, generator=gen)
    x = big[:, 2:2 + d]                                        # row stride d + 4
    if N > 3:
        big[3] = 0
    g = torch.randn(N, d, generator=gen)
    x64 = x.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(x64, 2, -1)
    (ref * g.double()).sum().backward()
    xg = big.cuda()[:, 2:2 + d].detach().requires_grad_(True)
    out = ops.row_normalize(xg)
    (out * g.cuda()).sum().backward()
    assert_close(out, ref, 1e-6, atol=1e-7)
    keep = torch.ones(N, dtype=torch.bool)
    if N > 3:
        keep[3] = False                                         # zero row: gradient = g / eps in both (1e12-scaled)
        assert torch.isfinite(xg.grad).all()
        assert_close(xg.grad.cpu()[3] * 1e-12, x64.grad[3] * 1e-12, 1e-5)
    assert_close(xg.grad.cpu()[keep], x64.grad[keep], 1e-5, atol=1e-6)


@pytest.mark.parametrize("B,K", [(64, 5), (1000, 25), (7, 1)])
def test_margin_loss_matches_reference_expression(B, K):
    """losses.margin_loss == torch.max(pos - neg, -margin).mean() + margin with the reference's n-major consumption of the
    negative block (src/jmac_model.py:351-378), forward and backward; ties at diff == -margin get half the gradient."""
    from jmac_amd import losses
    gen = torch.Generator(device="cuda").manual_seed(B + K)
    score = (torch.randn(B + B * K, device="cuda", generator=gen) * 4).requires_grad_(True)
    margin = torch.nn.Parameter(torch.tensor([5.0], device="cuda"), requires_grad=False)
    with torch.no_grad():                                      # a clipped pair and an exact tie
        score[B] = score[0] + 9.0
        if K > 1:
            score[B + B + 1] = score[1] + 5.0
    got = losses.margin_loss(score, B, margin)
    assert got.shape == (1,)
    (got * 3.0).sum().backward()
    g_got, score.grad = score.grad.clone(), None
    s64 = score.detach().double().requires_grad_(True)
    pos = s64[:B].view(-1, B).permute(1, 0)
    neg = s64[B:].view(-1, B).permute(1, 0)
    ref = torch.max(pos - neg, -margin.double()).mean() + margin.double()
    (ref * 3.0).sum().backward()
    assert abs(float(got) - float(ref)) <= 1e-6 * abs(float(ref))
    assert torch.allclose(g_got.double(), s64.grad, rtol=1e-5, atol=1e-9)
    assert float(g_got[B]) == 0.0                              # clipped at -margin: no gradient
    # bitwise reproducible
    assert torch.equal(losses.margin_loss(score.detach(), B, margin), losses.margin_loss(score.detach(), B, margin))
    # ragged input: the reference's own expression
    rag = torch.randn(B + B * K + 3, device="cuda", generator=gen)
    if (3 % B) != 0:
        with pytest.raises(RuntimeError):
            losses.margin_loss(rag, B, margin)


@pytest.mark.parametrize("B,K,d", [(64, 5, 40), (250, 25, 300), (7, 1, 12)])
def test_triple_l1_margin_loss_is_the_two_ops_in_one_node(B, K, d):
    """losses.triple_l1_margin_loss == margin_loss(triple_l1_score(...)): same loss (bitwise: the same two forward kernels),
    same gradients for both tables (the score gradient is derived inside the L1 adjoint; float atomics: tolerance), on the
    reference's batch layout (sub.repeat(K+1), rel.repeat(K+1), cat(obj, negatives); train.py:347-352)."""
    from jmac_amd import losses
    gen = torch.Generator(device="cuda").manual_seed(B * K + d)
    n, nr = 5 * B, 17
    ent0 = torch.randn(n, d, device="cuda", generator=gen)
    rel0 = torch.randn(nr, d, device="cuda", generator=gen)
    bh = torch.randint(0, n, (B,), device="cuda", generator=gen)
    br = torch.randint(0, nr, (B,), device="cuda", generator=gen)
    h, r = bh.repeat(K + 1), br.repeat(K + 1)
    t = torch.randint(0, n, (B * (K + 1),), device="cuda", generator=gen)
    margin = torch.nn.Parameter(torch.tensor([float(d) * 0.2], device="cuda"), requires_grad=False)
    res = []
    for fused in (True, False):
        ent, rel = ent0.clone().requires_grad_(True), rel0.clone().requires_grad_(True)
        if fused:
            loss = losses.triple_l1_margin_loss(ent, rel, h, r, t, B, margin)
        else:
            loss = losses.margin_loss(losses.triple_l1_score(ent, rel, h, r, t, period=B), B, margin)
        (loss * 1.7).sum().backward()
        res.append((loss.detach(), ent.grad, rel.grad))
    assert torch.equal(res[0][0], res[1][0])
    scale = float(res[1][1].abs().max())
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6 * scale)
    # (17 relation rows take ~B / 17 float-atomic contributions each, in an order that differs from run to run)
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=2e-5 * float(res[1][2].abs().max()))
    assert float(res[0][1].abs().max()) > 0
    # a batch that is not B (K + 1) long takes the two ops (and the reference's own expression behind them)
    with pytest.raises(RuntimeError):
        losses.triple_l1_margin_loss(ent0, rel0, h[:-1], r[:-1], t[:-1], B, margin)


def _layer_loss_case(B, K, d, seed, n=None, nr=17, L=0):
    gen = torch.Generator().manual_seed(seed)
    n = n or 5 * B
    ent = torch.randn(n, d, generator=gen)
    rel = torch.randn(nr, d, generator=gen)
    bh = torch.randint(0, n, (B,), generator=gen)
    br = torch.randint(0, nr, (B,), generator=gen)
    h, r = bh.repeat(K + 1), br.repeat(K + 1)
    t = torch.randint(0, n, (B * (K + 1),), generator=gen)
    links = torch.randint(0, n // 2, (L, 2), generator=gen) if L else None
    return ent, rel, h, r, t, links


@pytest.mark.parametrize("B,K,d,L", [(64, 5, 40, 0), (250, 25, 300, 300), (1000, 25, 256, 2264), (7, 1, 12, 5)])
def test_completion_layer_loss_is_the_separate_ops_in_one_node(B, K, d, L):
    """losses.completion_layer_loss (round 5: one node per layer's term of completion_loss, src/jmac_model.py:331-380) against
    (a) the oracle's float64 restatement -- margin ranking loss of the L1 triple scores + alignment_loss_simple on the links + the
    running loss -- forward and both table gradients, and (b) the separate ops of this library composed with torch adds: the loss
    to rounding, the L1 term's gradients BITWISE (the same integers times the same factor), the sum with the cosine rows to the
    last bit or two.  The persistent count tables are zero again afterwards."""
    from jmac_amd import losses
    ent, rel, h, r, t, links = _layer_loss_case(B, K, d, B * K + d, L=L)
    n = ent.shape[0]
    margin = torch.nn.Parameter(torch.tensor([float(d) * 0.2]), requires_grad=False)
    prev = torch.tensor([0.37])
    w0, w1 = (0, n // 2), (n // 2, n - n // 2)                     # the two sides of a link: two windows of ONE stacked table
    # (a) the oracle, float64
    e64, r64 = ent.double().requires_grad_(True), rel.double().requires_grad_(True)
    score = orc.triple_l1_score(e64, r64, h, r, t)
    pos = score[:B].view(-1, B).permute(1, 0)
    neg = score[B:].view(-1, B).permute(1, 0)
    ref = torch.max(pos - neg, -margin.double()).mean() + margin.double() + prev.double()
    if L:
        ref = ref + orc.pair_cosine_distance(e64[w0[0]:w0[0] + w0[1]], links[:, 0], e64[w1[0]:w1[0] + w1[1]], links[:, 1]).mean()
    (ref * 1.7).sum().backward()
    dev = torch.device("cuda")
    hd, rd, td = h.to(dev), r.to(dev), t.to(dev)
    lk = (links[:, 0].contiguous().to(dev), links[:, 1].contiguous().to(dev), w0, w1) if L else None
    mg, pv = margin.to(dev), prev.to(dev)
    res = []
    for fused in (True, False):
        eg, rg = ent.to(dev).requires_grad_(True), rel.to(dev).requires_grad_(True)
        if fused:
            loss = losses.completion_layer_loss(eg, rg, hd, rd, td, B, mg, links=lk, add_to=pv)
        else:
            loss = losses.triple_l1_margin_loss(eg, rg, hd, rd, td, B, mg) + pv
            if L:
                loss = loss + losses.pair_cosine_distance(eg, lk[0], eg, lk[1], w0, w1).mean()
        (loss * 1.7).sum().backward()
        res.append((loss.detach(), eg.grad, rg.grad))
    assert tuple(res[0][0].shape) == (1,)
    assert abs(float(res[0][0]) - float(ref)) <= 1e-5 * abs(float(ref))
    assert abs(float(res[0][0]) - float(res[1][0])) <= 1e-6 * abs(float(ref))
    assert_close(res[0][1], e64.grad, 1e-5, 1e-9, "d ent vs oracle")
    assert_close(res[0][2], r64.grad, 1e-5, 1e-9, "d rel vs oracle")
    assert torch.equal(res[0][2], res[1][2])                           # d rel: the same integers times the same factor
    if L == 0:
        assert torch.equal(res[0][1], res[1][1])                       # d ent, L1 term alone: likewise bitwise
    else:                                                              # with the links: the mean's 1/L reaches the pairs through another
        scale = float(res[1][1].abs().max())                           # expression (torch's expand / div): last-bit differences only
        assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-7 * scale
    for c in losses._CNT.values():
        assert float(c[0].abs().max()) == 0.0 and float(c[1].abs().max()) == 0.0
    # bitwise reproducible
    eg, rg = ent.to(dev).requires_grad_(True), rel.to(dev).requires_grad_(True)
    (losses.completion_layer_loss(eg, rg, hd, rd, td, B, mg, links=lk, add_to=pv) * 1.7).sum().backward()
    assert torch.equal(eg.grad, res[0][1]) and torch.equal(rg.grad, res[0][2])


@pytest.mark.parametrize("N1,N2,d,L,same", [(50, 60, 300, 500, False), (300, 300, 256, 5000, True), (9, 9, 5, 17, True), (30, 40, 64, 1, False)])
def test_pair_cosine_mean_fwd_bwd(N1, N2, d, L, same):
    """losses.pair_cosine_mean == pair_cosine_distance(...).mean() (+ add_to) against the oracle in float64; gradients bitwise
    equal to the composed form's where that form is deterministic too; rows no pair touches come out as exact zeros (the rows
    kernel is the first writer of the whole table)."""
    from jmac_amd import losses
    gen = torch.Generator().manual_seed(N1 + L)
    e1 = torch.randn(N1, d, generator=gen)
    e2 = e1 if same else torch.randn(N2, d, generator=gen)
    i1 = torch.randint(0, N1 - 2, (L,), generator=gen)            # the last two rows of e1 are never touched
    i2 = torch.randint(0, (N1 if same else N2) - 2, (L,), generator=gen)
    prev = torch.tensor([1.25])
    a64 = e1.double().requires_grad_(True)
    b64 = a64 if same else e2.double().requires_grad_(True)
    ref = orc.pair_cosine_distance(a64, i1, b64, i2).mean() + prev.double()
    (ref * 0.3).sum().backward()
    dev = torch.device("cuda")
    ag = e1.to(dev).requires_grad_(True)
    bg = ag if same else e2.to(dev).requires_grad_(True)
    out = losses.pair_cosine_mean(ag, i1.to(dev), bg, i2.to(dev), add_to=prev.to(dev))
    (out * 0.3).sum().backward()
    assert tuple(out.shape) == (1,) and abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    assert_close(ag.grad, a64.grad, 1e-5, 1e-9, "d e1")
    assert float(ag.grad[-2:].abs().max()) == 0.0
    if not same:
        assert_close(bg.grad, b64.grad, 1e-5, 1e-9, "d e2")
    ag2 = e1.to(dev).requires_grad_(True)
    bg2 = ag2 if same else e2.to(dev).requires_grad_(True)
    (losses.pair_cosine_distance(ag2, i1.to(dev), bg2, i2.to(dev)).mean() * 0.3).sum().backward()
    assert_close(ag.grad, ag2.grad, 1e-6, 1e-9, "rows form (scalar gradient) vs the vector form")


def test_pair_cosine_with_an_index_pair_first_seen_inside_a_capture():
    """the incidence record of an index pair that a stream capture meets for the first time is sorted INSIDE the
    capture -- with no host read on the way (the row pointer used to come from torch.bincount, which reads its maximum back and
    aborted the capture).  The replayed graph's loss and gradient equal the eager ones on the same tensors."""
    from jmac_amd import losses
    gen = torch.Generator(device="cuda").manual_seed(21)
    e = (torch.randn(500, 300, device="cuda", generator=gen)).requires_grad_(True)
    i1 = torch.randint(0, 500, (777,), device="cuda", generator=gen)
    i2 = torch.randint(0, 500, (777,), device="cuda", generator=gen)
    static_loss = torch.zeros(1, device="cuda")
    static_grad = torch.zeros_like(e)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                          # warm-up on OTHER index tensors: the capture's pair stays unseen
        w1, w2 = i1.clone(), i2.clone()
        for _ in range(2):
            (g,) = torch.autograd.grad(losses.pair_cosine_mean(e, w1, e, w2), [e])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = losses.pair_cosine_mean(e, i1, e, i2)                    # (i1, i2): first seen here
        (g,) = torch.autograd.grad(loss, [e])
        static_loss.copy_(loss.reshape(1))
        static_grad.copy_(g)
    graph.replay()
    torch.cuda.synchronize()
    want = losses.pair_cosine_mean(e, i1, e, i2)
    (gw,) = torch.autograd.grad(want, [e])
    assert torch.equal(static_loss, want.reshape(1).detach())
    assert torch.equal(static_grad, gw)
    e64 = e.detach().double().cpu().requires_grad_(True)
    ref = orc.pair_cosine_distance(e64, i1.cpu(), e64, i2.cpu()).mean()
    ref.backward()
    assert_close(static_loss, ref.detach().reshape(1), 1e-5, what="loss")
    assert_close(static_grad, e64.grad, 1e-5, what="d e")
