"""GPU: the block-batched encoder (JMAC.forward_stacked) -- several KGs through the layer kernels as ONE launch set on the
block-diagonal union of their graphs, BatchNorm statistics per KG.

The reference's training step encodes two KGs per batch with the same layer weights, one forward_base call after the other
(src/jmac_model.py:325-326 completion_loss, :263-264 alignment_loss); each call normalises with ITS rows' batch statistics
and moves the running estimates once.  Held here to

  * the same model making the separate calls (every loss, every gradient, the BatchNorm buffers, num_batches_tracked), for
    adjacent / non-adjacent / reversed KG pairs, three blocks, both encoders (with and without name information);
  * the ORACLE making the two separate forward_name calls on the REAL DBP-5L el + ja KGs at d = 300 (bench.py's PairWorkload:
    the step the `pair` object of the bench line times): forward in fp32 and float64, every parameter gradient against
    float64 on the GPU's side of every kink (as tests/test_gpu_ja_oracle.py does for one KG), BatchNorm buffers, all at 1e-4.
"""
import argparse
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import assert_close, expand_rel_act, random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4


# ---- small synthetic model with three KGs ---------------------------------------------------------------------------------
SIZES, NREL, D, DI = (150, 90, 120), 7, 32, 12


def _small_model(no_name=False, seed=0):
    from jmac_amd.model import JMAC
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = types.SimpleNamespace(dim=D, dropout=0.0, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=4,
                                 margin_align=1.0, margin_completion=5.0, batch_size=16, no_name_info=no_name, device="cuda")
    n = sum(SIZES)
    m = JMAC(args, rng.standard_normal((n, DI)).astype(np.float32), NREL * len(SIZES), n).cuda()
    with torch.no_grad():                                   # non-trivial BatchNorm affine parameters and estimates
        for lay in (m.conv1_alignment, m.conv2_alignment, m.conv1_completion):
            lay.bn.weight.uniform_(0.5, 1.5)
            lay.bn.bias.uniform_(-0.2, 0.2)
            lay.bn.running_mean.uniform_(-0.1, 0.1)
            lay.bn.running_var.uniform_(0.5, 1.5)
    graphs, eb, rb = [], [], []
    e0 = r0 = 0
    for k, nk in enumerate(SIZES):
        ei, et = random_graph(rng, nk, NREL, 5 * nk, hub=40 if k == 0 else None)
        graphs.append((torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda()))
        eb.append([e0, e0 + nk])
        rb.append([r0, r0 + NREL])
        e0 += nk
        r0 += NREL
    m.train()
    return m, graphs, eb, rb, rng


def _batch(rng, nk, B, K):
    h, r, t = rng.integers(0, nk, B), rng.integers(0, NREL, B), rng.integers(0, nk, B)
    return {"batch_h": torch.from_numpy(np.tile(h, K + 1)).cuda(), "batch_r": torch.from_numpy(np.tile(r, K + 1)).cuda(),
            "batch_t": torch.from_numpy(np.concatenate([t, rng.integers(0, nk, B * K)])).cuda()}


def _run(m, state, fn):
    """loss, grads, buffers of one fn() from ``state``."""
    m.load_state_dict(state, strict=True)
    m.zero_grad(set_to_none=True)
    loss = fn()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    bufs = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    return loss.detach().clone(), grads, bufs


def _compare(a, b, what, rtol=2e-5):
    la, ga, ba = a
    lb, gb, bb = b
    assert abs(float(la) - float(lb)) <= rtol * abs(float(lb)), (what, float(la), float(lb))
    gscale = max(float(g.abs().max()) for g in gb.values() if g is not None)
    for k in gb:
        if gb[k] is None:
            assert ga[k] is None or float(ga[k].abs().max()) == 0.0, (what, k)
            continue
        assert ga[k] is not None, (what, k)
        assert_close(ga[k], gb[k], rtol, 1e-6 * gscale, "%s grad %s" % (what, k))
    for k in bb:
        if "num_batches" in k:
            assert int(ba[k]) == int(bb[k]), (what, k, int(ba[k]), int(bb[k]))
        else:
            assert_close(ba[k], bb[k], rtol, 1e-7, "%s buffer %s" % (what, k))


@pytest.mark.parametrize("no_name", [False, True], ids=["forward_name", "forward_no_name"])
@pytest.mark.parametrize("pair", [(0, 1), (1, 2), (0, 2), (2, 0), (1, 1)], ids=lambda p: "kg%d-kg%d" % p)
def test_pair_losses_match_separate_calls(pair, no_name):
    """completion_loss (source and target batch) and alignment_loss through the stacked launch set == through two
    forward_base calls: adjacent blocks (one slice of the tables), non-adjacent and reversed ones (stack order != call order:
    the running estimates must still move in CALL order), and a KG paired with itself."""
    m, graphs, eb, rb, rng = _small_model(no_name)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    k1, k2 = pair
    n1, n2 = SIZES[k1], SIZES[k2]
    B, K = 16, 4
    L = 20
    links = torch.from_numpy(np.stack([rng.integers(0, n1, L), rng.integers(0, n2, L)], 1)).cuda()
    feed = {"links": links, "ent_bases1": eb[k1], "rel_bases1": rb[k1], "ent_bases2": eb[k2], "rel_bases2": rb[k2],
            "neg_left": rng.integers(0, n1, L * 4).astype(np.float64), "neg_right": rng.integers(0, n2, L * 4).astype(np.float64),
            "neg2_left": rng.integers(0, n1, L * 4), "neg2_right": rng.integers(0, n2, L * 4)}
    (ei1, et1), (ei2, et2) = graphs[k1], graphs[k2]
    cases = [("completion source", lambda d=_batch(rng, n1, B, K): m.completion_loss(d, ei1, et1, ei2, et2, feed, True)),
             ("completion target", lambda d=_batch(rng, n2, B, K): m.completion_loss(d, ei1, et1, ei2, et2, feed, False))]
    if not no_name:
        cases.append(("alignment", lambda: m.alignment_loss(feed, ei1, et1, ei2, et2)))
    for what, fn in cases:
        m.batched_pairs = True
        assert m.forward_stacked([(ei1, et1, eb[k1], rb[k1]), (ei2, et2, eb[k2], rb[k2])]) is not None
        got = _run(m, state, fn)
        m.batched_pairs = False
        ref = _run(m, state, fn)
        _compare(got, ref, "%s %s" % (what, pair))
        assert int(got[2]["conv1_completion.bn.num_batches_tracked"]) == int(state["conv1_completion.bn.num_batches_tracked"]) + 2


def test_three_blocks_and_eval_mode():
    """forward_blocks on all three KGs in a scrambled call order: outputs, gradients of a loss over all blocks and the
    BatchNorm buffers equal the three separate calls (train mode); in eval mode (running estimates, no batch statistics)
    the stacked rows equal the separate calls as well, and get_emb_blocks == get_emb per KG."""
    m, graphs, eb, rb, rng = _small_model(False)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    order = [2, 0, 1]
    blocks = [(graphs[k][0], graphs[k][1], eb[k], rb[k]) for k in order]
    ws = [torch.randn(SIZES[k], D, device="cuda") for k in order]

    def total():
        outs = m.forward_blocks(blocks)
        return sum((o[0] * w).sum() + (o[1][1] * w).sum() * 0.5 + o[2][1].sum() * 0.01 for o, w in zip(outs, ws))
    m.batched_pairs = True
    got = _run(m, state, total)
    m.batched_pairs = False
    ref = _run(m, state, total)
    _compare(got, ref, "three blocks")
    assert int(got[2]["conv1_alignment.bn.num_batches_tracked"]) == int(state["conv1_alignment.bn.num_batches_tracked"]) + 3
    m.load_state_dict(state, strict=True)
    m.eval()
    with torch.no_grad():
        m.batched_pairs = True
        a = m.forward_blocks(blocks)
        ea = m.get_emb_blocks(blocks, pyt=True)
        m.batched_pairs = False
        b = m.forward_blocks(blocks)
        for (oa, ob), k in zip(zip(a, b), order):
            assert_close(oa[0], ob[0], 2e-5, 1e-7, "eval align_out kg%d" % k)
            assert_close(oa[1][1], ob[1][1], 2e-5, 1e-7, "eval c1 kg%d" % k)
            assert_close(oa[2][1], ob[2][1], 2e-5, 1e-7, "eval rel_c1 kg%d" % k)
        for (xa, xc), k in zip(ea, order):
            ra, rc = m.get_emb(graphs[k][0], graphs[k][1], eb[k], rb[k], pyt=True)
            assert_close(xa, ra, 2e-5, 1e-7, "get_emb align kg%d" % k)
            assert_close(xc, rc, 2e-5, 1e-7, "get_emb completion kg%d" % k)


def test_segmented_bn_kernels_against_torch():
    """jmac_bn_tanh_seg_{fwd2,bwd2}_f32 against torch's BatchNorm1d applied block by block (float64), call order != stack
    order; blocks of 1 row up to thousands; d = 300."""
    from jmac_amd import encoder
    torch.manual_seed(3)
    sizes, order, d = (700, 1, 2500, 37), (2, 0, 3, 1), 300
    n = sum(sizes)
    x = (torch.randn(n, d, device="cuda") * 0.3 + 0.1).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(d).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.1, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    wr, br = bn.weight.detach().cpu().double().requires_grad_(True), bn.bias.detach().cpu().double().requires_grad_(True)
    rm, rv = bn.running_mean.detach().cpu().double(), bn.running_var.detach().cpu().double()
    seg = encoder.RowBlocks(sizes, order)
    y, y2 = torch.empty(n, d, device="cuda"), torch.empty(n, 2 * d, device="cuda")
    mean, invstd, _ = encoder._bn_fwd(x.detach(), bn, True, y, y2[:, d:], seg)
    g1, g2 = torch.randn(n, d, device="cuda"), torch.randn(n, d, device="cuda")
    gx, gbw = encoder._bn_bwd(x.detach(), y, g1, g2, bn.weight, mean, invstd, True, seg)
    torch.cuda.synchronize()
    # reference: block after block in CALL order (the running estimates are sequential), blocks of one row by hand
    xr = x.detach().cpu().double().requires_grad_(True)
    outs = [None] * len(sizes)
    for pos in order:
        lo = seg.offsets[pos]
        xb = xr[lo:lo + sizes[pos]]
        if sizes[pos] == 1:                                   # torch refuses one row per channel in train mode; the formula does not
            mu, var = xb.mean(0), xb.var(0, unbiased=False)
            rm, rv = 0.9 * rm + 0.1 * mu.detach(), 0.9 * rv + 0.1 * var.detach()
            outs[pos] = torch.tanh((xb - mu) / torch.sqrt(var + bn.eps) * wr + br)
        else:
            rm, rv = rm.clone(), rv.clone()                   # this call's own buffers (updated in place by it, by nothing later)
            outs[pos] = torch.tanh(torch.nn.functional.batch_norm(xb, rm, rv, wr, br, True, 0.1, bn.eps))
    yr = torch.cat(outs, 0)
    (yr * (g1 + g2).cpu().double()).sum().backward()
    assert_close(y, yr, 2e-5, 1e-6, "seg bn forward")
    assert torch.equal(y, y2[:, d:])
    assert_close(bn.running_mean, rm, 2e-5, 1e-7, "running_mean")
    assert_close(bn.running_var, rv, 2e-5, 1e-7, "running_var")
    assert int(bn.num_batches_tracked) == len(sizes)
    assert_close(gx, xr.grad, 1e-4, 1e-6, "seg bn gx")
    assert_close(gbw[:d], br.grad, 1e-4, 1e-6, "grad bias")
    assert_close(gbw[d:], wr.grad, 1e-4, 1e-6, "grad weight")


# ---- the real el + ja pair at d = 300 against the oracle ---------------------------------------------------------------------
def _pair_workload(d=300):
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(dim=d, batch=1000, negatives=25, bwd_mode=1)
    w = bench.PairWorkload(a, torch.device("cuda"), seed=1234, batched=True)
    w.model.completion_dropout.p = 0.0
    return w


def _kg_masks(w, captured, k):
    """The oracle's kink masks of KG k cut out of the STACKED tables the node gathered (block k of the stack)."""
    n1, n2 = w.n
    nr, d = w.nr, w.d
    eo, ro = (0, 0) if k == 0 else (n1, nr)                   # el is block 0, ja block 1 (table order)
    nk = w.n[k]
    ei, et = (w.g1, w.g2)[k]
    masks = {}
    for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
        PQZ, RR = captured[name + ".tables"]
        h = PQZ[ei[0] + eo, :d] + (PQZ[ei[1] + eo, d:2 * d] - RR[et + ro, :d])
        masks[name] = (h > 0).cpu()
        # relation-transform activations: reported on the compact rows of the STACKED relation table (the rows the union graph's
        # edges name, then the loop row) -> all 2 nr + 1 rows -> this KG's nr rows + the loop row
        T = expand_rel_act(captured[name + ".rel_act"], captured.get("rel_used"), 2 * nr)
        masks[name + ".rel"] = torch.cat((T[ro:ro + nr], T[-1:]), 0)
    masks["rel_linear11"] = (captured["rel_linear11.act"][ro:ro + nr] > 0).cpu()
    masks["rel_linear11_uni"] = expand_rel_act(captured["rel_linear11_uni.act"], captured.get("rel_used"), 2 * nr, loop=False)[ro:ro + nr]
    return masks


def test_pair_step_matches_oracle_real_el_ja():
    w = _pair_workload(300)
    assert w.n == (5231, 11805) and w.E == (12822, 17979) and w.nr == 961
    m = w.model
    from jmac_amd import encoder
    captured = {}
    encoder.CAPTURE = captured
    try:
        w.opt.zero_grad(set_to_none=True)
        st = m.forward_stacked(w.blocks())
        assert st is not None
        loss = w.loss()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        encoder.CAPTURE = None
    # (forward_stacked above moved the BatchNorm buffers once more than the step: buffers are checked in a clean pass below)
    grads_gpu = {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in m.named_parameters()}
    masks = [_kg_masks(w, captured, 0), _kg_masks(w, captured, 1)]
    with torch.no_grad():
        (e0, nn2) = st.ent_win[1]
        (r0, _) = st.rel_win[1]
        h, r, t = w.data["batch_h"], w.data["batch_r"], w.data["batch_t"]
        l1_masks = [(((c[h + e0] + rl[r + r0]) - c[t + e0]) > 0).cpu() for c, rl in zip(st.comp, st.rel)]
    # forward: fp32 and float64 oracle, two separate forward_name calls
    for dt in (torch.float32, torch.float64):
        o_loss, outs, _, bn = w.oracle_pass(dt)
        assert abs(float(loss) - float(o_loss)) <= RTOL * abs(float(o_loss)), (dt, float(loss), float(o_loss))
        for k in range(2):
            al, comp, rel = st.block(k)
            assert_close(al, outs[k][0], RTOL, 1e-7, "align_out kg%d %s" % (k, dt))
            assert_close(comp[1], outs[k][1][1], RTOL, 1e-7, "completion layer 1 kg%d %s" % (k, dt))
            assert_close(rel[1], outs[k][2][1], RTOL, 1e-7, "rel layer 1 kg%d %s" % (k, dt))
    # backward: float64 oracle on the GPU's side of every kink
    o_loss, _, grads, _ = w.oracle_pass(torch.float64, kink_masks=masks, backward=True, l1_sign_masks=l1_masks)
    assert w.l1_flips <= 8, w.l1_flips
    assert abs(float(loss) - float(o_loss)) <= RTOL * abs(float(o_loss))
    gscale = max(float(g.abs().max()) for g in grads.values() if g is not None)
    checked = 0
    for name, ref in grads.items():
        got = grads_gpu.get(name)
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, name
            continue
        got = got if got is not None else torch.zeros_like(ref)
        atol = 1e-4 * gscale if name.endswith("loop_rel") else 1e-9
        assert_close(got, ref, RTOL, atol, "grad " + name)
        checked += 1
    assert checked >= 12, checked
    # BatchNorm buffers after ONE step from the initial state: both KGs' statistics, in call order
    dev = w.links.device
    m.load_state_dict({k: v.to(dev) for k, v in w.state_cpu.items()}, strict=True)
    with torch.no_grad():
        w.loss()
    _, _, _, bn = w.oracle_pass(torch.float64)
    for k, ref in bn.items():
        assert_close(m.state_dict()[k], ref, RTOL, 1e-7, "buffer " + k)
    assert int(m.conv1_completion.bn.num_batches_tracked) == 2
    # and the parity object bench.py puts on its line
    m.load_state_dict({k: v.to(dev) for k, v in w.state_cpu.items()}, strict=True)
    res = w.check_against_oracle()
    assert res["ok"], res


def test_pair_step_captures_and_replays():
    """The pair step as a hipGraph (what bench.py times): replays keep training (loss falls) and the captured step equals
    the eager one from the same state."""
    w = _pair_workload(300)
    sys.path.insert(0, ROOT)
    import bench
    g = bench.try_capture(w)
    dev = w.links.device
    w.model.load_state_dict({k: v.to(dev) for k, v in w.state_cpu.items()}, strict=True)
    w.model.completion_dropout.p = 0.0
    l0 = float(w.loss().detach())
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        l1 = float(w.loss())
    assert np.isfinite(l1) and l1 < l0, (l0, l1)


def test_stacked_active_rows_equal_full_products():
    """The class order per KG block (encoder.ACTIVE_ROWS_STACKED: built, measured slower than the full products on stacked KGs,
    off by default) on the real el + ja pair step: loss and every gradient equal the default form to rounding, the per-KG
    BatchNorm buffers included."""
    from jmac_amd import encoder
    w = _pair_workload(300)
    w.model.completion_dropout.p = 0.0
    dev = w.links.device
    res = {}
    for flag in (True, False):
        encoder.ACTIVE_ROWS_STACKED = flag
        try:
            w.model.load_state_dict({k: v.to(dev) for k, v in w.state_cpu.items()}, strict=True)
            w.opt.zero_grad(set_to_none=True)
            loss = w.loss()
            loss.backward()
            torch.cuda.synchronize()
            res[flag] = (float(loss), {k: p.grad.clone() for k, p in w.model.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in w.model.state_dict().items() if "running" in k})
        finally:
            encoder.ACTIVE_ROWS_STACKED = False
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
    assert set(res[True][1]) == set(res[False][1])
    gscale = max(float(g.abs().max()) for g in res[False][1].values())
    for k, ref in res[False][1].items():
        # loop_rel's gradient is mathematically zero under train-mode BN (rounding noise of either form): on the others' scale
        atol = 1e-5 * gscale if k.endswith("loop_rel") else 1e-9
        assert float((res[True][1][k] - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-30) + atol, k
    for k, ref in res[False][2].items():
        assert torch.allclose(res[True][2][k], ref, rtol=1e-5, atol=1e-7), k
