"""GPU, the workload that carries the headline number: bench.py's own DBP-5L-ja-shaped step (N=11 805, E=17 979 and the
35 958-edge bidirectional graph; d=300 = BASELINE configs[1] and d=256 = the reference's default / configs[0]) --
one full ``forward_base`` (three RelationAwareLayer calls) + losses + backward through the HIP path, against the oracle
on the same seeded inputs: forward in fp32 and float64, every parameter gradient in float64, all at 1e-4.

Gradients are compared on the SAME side of every LeakyReLU kink: the attention pre-activations h_e (E*d*3 ~ 3*10^7
values) include a few of magnitude ~1e-7 that fp32 and float64 round to different signs; the forward is continuous
there, the derivative is not.  The float64 oracle is therefore given the sign pattern of the GPU's own fp32 tables
(oracle ``kink_mask``); the number of such elements is asserted to be tiny, and nothing else is relaxed.  The same holds
for the five LeakyReLUs of the relation side (the layers' relation transforms and the two relation MLPs: 5 x 962 x d
pre-activations; none has flipped on any of the eight cases so far) and for the terms of the L1 triple score (one flipped
term moved d loss / d ent_init_att_completion by 2e-3 of its scale on one case): their sign patterns are handed over as
well, and counted."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import assert_close, expand_rel_act, rel_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4


def _workload(d, bidir, data="synthetic"):
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(dim=d, batch=1000, negatives=25, bwd_mode=1)
    w = bench.JaWorkload(a, torch.device("cuda"), seed=1234, bidirectional=bidir, data=data)
    w.model.completion_dropout.p = 0.0          # parity is checked with dropout off; BN stays in train mode
    return w


def _gpu_kink_masks(w, captured):
    """Per layer: sign pattern of h_e = P[i] + (Q[j] - Rq[t]) computed from the SAME fp32 tables the kernel gathers."""
    masks, ei, et = {}, w.ei, w.et
    with torch.no_grad():
        for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
            x, r = captured[name]
            lay = getattr(w.model, name)
            d = dp = lay.out_channels
            if name + ".tables" in captured:         # fused node: the tables it gathered (a recomputation from a contiguous
                PQZ, RR = captured[name + ".tables"]  # copy of its strided operand may round differently)
            else:
                PQZ, RR, _, dp = lay._tables(x, lay.transform_relations(r))
            h = PQZ[ei[0], :d] + (PQZ[ei[1], dp:dp + d] - RR[et, :d])
            masks[name] = (h > 0).cpu()
            # the relation side's own LeakyReLUs: between the two relation transforms of the layer (src/jmac_model.py:41) ...
            # (the fused node reports the activation it computed: LeakyReLU / ReLU keep the sign of their argument)
            if name + ".rel_act" in captured:                   # compact relation rows (the rows the edges name) -> all rows
                masks[name + ".rel"] = expand_rel_act(captured[name + ".rel_act"], captured.get("rel_used"), r.shape[0])
            else:
                masks[name + ".rel"] = (torch.mm(torch.cat([r, lay.loop_rel], 0), lay.rel_transform_weight1) > 0).cpu()
        m = w.model                               # ... and inside the two relation MLPs (src/jmac_model.py:195-196)
        for key, table, weight in (("rel_linear11", m.rel_init_att_completion, m.rel_linear11),
                                   ("rel_linear11_uni", m.rel_init_att_alignment, m.rel_linear11_uni)):
            if key + ".act" in captured:               # rel_linear11_uni feeds conv2's chain only: reported on the compact rows
                used = captured.get("rel_used") if key == "rel_linear11_uni" else None
                masks[key] = expand_rel_act(captured[key + ".act"], used, table.shape[0], loop=False)
            else:
                masks[key] = (torch.mm(table, weight) > 0).cpu()
    return masks


@pytest.mark.parametrize("data", ["real", "synthetic"])
@pytest.mark.parametrize("bidir", [False, True], ids=["train-graph", "bidirectional"])
@pytest.mark.parametrize("d", [300, 256])
def test_bench_workload_step_matches_oracle(d, bidir, data):
    """``data='real'``: the REAL DBP-5L ja KG from tests/golden/dbp5l_ja_el_data.npz -- the train-mode graph bench.py times
    (17 979 edges, train.py:130-132) and the loader's bidirectional form (35 958 edges, src/utils.py:127-149) with its
    1 221-edge hub row and 4 332 isolated entities -- at the BASELINE dims; ``'synthetic'``: the seeded graph of the same shape."""
    w = _workload(d, bidir, data)
    assert w.N == 11805 and w.E == (35958 if bidir else 17979)
    if data == "real":
        deg = np.bincount(w.ei[0].cpu().numpy(), minlength=w.N)
        assert (int(deg.max()), int((deg == 0).sum())) == ((1221, 4332) if bidir else (28, 6380))
    captured, hooks = {}, []
    from jmac_amd import encoder
    fused = w.model._fused(300)
    assert fused                                                       # the bench step runs the fused encoder node
    if not fused:                       # op-by-op path: forward pre-hooks show each layer call's inputs (a layer WITH hooks
        for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):   # is never taken by the node: it would skip them)
            def pre(mod, args, name=name):
                captured[name] = (args[0].detach(), args[1].detach())
            hooks.append(getattr(w.model, name).register_forward_pre_hook(pre))
    encoder.CAPTURE = captured if fused else None                      # the fused node calls no layer module: it reports
    try:                                                                # the same (ent_emb, rel_emb) pairs itself
        w.opt.zero_grad(set_to_none=True)
        loss, align_out, comp, rel = w.forward_loss()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        encoder.CAPTURE = None
    for h in hooks:
        h.remove()
    assert {"conv1_alignment", "conv1_completion", "conv2_alignment"} <= set(captured)
    assert "conv1_alignment.tables" in captured                        # ... and it was the node that ran
    masks = _gpu_kink_masks(w, captured)

    # forward: plain oracle, fp32 and float64 (no mask involved: the forward is continuous at the kink)
    for dt in (torch.float32, torch.float64):
        o_loss, o_align, o_comp, _ = w.oracle_pass(dt)
        assert abs(float(loss) - float(o_loss)) <= RTOL * abs(float(o_loss)), (dt, float(loss), float(o_loss))
        assert_close(align_out, o_align, RTOL, 1e-7, "align_out %s" % dt)
        assert_close(comp[1], o_comp[1], RTOL, 1e-7, "completion layer 1 %s" % dt)

    # how many pre-activations sit on the other side of the kink in float64 (own signs)?  a handful out of E*d*3
    import oracle.jmac_oracle as orc
    flips = 0
    st64 = {k: v.double() if v.dtype.is_floating_point else v for k, v in w.state_cpu.items()}
    for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
        x, r = captured[name]
        p = orc._sub(st64, name)
        x64, r64 = x.cpu().double(), r.cpu().double()
        rel64 = orc.transform_relations(p, r64, 0.05, "leaky_relu")
        wt, wb = p["w_att"][:d], p["w_att"][d:]
        ei, et = w.ei.cpu(), w.et.cpu()
        h64 = (x64 @ wt)[ei[0]] + (x64 @ wb)[ei[1]] - (rel64 @ wb)[et]
        flips += int(((h64 > 0) != masks[name]).sum())
    assert flips <= 64, flips                                        # ~1e-6 of the 3 * E * d pre-activations
    # the same count for the relation side's LeakyReLUs (5 x 962 x d pre-activations)
    rflips = 0
    for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
        x, r = captured[name]
        p = orc._sub(st64, name)
        pre = torch.cat([r.cpu().double(), p["loop_rel"]], 0) @ p["rel_transform_weight1"]
        rows = rel_rows(captured.get("rel_used"), r.shape[0])                # counted where the node evaluated the activation
        rflips += int(((pre > 0) != masks[name + ".rel"])[rows].sum())
    rflips += int(((st64["rel_init_att_completion"] @ st64["rel_linear11"] > 0) != masks["rel_linear11"]).sum())
    rows = rel_rows(captured.get("rel_used"), w.nr, loop=False)
    rflips += int(((st64["rel_init_att_alignment"] @ st64["rel_linear11_uni"] > 0) != masks["rel_linear11_uni"])[rows].sum())
    assert rflips <= 8, rflips

    # the L1 triple score |(h + r) - t| (src/jmac_model.py:345-350) has the same kind of kink at 0 in each of its 26 000 x d
    # terms per layer: an element of magnitude ~1e-8 whose sign fp32 and float64 disagree on moves d loss / d score by 2/B
    # on that coordinate.  The GPU's own sign pattern (same fp32 operation order as the kernel) goes to the oracle, counted.
    with torch.no_grad():
        l1_masks = [(((c[w.h] + r_[w.r]) - c[w.t]) > 0).cpu() for c, r_ in zip(comp, rel)]
    # backward: float64 oracle on the GPU's side of every kink, every parameter, 1e-4
    o_loss, _, _, grads = w.oracle_pass(torch.float64, kink_masks=masks, backward=True, l1_sign_masks=l1_masks)
    assert w.l1_flips <= 8, w.l1_flips                               # of 2 x 26 000 x d terms
    assert abs(float(loss) - float(o_loss)) <= RTOL * abs(float(o_loss))
    gscale = max(float(g.abs().max()) for g in grads.values() if g is not None)
    checked = 0
    for name, prm in w.model.named_parameters():
        ref = grads.get(name)
        if ref is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0 or not prm.requires_grad, name
            continue
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        # loop_rel's gradient is mathematically zero under train-mode BN (a constant row shift cancels in the batch
        # mean): compare it on the scale of the other relation-side gradients
        atol = 1e-4 * gscale if name.endswith("loop_rel") else 1e-9
        assert_close(got, ref, RTOL, atol, "grad " + name)
        checked += 1
    assert checked >= 25, checked
    print("ja %s d=%d bidir=%s: loss %.6f, kink flips fp32-vs-f64 %d of %d (attention), %d of %d (relation side)"
          % (data, d, bidir, float(loss), flips, 3 * w.E * d, rflips, 5 * (w.nr + 1) * d),
          "; L1 score terms on the other side of 0: %d" % w.l1_flips)
