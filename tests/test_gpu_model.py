"""GPU parity: the JMAC encoder / scoring call sites (jmac_amd.model.JMAC) against the reference's golden
vectors: forward_name, get_emb, forward_linkpred, completion_loss, alignment_loss (+ selected grads)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import assert_close, load_golden, make_args, t


def _model(g, with_bn_before=False):
    from jmac_amd.model import JMAC
    args = make_args(dim=int(g["d"]), dropout=0.0, num_gcn_layer=2, num_negative=5, margin_align=1.0,
                     margin_completion=5.0, batch_size=40, no_name_info=False, device="cuda")
    m = JMAC(args, g["name_emb"], 2 * int(g["nrel"]), int(g["n1"]) + int(g["n2"]))
    sd = {k[len("state."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state.")}
    if with_bn_before:
        for k in list(sd):
            if ("bn_before." + k) in g:
                sd[k] = torch.from_numpy(g["bn_before." + k])
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def _graphs(g):
    n1, n2, nrel = int(g["n1"]), int(g["n2"]), int(g["nrel"])
    return (t(g["e1_index"], "cuda"), t(g["e1_type"], "cuda"), [0, n1], [0, nrel],
            t(g["e2_index"], "cuda"), t(g["e2_type"], "cuda"), [n1, n1 + n2], [nrel, 2 * nrel])


def test_encoder_eval_and_linkpred():
    g = load_golden("model_small")
    m = _model(g).eval()
    e1i, e1t, eb1, rb1, e2i, e2t, eb2, rb2 = _graphs(g)
    with torch.no_grad():
        a1, c1, r1 = m.forward_base(e1i, e1t, eb1, rb1)
        a2, c2, _ = m.forward_base(e2i, e2t, eb2, rb2)
        assert_close(a1, g["align1"], 1e-4) and assert_close(a2, g["align2"], 1e-4)
        assert_close(c1[1], g["comp1_l1"], 1e-4) and assert_close(c2[1], g["comp2_l1"], 1e-4)
        assert_close(r1[1], g["rel1_l1"], 1e-4)
        ea, ec = m.get_emb(e1i, e1t, eb1, rb1, pyt=True)
        assert_close(ea, g["emb1_align"], 1e-4) and assert_close(ec, g["emb1_comp"], 1e-4)
        n1 = int(g["n1"])
        dist = m.forward_linkpred(g["lp_h"].tolist(), g["lp_r"].tolist(), e1i, e1t, list(range(n1)), eb1, rb1)
        assert_close(dist, g["lp_dist"], 1e-4)


def test_losses_and_grads():
    g = load_golden("model_small")
    e1i, e1t, eb1, rb1, e2i, e2t, eb2, rb2 = _graphs(g)
    links = g["links"]
    k = g["neg_right"].shape[0] // len(links)
    feed = {"neg_left": np.repeat(links[:, 0], k).astype(np.float64), "neg_right": t(g["neg_right"]),
            "neg2_left": t(g["neg2_left"]), "neg2_right": np.repeat(links[:, 1], k).astype(np.float64),
            "links": links, "ent_bases1": eb1, "ent_bases2": eb2, "rel_bases1": rb1, "rel_bases2": rb2}
    data = {"batch_h": t(g["batch_h"], "cuda"), "batch_r": t(g["batch_r"], "cuda"), "batch_t": t(g["batch_t"], "cuda")}
    m = _model(g, with_bn_before=True).train()
    loss = m.completion_loss(data, e1i, e1t, e2i, e2t, feed, True)
    assert abs(loss.item() - float(g["completion_loss"])) < 1e-4 * abs(float(g["completion_loss"]))
    loss.backward()
    assert_close(m.ent_init_att_completion.grad, g["closs_grad_ent"], 1e-4, 1e-7)
    assert_close(m.conv1_completion.w_att.grad, g["closs_grad_w_att"], 1e-4, 1e-7)
    assert_close(m.conv1_completion.a_att.grad, g["closs_grad_a_att"], 1e-4, 1e-7)
    assert_close(m.rel_init_att_completion.grad, g["closs_grad_rel"], 1e-4, 1e-7)

    m = _model(g, with_bn_before=True).train()
    al = m.alignment_loss(feed, e1i, e1t, e2i, e2t)
    assert abs(al.item() - float(g["alignment_loss"])) < 1e-4 * abs(float(g["alignment_loss"]))
    al.backward()
    assert_close(m.name_linear.grad, g["aloss_grad_name_linear"], 1e-4, 1e-7)
    assert_close(m.conv2_alignment.gcn_weight.grad, g["aloss_grad_conv2_gcn"], 1e-4, 1e-7)
    assert_close(m.ent_init_att_completion.grad, g["aloss_grad_ent"], 1e-4, 1e-7)


# ---- DBPv1 variant (row a17): JMAC_DBPv1/models/jmac_model.py JMAC_MODEL ---------------------------------------
def _dbpv1(g, with_bn_before=False):
    import types
    from jmac_amd.model_dbpv1 import JMAC_MODEL
    args = types.SimpleNamespace(emb_dim=int(g["d"]), completion_dropout_rate=0.0, leaky_relu_w=0.05, opn="sub",
                                 num_gcn_layer=2, num_negative=4, margin_align=1.0, margin_completion=5.0,
                                 completion_batch_size=30)
    m = JMAC_MODEL(int(g["n"]), int(g["nrel"]), g["ent_info_att"], args)
    sd = {k[len("state."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state.")}
    if with_bn_before:
        for k in list(sd):
            if ("bn_before." + k) in g:
                sd[k] = torch.from_numpy(g["bn_before." + k])
    m.load_state_dict(sd, strict=True)                                   # reference checkpoint loads as is
    return m.cuda()


def test_dbpv1_model_matches_reference_golden():
    g = load_golden("model_dbpv1")
    ei, et = t(g["edge_index"], "cuda"), t(g["edge_type"], "cuda")
    m = _dbpv1(g).eval()
    with torch.no_grad():
        a, comp, rel = m.forward_base(ei, et)
        assert_close(a, g["align"], 1e-4) and assert_close(comp[1], g["comp_l1"], 1e-4) and assert_close(rel[1], g["rel_l1"], 1e-4)
        ea, ec = m.get_emb(ei, et, pyt=True)
        assert_close(ea, g["emb_align"], 1e-4) and assert_close(ec, g["emb_comp"], 1e-4)
    feed = {"links": g["links"], "neg_left": g["neg_left"], "neg_right": t(g["neg_right"]), "neg2_left": t(g["neg2_left"]),
            "neg2_right": g["neg2_right"]}
    data = {"batch_h": t(g["batch_h"], "cuda"), "batch_r": t(g["batch_r"], "cuda"), "batch_t": t(g["batch_t"], "cuda")}
    m = _dbpv1(g, with_bn_before=True).train()
    loss = m.completion_loss(data, ei, et, feed)
    assert abs(loss.item() - float(g["completion_loss"])) < 1e-4 * abs(float(g["completion_loss"]))
    loss.backward()
    assert_close(m.ent_completion_att.grad, g["closs_grad_ent"], 1e-4, 1e-7)
    assert_close(m.rel_completion_att.grad, g["closs_grad_rel"], 1e-4, 1e-7)
    assert_close(m.conv1_completion.w_att.grad, g["closs_grad_w_att"], 1e-4, 1e-7)
    m = _dbpv1(g, with_bn_before=True).train()
    al = m.alignment_loss(feed, ei, et)
    assert abs(al.item() - float(g["alignment_loss"])) < 1e-4 * abs(float(g["alignment_loss"]))
    al.backward()
    assert_close(m.ent_completion_att.grad, g["aloss_grad_ent"], 1e-4, 1e-7)
    assert_close(m.all_linear_comp.grad, g["aloss_grad_all_linear"], 1e-4, 1e-7)
    assert_close(m.rel_info_att.grad, g["aloss_grad_rel_info"], 1e-4, 1e-7)


def test_forward_no_name_pred_head_and_subset_candidates():
    """The branches train.py's defaults do not take: --no_name_info (src/jmac_model.py:207-220), pred_head=True
    (:308-309) and an all_index that is a proper subset (:304-305), against the oracle on the golden model's weights."""
    import types
    import oracle.jmac_oracle as orc
    from jmac_amd.model import JMAC
    g = load_golden("model_small")
    args = make_args(dim=int(g["d"]), dropout=0.0, num_gcn_layer=2, num_negative=5, margin_align=1.0,
                     margin_completion=5.0, batch_size=40, no_name_info=True, device="cuda")
    m = JMAC(args, g["name_emb"], 2 * int(g["nrel"]), int(g["n1"]) + int(g["n2"]))
    sd = {k[len("state."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state.")}
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    e1i, e1t, eb1, rb1 = _graphs(g)[:4]
    p = {k: v.double() for k, v in sd.items()}
    bn = {k: v.clone() for k, v in p.items() if "running" in k}
    ref_out, ref_comp, ref_rel = orc.forward_no_name(p, e1i.cpu(), e1t.cpu(), eb1, rb1, 2, 0.05, "sub", False, bn)
    with torch.no_grad():
        out, comp, rel = m.forward_base(e1i, e1t, eb1, rb1)
        assert m.forward_base.__func__ is JMAC.forward_no_name
        assert_close(out, ref_out, 1e-4) and assert_close(comp[1], ref_comp[1], 1e-4) and assert_close(rel[1], ref_rel[1], 1e-4)
        h, r = g["lp_h"].tolist(), g["lp_r"].tolist()
        n1 = int(g["n1"])
        d_head = m.forward_linkpred(h, r, e1i, e1t, list(range(n1)), eb1, rb1, pred_head=True)
        assert_close(d_head, orc.linkpred_dist(ref_comp, ref_rel, h, r, pred_head=True), 1e-4)
        sub = list(range(5, n1, 3))                                   # candidates = a subset of the entities
        hs = [x % len(sub) for x in h]                                # e_index addresses the SUBSET table (:304-306)
        d_sub = m.forward_linkpred(hs, r, e1i, e1t, sub, eb1, rb1)
        want = 0
        for ent, rl in zip(ref_comp, ref_rel):
            ent_s = ent[torch.tensor(sub)]
            want = want + torch.cdist(ent_s[torch.tensor(hs)] + rl[torch.tensor(r)], ent_s, p=1)
        assert_close(d_sub, want, 1e-4)
