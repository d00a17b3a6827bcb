"""CPU: the oracle (oracle/jmac_oracle.py) against the golden vectors captured from the reference
(tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import oracle.jmac_oracle as orc
from util import LAYER_CASES, assert_close, layer_grads, layer_params, load_golden, rel_err, t


@pytest.fixture(autouse=True)
def _one_thread():
    """The fixtures are generated on one thread (tests/golden/gen_golden.py); the oracle restates the same arithmetic, so on
    one thread the comparison does not depend on how the host splits its float sums."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("case", LAYER_CASES + ["layer_dbpv1"])
def test_layer_forward_backward_matches_reference(case):
    g = load_golden(case)
    p = {k: v.clone().requires_grad_(True) for k, v in layer_params(g).items()}
    X = t(g["X"]).requires_grad_(True)
    R = t(g["R"]).requires_grad_(True)
    ei, et = t(g["edge_index"]), t(g["edge_type"])
    rel_act = "relu" if case == "layer_dbpv1" else "leaky_relu"
    d = int(g["d"])
    rm, rv = torch.zeros(d), torch.ones(d)
    out = orc.layer_forward(p, X, R, ei, et, float(g["slope"]), "sub", rel_act, True, rm, rv)
    # measured on one thread in the build container: every quantity below is BIT-IDENTICAL to the fixture (error 0.0) on all
    # eight cases; 1e-6 / 1e-5 is head-room for a host whose BLAS splits its sums differently, not a measured error
    assert rel_err(out, g["out_train"]) < 1e-6
    assert rel_err(rm, g["running_mean_after"]) < 1e-6 and rel_err(rv, g["running_var_after"]) < 1e-6
    (out * t(g["G"])).sum().backward()
    assert rel_err(X.grad, g["grad_X"]) < 1e-5
    assert rel_err(R.grad, g["grad_R"]) < 1e-5
    for name, ref in layer_grads(g).items():
        got = p[name].grad if p[name].grad is not None else torch.zeros_like(p[name])
        # loop_rel's gradient is mathematically zero here: judge its noise against grad_R's scale
        atol = 1e-4 * float(np.abs(g["grad_R"]).max()) + 1e-6 if name == "loop_rel" else 1e-6
        assert_close(got, ref, 1e-5, atol, name)
    # eval mode uses the running statistics left by the training step
    out_eval = orc.layer_forward({k: v.detach() for k, v in p.items()}, X.detach(), R.detach(), ei, et,
                                 float(g["slope"]), "sub", rel_act, False, rm, rv)
    assert rel_err(out_eval, g["out_eval"]) < 1e-6
    nb, _, _ = orc.layer_pre_bn({k: v.detach() for k, v in p.items()}, X.detach(), R.detach(), ei, et,
                                float(g["slope"]), "sub", rel_act)
    assert rel_err(nb, g["nb"]) < 1e-6 or np.abs(g["nb"]).max() == 0


@pytest.mark.parametrize("case", LAYER_CASES + ["layer_dbpv1"])
def test_factorised_identity(case):
    """The algebra the HIP kernels rely on equals the reference formulation."""
    g = load_golden(case)
    p = layer_params(g)
    rel_act = "relu" if case == "layer_dbpv1" else "leaky_relu"
    args = (p, t(g["X"]), t(g["R"]), t(g["edge_index"]), t(g["edge_type"]), float(g["slope"]))
    _, _, pre = orc.layer_pre_bn(*args, "sub", rel_act)
    fac = orc.factorised_pre_bn(*args, rel_act)
    assert rel_err(fac, pre) < 2e-5


def _model_state(g):
    return {k[len("state."):]: t(v) for k, v in g.items() if k.startswith("state.")}


def test_model_encoder_and_linkpred():
    g = load_golden("model_small")
    st = _model_state(g)
    n1, n2, nrel = int(g["n1"]), int(g["n2"]), int(g["nrel"])
    bn = {k: v.clone() for k, v in st.items() if "running" in k}
    name = t(g["name_emb"])
    a1, c1, r1 = orc.forward_name(st, name, t(g["e1_index"]), t(g["e1_type"]), [0, n1], [0, nrel], bn_state=bn)
    a2, c2, _ = orc.forward_name(st, name, t(g["e2_index"]), t(g["e2_type"]), [n1, n1 + n2], [nrel, 2 * nrel], bn_state=bn)
    assert rel_err(a1, g["align1"]) < 1e-5 and rel_err(a2, g["align2"]) < 1e-5
    assert rel_err(c1[1], g["comp1_l1"]) < 1e-5 and rel_err(c2[1], g["comp2_l1"]) < 1e-5
    assert rel_err(r1[1], g["rel1_l1"]) < 1e-5
    ea, ec = orc.get_emb(a1, c1)
    assert rel_err(ea, g["emb1_align"]) < 1e-5 and rel_err(ec, g["emb1_comp"]) < 1e-5
    dist = orc.linkpred_dist(c1, r1, g["lp_h"].tolist(), g["lp_r"].tolist())
    assert rel_err(dist, g["lp_dist"]) < 1e-5
    # plain-definition L1 equals cdist
    er = c1[1][t(g["lp_h"])] + r1[1][t(g["lp_r"])]
    assert rel_err(orc.l1_scores(er, c1[1]), torch.cdist(er, c1[1], p=1)) < 1e-6


def test_filtered_ranks_and_metrics():
    g = load_golden("model_small")
    dist = t(g["lp_dist"])
    gold = g["lp_t"].tolist()
    for filt in (0, 1):
        ranks = orc.filtered_ranks(dist, gold, g["filt_ptr"] if filt else None, g["filt_idx"] if filt else None)
        assert (ranks == g["ranks_filt%d" % filt]).all()
        assert np.allclose(orc.ranking_metrics(ranks), g["eval_filt%d" % filt], atol=1e-12)
    assert (g["ranks_filt1"] <= g["ranks_filt0"]).all()


def test_get_neg_and_alignment_quality():
    g = load_golden("model_small")
    e1, e2 = t(g["emb1_align"]), t(g["emb2_align"])
    links = g["links"]
    k = g["neg_right"].shape[0] // len(links)
    assert (orc.get_neg(links[:, 0].tolist(), e1, e2, k).numpy() == g["neg_right"]).all()
    assert (orc.get_neg(links[:, 1].tolist(), e2, e1, k).numpy() == g["neg2_left"]).all()
    sim = e1[t(links[:, 0])] @ e2.t()
    assert (orc.topk_lowest_index(sim, k).reshape(-1).numpy() == g["neg_right"]).all()
    ent, sm1, sm2 = orc.alignment_quality(e1, e2, g["aq_list1"].tolist(), g["aq_list2"].tolist())
    assert abs(ent.item() - float(g["aq_entropy"])) < 1e-5
    assert rel_err(sm1, g["aq_softmax_rows"]) < 1e-5 and rel_err(sm2, g["aq_softmax_cols"]) < 1e-5


def test_losses():
    g = load_golden("model_small")
    st = _model_state(g)
    for k in list(st):
        if ("bn_before." + k) in g:
            st[k] = t(g["bn_before." + k])
    n1, n2, nrel = int(g["n1"]), int(g["n2"]), int(g["nrel"])
    name = t(g["name_emb"])
    leaf = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in st.items()}

    def enc(bn):
        o1 = orc.forward_name(leaf, name, t(g["e1_index"]), t(g["e1_type"]), [0, n1], [0, nrel], training=True, bn_state=bn)
        o2 = orc.forward_name(leaf, name, t(g["e2_index"]), t(g["e2_type"]), [n1, n1 + n2], [nrel, 2 * nrel], training=True, bn_state=bn)
        return o1, o2

    bn = {k: v.clone() for k, v in st.items() if "running" in k}
    (a1, c1, r1), (a2, c2, r2) = enc(bn)
    B = int(g["batch_h"].shape[0] // (g["neg_right"].shape[0] // len(g["links"]) + 1))
    loss = orc.completion_loss(c1, r1, c2, r2, t(g["batch_h"]), t(g["batch_r"]), t(g["batch_t"]), g["links"], B,
                               float(st["margin_completion"]), True)
    assert abs(loss.item() - float(g["completion_loss"])) < 1e-4 * abs(float(g["completion_loss"]))
    loss.backward()
    assert rel_err(leaf["ent_init_att_completion"].grad, g["closs_grad_ent"]) < 1e-4
    assert rel_err(leaf["conv1_completion.w_att"].grad, g["closs_grad_w_att"]) < 1e-4
    assert rel_err(leaf["conv1_completion.a_att"].grad, g["closs_grad_a_att"]) < 1e-4
    assert rel_err(leaf["rel_init_att_completion"].grad, g["closs_grad_rel"]) < 1e-4

    for v in leaf.values():
        v.grad = None
    bn = {k: t(g["bn_before." + k]) for k in st if ("bn_before." + k) in g}
    (a1, _, _), (a2, _, _) = enc(bn)
    links = g["links"]
    k = g["neg_right"].shape[0] // len(links)
    pos = np.repeat(links[:, 0], k)
    pos2 = np.repeat(links[:, 1], k)
    al = orc.alignment_loss(a1, a2, links, pos, g["neg_right"], g["neg2_left"], pos2, k, 1.0)
    assert abs(al.item() - float(g["alignment_loss"])) < 1e-4 * abs(float(g["alignment_loss"]))
    al.backward()
    assert rel_err(leaf["name_linear"].grad, g["aloss_grad_name_linear"]) < 1e-4
    assert rel_err(leaf["conv2_alignment.gcn_weight"].grad, g["aloss_grad_conv2_gcn"]) < 1e-4


def test_alignment_eval_matches_reference():
    g = load_golden("align_eval")
    e1, e2 = t(g["e1"]), t(g["e2"])
    for k in (0, 10):
        top_k, hits, mr, mrr, s = orc.alignment_test(e1, e2, (1, 5, 10), k)
        assert np.allclose(hits, g["hits_csls%d" % k], atol=1e-9)
        assert abs(mr - float(g["mr_csls%d" % k])) < 1e-9 and abs(mrr - float(g["mrr_csls%d" % k])) < 1e-9
        # the reference's np.partition-based neighbourhood mean may swap the k-th for the (k+1)-th neighbour
        assert np.abs(s.numpy() - g["sim_csls%d" % k]).max() < (1e-6 if k == 0 else 2e-2)


def test_dbpv1_model_oracle_matches_reference_golden():
    """Row a17: the oracle's restatement of JMAC_DBPv1's JMAC_MODEL (forward_base, get_emb, completion_loss with
    L2-normalised rows) against the reference's captured outputs."""
    g = load_golden("model_dbpv1")
    p = {k[len("state."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state.")}
    bn = {k: v.clone() for k, v in p.items() if "running" in k}
    ei, et = torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_type"])
    info = torch.from_numpy(g["ent_info_att"])
    a, comp, rel = orc.dbpv1_forward_base(p, info, ei, et, 2, 0.05, False, bn)
    for got, key in ((a, "align"), (comp[1], "comp_l1"), (rel[1], "rel_l1")):
        assert (got - torch.from_numpy(g[key])).abs().max().item() <= 2e-6 * max(1.0, float(np.abs(g[key]).max()))
    ea, ec = orc.get_emb(a, comp)
    assert (ea - torch.from_numpy(g["emb_align"])).abs().max().item() < 2e-6
    assert (ec - torch.from_numpy(g["emb_comp"])).abs().max().item() < 2e-6
    # completion loss in train mode from the BN state the reference started from
    pb = dict(p)
    for k in list(pb):
        if ("bn_before." + k) in g:
            pb[k] = torch.from_numpy(g["bn_before." + k])
    bnb = {k: v.clone() for k, v in pb.items() if "running" in k}
    _, comp_t, rel_t = orc.dbpv1_forward_base(pb, info, ei, et, 2, 0.05, True, bnb)
    loss = orc.dbpv1_completion_loss(comp_t, rel_t, torch.from_numpy(g["batch_h"]), torch.from_numpy(g["batch_r"]),
                                     torch.from_numpy(g["batch_t"]), g["links"], 30, 5.0)
    assert abs(float(loss) - float(g["completion_loss"])) < 2e-6 * abs(float(g["completion_loss"]))


def test_dbpv1_scoring_call_sites():
    """Row a18: the oracle's restatements of JMAC_DBPv1/modules/utils/util.py:35-58 and trainer/jmac_trainer.py:281-300
    against the fixture captured from the reference's own functions (tests/golden/gen_golden.py:gen_dbpv1_scoring)."""
    g = load_golden("scoring_dbpv1")
    emb, k, links = t(g["emb"]), int(g["k"]), g["links"]
    assert float(g["min_topk_gap"]) > 1e-5                       # index parity is well defined on this fixture
    assert (orc.dbpv1_get_neg(links[:, 1], emb, k).numpy() == g["neg2_left"]).all()
    assert (orc.dbpv1_get_neg(links[:, 0], emb, k).numpy() == g["neg_right"]).all()
    # the seed itself is its own nearest neighbour (one table on both sides)
    assert (g["neg_right"].reshape(len(links), k)[:, 0] == links[:, 0]).all()
    ent, p1, p2 = orc.dbpv1_alignment_quality(emb, g["list1"], g["list2"])
    assert abs(float(ent) - float(g["entropy"])) < 1e-6 * abs(float(g["entropy"]))
    assert_close(p1, g["softmax_simi"], 1e-6)
    assert_close(p2, g["softmax_simi2"], 1e-6)
