#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE itself (imported from /root/reference, build
container only) on seeded inputs and stores inputs + reference outputs as small .npz fixtures.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py          # writes tests/golden/*.npz

Nothing from the reference is copied: it is imported, called, and only arrays are saved.
Accommodations applied at import time:
  1. ``torch_scatter`` (third party, absent) -> oracle/_shim/torch_scatter (published algorithm);
  2. ``np.int`` / ``np.float`` / ``np.bool`` aliases restored (removed in numpy >= 1.24);
  3. ``train.py`` has a syntax error at line 101 -> its text is read, that one line is repaired in
     memory, and the module is exec'd (cwd is a temp dir: its Logger mkdirs ``logging/``);
  4. the DBPv1 variant lives in a second package tree with clashing module names -> generated in a
     child process with its own sys.path.
The GPU box never runs this file; it only reads the fixtures.
"""
import argparse
import logging
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("JMAC_REFERENCE", "/root/reference")

np.int = int        # noqa  (accommodation 2)
np.float = float    # noqa
if not hasattr(np, "bool"):
    np.bool = bool  # noqa

import torch  # noqa: E402

# ONE thread and deterministic kernels: the reference's scatter / index backward sums floats in thread order, and 120 Adam steps
# amplify such differences -- with this the committed fixtures regenerate bit-identically on any host load
torch.set_num_threads(1)
torch.use_deterministic_algorithms(True, warn_only=True)


def _paths(variant):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle", "_shim"))
    if variant == "dbpv1":
        sys.path.insert(0, os.path.join(REF, "JMAC_DBPv1"))
    else:
        sys.path.insert(0, REF)


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------
# graphs
# ----------------------------------------------------------------------------------------------
def tiny_graph():
    # 10 nodes. node 0: hub (6 in-edges incl. duplicate and a self edge); nodes 5,7,9: no in-edge.
    dst = [0, 0, 0, 0, 0, 0, 1, 2, 2, 3, 4, 4, 6, 8, 8, 8]
    src = [1, 2, 2, 0, 9, 5, 0, 3, 3, 3, 1, 7, 6, 0, 1, 2]
    typ = [0, 1, 1, 2, 3, 0, 0, 2, 2, 1, 3, 3, 0, 1, 2, 3]
    return 10, 4, np.array([dst, src], dtype=np.int64), np.array(typ, dtype=np.int64)


def random_graph(rng, n, nr, e, hub_frac=0.2):
    w = 1.0 / np.arange(1, n + 1) ** 0.9
    w = w / w.sum()
    dst = rng.choice(n, size=e, p=w)
    dst[: int(e * hub_frac)] = rng.integers(0, max(1, n // 50), int(e * hub_frac))
    src = rng.integers(0, n, e)
    typ = rng.integers(0, nr, e)
    # leave the top tenth of ids without in-edges (empty segments at the end, too)
    dst = np.where(dst >= n - n // 10, dst // 2, dst)
    perm = rng.permutation(e)
    return np.stack([dst[perm], src[perm]]).astype(np.int64), typ[perm].astype(np.int64)


def ja_slice(limit, bidirectional):
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "datasetdbp5l/kg/ja-train.tsv"), sep="\t", header=None).values
    df = df[(df[:, 0] < limit) & (df[:, 2] < limit)]
    h, r, t = df[:, 0], df[:, 1], df[:, 2]
    if bidirectional:      # src/utils.py:127-149: sender = h|t, receiver = t|h, same relation id
        ei = np.stack([np.concatenate([h, t]), np.concatenate([t, h])])
        et = np.concatenate([r, r])
    else:                  # train.py:130-132: [head, tail]
        ei = np.stack([h, t])
        et = r
    return ei.astype(np.int64), et.astype(np.int64)


# ----------------------------------------------------------------------------------------------
# layer fixtures
# ----------------------------------------------------------------------------------------------
def run_layer_case(layer_cls, make_layer, name, n, nr, d, ei, et, seed, x_scale=1.0):
    """Reference layer fwd (train + eval BN) and bwd on seeded inputs; returns dict of arrays."""
    import oracle.jmac_oracle as orc
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    layer = make_layer(d)
    with torch.no_grad():   # non-trivial BN affine so that its grads are exercised
        layer.bn.weight.copy_(1 + 0.1 * torch.randn(d, generator=g))
        layer.bn.bias.copy_(0.1 * torch.randn(d, generator=g))
    X = (torch.randn(n, d, generator=g) * x_scale / np.sqrt(d) * 4).requires_grad_(True)
    R = (torch.randn(nr, d, generator=g) / np.sqrt(d) * 4).requires_grad_(True)
    G = torch.randn(n, d, generator=g)
    edge_index = torch.from_numpy(ei)
    edge_type = torch.from_numpy(et)

    layer.train()
    out = layer(X, R, edge_index, edge_type)
    (out * G).sum().backward()
    arrays = dict(
        n=n, nr=nr, d=d, slope=layer.atv_mlp.negative_slope,
        edge_index=ei, edge_type=et, X=_np(X), R=_np(R), G=_np(G),
        out_train=_np(out), grad_X=_np(X.grad), grad_R=_np(R.grad),
        running_mean_after=_np(layer.bn.running_mean), running_var_after=_np(layer.bn.running_var),
    )
    for pname, p in layer.named_parameters():
        arrays["param." + pname] = _np(p)
        arrays["grad." + pname] = _np(p.grad) if p.grad is not None else np.zeros(p.shape, np.float32)

    # pre-BN neighbour aggregate, captured through the reference's own propagate()
    with torch.no_grad():
        rel = torch.cat([R, layer.loop_rel], 0) @ layer.rel_transform_weight1
        rel = (layer.act_rel(rel) if hasattr(layer, "act_rel") else layer.atv_mlp(rel)) @ layer.rel_transform_weight2
        norm = layer.compute_norm(edge_index, n)
        nb = layer.propagate("add", edge_index, x=X, edge_type=edge_type, rel_embed=rel,
                             edge_norm=norm, mode="in")
        arrays["nb"] = _np(nb)
        arrays["rel_transformed"] = _np(rel)

    layer.eval()
    with torch.no_grad():
        arrays["out_eval"] = _np(layer(X, R, edge_index, edge_type))

    # cross-check the oracle restatement right here (fails generation if it drifts)
    p = {k[len("param."):]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith("param.")}
    rel_act = "relu" if hasattr(layer, "act_rel") else "leaky_relu"
    o = orc.layer_forward(p, X.detach(), R.detach(), edge_index, edge_type, float(arrays["slope"]),
                          "sub", rel_act, True)
    err = (o - out.detach()).abs().max().item()
    assert err < 5e-6, (name, err)
    print("  %-18s N=%d E=%d d=%d  oracle-vs-reference max|err| = %.2e" % (name, n, ei.shape[1], d, err))
    return arrays


def gen_jafull():
    """The reference's layer on the WHOLE real DBP-5L ja graph in its bidirectional loader form (src/utils.py:127-149): N = 11 805,
    E = 35 958, in-degree up to 1 221 (hub rows: split and cooperative segments on the device), 4 332 isolated nodes.  d = 8
    keeps the fixture small; the arithmetic per edge is the same at every d."""
    from src.jmac_model import RelationAwareLayer
    args = types.SimpleNamespace(leaky_relu_w=0.05, comp_op="sub")
    mk = lambda d: RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=args)
    n = sum(1 for _ in open(os.path.join(REF, "datasetdbp5l/entity/ja.tsv")))
    ei, et = ja_slice(n, bidirectional=True)
    assert n == 11805 and ei.shape[1] == 35958
    arr = run_layer_case(RelationAwareLayer, mk, "ja_full", n, 961, 8, ei, et, 8)
    arr["edge_index"], arr["edge_type"] = ei.astype(np.int32), et.astype(np.int32)
    deg = np.bincount(ei[0], minlength=n)
    print("  max in-degree %d, isolated %d" % (deg.max(), int((deg == 0).sum())))
    _save("layer_ja_full", **arr)


def gen_root():
    from src.jmac_model import RelationAwareLayer, JMAC
    from src.knowledgegraph import KnowledgeGraph
    from src.validate import CompletionEvaluator
    from modules.utils.util import get_neg
    import oracle.jmac_oracle as orc

    args = types.SimpleNamespace(leaky_relu_w=0.05, comp_op="sub")
    mk = lambda d: RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=args)
    rng = np.random.default_rng(7)

    print("layer fixtures (reference: src/jmac_model.py RelationAwareLayer)")
    n, nr, ei, et = tiny_graph()
    _save("layer_tiny", **run_layer_case(RelationAwareLayer, mk, "tiny", n, nr, 8, ei, et, 1))
    ei, et = random_graph(rng, 200, 12, 900)
    _save("layer_rand200", **run_layer_case(RelationAwareLayer, mk, "rand200", 200, 12, 32, ei, et, 2))
    ei, et = random_graph(rng, 160, 30, 1100)
    _save("layer_d300", **run_layer_case(RelationAwareLayer, mk, "d300", 160, 30, 300, ei, et, 3))
    ei, et = ja_slice(1000, bidirectional=False)
    _save("layer_ja_train", **run_layer_case(RelationAwareLayer, mk, "ja_train", 1000, 961, 32, ei, et, 5))
    ei, et = ja_slice(1000, bidirectional=True)
    _save("layer_ja_bidir", **run_layer_case(RelationAwareLayer, mk, "ja_bidir", 1000, 961, 32, ei, et, 6))
    # no edges at all: every segment empty
    _save("layer_noedge", **run_layer_case(RelationAwareLayer, mk, "noedge", 17, 3, 12,
                                           np.zeros((2, 0), np.int64), np.zeros((0,), np.int64), 7))

    # ------------------------------------------------------------------------------------------
    # model-level fixture: two small KGs in global tables (bases), name info on, 2 GCN layers
    # ------------------------------------------------------------------------------------------
    print("model fixture (reference: JMAC.forward_name/get_emb/forward_linkpred/losses, CompletionEvaluator)")
    torch.manual_seed(11)
    d, nrel = 48, 9
    n1, n2 = 130, 110
    margs = types.SimpleNamespace(dim=d, dropout=0.0, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2,
                                  num_negative=5, margin_align=1.0, margin_completion=5.0, batch_size=40,
                                  no_name_info=False, device=torch.device("cpu"))
    name_emb = rng.standard_normal((n1 + n2, 20)).astype(np.float32)
    model = JMAC(margs, name_emb, 2 * nrel, n1 + n2)

    def kg_triples(n, m):
        h = rng.integers(0, n, m); r = rng.integers(0, nrel - 1, m); t = rng.integers(0, n, m)
        return np.unique(np.stack([h, r, t], 1), axis=0)

    tr1 = kg_triples(n1, 420)
    tr2 = kg_triples(n2, 300)
    k1 = int(len(tr1) * 0.7); k2 = int(len(tr1) * 0.85)
    perm = rng.permutation(len(tr1)); tr1 = tr1[perm]
    kg1 = KnowledgeGraph("aa", tr1[:k1], tr1[k1:k2], tr1[k2:], n1, nrel, False, 0, 0, "cpu")
    kg1.upper_entity_base, kg1.upper_relation_base = n1, nrel
    kg2 = KnowledgeGraph("bb", tr2[:200], tr2[200:250], tr2[250:], n2, nrel, True, n1, nrel, "cpu")
    kg2.upper_entity_base, kg2.upper_relation_base = n1 + n2, 2 * nrel

    def bidir(tr):
        h, r, t = tr[:, 0], tr[:, 1], tr[:, 2]
        return np.stack([np.concatenate([h, t]), np.concatenate([t, h])]).astype(np.int64), np.concatenate([r, r]).astype(np.int64)

    kg1.edge_index, kg1.edge_type = bidir(kg1.train_data)
    kg2.edge_index, kg2.edge_type = bidir(kg2.train_data)
    e1i, e1t = torch.from_numpy(kg1.edge_index), torch.from_numpy(kg1.edge_type)
    # train-mode graph for kg2 is single direction (train.py:130-135)
    e2i = torch.from_numpy(np.stack([kg2.train_data[:, 0], kg2.train_data[:, 2]]).astype(np.int64))
    e2t = torch.from_numpy(kg2.train_data[:, 1].astype(np.int64))
    eb1, rb1 = [0, n1], [0, nrel]
    eb2, rb2 = [n1, n1 + n2], [nrel, 2 * nrel]

    # a few train-mode steps so BN running stats are non-trivial, then eval
    model.train()
    with torch.no_grad():
        for _ in range(2):
            model.forward_base(e1i, e1t, eb1, rb1)
            model.forward_base(e2i, e2t, eb2, rb2)
    model.eval()
    arrays = {"name_emb": name_emb, "n1": n1, "n2": n2, "nrel": nrel, "d": d,
              "e1_index": kg1.edge_index, "e1_type": kg1.edge_type, "e2_index": _np(e2i), "e2_type": _np(e2t),
              "train1": kg1.train_data, "val1": kg1.val_data, "test1": kg1.test_data}
    for k, v in model.state_dict().items():
        arrays["state." + k] = _np(v)
    with torch.no_grad():
        a1, c1, r1 = model.forward_base(e1i, e1t, eb1, rb1)
        a2, c2, r2 = model.forward_base(e2i, e2t, eb2, rb2)
        arrays.update(align1=_np(a1), comp1_l0=_np(c1[0]), comp1_l1=_np(c1[1]), rel1_l0=_np(r1[0]), rel1_l1=_np(r1[1]),
                      align2=_np(a2), comp2_l1=_np(c2[1]))
        g1a, g1c = model.get_emb(e1i, e1t, eb1, rb1, pyt=True)
        g2a, g2c = model.get_emb(e2i, e2t, eb2, rb2, pyt=True)
        arrays.update(emb1_align=_np(g1a), emb1_comp=_np(g1c), emb2_align=_np(g2a), emb2_comp=_np(g2c))
        hb = kg1.h_val.tolist(); rb = kg1.r_val.tolist(); tb = kg1.t_val.tolist()
        dist = model.forward_linkpred(hb, rb, e1i, e1t, list(range(n1)), eb1, rb1)
        arrays.update(lp_h=np.array(hb), lp_r=np.array(rb), lp_t=np.array(tb), lp_dist=_np(dist))
        fp, fi = orc.build_filter_csr(hb, rb, kg1.true_tail)
        arrays.update(filt_ptr=fp, filt_idx=fi)
        ev = CompletionEvaluator(kg1, model, "cpu", None)
        lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
        for filt in (False, True):
            h1, h10, mrr = ev.test(margs, is_val=True, filterr=filt, logger=lg)
            arrays["eval_filt%d" % int(filt)] = np.array([h1, h10, mrr], dtype=np.float64)
        # the oracle's rank function reproduces the evaluator's metrics on the reference's distances
        for filt in (False, True):
            ranks = orc.filtered_ranks(dist, tb, fp if filt else None, fi if filt else None)
            got = np.array(orc.ranking_metrics(ranks))
            assert np.allclose(got, arrays["eval_filt%d" % int(filt)], atol=1e-12), (got, arrays["eval_filt%d" % int(filt)])
            arrays["ranks_filt%d" % int(filt)] = ranks

    # get_neg / compute_alignment_quality on the normalised embeddings (train.py:183-184, :145)
    links = np.stack([rng.permutation(n1)[:30], rng.permutation(n2)[:30]], 1)
    neg_right = get_neg(links[:, 0].tolist(), g1a, g2a, margs.num_negative)
    neg2_left = get_neg(links[:, 1].tolist(), g2a, g1a, margs.num_negative)
    arrays.update(links=links, neg_right=_np(neg_right), neg2_left=_np(neg2_left))

    tmp = tempfile.mkdtemp()
    cwd = os.getcwd(); os.chdir(tmp)
    try:
        src = open(os.path.join(REF, "train.py")).read().split("\n")
        assert "--no_name_info, action" in src[100]
        src[100] = src[100].replace("'--no_name_info, action", "'--no_name_info', action")   # accommodation 3
        train_mod = types.ModuleType("ref_train")
        train_mod.__dict__["__name__"] = "ref_train"
        exec(compile("\n".join(src), "train.py", "exec"), train_mod.__dict__)
    finally:
        os.chdir(cwd)
    t1 = rng.permutation(n1)[:25].tolist(); t2 = rng.permutation(n2)[:25].tolist()
    ent, sm1, sm2 = train_mod.compute_alignment_quality(g1a, g2a, t1, t2)
    arrays.update(aq_list1=np.array(t1), aq_list2=np.array(t2), aq_entropy=np.float64(ent.item()),
                  aq_softmax_rows=_np(sm1), aq_softmax_cols=_np(sm2))

    # losses in train mode (dropout p = 0 so they are deterministic), with grads of a few leaves
    model.train()
    B, K = margs.batch_size, margs.num_negative
    trip = torch.from_numpy(kg1.train_data[:B].astype(np.int64))
    neg = torch.from_numpy(rng.integers(0, n1, (B, K)).astype(np.int64))
    sub, rel, obj = trip[:, 0], trip[:, 1], trip[:, 2]
    data = {"batch_h": sub.repeat(K + 1), "batch_r": rel.repeat(K + 1), "batch_t": torch.cat((obj, neg.view(-1)))}
    pos = np.ones((len(links), K)) * links[:, 0].reshape(-1, 1)
    pos2 = np.ones((len(links), K)) * links[:, 1].reshape(-1, 1)
    feed = {"neg_left": pos.reshape(-1), "neg_right": neg_right, "neg2_left": neg2_left, "neg2_right": pos2.reshape(-1),
            "links": links, "ent_bases1": eb1, "ent_bases2": eb2, "rel_bases1": rb1, "rel_bases2": rb2}
    bn_before = {k: _np(v) for k, v in model.state_dict().items() if "running" in k}
    model.zero_grad()
    closs = model.completion_loss(data, e1i, e1t, e2i, e2t, feed, True)
    closs.backward()
    arrays.update(batch_h=_np(data["batch_h"]), batch_r=_np(data["batch_r"]), batch_t=_np(data["batch_t"]),
                  completion_loss=np.float64(closs.item()),
                  closs_grad_ent=_np(model.ent_init_att_completion.grad),
                  closs_grad_w_att=_np(model.conv1_completion.w_att.grad),
                  closs_grad_a_att=_np(model.conv1_completion.a_att.grad),
                  closs_grad_rel=_np(model.rel_init_att_completion.grad))
    for k, v in bn_before.items():
        arrays["bn_before." + k] = v
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in bn_before.items()}})
    model.zero_grad()
    aloss = model.alignment_loss(feed, e1i, e1t, e2i, e2t)
    aloss.backward()
    arrays.update(alignment_loss=np.float64(aloss.item()),
                  aloss_grad_name_linear=_np(model.name_linear.grad),
                  aloss_grad_conv2_gcn=_np(model.conv2_alignment.gcn_weight.grad),
                  aloss_grad_ent=_np(model.ent_init_att_completion.grad))
    _save("model_small", **arrays)


def _induced_subgraph(lang, m):
    """The m highest-degree entities of a DBP-5L KG (train + val + test degree, ties -> lower id) and the triples
    among them, entities relabelled 0..m-1 in that order; relation ids are kept.  Returns (train, val, test, old_ids)."""
    import pandas as pd
    root = os.path.join(REF, "datasetdbp5l", "kg")
    parts = {sp: pd.read_csv(os.path.join(root, "%s-%s.tsv" % (lang, sp)), sep="\t", header=None).values.astype(np.int64)
             for sp in ("train", "val", "test")}
    allt = np.concatenate(list(parts.values()))
    n = int(max(allt[:, 0].max(), allt[:, 2].max())) + 1
    deg = np.bincount(allt[:, 0], minlength=n) + np.bincount(allt[:, 2], minlength=n)
    keep = np.argsort(-deg, kind="stable")[:m]
    new = -np.ones(n, dtype=np.int64)
    new[keep] = np.arange(m)
    out = []
    for sp in ("train", "val", "test"):
        t = parts[sp]
        ok = (new[t[:, 0]] >= 0) & (new[t[:, 2]] >= 0)
        t = t[ok]
        out.append(np.stack([new[t[:, 0]], t[:, 1], new[t[:, 2]]], 1))
    return out[0], out[1], out[2], keep


def gen_e2e():
    """End-to-end parity fixture (BASELINE metric "Hits@1 parity"): the REFERENCE's JMAC trained for a fixed, seeded
    number of steps (dropout 0, captured batches) on sub-graphs derived from the real DBP-5L ja / el KGs, then scored by the
    reference's own CompletionEvaluator.test (src/validate.py:22-80, filtered, validation split of the target KG).
    Stored: the initial state_dict, graphs, every batch, the per-step losses, Hits@1 / Hits@10 / MRR before and after
    and the final ranks -- so that the same steps can be replayed through the oracle (CPU) and the HIP path (GPU)."""
    from src.jmac_model import JMAC
    from src.knowledgegraph import KnowledgeGraph
    from src.validate import CompletionEvaluator
    import pandas as pd
    import oracle.jmac_oracle as orc
    rng = np.random.default_rng(2024)
    torch.manual_seed(2024)
    n1, n2, d, nrel = 700, 500, 32, 961                      # nrel: relations.txt lines + 1 (src/data_loader.py:214-215)
    tr1, va1, te1, old1 = _induced_subgraph("ja", n1)        # target KG (trained on train only)
    tr2, va2, te2, old2 = _induced_subgraph("el", n2)        # supporter KG (train + val, knowledgegraph.py:18-19)
    seeds = pd.read_csv(os.path.join(REF, "datasetdbp5l", "seed_train_pairs", "el-ja.tsv"), sep="\t", header=None).values
    seeds = seeds.astype(np.int64)                           # (el id, ja id), float-formatted on disk
    m2 = -np.ones(int(max(old2.max(), seeds[:, 0].max())) + 1, np.int64); m2[old2] = np.arange(n2)
    m1 = -np.ones(int(max(old1.max(), seeds[:, 1].max())) + 1, np.int64); m1[old1] = np.arange(n1)
    ok = (m2[seeds[:, 0]] >= 0) & (m1[seeds[:, 1]] >= 0)
    links = np.stack([m1[seeds[ok, 1]], m2[seeds[ok, 0]]], 1)        # (kg1 = ja, kg2 = el), LOCAL ids like train.py:183
    margs = types.SimpleNamespace(dim=d, dropout=0.0, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=5,
                                  margin_align=1.0, margin_completion=5.0, batch_size=64, no_name_info=False,
                                  device=torch.device("cpu"))
    name_emb = rng.standard_normal((n1 + n2, 20)).astype(np.float32)
    model = JMAC(margs, name_emb, 2 * nrel, n1 + n2)
    # loop_rel is frozen for this run.  Its gradient is mathematically ZERO under train-mode BatchNorm (it only shifts
    # every pre-BN row by the same vector, which the batch mean removes), so what Adam sees is fp32 rounding noise -- and
    # Adam turns any noise above its eps into full +-lr steps.  The walk is invisible in train mode but moves the
    # running means the evaluator then uses, i.e. it makes the final ranks a function of rounding order rather than of
    # the algorithm.  Freezing it (a legitimate use of the reference's own module) makes the run reproducible by any
    # correct fp32 implementation; everything else is exactly train.py's procedure.
    for lay in (model.conv1_alignment, model.conv2_alignment, model.conv1_completion):
        lay.loop_rel.requires_grad_(False)
    kg1 = KnowledgeGraph("ja", tr1, va1, te1, n1, nrel, False, 0, 0, "cpu")
    kg1.upper_entity_base, kg1.upper_relation_base = n1, nrel
    kg2 = KnowledgeGraph("el", tr2, va2, te2, n2, nrel, True, n1, nrel, "cpu")
    kg2.upper_entity_base, kg2.upper_relation_base = n1 + n2, 2 * nrel

    def train_graph(tr):                                      # align_data_processing, train.py:116-135: head <- tail
        return np.stack([tr[:, 0], tr[:, 2]]).astype(np.int64), tr[:, 1].astype(np.int64)
    ei1, et1 = train_graph(kg1.train_data)
    ei2, et2 = train_graph(kg2.train_data)
    kg1.edge_index, kg1.edge_type = ei1, et1                  # the evaluator scores on the graph it is handed
    e1i, e1t, e2i, e2t = (torch.from_numpy(x) for x in (ei1, et1, ei2, et2))
    eb1, rb1, eb2, rb2 = [0, n1], [0, nrel], [n1, n1 + n2], [nrel, 2 * nrel]
    K, B = margs.num_negative, margs.batch_size
    neg_l = np.repeat(links[:, 0], K); neg2_r = np.repeat(links[:, 1], K)
    feed = {"links": links, "neg_left": neg_l.astype(np.float64), "neg_right": torch.from_numpy(rng.integers(0, n2, len(links) * K)),
            "neg2_left": torch.from_numpy(rng.integers(0, n1, len(links) * K)), "neg2_right": neg2_r.astype(np.float64),
            "ent_bases1": eb1, "ent_bases2": eb2, "rel_bases1": rb1, "rel_bases2": rb2}
    arrays = {"n1": n1, "n2": n2, "d": d, "nrel": nrel, "name_emb": name_emb, "links": links, "e1_index": ei1, "e1_type": et1,
              "e2_index": ei2, "e2_type": et2, "train1": kg1.train_data, "val1": kg1.val_data, "test1": kg1.test_data,
              "train2": kg2.train_data, "neg_right": _np(feed["neg_right"]), "neg2_left": _np(feed["neg2_left"]),
              "lr": np.float64(5e-3), "batch_size": B, "num_negative": K}
    lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
    ev = CompletionEvaluator(kg1, model, "cpu", None)

    def one_step(kind, tr, nent, opt_a, opt_c):
        """One step of train.py's loop: an alignment step (train.py:367-378) or a completion batch (train.py:338-352)."""
        if kind == "a":
            opt_a.zero_grad()
            loss = model.alignment_loss(feed, e1i, e1t, e2i, e2t)
            loss.backward(); opt_a.step()
            return loss, None
        trip = torch.from_numpy(tr[rng.permutation(len(tr))[:B]].astype(np.int64))
        neg = torch.from_numpy(rng.integers(0, nent, (B, K)).astype(np.int64))
        data = {"batch_h": trip[:, 0].repeat(K + 1), "batch_r": trip[:, 1].repeat(K + 1),
                "batch_t": torch.cat((trip[:, 2], neg.view(-1)))}            # train.py:347-352
        opt_c.zero_grad()
        loss = model.completion_loss(data, e1i, e1t, e2i, e2t, feed, kind == "c1")
        loss.backward(); opt_c.step()
        return loss, data
    EPOCH = [("c1", kg1.train_data, n1)] * 6 + [("c2", kg2.train_data, n2)] * 3 + [("a", None, 0)]   # train.py:486-489

    # ---- phase 0 (not replayed): the reference trains ITSELF from its random initialisation until the filtered Hits@1 of the
    # validation split is well away from zero -- an untrained or barely trained L1 translation model ranks the head entity
    # itself first (h + r ~ h), i.e. Hits@1 = 1/len(val) whatever the implementation does, which pins nothing.  The state it
    # reaches is the replay's starting point (``state0``); the replayed steps start with fresh optimisers.
    model.eval()
    with torch.no_grad():
        arrays["metrics_untrained"] = np.array(ev.test(margs, is_val=True, filterr=True, logger=lg), dtype=np.float64)
    model.train()
    PRE = 25
    arrays["pretrain_steps"] = PRE * len(EPOCH)
    opt_a = torch.optim.Adam(model.parameters(), lr=float(arrays["lr"]))
    opt_c = torch.optim.Adam(model.parameters(), lr=float(arrays["lr"]))
    for _ in range(PRE):
        for kind, tr, nent in EPOCH:
            one_step(kind, tr, nent, opt_a, opt_c)
    for k, v in model.state_dict().items():
        arrays["state0." + k] = _np(v)
    model.eval()
    with torch.no_grad():
        arrays["metrics_before"] = np.array(ev.test(margs, is_val=True, filterr=True, logger=lg), dtype=np.float64)
    hb, rb, tb = kg1.h_val.tolist(), kg1.r_val.tolist(), kg1.t_val.tolist()
    fp, fi = orc.build_filter_csr(hb, rb, kg1.true_tail)
    arrays.update(filt_ptr=fp, filt_idx=fi)

    def evaluate(tag):
        """The reference's evaluator on the validation split (filtered) + the ranks behind its metrics + how decisive each
        rank is: the distance gap between the gold tail and its nearest unfiltered competitor."""
        model.eval()
        with torch.no_grad():
            arrays["metrics_" + tag] = np.array(ev.test(margs, is_val=True, filterr=True, logger=lg), dtype=np.float64)
            dist = model.forward_linkpred(hb, rb, e1i, e1t, list(range(n1)), eb1, rb1)
        ranks = orc.filtered_ranks(dist, tb, fp, fi)
        assert np.allclose(np.array(orc.ranking_metrics(ranks)), arrays["metrics_" + tag], atol=1e-12)
        dn = dist.numpy().astype(np.float64)
        gap = np.empty(len(tb))
        for j in range(len(tb)):
            comp = np.ones(n1, bool)
            comp[fi[fp[j]:fp[j + 1]]] = False
            comp[tb[j]] = False
            gap[j] = np.abs(dn[j, comp] - dn[j, tb[j]]).min()
        arrays["ranks_" + tag] = ranks
        arrays["rank_gap_" + tag] = gap
        model.train()

    # train.py:406-407: two Adam optimisers over ALL parameters
    opt_a = torch.optim.Adam(model.parameters(), lr=float(arrays["lr"]))
    opt_c = torch.optim.Adam(model.parameters(), lr=float(arrays["lr"]))
    model.train()
    # schedule: per epoch 6 completion steps on the target KG, 3 on the supporter KG, 1 alignment step (train.py:486-489)
    sched, bh, br, bt, losses = [], [], [], [], []
    CKPT = 3                                                   # epochs before the first evaluation (30 steps)
    arrays["ckpt_steps"] = CKPT * 10
    for epoch in range(12):
        if epoch == CKPT:
            evaluate("ckpt")
        for kind, tr, nent in EPOCH:
            loss, data = one_step(kind, tr, nent, opt_a, opt_c)
            if data is None:
                h = r = t = np.zeros(B * (K + 1), np.int64)
            else:
                h, r, t = (_np(data[x]) for x in ("batch_h", "batch_r", "batch_t"))
            sched.append({"c1": 0, "c2": 1, "a": 2}[kind]); bh.append(h); br.append(r); bt.append(t)
            losses.append(loss.item())
    arrays.update(sched=np.array(sched), batch_h=np.stack(bh), batch_r=np.stack(br), batch_t=np.stack(bt),
                  losses=np.array(losses, dtype=np.float64))
    evaluate("after")
    for k, v in model.state_dict().items():
        arrays["state1." + k] = _np(v)
    print("e2e: n1=%d E1=%d val=%d | n2=%d E2=%d | links=%d | steps=%d (+%d before state0) | loss %.4f -> %.4f | H@1/H@10/MRR %s -> %s -> %s -> %s" % (
        n1, ei1.shape[1], len(kg1.val_data), n2, ei2.shape[1], len(links), len(sched), int(arrays["pretrain_steps"]), losses[0],
        losses[-1], np.round(arrays["metrics_untrained"], 4), np.round(arrays["metrics_before"], 4),
        np.round(arrays["metrics_ckpt"], 4), np.round(arrays["metrics_after"], 4)))
    _save("e2e_ja_sub", **arrays)


def gen_aligneval():
    """modules.finding.evaluation.test on the model_small embeddings (train.py:105-113 settings)."""
    from modules.finding.evaluation import test
    from modules.finding.similarity import sim
    g = dict(np.load(os.path.join(HERE, "model_small.npz")))
    rng = np.random.default_rng(33)
    n = 100
    e1 = g["emb1_align"][:n].astype(np.float32)
    # a noisy copy of e1 plus the other KG's embedding: a non-trivial but learnable alignment
    e2 = (0.6 * e1 + 0.4 * g["emb2_align"][:n] + 0.05 * rng.standard_normal(e1.shape)).astype(np.float32)
    lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
    out = {"e1": e1, "e2": e2}
    for k in (0, 10):
        top_k, hits, mr, mrr = test(e1, e2, None, [1, 5, 10], 1, metric="cosine", normalize=False, csls_k=k, accurate=True, logger=lg)
        out["hits_csls%d" % k] = np.asarray(hits, dtype=np.float64)
        out["mr_csls%d" % k] = np.float64(mr)
        out["mrr_csls%d" % k] = np.float64(mrr)
        out["sim_csls%d" % k] = sim(e1, e2, metric="cosine", normalize=False, csls_k=k).astype(np.float32)
    _save("align_eval", **out)


def _load_ref_train():
    """train.py cannot be imported (syntax error at :101): read, repair that line in memory, exec (accommodation 3)."""
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd(); os.chdir(tmp)
    try:
        src = open(os.path.join(REF, "train.py")).read().split("\n")
        assert "--no_name_info, action" in src[100]
        src[100] = src[100].replace("'--no_name_info, action", "'--no_name_info', action")
        train_mod = types.ModuleType("ref_train")
        train_mod.__dict__["__name__"] = "ref_train"
        exec(compile("\n".join(src), "train.py", "exec"), train_mod.__dict__)
    finally:
        os.chdir(cwd)
    return train_mod


def gen_entr():
    """EnTr bookkeeping (row f4): train.py:138-211 seed_enlargement_triple_transferring (first-visit branch: no
    multinomial draw, so it is deterministic) and train.py:297-325 transfer_knowledge, on the model_small embeddings."""
    train_mod = _load_ref_train()
    g = dict(np.load(os.path.join(HERE, "model_small.npz")))
    rng = np.random.default_rng(55)
    n1, n2, nrel = int(g["n1"]), int(g["n2"]), int(g["nrel"])
    out1, out2 = torch.from_numpy(g["emb1_align"]), torch.from_numpy(g["emb2_align"])
    triples1 = g["train1"].astype(np.int64)
    h = rng.integers(0, n2, 260); r = rng.integers(0, nrel - 1, 260); t = rng.integers(0, n2, 260)
    triples2 = np.unique(np.stack([h, r, t], 1), axis=0)[rng.permutation(250)[:230]].astype(np.int64)
    # links: 70 pairs; entity 0 on either side (the reference's `links.get(x)` truthiness quirk), one source listed twice
    src = rng.permutation(n1)[:70]; dst = rng.permutation(n2)[:70]
    src[3], dst[3] = 5, 0
    src[4], dst[4] = 0, 7
    src[9] = src[8]
    links = np.stack([src, dst], 1).astype(np.int64)
    # make sure several triples have both ends linked, in both directions, some colliding with existing triples
    for i in range(12):
        a, b = rng.integers(0, 70, 2)
        triples1[i] = [links[a, 0], rng.integers(0, nrel - 1), links[b, 0]]
        triples2[i] = [links[b, 1], rng.integers(0, nrel - 1), links[a, 1]]
    triples2[12] = [links[20, 1], triples1[0, 1], links[21, 1]]
    triples1[12] = [links[20, 0], triples1[0, 1], links[21, 0]]          # its image already exists in KG 2
    triples1[13] = triples1[1]                                          # duplicate source triple: transferred once
    kg = lambda tr: types.SimpleNamespace(triple_keys=set("%d_%d_%d" % tuple(x) for x in tr.tolist()))
    kg1, kg2 = kg(triples1), kg(triples2)
    args = types.SimpleNamespace(num_negative=5, pair_sample_weight=0.2)
    test_src = rng.permutation(n1)[:25].tolist(); test_dst = rng.permutation(n2)[:25].tolist()
    ge, gs = [-1], [links]
    nt1, nt2, nk1, nk2, feed, gs_out = train_mod.seed_enlargement_triple_transferring(
        out1, out2, test_src, test_dst, ge, 0, links, triples1.tolist(), triples2.tolist(), gs, [0, n1], [0, nrel],
        [n1, n1 + n2], [nrel, 2 * nrel], kg1, kg2, args)
    key = lambda ks: np.array(sorted(tuple(int(v) for v in k.split("_")) for k in ks), dtype=np.int64)
    _save("entr_small", n1=n1, n2=n2, nrel=nrel, triples1=triples1, triples2=triples2, links=links,
          test_src=np.array(test_src), test_dst=np.array(test_dst),
          new_triples1=np.array([list(x) for x in nt1], dtype=np.int64), new_triples2=np.array([list(x) for x in nt2], dtype=np.int64),
          keys1=key(nk1), keys2=key(nk2), entropy=np.float64(float(ge[0])),
          neg_left=np.asarray(feed["neg_left"], dtype=np.float64), neg_right=_np(feed["neg_right"]),
          neg2_left=_np(feed["neg2_left"]), neg2_right=np.asarray(feed["neg2_right"], dtype=np.float64),
          feed_links=np.asarray(feed["links"], dtype=np.int64))


def gen_dataset():
    """Dataset reader (row f2, formats): a seeded SYNTHETIC mini dataset in the DBP-5L on-disk format
    (tests/golden/dbp5l_mini/: entity/<lang>.tsv, kg/<lang>-{train,val,test}.tsv, seed_{train,test}_pairs/<l1>-<l2>.tsv
    with float-formatted ids, relations.txt) is written, then the REFERENCE's loader
    (src/data_loader.py:158-221 ParseData.create_KG_objects_and_alignment -> src/utils.py:112-154) reads it."""
    from src.data_loader import ParseData
    root = os.path.join(HERE, "dbp5l_mini")
    rng = np.random.default_rng(77)
    sizes = {"el": 40, "ja": 55, "en": 70}
    nrel = 12
    for sub in ("entity", "kg", "seed_train_pairs", "seed_test_pairs"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    with open(os.path.join(root, "relations.txt"), "w") as f:
        f.write("".join("http://example.org/property/r%d\n" % i for i in range(nrel)))
    for lang, n in sizes.items():
        with open(os.path.join(root, "entity", lang + ".tsv"), "w") as f:
            f.write("".join("%s entity %d\n" % (lang, i) for i in range(n)))
        tr = np.unique(np.stack([rng.integers(0, n, 4 * n), rng.integers(0, nrel, 4 * n), rng.integers(0, n, 4 * n)], 1), axis=0)
        tr = tr[rng.permutation(len(tr))]
        a, b = int(len(tr) * 0.6), int(len(tr) * 0.85)
        for name, part in (("train", tr[:a]), ("val", tr[a:b]), ("test", tr[b:])):
            np.savetxt(os.path.join(root, "kg", "%s-%s.tsv" % (lang, name)), part, fmt="%d", delimiter="\t")
    for l1, l2 in (("el", "ja"), ("ja", "en"), ("el", "en")):
        m = min(sizes[l1], sizes[l2])
        pairs = np.stack([rng.permutation(sizes[l1])[:m], rng.permutation(sizes[l2])[:m]], 1).astype(np.float64)
        np.savetxt(os.path.join(root, "seed_train_pairs", "%s-%s.tsv" % (l1, l2)), pairs[: m // 2], fmt="%.1f", delimiter="\t")
        np.savetxt(os.path.join(root, "seed_test_pairs", "%s-%s.tsv" % (l1, l2)), pairs[m // 2:], fmt="%.1f", delimiter="\t")
    lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
    out = {}
    for target in ("ja", "en"):
        pd_ = ParseData(types.SimpleNamespace(data_path=root, target_language=target, device="cpu"), lg)
        kgs, s_train, s_test = pd_.create_KG_objects_and_alignment()
        out["%s.kg_names" % target] = np.array(pd_.kg_names)
        out["%s.num_entities" % target] = np.int64(pd_.num_entities)
        for lang, kg in kgs.items():
            pre = "%s.%s." % (target, lang)
            out[pre + "train"], out[pre + "val"], out[pre + "test"] = kg.train_data, kg.val_data, kg.test_data
            out[pre + "meta"] = np.array([kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base,
                                          kg.relation_id_base, kg.upper_entity_base, kg.upper_relation_base], dtype=np.int64)
            out[pre + "edge_index"], out[pre + "edge_type"] = np.asarray(kg.edge_index), np.asarray(kg.edge_type)
            if not kg.is_supporter_kg:      # filter dictionary of the evaluator (knowledgegraph.py:45-46,62-86)
                keys = sorted(kg.true_tail.keys())
                out[pre + "true_tail_keys"] = np.array(keys, dtype=np.int64)
                out[pre + "true_tail_ptr"] = np.cumsum([0] + [len(kg.true_tail[k]) for k in keys]).astype(np.int64)
                out[pre + "true_tail_idx"] = np.array([t for k in keys for t in kg.true_tail[k]], dtype=np.int64)
        for name, seeds in (("seeds_train", s_train), ("seeds_test", s_test)):
            for (l1, l2), v in seeds.items():
                out["%s.%s.%s-%s" % (target, name, l1, l2)] = v
    _save("dbp5l_mini", **out)


def array_digest(a):
    """Order-sensitive 64-bit digest of an integer array (plain numpy arithmetic; the tests recompute it)."""
    v = np.ascontiguousarray(a).astype(np.uint64).reshape(-1)
    w = (np.arange(1, v.size + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    return np.uint64((v * w).sum(dtype=np.uint64) ^ np.uint64(v.size))


def gen_realpair():
    """BASELINE config 1 / SURVEY 8(d) "Config 1": the REAL DBP-5L el (supporter) and ja (target) KGs with their seed pairs as
    integer arrays (dbp5l_ja_el_data.npz: triples and seed pairs; entity / relation names are not kept -- the loaders only
    count those lines).  tests/util.py:write_dbp5l_dir turns the arrays back into the dataset's on-disk format in a temporary
    directory; the REFERENCE's loader reads that directory here and its arrays are pinned by shape and digest
    (dbp5l_ja_el.npz)."""
    from src.data_loader import ParseData
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from util import write_dbp5l_dir
    src = os.path.join(REF, "datasetdbp5l")
    data = {"langs": np.array(["el", "ja"]), "seed_pair": np.array(["el", "ja"]),
            "n_relation_lines": np.int64(sum(1 for _ in open(os.path.join(src, "relations.txt"))))}
    for lang in ("el", "ja"):
        data[lang + ".num_entity"] = np.int64(sum(1 for _ in open(os.path.join(src, "entity", lang + ".tsv"))))
        for part in ("train", "val", "test"):
            data["%s.%s" % (lang, part)] = np.loadtxt(os.path.join(src, "kg", "%s-%s.tsv" % (lang, part)), dtype=np.int64,
                                                      delimiter="\t").reshape(-1, 3).astype(np.int32)
    for sub in ("seed_train_pairs", "seed_test_pairs"):
        data[sub] = np.loadtxt(os.path.join(src, sub, "el-ja.tsv"), dtype=np.float64, delimiter="\t").reshape(-1, 2).astype(np.int32)
    _save("dbp5l_ja_el_data", **data)
    root = write_dbp5l_dir(tempfile.mkdtemp(prefix="dbp5l_ja_el_"), data)
    lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
    pd_ = ParseData(types.SimpleNamespace(data_path=root, target_language="ja", device="cpu"), lg)
    kgs, s_train, s_test = pd_.create_KG_objects_and_alignment()
    out = {"kg_names": np.array(pd_.kg_names), "num_entities": np.int64(pd_.num_entities)}
    for lang, kg in kgs.items():
        ei, et = kg.edge_index.cpu().numpy() if hasattr(kg.edge_index, "cpu") else np.asarray(kg.edge_index), \
            kg.edge_type.cpu().numpy() if hasattr(kg.edge_type, "cpu") else np.asarray(kg.edge_type)
        out[lang + ".meta"] = np.array([kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base,
                                        kg.relation_id_base, kg.upper_entity_base, kg.upper_relation_base], dtype=np.int64)
        out[lang + ".shapes"] = np.array([len(kg.train_data), len(kg.val_data), len(kg.test_data), ei.shape[1]], dtype=np.int64)
        out[lang + ".digests"] = np.array([array_digest(kg.train_data), array_digest(kg.val_data), array_digest(kg.test_data),
                                           array_digest(ei), array_digest(et)], dtype=np.uint64)
        deg_in = np.bincount(ei[0], minlength=kg.num_entity)
        out[lang + ".degree"] = np.array([deg_in.max(), int((deg_in == 0).sum())], dtype=np.int64)
        if not kg.is_supporter_kg:
            tt = kg.true_tail if hasattr(kg, "true_tail") else {}
            out[lang + ".true_tail"] = np.array([len(tt), sum(len(v) for v in tt.values())], dtype=np.int64)
    for tag, sd in (("seeds_train", s_train), ("seeds_test", s_test)):
        (k, v), = sd.items()
        out[tag + ".pair"] = np.array(list(k))
        out[tag] = np.asarray(v.cpu().numpy() if hasattr(v, "cpu") else v, dtype=np.int64)
    _save("dbp5l_ja_el", **out)
    for k in sorted(out):
        if k.endswith("shapes") or k.endswith("meta") or k.endswith("degree"):
            print("  ", k, out[k])


def gen_realall():
    """BASELINE config 3's data: ALL FIVE real DBP-5L KGs (el, en, es, fr, ja: train / val / test triples) and the ten seed-pair
    files as integer arrays (dbp5l_all_data.npz; entity / relation names are not kept -- the loaders only count those lines).
    As for the el / ja pair (gen_realpair), tests/util.py:write_dbp5l_dir turns the arrays back into the dataset's on-disk
    format in a temporary directory, the REFERENCE's loader reads that directory here (target_language = ja) and its arrays
    are pinned by shape, id bases, digests and degrees (dbp5l_all.npz)."""
    from src.data_loader import ParseData
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from util import write_dbp5l_dir
    src = os.path.join(REF, "datasetdbp5l")
    langs = sorted(f[:2] for f in os.listdir(os.path.join(src, "entity")) if f.endswith(".tsv"))
    data = {"langs": np.array(langs), "n_relation_lines": np.int64(sum(1 for _ in open(os.path.join(src, "relations.txt"))))}
    for lang in langs:
        data[lang + ".num_entity"] = np.int64(sum(1 for _ in open(os.path.join(src, "entity", lang + ".tsv"))))
        for part in ("train", "val", "test"):
            data["%s.%s" % (lang, part)] = np.loadtxt(os.path.join(src, "kg", "%s-%s.tsv" % (lang, part)), dtype=np.int64,
                                                      delimiter="\t").reshape(-1, 3).astype(np.int32)
    pairs = sorted(f[:5] for f in os.listdir(os.path.join(src, "seed_train_pairs")) if f.endswith(".tsv"))
    data["seed_pairs"] = np.array(pairs)
    for sub in ("seed_train_pairs", "seed_test_pairs"):
        for pr in pairs:
            data["%s.%s" % (sub, pr)] = np.loadtxt(os.path.join(src, sub, pr + ".tsv"), dtype=np.float64,
                                                   delimiter="\t").reshape(-1, 2).astype(np.int32)
    _save("dbp5l_all_data", **data)
    root = write_dbp5l_dir(tempfile.mkdtemp(prefix="dbp5l_all_"), data)
    lg = logging.getLogger("golden"); lg.setLevel(logging.ERROR)
    pd_ = ParseData(types.SimpleNamespace(data_path=root, target_language="ja", device="cpu"), lg)
    kgs, s_train, s_test = pd_.create_KG_objects_and_alignment()
    out = {"kg_names": np.array(pd_.kg_names), "num_entities": np.int64(pd_.num_entities)}
    for lang, kg in kgs.items():
        ei = kg.edge_index.cpu().numpy() if hasattr(kg.edge_index, "cpu") else np.asarray(kg.edge_index)
        et = kg.edge_type.cpu().numpy() if hasattr(kg.edge_type, "cpu") else np.asarray(kg.edge_type)
        out[lang + ".meta"] = np.array([kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base,
                                        kg.relation_id_base, kg.upper_entity_base, kg.upper_relation_base], dtype=np.int64)
        out[lang + ".shapes"] = np.array([len(kg.train_data), len(kg.val_data), len(kg.test_data), ei.shape[1]], dtype=np.int64)
        out[lang + ".digests"] = np.array([array_digest(kg.train_data), array_digest(kg.val_data), array_digest(kg.test_data),
                                           array_digest(ei), array_digest(et)], dtype=np.uint64)
        deg_in = np.bincount(ei[0], minlength=kg.num_entity)
        out[lang + ".degree"] = np.array([deg_in.max(), int((deg_in == 0).sum())], dtype=np.int64)
        tdeg = np.bincount(np.asarray(kg.train_data)[:, 0], minlength=kg.num_entity)       # train-mode graph: heads aggregate
        out[lang + ".train_degree"] = np.array([tdeg.max(), int((tdeg == 0).sum())], dtype=np.int64)
    for tag, sd in (("seeds_train", s_train), ("seeds_test", s_test)):
        keys = sorted(sd)
        out[tag + ".pairs"] = np.array(["%s-%s" % k for k in keys])
        out[tag + ".sizes"] = np.array([len(sd[k]) for k in keys], dtype=np.int64)
        out[tag + ".digests"] = np.array([array_digest(np.asarray(sd[k].cpu().numpy() if hasattr(sd[k], "cpu") else sd[k], dtype=np.int64))
                                          for k in keys], dtype=np.uint64)
    _save("dbp5l_all", **out)
    for k in sorted(out):
        if k.endswith("shapes") or k.endswith("meta") or k.endswith("degree"):
            print("  ", k, out[k])


def gen_dbpv1_model():
    """Model-level fixture of the DBPv1 variant (row a17): JMAC_DBPv1/models/jmac_model.py:116-277 JMAC_MODEL on one
    merged graph with inverse edges (jmac_trainer.py:93-96): forward_base, get_emb, completion_loss (rows L2-normalised
    before the L1 score, :245-247), alignment_loss, with grads."""
    from models.jmac_model import JMAC_MODEL
    rng = np.random.default_rng(91)
    torch.manual_seed(19)
    n, nrel, d = 150, 7, 40
    args = types.SimpleNamespace(emb_dim=d, completion_dropout_rate=0.0, leaky_relu_w=0.05, opn="sub", num_gcn_layer=2,
                                 num_negative=4, margin_align=1.0, margin_completion=5.0, completion_batch_size=30)
    info = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    model = JMAC_MODEL(n, nrel, info, args)
    tr = np.unique(np.stack([rng.integers(0, n, 500), rng.integers(0, nrel, 500), rng.integers(0, n, 500)], 1), axis=0)
    # inverse edges get relation id r + num_rel (jmac_trainer.py:93-96)
    ei = np.stack([np.concatenate([tr[:, 0], tr[:, 2]]), np.concatenate([tr[:, 2], tr[:, 0]])]).astype(np.int64)
    et = np.concatenate([tr[:, 1], tr[:, 1] + nrel]).astype(np.int64)
    eit, ett = torch.from_numpy(ei), torch.from_numpy(et)
    model.train()
    with torch.no_grad():
        for _ in range(2):
            model.forward_base(eit, ett)
    arrays = {"n": n, "nrel": nrel, "d": d, "edge_index": ei, "edge_type": et, "ent_info_att": _np(info)}
    model.eval()
    for k, v in model.state_dict().items():
        arrays["state." + k] = _np(v)
    with torch.no_grad():
        a, comp, rel = model.forward_base(eit, ett)
        ga, gc = model.get_emb(eit, ett, pyt=True)
    arrays.update(align=_np(a), comp_l0=_np(comp[0]), comp_l1=_np(comp[1]), rel_l0=_np(rel[0]), rel_l1=_np(rel[1]),
                  emb_align=_np(ga), emb_comp=_np(gc))
    model.train()
    B, K = args.completion_batch_size, args.num_negative
    trip = torch.from_numpy(tr[:B].astype(np.int64))
    neg = torch.from_numpy(rng.integers(0, n, (B, K)).astype(np.int64))
    data = {"batch_h": trip[:, 0].repeat(K + 1), "batch_r": trip[:, 1].repeat(K + 1),
            "batch_t": torch.cat((trip[:, 2], neg.view(-1)))}
    links = np.stack([rng.permutation(n // 2)[:20], n // 2 + rng.permutation(n // 2)[:20]], 1)
    feed = {"links": links,
            "neg_left": np.repeat(links[:, 0], K), "neg_right": torch.from_numpy(rng.integers(0, n, 20 * K)),
            "neg2_left": torch.from_numpy(rng.integers(0, n, 20 * K)), "neg2_right": np.repeat(links[:, 1], K)}
    bn_before = {k: _np(v) for k, v in model.state_dict().items() if "running" in k}
    model.zero_grad()
    closs = model.completion_loss(data, eit, ett, feed)
    closs.backward()
    arrays.update(batch_h=_np(data["batch_h"]), batch_r=_np(data["batch_r"]), batch_t=_np(data["batch_t"]), links=links,
                  neg_left=feed["neg_left"], neg_right=_np(feed["neg_right"]), neg2_left=_np(feed["neg2_left"]),
                  neg2_right=feed["neg2_right"], completion_loss=np.float64(closs.item()),
                  closs_grad_ent=_np(model.ent_completion_att.grad), closs_grad_rel=_np(model.rel_completion_att.grad),
                  closs_grad_w_att=_np(model.conv1_completion.w_att.grad))
    for k, v in bn_before.items():
        arrays["bn_before." + k] = v
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in bn_before.items()}})
    model.zero_grad()
    aloss = model.alignment_loss(feed, eit, ett)
    aloss.backward()
    arrays.update(alignment_loss=np.float64(aloss.item()), aloss_grad_ent=_np(model.ent_completion_att.grad),
                  aloss_grad_all_linear=_np(model.all_linear_comp.grad), aloss_grad_rel_info=_np(model.rel_info_att.grad))
    _save("model_dbpv1", **arrays)


def gen_dbpv1_scoring():
    """Row a18: the DBPv1 scoring call sites on ONE embedding table -- get_neg(ILL, output_layer, k)
    (JMAC_DBPv1/modules/utils/util.py:35-58: top-k over ALL entities, own KG and self included) and
    Trainer.compute_alignment_quality(embedding, list1, list2) (JMAC_DBPv1/trainer/jmac_trainer.py:281-300: entropy +
    the two [T,T] softmax matrices).  Both are called as the trainer calls them (:201, :236-237): on the L2-normalised
    output of get_emb.  Rows are well separated (random unit vectors, d=48), so the top-k order is far from ties."""
    from modules.utils.util import get_neg
    from trainer.jmac_trainer import Trainer
    n, d, k, nl = 400, 48, 10, 48
    # seeds of the two KGs live in the two halves of the merged id space (kgs.train_links, jmac_trainer.py:230-237).
    # The fixture is "bit-exact index" material only if every seed row's k+1 leading similarities are separated by more
    # than the rounding of a differently ordered fp32 accumulation (~1e-7 on unit vectors): take the first generator
    # seed whose smallest such gap exceeds 1e-5.
    for seed in range(77, 1077):
        rng = np.random.default_rng(seed)
        emb = rng.standard_normal((n, d)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        links = np.stack([rng.permutation(n // 2)[:nl], n // 2 + rng.permutation(n // 2)[:nl]], 1).astype(np.int64)
        e = torch.from_numpy(emb)
        gap = np.inf
        for col in (0, 1):
            srt = torch.sort(e[links[:, col]] @ e.t(), dim=1, descending=True)[0]
            gap = min(gap, float((srt[:, :k] - srt[:, 1:k + 1]).min()))
        if gap > 1e-5:
            break
    print("scoring_dbpv1: generator seed %d, smallest top-(k+1) gap %.3e" % (seed, gap))
    neg2_left = get_neg(links[:, 1], e, k)           # :236
    neg_right = get_neg(links[:, 0], e, k)           # :237
    list1 = rng.permutation(n // 2)[:90].tolist()    # valid + test entities of KG1 / KG2 (:201)
    list2 = (n // 2 + rng.permutation(n // 2)[:90]).tolist()
    entropy, s1, s2 = Trainer.compute_alignment_quality(None, e, list1, list2)
    _save("scoring_dbpv1", n=n, d=d, k=k, gen_seed=seed, emb=emb, links=links, neg2_left=_np(neg2_left), neg_right=_np(neg_right),
          min_topk_gap=np.float64(gap), list1=np.asarray(list1), list2=np.asarray(list2), entropy=np.float64(entropy.item()),
          softmax_simi=_np(s1), softmax_simi2=_np(s2))


def gen_dbpv1():
    from models.jmac_model import RelationalAwareLayer
    args = types.SimpleNamespace(leaky_relu_w=0.05, opn="sub")
    mk = lambda d: RelationalAwareLayer(d, d, 14, rel_dim=d, act=torch.tanh, args=args)
    rng = np.random.default_rng(21)
    print("layer fixture (reference: JMAC_DBPv1/models/jmac_model.py RelationalAwareLayer)")
    ei, et = random_graph(rng, 180, 14, 800)
    _save("layer_dbpv1", **run_layer_case(RelationalAwareLayer, mk, "dbpv1", 180, 14, 40, ei, et, 8))
    gen_dbpv1_model()
    gen_dbpv1_scoring()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="all", choices=["all", "root", "jafull", "dbpv1", "aligneval", "entr", "dataset", "realpair", "realall", "e2e"])
    a = ap.parse_args()
    if a.variant == "all":
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for v in ("root", "jafull", "dbpv1", "aligneval", "entr", "dataset", "realpair", "realall", "e2e"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--variant", v], env=env)
    else:
        _paths(a.variant)
        {"root": gen_root, "jafull": gen_jafull, "dbpv1": gen_dbpv1, "aligneval": gen_aligneval, "entr": gen_entr, "dataset": gen_dataset, "realpair": gen_realpair, "realall": gen_realall,
         "e2e": gen_e2e}[a.variant]()
