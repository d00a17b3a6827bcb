"""Minimal training / evaluation harness that drives the hot path the way the reference's train.py does.

Not a re-implementation of train.py (argparse, Logger, DataLoader worker processes and the fastText name
embeddings are out of scope): just enough host code to run, on the HIP device,

    train_completion_component   train.py:328-364    batches sub.repeat(K+1) / rel.repeat(K+1) / cat(obj, neg)
    train_alignment_component    train.py:367-378    one full-batch alignment step
    CompletionEvaluator.test     src/validate.py:22-80   filtered ranking, Hits@1 / Hits@10 / MRR
    one epoch over the KG pairs  train.py:423-512    get_emb -> EnTr -> completion -> alignment

end to end, so the integration of data.py / graph.py / model.py / losses.py / scoring.py / entr.py is tested as a
whole (tests/test_gpu_harness.py).  Negatives are drawn uniformly on the device and tails that are true for (h, r)
are redrawn once -- the reference's per-triple numpy mask (modules/load/data_loader.py:35-46) runs in DataLoader
workers and is not part of the hot path.
"""
from __future__ import annotations

import types
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import entr, scoring
from .data import KnowledgeGraph
from .model import JMAC


def make_args(dim=300, batch_size=1000, num_negative=25, device="cuda", **kw):
    a = dict(dim=dim, dropout=0.4, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=num_negative,
             margin_align=1.0, margin_completion=5.0, batch_size=batch_size, no_name_info=False, device=device,
             pair_sample_weight=0.2, lr=1e-3)                     # train.py:57-102 defaults for the fields used here
    a.update(kw)
    return types.SimpleNamespace(**a)


def process_input_data(kg: KnowledgeGraph, device):
    """train.py:214-228: first visit uses the loader's bidirectional graph, later visits the transferred triples."""
    if not len(kg.transferred_triples):
        triples = kg.train_data
        kg.triple_keys = entr.encode_triples(triples)
        ei, et = torch.from_numpy(kg.edge_index).to(device), torch.from_numpy(kg.edge_type).to(device)
    else:
        triples = kg.transferred_triples
        ei, et = entr.align_data_processing(triples, device)
    return ei, et, [kg.entity_id_base, kg.upper_entity_base], [kg.relation_id_base, kg.upper_relation_base], triples


def completion_batches(triples: np.ndarray, num_ent: int, batch_size: int, k: int, device, generator=None):
    """(triple [B,3], neg [B,k]) batches of full size only (train.py:343-346 skips ragged last batches)."""
    t = torch.from_numpy(np.asarray(triples, dtype=np.int64)).to(device)
    perm = torch.randperm(len(t), device=device, generator=generator)
    for s in range(0, len(t) - batch_size + 1, batch_size):
        tr = t[perm[s:s + batch_size]]
        neg = torch.randint(0, num_ent, (batch_size, k), device=device, generator=generator)
        clash = neg == tr[:, 2:3]
        neg = torch.where(clash, torch.randint(0, num_ent, neg.shape, device=device, generator=generator), neg)
        yield tr, neg


def train_completion_component(model: JMAC, opt, ei1, et1, ei2, et2, feeddict, triples1, triples2, n1, n2, args, generator=None):
    losses = []
    for triples, n_ent, source in ((triples1, n1, True), (triples2, n2, False)):
        for tr, neg in completion_batches(triples, n_ent, args.batch_size, args.num_negative, ei1.device, generator):
            opt.zero_grad(set_to_none=True)
            sub, rel, obj = tr[:, 0], tr[:, 1], tr[:, 2]
            k = args.num_negative
            data = {"batch_h": sub.repeat(k + 1), "batch_r": rel.repeat(k + 1), "batch_t": torch.cat((obj, neg.view(-1)))}
            loss = model.completion_loss(data, ei1, et1, ei2, et2, feeddict, source)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
    return float(torch.stack(losses).mean()) if losses else float("nan")


def train_alignment_component(model: JMAC, opt, ei1, et1, ei2, et2, feeddict):
    if not len(feeddict["links"]):
        return 0.0
    opt.zero_grad(set_to_none=True)
    loss = model.alignment_loss(feeddict, ei1, et1, ei2, et2)
    loss.backward()
    opt.step()
    return float(loss.detach())


@torch.no_grad()
def evaluate_completion(model: JMAC, kg: KnowledgeGraph, ei, et, args, split="val", filtered=True, fused=True, eval_batch=None):
    """CompletionEvaluator.test (src/validate.py:22-80) with the encoder run once instead of once per batch.
    The reference scores 1 000 queries at a time (``args.batch_size``) because it materialises their [B, N] distance matrix.  The
    fused path has no matrix and a query's rank does not depend on what else is in its call, so it takes ``eval_batch`` queries
    per call -- default: up to 16 384, i.e. a whole DBP-5L split at once (the per-call query preparation and the last partial
    round of tiles are paid once); the materialised path keeps the reference's batches."""
    model.eval()
    data = {"val": kg.val_data, "test": kg.test_data, "train": kg.train_data}[split]
    eb, rb = [kg.entity_id_base, kg.upper_entity_base], [kg.relation_id_base, kg.upper_relation_base]
    cached = model.forward_base(ei, et, eb, rb)
    ranks = []
    step = int(eval_batch or (16384 if fused else args.batch_size))
    for s in range(0, len(data), step):
        b = data[s:s + step]
        h, r, t = b[:, 0].tolist(), b[:, 1].tolist(), b[:, 2].tolist()
        fp = fi = None
        if filtered:
            fp, fi = scoring.build_filter_csr(h, r, kg.true_tail, ei.device)
        if fused:       # ranks without the [B, N] matrix
            ranks.append(model.linkpred_ranks(h, r, t, ei, et, eb, rb, fp, fi, cached=cached))
        else:           # the reference's two steps: forward_linkpred, then the ranking loop
            dist = model.forward_linkpred(h, r, ei, et, range(kg.num_entity), eb, rb, cached=cached)
            ranks.append(scoring.filtered_rank(dist, torch.as_tensor(t, dtype=torch.int32, device=dist.device), fp, fi))
    rk = torch.cat(ranks).double()
    model.train()
    return float((rk <= 1).double().mean()), float((rk <= 10).double().mean()), float((1.0 / rk).mean())


def train_epoch(model: JMAC, kgs: Dict[str, KnowledgeGraph], seeds_train: Dict[Tuple[str, str], np.ndarray],
                seeds_test: Dict[Tuple[str, str], np.ndarray], opt_c, opt_a, args, state: dict, refresh: bool,
                generator=None) -> List[dict]:
    """One pass over the KG pairs (train.py:426-496).  ``state`` keeps the per-pair feeddicts / graphs / entropies."""
    dev = torch.device(args.device)
    log = []
    for idx, ((l1, l2), links) in enumerate(sorted(seeds_train.items())):
        kg1, kg2 = kgs[l1], kgs[l2]
        ei1, et1, eb1, rb1, tr1 = process_input_data(kg1, dev)
        ei2, et2, eb2, rb2, tr2 = process_input_data(kg2, dev)
        st = state.setdefault(idx, {"entropy": [-1], "seeds": [links]})
        if refresh or "feeddict" not in st:
            model.eval()
            with torch.no_grad():                      # train.py:450-451: get_emb once per KG of the pair -- one encoder pass here
                (a1, _), (a2, _) = model.get_emb_blocks([(ei1, et1, eb1, rb1), (ei2, et2, eb2, rb2)])
                o1, o2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
            model.train()
            test_pairs = seeds_test.get((l1, l2), links)
            new1, new2, k1, k2, feed, _ = entr.seed_enlargement_triple_transferring(
                o1, o2, test_pairs[:, 0].tolist(), test_pairs[:, 1].tolist(), st["entropy"], 0, st["seeds"][0], tr1, tr2,
                st["seeds"], eb1, rb1, eb2, rb2, kg1, kg2, args, generator=generator)
            kg1.triple_keys, kg2.triple_keys = k1, k2
            kg1.transferred_triples, kg2.transferred_triples = new1, new2
            st["feeddict"] = feed
            st["g1"] = entr.align_data_processing(new1, dev)           # train-mode graph: head <- tail, one direction
            st["g2"] = entr.align_data_processing(new2, dev)
            st["tr"] = (new1, new2)
        (ei1, et1), (ei2, et2) = st["g1"], st["g2"]
        closs = train_completion_component(model, opt_c, ei1, et1, ei2, et2, st["feeddict"], st["tr"][0], st["tr"][1],
                                           kg1.num_entity, kg2.num_entity, args, generator)
        aloss = train_alignment_component(model, opt_a, ei1, et1, ei2, et2, st["feeddict"])
        log.append({"pair": (l1, l2), "completion_loss": closs, "align_loss": aloss, "links": len(st["feeddict"]["links"]),
                    "triples": (len(st["tr"][0]), len(st["tr"][1])), "entropy": st["entropy"][0]})
    return log
