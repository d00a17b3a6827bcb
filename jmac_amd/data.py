"""DBP-5L on-disk format -> arrays ("graph ingest / formats").

Host-side mirror of ``ParseData.create_KG_objects_and_alignment`` / ``load_kg_data`` /
``load_all_to_all_seed_align_links`` (src/data_loader.py:137-221), ``get_kg_edges_for_each``
(src/utils.py:112-149) and the ``KnowledgeGraph`` container's derived fields (src/knowledgegraph.py:8-86):

    <root>/entity/<lang>.tsv                       one entity per line (only the COUNT is used, :205-209)
    <root>/kg/<lang>-{train,val,test}.tsv          ``h \\t r \\t t`` integer triples, ids local to the KG
    <root>/seed_{train,test}_pairs/<l1>-<l2>.tsv   aligned pairs, ids written as floats (``928.0 \\t 912.0``)
    <root>/relations.txt                           one relation per line; num_relation = lines + 1 (:211-212)

Everything here is numpy on the host (the reference does this once at start-up); the arrays feed
``jmac_amd.graph.RelGraph`` / ``jmac_amd.entr.align_data_processing`` on the device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


def _read_ints(path: str, cols: int) -> np.ndarray:
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([int(float(x)) for x in line.split("\t")[:cols]])    # seed files hold '928.0'
    return np.asarray(rows, dtype=np.int64).reshape(-1, cols)


def get_language_list(data_dir: str) -> List[str]:
    """src/utils.py:95-107: sorted first two characters of entity/*.tsv."""
    files = sorted(f for f in os.listdir(os.path.join(data_dir, "entity")) if f[-3:] == "tsv")
    return [f[:2] for f in files]


def get_kg_edges_for_each(kg_dir: str, language: str, is_target_KG: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """src/utils.py:112-149: bidirectional edge list (the reverse edge reuses the relation id); supporter KGs add
    their validation triples.  Row 0 = senders, row 1 = receivers, as the reference stacks them."""
    tr = _read_ints(os.path.join(kg_dir, language + "-train.tsv"), 3)
    send = [tr[:, 0], tr[:, 2]]
    recv = [tr[:, 2], tr[:, 0]]
    typ = [tr[:, 1], tr[:, 1]]
    if not is_target_KG:
        va = _read_ints(os.path.join(kg_dir, language + "-val.tsv"), 3)
        send += [va[:, 0], va[:, 2]]
        recv += [va[:, 2], va[:, 0]]
        typ += [va[:, 1], va[:, 1]]
    return np.vstack((np.concatenate(send), np.concatenate(recv))), np.concatenate(typ)


def edges_from_triples(triples: np.ndarray, bidirectional: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """(edge_index [2,E] int64, edge_type [E] int64) of a triple list: the train-mode graph of align_data_processing
    (train.py:116-135: row 0 = heads = aggregation destinations, row 1 = tails), or the loader's bidirectional form
    (src/utils.py:127-149: every triple in both directions, the reverse edge reusing the relation id)."""
    t = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    if not bidirectional:
        return np.ascontiguousarray(t[:, [0, 2]].T), np.ascontiguousarray(t[:, 1])
    return (np.vstack((np.concatenate((t[:, 0], t[:, 2])), np.concatenate((t[:, 2], t[:, 0])))),
            np.concatenate((t[:, 1], t[:, 1])))


def load_dbp5l_arrays(path: str) -> Dict[str, np.ndarray]:
    """The DBP-5L triples / seed pairs kept as integer arrays in one .npz (``<lang>.train|val|test`` [T,3],
    ``<lang>.num_entity``, ``n_relation_lines``, ``seed_{train,test}_pairs``): the same content as the on-disk
    directory ``load_dbp5l`` reads, without the text files."""
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def true_tail_dict(triples: np.ndarray) -> Dict[Tuple[int, int], np.ndarray]:
    """get_true_tail, src/knowledgegraph.py:62-86: {(h, r): distinct tails} (the evaluator's filter)."""
    t = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    if not len(t):
        return {}
    order = np.lexsort((t[:, 2], t[:, 1], t[:, 0]))
    t = np.unique(t[order], axis=0)
    key_change = np.flatnonzero(np.any(t[1:, :2] != t[:-1, :2], axis=1)) + 1
    starts = np.concatenate([[0], key_change])
    ends = np.concatenate([key_change, [len(t)]])
    return {(int(t[s, 0]), int(t[s, 1])): t[s:e, 2].copy() for s, e in zip(starts, ends)}


@dataclass
class KnowledgeGraph:
    """Fields of src/knowledgegraph.py:8-60 that the hot path and its callers read."""
    lang: str
    train_data: np.ndarray            # supporter KGs: train + val (knowledgegraph.py:18-19)
    val_data: np.ndarray
    test_data: np.ndarray
    num_entity: int
    num_relation: int
    is_supporter_kg: bool
    entity_id_base: int
    relation_id_base: int
    upper_entity_base: int = 0
    upper_relation_base: int = 0
    edge_index: np.ndarray = None     # bidirectional loader graph (src/utils.py:112-149)
    edge_type: np.ndarray = None
    true_tail: Dict = field(default_factory=dict)
    transferred_triples: list = field(default_factory=list)
    triple_keys: np.ndarray = None


def load_dbp5l(data_path: str, target_language: str):
    """(kg_object_dict, seeds_train, seeds_test, num_entities) as ParseData.create_KG_objects_and_alignment builds them."""
    names = get_language_list(data_path)
    rel_lines = [l for l in open(os.path.join(data_path, "relations.txt"))]
    num_rel = len(rel_lines) + 1
    kgs: Dict[str, KnowledgeGraph] = {}
    ebase = rbase = 0
    for lang in names:
        tr = _read_ints(os.path.join(data_path, "kg", lang + "-train.tsv"), 3)
        va = _read_ints(os.path.join(data_path, "kg", lang + "-val.tsv"), 3)
        te = _read_ints(os.path.join(data_path, "kg", lang + "-test.tsv"), 3)
        n_ent = sum(1 for _ in open(os.path.join(data_path, "entity", lang + ".tsv")))
        sup = lang != target_language
        kg = KnowledgeGraph(lang, np.concatenate((tr, va)) if sup else tr, va, te, n_ent, num_rel, sup, ebase, rbase)
        if not sup:
            kg.true_tail = true_tail_dict(np.concatenate((tr, va, te), axis=0))      # knowledgegraph.py:45-46
        ebase += n_ent
        rbase += num_rel
        kg.upper_entity_base, kg.upper_relation_base = ebase, rbase
        kg.edge_index, kg.edge_type = get_kg_edges_for_each(os.path.join(data_path, "kg"), lang, not sup)
        kgs[lang] = kg

    def seeds(sub):
        out = {}
        d = os.path.join(data_path, sub)
        for f in os.listdir(d):
            out[(f[0:2], f[3:5])] = _read_ints(os.path.join(d, f), 2)
        return out
    return kgs, seeds("seed_train_pairs"), seeds("seed_test_pairs"), ebase


def kgs_from_arrays(z: Dict[str, np.ndarray], target_language: str):
    """``load_dbp5l`` on the integer-array form of the dataset (``load_dbp5l_arrays``: tests/golden/dbp5l_all_data.npz holds all
    five KGs and the ten seed-pair files, dbp5l_ja_el_data.npz the el / ja pair): the same KnowledgeGraph objects, id bases
    (src/data_loader.py:162-181) and seed dictionaries, without the text files."""
    names = sorted(str(x) for x in z["langs"])
    num_rel = int(z["n_relation_lines"]) + 1
    kgs: Dict[str, KnowledgeGraph] = {}
    ebase = rbase = 0
    for lang in names:
        tr, va, te = (np.asarray(z["%s.%s" % (lang, part)], dtype=np.int64).reshape(-1, 3) for part in ("train", "val", "test"))
        n_ent = int(z[lang + ".num_entity"])
        sup = lang != target_language
        kg = KnowledgeGraph(lang, np.concatenate((tr, va)) if sup else tr, va, te, n_ent, num_rel, sup, ebase, rbase)
        if not sup:
            kg.true_tail = true_tail_dict(np.concatenate((tr, va, te), axis=0))
        ebase += n_ent
        rbase += num_rel
        kg.upper_entity_base, kg.upper_relation_base = ebase, rbase
        send, recv, typ = [tr[:, 0], tr[:, 2]], [tr[:, 2], tr[:, 0]], [tr[:, 1], tr[:, 1]]     # src/utils.py:127-149
        if sup:
            send += [va[:, 0], va[:, 2]]
            recv += [va[:, 2], va[:, 0]]
            typ += [va[:, 1], va[:, 1]]
        kg.edge_index, kg.edge_type = np.vstack((np.concatenate(send), np.concatenate(recv))), np.concatenate(typ)
        kgs[lang] = kg
    seeds = {"seed_train_pairs": {}, "seed_test_pairs": {}}
    if "seed_pairs" in z:
        for pr in [str(x) for x in z["seed_pairs"]]:
            for sub in seeds:
                seeds[sub][(pr[0:2], pr[3:5])] = np.asarray(z["%s.%s" % (sub, pr)], dtype=np.int64).reshape(-1, 2)
    else:
        pr = tuple(str(x) for x in z["seed_pair"])
        for sub in seeds:
            seeds[sub][pr] = np.asarray(z[sub], dtype=np.int64).reshape(-1, 2)
    return kgs, seeds["seed_train_pairs"], seeds["seed_test_pairs"], ebase


def union_edges(kgs: Dict[str, KnowledgeGraph], bidirectional: bool = False):
    """BASELINE config 3: the block-diagonal union of the KGs in the model's id spaces -- entity ids offset by
    ``entity_id_base``, relation ids by ``relation_id_base`` (src/data_loader.py:162-181) -- as ONE typed edge list.
    ``bidirectional=False``: the train-mode graph of every KG (train.py:116-135 on ``train_data``: supporters train + val,
    the target train); ``True``: the loader's bidirectional graphs (src/utils.py:112-149).
    Returns (edge_index [2,E], edge_type [E], N, nr, ent_bases, rel_bases) with ``ent_bases[k] .. ent_bases[k+1]`` the rows of
    the k-th KG in sorted-name order."""
    eis, ets, ent_bases, rel_bases = [], [], [0], [0]
    for lang in sorted(kgs):
        kg = kgs[lang]
        assert kg.entity_id_base == ent_bases[-1] and kg.relation_id_base == rel_bases[-1]
        ei, et = (kg.edge_index, kg.edge_type) if bidirectional else edges_from_triples(kg.train_data, False)
        eis.append(np.asarray(ei, dtype=np.int64) + kg.entity_id_base)
        ets.append(np.asarray(et, dtype=np.int64) + kg.relation_id_base)
        ent_bases.append(kg.upper_entity_base)
        rel_bases.append(kg.upper_relation_base)
    return np.concatenate(eis, axis=1), np.concatenate(ets), ent_bases[-1], rel_bases[-1], ent_bases, rel_bases
