"""EnTr bookkeeping: alignment-seed enlargement and triple transfer.

Host-side mirror of ``seed_enlargement_triple_transferring`` (train.py:138-211), ``transfer_knowledge``
(train.py:297-325) and ``align_data_processing`` (train.py:116-135).  The similarity work runs on the HIP kernels
(``scoring.alignment_quality`` = compute_alignment_quality, ``scoring.get_neg``); the python dict / set / string-key
loops of the reference become array operations with the same results IN THE SAME ORDER, including the reference's
quirks:

* ``links.get(head) and links.get(tail)`` tests the MAPPED id for truthiness, so a triple whose head or tail maps
  to entity 0 is never transferred (train.py:309,317);
* ``{ele[0]: ele[1] for ele in pairs}``: a source listed twice keeps its LAST target; the inverse dict is built by
  iterating that dict (first-occurrence key order), again last writer wins (train.py:202,301);
* a transferred triple is appended only the first time its key is new to the target KG's key set (train.py:310-313).

Triple keys are int64 codes ``h << 42 | r << 21 | t`` (ids < 2^21) instead of ``'h_r_t'`` strings;
``keys_from_strings`` / ``keys_to_strings`` convert for interop with reference-side state.
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import numpy as np
import torch

from . import scoring

_SHIFT = 21
_MAXID = (1 << _SHIFT) - 1


def _tri(a) -> np.ndarray:
    a = np.asarray(a, dtype=np.int64)
    return a.reshape(-1, 3)


def encode_triples(triples) -> np.ndarray:
    t = _tri(triples)
    if t.size and (t.min() < 0 or t.max() > _MAXID):
        raise ValueError("triple ids must lie in [0, 2^21)")
    return (t[:, 0] << (2 * _SHIFT)) | (t[:, 1] << _SHIFT) | t[:, 2]


def keys_from_strings(keys: Iterable[str]) -> np.ndarray:
    """A reference-side ``kg.triple_keys`` set of 'h_r_t' strings -> int64 codes."""
    rows = [tuple(int(v) for v in k.split("_")) for k in keys]
    return np.unique(encode_triples(np.array(rows, dtype=np.int64).reshape(-1, 3)))


def keys_to_strings(codes: np.ndarray) -> set:
    c = np.asarray(codes, dtype=np.int64)
    return set("%d_%d_%d" % (h, r, t) for h, r, t in zip(c >> (2 * _SHIFT), (c >> _SHIFT) & _MAXID, c & _MAXID))


def _as_codes(keys) -> np.ndarray:
    if isinstance(keys, np.ndarray):
        return keys.astype(np.int64, copy=False)
    if isinstance(keys, (set, frozenset, list, tuple)) and (len(keys) == 0 or isinstance(next(iter(keys)), str)):
        return keys_from_strings(keys) if len(keys) else np.zeros(0, dtype=np.int64)
    return np.asarray(list(keys), dtype=np.int64)


def link_maps(links, n_src: int = 0, n_dst: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(fwd, inv) lookup tables of the reference's ``links`` / ``inverse_links`` dicts; -1 = no entry."""
    lk = np.asarray(links, dtype=np.int64).reshape(-1, 2)
    n_src = max(int(n_src), int(lk[:, 0].max()) + 1 if lk.size else 0)
    n_dst = max(int(n_dst), int(lk[:, 1].max()) + 1 if lk.size else 0)
    fwd = np.full(n_src, -1, dtype=np.int64)
    inv = np.full(n_dst, -1, dtype=np.int64)
    if lk.size:
        fwd[lk[:, 0]] = lk[:, 1]                              # repeated source: the last pair wins (dict semantics)
        _, first = np.unique(lk[:, 0], return_index=True)
        keys = lk[np.sort(first), 0]                          # dict iteration order = first occurrence of each key
        inv[fwd[keys]] = keys                                 # repeated target: the later key wins
    return fwd, inv


def _transfer_one(triples: np.ndarray, mapping: np.ndarray, target_keys: np.ndarray) -> np.ndarray:
    if not len(triples) or not len(mapping):
        return np.zeros((0, 3), dtype=np.int64)
    h, r, t = triples[:, 0], triples[:, 1], triples[:, 2]
    inb = (h < len(mapping)) & (t < len(mapping))
    mh = np.where(inb, mapping[np.minimum(h, len(mapping) - 1)], -1)
    mt = np.where(inb, mapping[np.minimum(t, len(mapping) - 1)], -1)
    ok = (mh > 0) & (mt > 0)                                  # `.get()` truthiness: missing (None) and id 0 both fail
    cand = np.stack([mh[ok], r[ok], mt[ok]], 1)
    codes = encode_triples(cand)
    fresh = ~np.isin(codes, target_keys)
    cand, codes = cand[fresh], codes[fresh]
    _, first = np.unique(codes, return_index=True)            # keep the first occurrence, in list order
    return cand[np.sort(first)]


def transfer_knowledge(triple_list_src, triple_list_dst, links, triple_keys1, triple_keys2, n_src: int = 0, n_dst: int = 0):
    """train.py:297-325.  ``links``: [L,2] (source id, target id) pairs in the order the reference builds its dict from.
    Returns (triples_src + transferred-in, triples_dst + transferred-in, keys1, keys2) as int64 arrays."""
    src, dst = _tri(triple_list_src), _tri(triple_list_dst)
    k1, k2 = _as_codes(triple_keys1), _as_codes(triple_keys2)
    fwd, inv = link_maps(links, n_src, n_dst)
    add_dst = _transfer_one(src, fwd, k2)                     # source facts re-expressed in the target KG
    add_src = _transfer_one(dst, inv, k1)
    k1 = np.concatenate([k1, encode_triples(add_src)])
    k2 = np.concatenate([k2, encode_triples(add_dst)])
    return np.concatenate([src, add_src]), np.concatenate([dst, add_dst]), k1, k2


def align_data_processing(triple_list, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """train.py:116-135: edge_index [2,E] = (head, tail) -- head is the aggregation destination -- and edge_type [E]."""
    t = _tri(triple_list)
    ei = torch.from_numpy(np.ascontiguousarray(t[:, [0, 2]].T)).to(device)
    return ei, torch.from_numpy(np.ascontiguousarray(t[:, 1])).to(device)


def seed_enlargement_triple_transferring(output1, output2, align_test_src, align_test_dst, global_entropies, seed_index,
                                         train_align_pairs, triples1, triples2, global_seeds, ent_bases1, rel_bases1,
                                         ent_bases2, rel_bases2, kg1, kg2, args, generator=None):
    """train.py:138-211 with the same arguments and return tuple.  ``output1/2``: the L2-normalised alignment
    embeddings of ``get_emb`` ON THE HIP DEVICE; ``kg1/kg2`` need a ``triple_keys`` attribute (int64 codes, or the
    reference's set of strings).  ``generator``: optional torch.Generator for the multinomial draw (train.py:166)."""
    entropy_t, simi, _ = scoring.alignment_quality(output1, output2, align_test_src, align_test_dst)
    entropy = float(entropy_t)
    prev = global_entropies[seed_index]
    additional = np.zeros((0, 2), dtype=np.int64)
    if prev == -1:
        global_entropies[seed_index] = entropy
    else:
        sample_percent = (prev - entropy) / prev * args.pair_sample_weight
        if sample_percent < 0:
            sample_percent = 0
            global_entropies[seed_index] = entropy
        num_pairs = int(sample_percent * len(align_test_src))
        if num_pairs > 0:
            max_values = simi.max(dim=1)[0]
            src_nodes = max_values.multinomial(num_pairs, replacement=False, generator=generator)
            dst_nodes = simi.index_select(0, src_nodes).max(dim=1)[1]
            additional = torch.stack([src_nodes, dst_nodes], 1).cpu().numpy().astype(np.int64)
    pairs = np.asarray(train_align_pairs, dtype=np.int64).reshape(-1, 2)
    if len(additional):
        pairs = additional if not len(pairs) else np.concatenate([pairs, additional], axis=0)
        global_seeds[seed_index] = pairs
    feeddict = {"neg_left": [], "neg_right": [], "neg2_left": [], "neg2_right": [], "links": [], "ent_bases1": ent_bases1,
                "ent_bases2": ent_bases2, "rel_bases1": rel_bases1, "rel_bases2": rel_bases2}
    if len(pairs):
        k = args.num_negative
        feeddict.update(
            neg2_left=scoring.get_neg(pairs[:, 1].tolist(), output2, output1, k),       # train.py:183
            neg_right=scoring.get_neg(pairs[:, 0].tolist(), output1, output2, k),       # :184
            neg_left=np.repeat(pairs[:, 0], k).astype(np.float64),                      # :189-193 (float arrays)
            neg2_right=np.repeat(pairs[:, 1], k).astype(np.float64),
            links=pairs)
    new1, new2, k1, k2 = transfer_knowledge(triples1, triples2, pairs, kg1.triple_keys, kg2.triple_keys,
                                            output1.shape[0], output2.shape[0])
    return new1, new2, k1, k2, feeddict, global_seeds
