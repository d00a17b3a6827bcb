"""Seeded synthetic knowledge graphs of the shapes BASELINE.json names (no datasets travel to the GPU box).

Shapes follow SURVEY.md section 8d (measured on datasetdbp5l/): ``ja`` has N=11 805 entities, 961 relation
rows, 17 979 train triples; in the train-mode graph (train.py:130-132: destination = head, source =
tail) 54 % of the nodes have no in-edge, segment length max 28, mean 1.52.
"""
from __future__ import annotations

import numpy as np

DBP5L = {  # lang: (N, train triples, distinct relation ids used ~)
    "el": (5231, 8670), "en": (13996, 48652), "es": (12381, 33036), "fr": (13176, 30139), "ja": (11805, 17979),
}
NUM_REL = 961     # relations.txt lines + 1 (src/data_loader.py:214-215)


def dbp5l_like(lang: str = "ja", seed: int = 1234, bidirectional: bool = False, num_triples: int = None):
    """(edge_index [2,E] int64, edge_type [E] int64, N, nr) with the degree profile of a DBP-5L train graph."""
    n, e = DBP5L[lang]
    if num_triples is not None:
        e = int(num_triples)
    rng = np.random.default_rng(seed)
    # heads: 46 % of the nodes are non-empty destinations; segment lengths geometric-ish, capped at 28
    heads = rng.permutation(n)[: int(round(n * 0.46))]
    w = rng.geometric(0.38, size=heads.size).astype(np.float64)
    w = np.minimum(w, 28.0)
    dst = rng.choice(heads, size=e, p=w / w.sum())
    src = rng.integers(0, n, size=e)
    rel_ids = rng.permutation(NUM_REL - 1)[:400]                       # ~160-833 distinct ids per KG
    zipf = 1.0 / np.arange(1, rel_ids.size + 1)
    typ = rel_ids[rng.choice(rel_ids.size, size=e, p=zipf / zipf.sum())]
    if bidirectional:                                                  # src/utils.py:127-149
        dst, src, typ = np.concatenate([dst, src]), np.concatenate([src, dst]), np.concatenate([typ, typ])
    return np.stack([dst, src]).astype(np.int64), typ.astype(np.int64), n, NUM_REL


# train + validation triples: what a SUPPORTER KG trains on (src/knowledgegraph.py:18-19; SURVEY 8d shape table)
DBP5L_TRAIN_VAL = {"el": 12822, "en": 72703, "es": 49256, "fr": 44844, "ja": 26612}


def dbp5l_union(seed: int = 1234, bidirectional: bool = False, target: str = None):
    """Config 3: block-diagonal union of the five DBP-5L-shaped KGs with the reference's id offsets
    (entity_id_base / relation_id_base, src/data_loader.py:162-181): N = 56 589, 5 x 961 relation rows (+1 loop row
    added by the layer).  ``target``: that KG carries its train triples, the other four train + validation (the edge
    counts the reference trains with: E = 197 604 for target 'ja'); None: train triples everywhere (E = 138 476).
    Returns (edge_index, edge_type, N, nr, ent_bases, rel_bases)."""
    eis, ets, ent_bases, rel_bases = [], [], [0], [0]
    for k, lang in enumerate(("el", "en", "es", "fr", "ja")):
        e_lang = DBP5L_TRAIN_VAL[lang] if (target is not None and lang != target) else None
        ei, et, n, nr = dbp5l_like(lang, seed + k, bidirectional, e_lang)
        eis.append(ei + ent_bases[-1])
        ets.append(et + rel_bases[-1])
        ent_bases.append(ent_bases[-1] + n)
        rel_bases.append(rel_bases[-1] + nr)
    return np.concatenate(eis, axis=1), np.concatenate(ets), ent_bases[-1], rel_bases[-1], ent_bases, rel_bases


def power_law_graph(n: int, e: int, nr: int, seed: int = 1234, alpha: float = 2.1, max_deg: int = 100_000):
    """Config 4: destinations Zipf-like (exponent ~alpha, truncated), sources uniform, types Zipf over nr."""
    rng = np.random.default_rng(seed)
    # per-node weight ~ Pareto so that in-degree follows a power law with hubs
    w = (1.0 - rng.random(n)) ** (-1.0 / (alpha - 1.0))
    w = np.minimum(w, float(max_deg))
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    dst = np.searchsorted(cdf, rng.random(e), side="right").astype(np.int64)
    np.clip(dst, 0, n - 1, out=dst)
    src = rng.integers(0, n, size=e, dtype=np.int64)
    zipf = 1.0 / np.arange(1, nr + 1)
    tcdf = np.cumsum(zipf)
    tcdf /= tcdf[-1]
    typ = np.searchsorted(tcdf, rng.random(e), side="right").astype(np.int64)
    np.clip(typ, 0, nr - 1, out=typ)
    return np.stack([dst, src]), typ, n, nr + 1


def fwd_algorithmic_bytes(n: int, e: int, d: int, elem: int = 4) -> int:
    """E*(2*d*s + 8) + N*(2*d*s + 12): per edge the [Q|Z] row + col + type; per node
    the P row in, the output row out, rowptr + max/den.  Relation rows (L2-resident) are excluded.
    With bf16 tables (elem=2) the output row stays fp32: N*(d*2 + d*4 + 12)."""
    return e * (2 * d * elem + 8) + n * (d * elem + d * 4 + 12)


def bwd_algorithmic_bytes(n: int, e: int, d: int, elem: int = 4) -> int:
    """SURVEY.md section 8d, backward: E*(2*d*s + 8) [re-gather of the [Q|Z] rows + col + type]
    + E*2*d*4 [the dQ / dZ contributions, counted once as a write] + N*(3*d*4 + 16) [G and out in, dP out, max/den/rowptr]."""
    return e * (2 * d * elem + 8) + e * 2 * d * 4 + n * (3 * d * 4 + 16)


def bwd_implementation_bytes(n: int, e: int, d: int, elem: int = 4) -> int:
    """What the deterministic three-pass backward of this library actually moves: pass A re-gathers [Q|Z]
    and writes 72 B of per-edge records; passes B and C each gather one G row (d) + the record per edge; per node: P, G,
    out, Z in, dP out (pass A) and the [dQ|dZ] row out (pass B).  Diagnostic only: fractions are quoted on
    bwd_algorithmic_bytes."""
    per_edge = (2 * d * elem + 8 + 72) + 2 * (d * elem + 72 + 8)
    per_node = 5 * d * elem + 16 + 2 * d * elem
    return e * per_edge + n * per_node
