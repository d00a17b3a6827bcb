"""EnTr bookkeeping (row f4), host logic -- no GPU: the vectorised transfer_knowledge / link maps of jmac_amd/entr.py
against (a) the reference's own outputs captured in tests/golden/entr_small.npz (train.py:138-211, 297-325 run by
gen_golden.py) and (b) the oracle's loop restatement on random cases, including the reference's quirks."""
import numpy as np
import pytest

import oracle.jmac_oracle as orc
from conftest import load_golden


def _tuples(a):
    return [tuple(int(v) for v in x) for x in np.asarray(a).reshape(-1, 3)]


def test_transfer_knowledge_matches_reference_golden():
    from jmac_amd import entr
    g = load_golden("entr_small")
    k1 = entr.encode_triples(g["triples1"])
    k2 = entr.encode_triples(g["triples2"])
    n1, n2, a1, a2 = entr.transfer_knowledge(g["triples1"], g["triples2"], g["links"], k1, k2)
    assert np.array_equal(n1, g["new_triples1"])                  # same triples IN THE SAME ORDER as the reference
    assert np.array_equal(n2, g["new_triples2"])
    assert np.array_equal(np.unique(a1), np.unique(entr.encode_triples(g["keys1"])))
    assert np.array_equal(np.unique(a2), np.unique(entr.encode_triples(g["keys2"])))
    # the fixture exercises the quirks: entity 0 as a mapped id, a source listed twice, an image that already exists
    assert (g["links"][:, 1] == 0).any() and (g["links"][:, 0] == 0).any()
    assert len(np.unique(g["links"][:, 0])) < len(g["links"])


def test_string_keys_interop():
    from jmac_amd import entr
    g = load_golden("entr_small")
    strs = set("%d_%d_%d" % tuple(x) for x in g["triples2"].tolist())
    codes = entr.keys_from_strings(strs)
    assert entr.keys_to_strings(codes) == strs
    out = entr.transfer_knowledge(g["triples1"], g["triples2"], g["links"], set("%d_%d_%d" % tuple(x) for x in g["triples1"].tolist()), strs)
    assert np.array_equal(out[1], g["new_triples2"])


@pytest.mark.parametrize("seed", range(6))
def test_transfer_knowledge_random_vs_loop_oracle(seed):
    from jmac_amd import entr
    rng = np.random.default_rng(seed)
    n1, n2, nr = 40, 35, 5
    t1 = np.stack([rng.integers(0, n1, 300), rng.integers(0, nr, 300), rng.integers(0, n1, 300)], 1)
    t2 = np.stack([rng.integers(0, n2, 250), rng.integers(0, nr, 250), rng.integers(0, n2, 250)], 1)
    L = int(rng.integers(0, 30))
    pairs = np.stack([rng.integers(0, n1, L), rng.integers(0, n2, L)], 1)      # repeats on both sides, id 0 included
    ks1, ks2 = set(_tuples(t1)), set(_tuples(t2))
    w1, w2 = orc.transfer_knowledge(t1, t2, pairs, ks1, ks2)
    g1, g2, c1, c2 = entr.transfer_knowledge(t1, t2, pairs, entr.encode_triples(t1), entr.encode_triples(t2), n1, n2)
    assert _tuples(g1) == w1 and _tuples(g2) == w2
    assert set(np.unique(c1).tolist()) == set(entr.encode_triples(np.array(sorted(ks1))).tolist())
    assert set(np.unique(c2).tolist()) == set(entr.encode_triples(np.array(sorted(ks2))).tolist())


def test_link_maps_dict_semantics():
    from jmac_amd import entr
    pairs = np.array([[3, 9], [5, 0], [3, 4], [7, 4], [0, 2]])
    fwd, inv = entr.link_maps(pairs)
    links = {int(a): int(b) for a, b in pairs}
    inverse = {v: k for k, v in links.items()}
    assert {i: int(v) for i, v in enumerate(fwd) if v >= 0} == links
    assert {i: int(v) for i, v in enumerate(inv) if v >= 0} == inverse


def test_align_data_processing_direction():
    import torch
    from jmac_amd import entr
    ei, et = entr.align_data_processing([[1, 7, 2], [3, 8, 4]], "cpu")          # train.py:130-132: [head, tail]
    assert ei.tolist() == [[1, 3], [2, 4]] and et.tolist() == [7, 8] and ei.dtype == torch.int64
