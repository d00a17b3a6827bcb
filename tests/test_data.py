"""Dataset reader (row f2, formats), host logic -- no GPU: jmac_amd/data.py against the reference's own loader run
on the synthetic DBP-5L-format mini dataset (tests/golden/dbp5l_mini/, golden dbp5l_mini.npz from gen_golden.py)."""
import os

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dbp5l_mini")


@pytest.mark.parametrize("target", ["ja", "en"])
def test_reader_matches_reference_loader(target):
    from jmac_amd import data
    g = load_golden("dbp5l_mini")
    kgs, s_train, s_test, n_ent = data.load_dbp5l(ROOT, target)
    assert list(kgs) == g["%s.kg_names" % target].tolist() and n_ent == int(g["%s.num_entities" % target])
    for lang, kg in kgs.items():
        pre = "%s.%s." % (target, lang)
        assert np.array_equal(kg.train_data, g[pre + "train"])            # supporters: train + val concatenated
        assert np.array_equal(kg.val_data, g[pre + "val"]) and np.array_equal(kg.test_data, g[pre + "test"])
        meta = [kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base, kg.relation_id_base,
                kg.upper_entity_base, kg.upper_relation_base]
        assert meta == g[pre + "meta"].tolist()
        assert np.array_equal(kg.edge_index, g[pre + "edge_index"]) and np.array_equal(kg.edge_type, g[pre + "edge_type"])
        if not kg.is_supporter_kg:
            keys = [tuple(k) for k in g[pre + "true_tail_keys"].tolist()]
            assert sorted(kg.true_tail) == keys
            ptr, idx = g[pre + "true_tail_ptr"], g[pre + "true_tail_idx"]
            for i, k in enumerate(keys):
                assert sorted(kg.true_tail[k].tolist()) == sorted(idx[ptr[i]:ptr[i + 1]].tolist())
    for name, seeds in (("seeds_train", s_train), ("seeds_test", s_test)):
        want = {k.split(".")[-1]: g[k] for k in g if k.startswith("%s.%s." % (target, name))}
        assert {"%s-%s" % k for k in seeds} == set(want)
        for (l1, l2), v in seeds.items():
            assert np.array_equal(v, want["%s-%s" % (l1, l2)])           # float-formatted ids parsed to ints


def test_relation_count_and_direction():
    from jmac_amd import data
    kgs, _, _, _ = data.load_dbp5l(ROOT, "ja")
    assert kgs["ja"].num_relation == 13                                   # relations.txt lines + 1 (data_loader.py:211-212)
    tr = kgs["ja"].train_data
    e = len(tr)
    ei, et = kgs["ja"].edge_index, kgs["ja"].edge_type
    assert ei.shape == (2, 2 * e)                                         # target KG: train only, both directions
    assert np.array_equal(ei[0, :e], tr[:, 0]) and np.array_equal(ei[1, :e], tr[:, 2])
    assert np.array_equal(ei[0, e:], tr[:, 2]) and np.array_equal(et[e:], tr[:, 1])   # reverse edge reuses the id
