"""BASELINE config 1's data: the real DBP-5L el / ja KGs and their seed pairs (tests/golden/dbp5l_ja_el_data.npz, written back
into the dataset's own on-disk format by util.write_dbp5l_dir) read by jmac_amd.data.load_dbp5l must give the arrays the
REFERENCE's loader built from the same files
(src/data_loader.py:158-221, src/utils.py:112-149, src/knowledgegraph.py:18-19,45-46; pinned in dbp5l_ja_el.npz), and the
shapes SURVEY.md section 8(d) tabulates."""

import numpy as np

from conftest import load_golden
from util import write_dbp5l_dir


def array_digest(a):
    """Same arithmetic as tests/golden/gen_golden.py:array_digest."""
    v = np.ascontiguousarray(a).astype(np.uint64).reshape(-1)
    w = (np.arange(1, v.size + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    return np.uint64((v * w).sum(dtype=np.uint64) ^ np.uint64(v.size))


def test_real_ja_el_dataset_matches_reference_loader(tmp_path):
    from jmac_amd import data
    g = load_golden("dbp5l_ja_el")
    root = write_dbp5l_dir(str(tmp_path / "dbp5l_ja_el"), load_golden("dbp5l_ja_el_data"))
    kgs, s_train, s_test, n_ent = data.load_dbp5l(root, "ja")
    assert list(kgs) == list(g["kg_names"]) == ["el", "ja"] and n_ent == int(g["num_entities"]) == 17036
    for lang, kg in kgs.items():
        meta = [kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base, kg.relation_id_base,
                kg.upper_entity_base, kg.upper_relation_base]
        assert meta == g[lang + ".meta"].tolist()
        assert [len(kg.train_data), len(kg.val_data), len(kg.test_data), kg.edge_index.shape[1]] == g[lang + ".shapes"].tolist()
        dig = [array_digest(kg.train_data), array_digest(kg.val_data), array_digest(kg.test_data), array_digest(kg.edge_index),
               array_digest(kg.edge_type)]
        assert [int(x) for x in dig] == [int(x) for x in g[lang + ".digests"]], lang
        deg = np.bincount(kg.edge_index[0], minlength=kg.num_entity)
        assert [int(deg.max()), int((deg == 0).sum())] == g[lang + ".degree"].tolist()
    # ja N = 11 805, train 17 979, val 8 633, test 2 162, bidirectional E = 35 958, max degree 1 221,
    # 4 332 isolated nodes; el (supporter: train + val) 12 822 triples, 25 644 bidirectional edges, max degree 673
    ja, el = kgs["ja"], kgs["el"]
    assert (ja.num_entity, len(ja.train_data), len(ja.val_data), len(ja.test_data), ja.edge_index.shape[1]) == (11805, 17979, 8633, 2162, 35958)
    assert (el.num_entity, len(el.train_data), el.edge_index.shape[1]) == (5231, 12822, 25644)
    tt = ja.true_tail
    assert [len(tt), sum(len(v) for v in tt.values())] == g["ja.true_tail"].tolist()
    for tag, sd in (("seeds_train", s_train), ("seeds_test", s_test)):
        (k, v), = sd.items()
        assert list(k) == list(g[tag + ".pair"]) == ["el", "ja"]
        assert (np.asarray(v) == g[tag]).all() and len(v) == 1112


def test_all_five_real_kgs_match_reference_loader(tmp_path):
    """BASELINE config 3's data: all five real DBP-5L KGs + the ten seed-pair files (tests/golden/dbp5l_all_data.npz), written back
    into the dataset's on-disk format and read by jmac_amd.data.load_dbp5l, AND built straight from the arrays
    (data.kgs_from_arrays): both must give what the REFERENCE's loader built from the same files (dbp5l_all.npz: id bases
    src/data_loader.py:162-181, triple / edge digests, degrees), and the union graph of SURVEY.md 8(d) "Config 3"."""
    from jmac_amd import data
    g = load_golden("dbp5l_all")
    z = load_golden("dbp5l_all_data")
    root = write_dbp5l_dir(str(tmp_path / "dbp5l_all"), z)
    for kgs, s_train, s_test, n_ent in (data.load_dbp5l(root, "ja"), data.kgs_from_arrays(z, "ja")):
        assert list(kgs) == list(g["kg_names"]) == ["el", "en", "es", "fr", "ja"] and n_ent == int(g["num_entities"]) == 56589
        for lang, kg in kgs.items():
            meta = [kg.num_entity, kg.num_relation, int(kg.is_supporter_kg), kg.entity_id_base, kg.relation_id_base,
                    kg.upper_entity_base, kg.upper_relation_base]
            assert meta == g[lang + ".meta"].tolist(), lang
            assert [len(kg.train_data), len(kg.val_data), len(kg.test_data), kg.edge_index.shape[1]] == g[lang + ".shapes"].tolist()
            dig = [array_digest(kg.train_data), array_digest(kg.val_data), array_digest(kg.test_data), array_digest(kg.edge_index),
                   array_digest(kg.edge_type)]
            assert [int(x) for x in dig] == [int(x) for x in g[lang + ".digests"]], lang
            deg = np.bincount(kg.edge_index[0], minlength=kg.num_entity)
            assert [int(deg.max()), int((deg == 0).sum())] == g[lang + ".degree"].tolist()
            tdeg = np.bincount(kg.train_data[:, 0], minlength=kg.num_entity)
            assert [int(tdeg.max()), int((tdeg == 0).sum())] == g[lang + ".train_degree"].tolist()
        for tag, sd in (("seeds_train", s_train), ("seeds_test", s_test)):
            keys = sorted(sd)
            assert ["%s-%s" % k for k in keys] == list(g[tag + ".pairs"]) and len(keys) == 10
            assert [len(sd[k]) for k in keys] == g[tag + ".sizes"].tolist()
            assert [int(array_digest(sd[k])) for k in keys] == [int(x) for x in g[tag + ".digests"]]
        # bidirectional max degrees 673 / 3 119 / 3 880 / 4 219 (supporters, train + val) and 1 221 (ja, train only)
        assert [int(g[l + ".degree"][0]) for l in ("el", "en", "es", "fr", "ja")] == [673, 3119, 3880, 4219, 1221]
        ei, et, n, nr, eb, rb = data.union_edges(kgs)
        assert (n, nr, ei.shape[1]) == (56589, 5 * 961, 12822 + 72703 + 49256 + 44844 + 17979) and eb[-1] == n and rb[-1] == nr
        for k, lang in enumerate(sorted(kgs)):              # block-diagonal: every edge stays inside its KG's id ranges
            sel = (ei[0] >= eb[k]) & (ei[0] < eb[k + 1])
            assert sel.sum() == len(kgs[lang].train_data)
            assert ((ei[1][sel] >= eb[k]) & (ei[1][sel] < eb[k + 1])).all() and ((et[sel] >= rb[k]) & (et[sel] < rb[k + 1])).all()
        eib, _, _, _, _, _ = data.union_edges(kgs, bidirectional=True)
        assert eib.shape[1] == 25644 + 145406 + 98512 + 89688 + 35958
