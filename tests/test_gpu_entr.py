"""GPU: seed_enlargement_triple_transferring (row f4) on the HIP scoring kernels against the reference's captured
outputs (tests/golden/entr_small.npz: train.py:138-211 run on the model_small embeddings)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import load_golden


def _setup():
    g, m = load_golden("entr_small"), load_golden("model_small")
    o1, o2 = torch.from_numpy(m["emb1_align"]).cuda(), torch.from_numpy(m["emb2_align"]).cuda()
    args = types.SimpleNamespace(num_negative=5, pair_sample_weight=0.2)
    n1, n2, nrel = int(g["n1"]), int(g["n2"]), int(g["nrel"])
    bases = ([0, n1], [0, nrel], [n1, n1 + n2], [nrel, 2 * nrel])
    return g, o1, o2, args, bases


def test_first_visit_matches_reference_golden():
    from jmac_amd import entr
    g, o1, o2, args, (eb1, rb1, eb2, rb2) = _setup()
    kg1 = types.SimpleNamespace(triple_keys=entr.encode_triples(g["triples1"]))
    kg2 = types.SimpleNamespace(triple_keys=entr.encode_triples(g["triples2"]))
    ge, gs = [-1], [g["links"]]
    n1, n2, k1, k2, feed, gs_out = entr.seed_enlargement_triple_transferring(
        o1, o2, g["test_src"].tolist(), g["test_dst"].tolist(), ge, 0, g["links"], g["triples1"], g["triples2"], gs,
        eb1, rb1, eb2, rb2, kg1, kg2, args)
    assert abs(ge[0] - float(g["entropy"])) <= 1e-5 * float(g["entropy"])
    assert np.array_equal(n1, g["new_triples1"]) and np.array_equal(n2, g["new_triples2"])
    assert np.array_equal(feed["links"], g["feed_links"])
    assert np.array_equal(feed["neg_left"], g["neg_left"]) and np.array_equal(feed["neg2_right"], g["neg2_right"])
    assert torch.equal(feed["neg_right"].cpu(), torch.from_numpy(g["neg_right"]))       # top-k indices: bit-exact
    assert torch.equal(feed["neg2_left"].cpu(), torch.from_numpy(g["neg2_left"]))
    assert feed["ent_bases2"] == eb2 and feed["rel_bases1"] == rb1


def test_enlargement_branch_properties():
    """Second visit with a lower entropy: num_pairs = int((prev - H)/prev * w * |test|) new pairs (train.py:153-169),
    drawn without replacement, each paired with the arg-max of its softmax row."""
    from jmac_amd import entr, scoring
    g, o1, o2, args, (eb1, rb1, eb2, rb2) = _setup()
    args.pair_sample_weight = 2.0
    kg1 = types.SimpleNamespace(triple_keys=entr.encode_triples(g["triples1"]))
    kg2 = types.SimpleNamespace(triple_keys=entr.encode_triples(g["triples2"]))
    H = float(g["entropy"])
    ge, gs = [H * 1.25], [g["links"]]
    gen = torch.Generator(device="cuda").manual_seed(3)
    out = entr.seed_enlargement_triple_transferring(
        o1, o2, g["test_src"].tolist(), g["test_dst"].tolist(), ge, 0, g["links"], g["triples1"], g["triples2"], gs,
        eb1, rb1, eb2, rb2, kg1, kg2, args, generator=gen)
    pairs = out[4]["links"]
    extra = pairs[len(g["links"]):]
    want_pairs = len(extra)
    assert want_pairs in (9, 10)                                     # int(0.2 * 2.0 * 25 -+ the entropy's last-digit rounding)
    assert len(np.unique(extra[:, 0])) == want_pairs
    assert ge[0] == H * 1.25                                         # improved entropy: the stored value is kept
    _, simi, _ = scoring.alignment_quality(o1, o2, g["test_src"].tolist(), g["test_dst"].tolist())
    assert np.array_equal(simi[extra[:, 0]].argmax(1).cpu().numpy(), extra[:, 1])
    assert np.array_equal(gs[0], pairs) and len(out[4]["neg_right"]) == len(pairs) * args.num_negative
    # a worse entropy resets the stored value and adds nothing (train.py:154-156)
    ge2, gs2 = [H * 0.5], [g["links"]]
    out2 = entr.seed_enlargement_triple_transferring(
        o1, o2, g["test_src"].tolist(), g["test_dst"].tolist(), ge2, 0, g["links"], g["triples1"], g["triples2"], gs2,
        eb1, rb1, eb2, rb2, kg1, kg2, args)
    assert abs(ge2[0] - H) <= 1e-5 * H and len(out2[4]["links"]) == len(g["links"])
