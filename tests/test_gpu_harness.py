"""GPU integration: the whole path end to end on the synthetic DBP-5L-format mini dataset -- data.load_dbp5l ->
RelGraph -> JMAC (three HIP layers) -> fused losses -> backward -> Adam, EnTr (alignment_quality, get_neg,
transfer_knowledge) between epochs, filtered link-prediction evaluation -- driven like the reference's train.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dbp5l_mini")


def test_train_two_epochs_on_mini_dataset():
    from jmac_amd import data, harness
    from jmac_amd.model import JMAC
    torch.manual_seed(0)
    kgs, s_train, s_test, n_ent = data.load_dbp5l(ROOT, "ja")
    args = harness.make_args(dim=32, batch_size=32, num_negative=5, dropout=0.0, lr=5e-3, pair_sample_weight=2.0)
    rng = np.random.default_rng(0)
    name_emb = rng.standard_normal((n_ent, 24)).astype(np.float32)
    n_rel_total = sum(kg.num_relation for kg in kgs.values())
    model = JMAC(args, name_emb, n_rel_total, n_ent).cuda()
    opt_c = torch.optim.Adam(model.parameters(), lr=args.lr)
    opt_a = torch.optim.Adam(model.parameters(), lr=args.lr)
    gen = torch.Generator(device="cuda").manual_seed(1)
    ja = kgs["ja"]
    ei = torch.from_numpy(ja.edge_index).cuda()
    et = torch.from_numpy(ja.edge_type).cuda()
    h1_0, h10_0, mrr_0 = harness.evaluate_completion(model, ja, ei, et, args, "train")
    state, logs = {}, []
    model.train()
    for epoch in range(6):
        logs.append(harness.train_epoch(model, kgs, s_train, s_test, opt_c, opt_a, args, state, refresh=(epoch % 3 == 0),
                                        generator=gen))
    first = np.mean([p["completion_loss"] for p in logs[0]])
    last = np.mean([p["completion_loss"] for p in logs[-1]])
    assert np.isfinite(first) and np.isfinite(last) and last < first           # the completion loss goes down
    assert all(np.isfinite(p["align_loss"]) for e in logs for p in e)
    assert logs[-1][0]["align_loss"] < logs[0][0]["align_loss"]
    # triple transfer happened and is idempotent across refreshes (keys dedupe): counts never shrink
    for a, b in zip(logs[0], logs[-1]):
        assert b["triples"][0] >= a["triples"][0] and b["triples"][1] >= a["triples"][1]
    assert any(p["triples"][0] > len(kgs[p["pair"][0]].train_data) or p["triples"][1] > len(kgs[p["pair"][1]].train_data)
               for p in logs[0])
    h1, h10, mrr = harness.evaluate_completion(model, ja, ei, et, args, "val")
    assert 0.0 <= h1 <= h10 <= 1.0 and 0.0 < mrr <= 1.0
    # the triples are random (nothing to generalise to), but the ones trained on must now rank better than at init
    _, h10_t, mrr_t = harness.evaluate_completion(model, ja, ei, et, args, "train")
    assert mrr_t > mrr_0 and h10_t >= h10_0
    # evaluation is deterministic and the cached-encoder path equals the per-batch path of the reference
    assert harness.evaluate_completion(model, ja, ei, et, args, "val") == (h1, h10, mrr)
