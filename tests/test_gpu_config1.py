"""BASELINE config 1 on the device: the reference's own default run -- target_language=ja, d=256 (train.py:73), batch 1000,
25 negatives, dropout 0.4, two GCN layers -- on the REAL DBP-5L el / ja KGs (tests/golden/dbp5l_ja_el_data.npz), driven like train.py
(get_emb -> EnTr -> completion batches -> alignment step, src of truth: jmac_amd/harness.py), one epoch and then a few more.

What can be asserted without the reference's random streams (its DataLoader workers draw the negatives): the plumbing at
full size -- every kernel on the real graphs (hub rows of 1 221 / 673 edges, 4 332 isolated entities, 961 relations), the EnTr
transfer, the filtered evaluator over 11 805 candidates -- finite losses that go down, and link-prediction quality on the
VALIDATION triples that moves far from the untrained model's.  Numerical parity at this size is test_gpu_layer's
`layer_ja_full` case (the reference's layer on this very graph) and tests/test_gpu_ja_oracle.py."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config1_epochs_on_real_ja(tmp_path):
    from conftest import load_golden
    from util import write_dbp5l_dir
    from jmac_amd import data, harness
    from jmac_amd.model import JMAC
    torch.manual_seed(0)
    root = write_dbp5l_dir(str(tmp_path / "dbp5l_ja_el"), load_golden("dbp5l_ja_el_data"))
    kgs, s_train, s_test, n_ent = data.load_dbp5l(root, "ja")
    args = harness.make_args(dim=256, batch_size=1000, num_negative=25, dropout=0.4, lr=1e-3)
    name_emb = np.random.default_rng(0).standard_normal((n_ent, 300)).astype(np.float32)   # synthetic N(0,1) names
    n_rel_total = sum(kg.num_relation for kg in kgs.values())
    model = JMAC(args, name_emb, n_rel_total, n_ent).cuda()
    opt_c = torch.optim.Adam(model.parameters(), lr=args.lr)
    opt_a = torch.optim.Adam(model.parameters(), lr=args.lr)
    gen = torch.Generator(device="cuda").manual_seed(1)
    ja = kgs["ja"]
    ei, et = torch.from_numpy(ja.edge_index).cuda(), torch.from_numpy(ja.edge_type).cuda()
    h1_0, h10_0, mrr_0 = harness.evaluate_completion(model, ja, ei, et, args, "val")
    assert h10_0 < 0.05                                              # untrained (11 805 candidates): the graph alone gives ~2 %
    state, logs = {}, []
    model.train()
    for epoch in range(4):
        logs.append(harness.train_epoch(model, kgs, s_train, s_test, opt_c, opt_a, args, state, refresh=(epoch == 0), generator=gen))
    first, last = logs[0][0], logs[-1][0]
    assert first["pair"] == ("el", "ja") and first["links"] >= 1112            # seed pairs (+ enlargement)
    assert first["triples"][0] >= 12822 and first["triples"][1] >= 17979       # el: train + val; ja: train (+ transferred)
    assert all(np.isfinite(p["completion_loss"]) and np.isfinite(p["align_loss"]) for e in logs for p in e)
    assert last["completion_loss"] < 0.8 * first["completion_loss"]
    assert last["align_loss"] < first["align_loss"]
    h1, h10, mrr = harness.evaluate_completion(model, ja, ei, et, args, "val")
    assert 0.0 <= h1 <= h10 <= 1.0
    print('config 1: val Hits@1 / Hits@10 / MRR  untrained %.4f %.4f %.4f  after 4 epochs %.4f %.4f %.4f' % (h1_0, h10_0, mrr_0, h1, h10, mrr))
    assert h10 > 3 * h10_0 and mrr > 3 * mrr_0, (h1, h10, mrr, h10_0, mrr_0)
