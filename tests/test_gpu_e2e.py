"""GPU, BASELINE metric "Hits@1 parity" end to end: the HIP path replays the reference's 120 seeded training steps
(tests/golden/e2e_ja_sub.npz: ja / el sub-graphs of DBP-5L, dropout 0, captured batches, two Adam optimisers as
train.py:406-407; the replay starts from the state the reference trained itself to in 250 earlier steps, filtered
Hits@1 ~ 10 %) through jmac_amd.model.JMAC -- three HIP layers, fused losses, deterministic backward -- and scores the
validation split with the HIP evaluator path (forward_linkpred + filtered_rank, src/validate.py:22-80).  It must land on the
reference's losses, ranks (30-step checkpoint: identical) and Hits@1 / Hits@10 / MRR (end of run: within seed noise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_replay import check_outcome, feeddict, fixture


@pytest.mark.parametrize("optimizer", ["torch.optim.Adam", "jmac_amd.optim.Adam"])
def test_hip_path_replays_reference_training_run(optimizer):
    """Both optimizers replay the run: the reference's own (torch.optim.Adam, train.py:406-407) and this library's one-launch form of
    the same update -- it has to land on the same losses, checkpoint ranks and metrics."""
    from jmac_amd import harness, optim, scoring
    Adam = torch.optim.Adam if optimizer == "torch.optim.Adam" else optim.Adam
    from jmac_amd.model import JMAC
    g, m = fixture()
    dev = torch.device("cuda")
    args = harness.make_args(dim=m["d"], batch_size=m["B"], num_negative=m["K"], dropout=0.0, device="cuda")
    model = JMAC(args, g["name_emb"], 2 * m["nrel"], m["n1"] + m["n2"])
    model.load_state_dict({k[len("state0."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state0.")}, strict=True)
    model = model.to(dev)
    model.ent_info_att = model.ent_info_att.to(dev)
    for lay in (model.conv1_alignment, model.conv2_alignment, model.conv1_completion):
        lay.loop_rel.requires_grad_(False)                    # as in the fixture (see gen_e2e: zero-gradient parameter + Adam)
    opt_a = Adam(model.parameters(), lr=m["lr"])  # train.py:406-407
    opt_c = Adam(model.parameters(), lr=m["lr"])
    e1i, e1t, e2i, e2t = (torch.from_numpy(g[k]).to(dev) for k in ("e1_index", "e1_type", "e2_index", "e2_type"))
    feed = feeddict(g, m)
    val = g["val1"]
    fp, fi = torch.from_numpy(g["filt_ptr"]).to(dev), torch.from_numpy(g["filt_idx"]).to(dev)

    # the evaluator path the harness runs by default: harness.evaluate_completion(fused=True) -> JMAC.linkpred_ranks ->
    # jmac_linkpred_rank_f32 (no [B, N] matrix; its per-candidate sum runs over both layers in one accumulator, the
    # reference adds two cdist matrices, src/jmac_model.py:312).  It is held to the reference's ranks as well.
    from jmac_amd.data import KnowledgeGraph, true_tail_dict
    kg1 = KnowledgeGraph("ja", g["train1"], val, g["test1"], m["n1"], m["nrel"], False, 0, 0, m["n1"], m["nrel"])
    kg1.true_tail = true_tail_dict(np.concatenate((g["train1"], val, g["test1"])))      # src/knowledgegraph.py:45-46

    def val_ranks():
        model.eval()
        with torch.no_grad():
            dist = model.forward_linkpred(val[:, 0].tolist(), val[:, 1].tolist(), e1i, e1t, range(m["n1"]), m["eb1"], m["rb1"])
            rk = scoring.filtered_rank(dist, val[:, 2], fp, fi).cpu().numpy()
            rk_fused = model.linkpred_ranks(val[:, 0].tolist(), val[:, 1].tolist(), val[:, 2].tolist(), e1i, e1t, m["eb1"],
                                            m["rb1"], fp, fi).cpu().numpy()
            met_fused = harness.evaluate_completion(model, kg1, e1i, e1t, args, split="val", filtered=True, fused=True)
        model.train()
        return rk, rk_fused, np.array(met_fused)

    model.train()
    losses, ranks_ckpt = [], None
    for s, kind in enumerate(g["sched"]):
        if s == int(g["ckpt_steps"]):
            ranks_ckpt = val_ranks()
        if kind == 2:
            opt_a.zero_grad()
            loss = model.alignment_loss(feed, e1i, e1t, e2i, e2t)
            loss.backward()
            opt_a.step()
        else:
            data = {k: torch.from_numpy(g[k][s]).to(dev) for k in ("batch_h", "batch_r", "batch_t")}
            opt_c.zero_grad()
            loss = model.completion_loss(data, e1i, e1t, e2i, e2t, feed, kind == 0)
            loss.backward()
            opt_c.step()
        losses.append(float(loss.detach()))
    ranks_ckpt, fused_ckpt, met_ckpt = ranks_ckpt
    ranks_after, fused_after, met_after = val_ranks()
    got = check_outcome(g, losses, ranks_ckpt, ranks_after, "hip", decided_gap=1e-4)
    # the fused evaluator: same acceptance (decided ranks identical at the checkpoint, metrics within seed noise at the end);
    # what harness.evaluate_completion reports IS the metric of those ranks (its own filter CSR from kg.true_tail included)
    from e2e_replay import metrics
    check_outcome(g, losses, fused_ckpt, fused_after, "hip-fused", decided_gap=1e-4)
    assert np.allclose(met_ckpt, metrics(fused_ckpt), atol=1e-12) and np.allclose(met_after, metrics(fused_after), atol=1e-12)
    assert met_ckpt[0] >= 0.05                               # Hits@1 is not the vacuous 1/len(val) here
    print("fused evaluator: Hits@1 %.4f Hits@10 %.4f MRR %.4f at the checkpoint (reference %s); ranks identical: %d / %d; differing "
          "from the materialised path: %d" % (met_ckpt[0], met_ckpt[1], met_ckpt[2], np.round(g["metrics_ckpt"], 4),
                                             int((fused_ckpt == g["ranks_ckpt"]).sum()), len(fused_ckpt),
                                             int((fused_ckpt != ranks_ckpt).sum())))
    print("HIP replay: Hits@1 %.4f Hits@10 %.4f MRR %.4f (reference %s); ranks identical at the checkpoint: %d / %d" % (
        got[0], got[1], got[2], np.round(g["metrics_after"], 4), int((ranks_ckpt == g["ranks_ckpt"]).sum()), len(ranks_ckpt)))
