"""GPU, BASELINE metric "Hits@1 parity" end to end: the HIP path replays the reference's 120 seeded training steps
(tests/golden/e2e_ja_sub.npz: ja / el sub-graphs of DBP-5L, dropout 0, captured batches, two Adam optimisers as
train.py:406-407) through jmac_amd.model.JMAC -- three HIP layers, fused losses, deterministic backward -- and scores the
validation split with the HIP evaluator path (forward_linkpred + filtered_rank, src/validate.py:22-80).  It must land on the
reference's losses, ranks (30-step checkpoint: identical) and Hits@1 / Hits@10 / MRR (end of run: within seed noise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_replay import check_outcome, feeddict, fixture


def test_hip_path_replays_reference_training_run():
    from jmac_amd import harness, scoring
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
    opt_a = torch.optim.Adam(model.parameters(), lr=m["lr"])  # train.py:406-407
    opt_c = torch.optim.Adam(model.parameters(), lr=m["lr"])
    e1i, e1t, e2i, e2t = (torch.from_numpy(g[k]).to(dev) for k in ("e1_index", "e1_type", "e2_index", "e2_type"))
    feed = feeddict(g, m)
    val = g["val1"]
    fp, fi = torch.from_numpy(g["filt_ptr"]).to(dev), torch.from_numpy(g["filt_idx"]).to(dev)

    def val_ranks():
        model.eval()
        with torch.no_grad():
            dist = model.forward_linkpred(val[:, 0].tolist(), val[:, 1].tolist(), e1i, e1t, range(m["n1"]), m["eb1"], m["rb1"])
            rk = scoring.filtered_rank(dist, val[:, 2], fp, fi).cpu().numpy()
        model.train()
        return rk

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
    got = check_outcome(g, losses, ranks_ckpt, val_ranks(), "hip")
    print("HIP replay: Hits@1 %.4f Hits@10 %.4f MRR %.4f (reference %s); ranks identical at the checkpoint: %d / %d" % (
        got[0], got[1], got[2], np.round(g["metrics_after"], 4), int((ranks_ckpt == g["ranks_ckpt"]).sum()), len(ranks_ckpt)))
