"""GPU: the training step is bitwise reproducible.

Every adjoint on the path sums in a fixed order or sums exact integers: the aggregation backward (records + plain stores), the
BatchNorm / normalise reductions (two-level, fixed order), the grouped relation-side products (partial tiles summed in wave
order), the triple-L1 margin adjoint (float atomics on INTEGER contributions: the loss' gradient is gloss / (2 B K) times an
integer matrix, so any order gives the same bits) and the pair-cosine adjoint (incidences sorted by gradient row, one plain store
per row).  Two runs of the same steps from the same state -- dropout ON, drawn from the same generator state -- must therefore
leave identical bits in every parameter, every Adam moment and every BatchNorm buffer."""
import argparse
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(w, steps, torch_adam=False):
    import bench
    dev = next(w.model.parameters()).device
    w.model.load_state_dict({k: v.to(dev) for k, v in w.state_cpu.items()}, strict=True)
    w.opt = bench.make_adam(w.model.parameters(), argparse.Namespace(torch_adam=torch_adam))
    torch.manual_seed(123)
    losses = []
    for _ in range(steps):
        losses.append(w.step().detach().clone())
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in w.model.named_parameters() if p.grad is not None}
    state = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    moments = [v.detach().clone() for st in w.opt.state.values() for v in st.values() if torch.is_tensor(v)]
    return losses, grads, state, moments


@pytest.mark.parametrize("torch_adam", [False, True])
@pytest.mark.parametrize("kind", ["ja", "pair"])
def test_two_full_steps_are_bitwise_identical(kind, torch_adam):
    """bench.py's own workloads on the REAL data at d = 300 (dropout 0.4 on): ``ja`` = the single-KG step of the headline
    (forward_base + completion / cosine losses + backward + Adam), ``pair`` = JMAC.completion_loss on the el + ja pair through the
    stacked launch set."""
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
    w = bench.JaWorkload(a, torch.device("cuda"), data="real") if kind == "ja" else bench.PairWorkload(a, torch.device("cuda"))
    assert w.model.completion_dropout.p == 0.4 and w.model.training
    a1, a2 = _run(w, 2, torch_adam), _run(w, 2, torch_adam)
    for x, y in zip(a1[0], a2[0]):
        assert torch.equal(x, y), (float(x), float(y))
    for name in a1[1]:
        assert torch.equal(a1[1][name], a2[1][name]), "grad " + name
    for name in a1[2]:
        assert torch.equal(a1[2][name], a2[2][name]), "state " + name
    assert len(a1[3]) == len(a2[3]) and all(torch.equal(x, y) for x, y in zip(a1[3], a2[3]))


def test_loss_adjoints_are_bitwise_reproducible_and_match_autograd():
    """The two loss adjoints on their own, with heavy index repetition (hot rows): ten runs give identical bits, and the values
    are torch autograd's on the reference's expressions (src/jmac_model.py:345-378, :245-247) in float64."""
    import numpy as np
    from jmac_amd import losses
    from util import assert_close
    rng = np.random.default_rng(0)
    n, nr, d, B, K = 400, 7, 300, 64, 25
    ent = (torch.randn(n, d) * 0.3).cuda().requires_grad_(True)
    rel = (torch.randn(nr, d) * 0.3).cuda().requires_grad_(True)
    hb, rb, tb = rng.integers(0, 20, B), rng.integers(0, nr, B), rng.integers(0, n, B)           # 20 hot heads
    h = torch.from_numpy(np.tile(hb, K + 1)).cuda()
    r = torch.from_numpy(np.tile(rb, K + 1)).cuda()
    t = torch.from_numpy(np.concatenate([tb, rng.integers(0, 30, B * K)])).cuda()                 # 30 hot tails
    margin = torch.tensor([5.0]).cuda()
    i1 = torch.from_numpy(rng.integers(0, 25, 500)).cuda()
    i2 = torch.from_numpy(rng.integers(0, n, 500)).cuda()

    def run():
        ent.grad = rel.grad = None
        loss = losses.triple_l1_margin_loss(ent, rel, h, r, t, B, margin) * 1.7 + losses.pair_cosine_distance(ent, i1, ent, i2).mean()
        loss.backward()
        return loss.detach().clone(), ent.grad.clone(), rel.grad.clone()
    first = run()
    for _ in range(9):
        again = run()
        assert all(torch.equal(x, y) for x, y in zip(first, again))
    e64, r64 = ent.detach().double().cpu().requires_grad_(True), rel.detach().double().cpu().requires_grad_(True)
    hc, rc, tc = h.cpu(), r.cpu(), t.cpu()
    score = torch.norm((e64[hc] + r64[rc]) - e64[tc], 1, -1)
    pos, neg = score[:B].view(-1, B).permute(1, 0), score[B:].view(-1, B).permute(1, 0)
    ref = (torch.max(pos - neg, torch.tensor([-5.0], dtype=torch.float64)).mean() + 5.0) * 1.7
    a_, b_ = torch.nn.functional.normalize(e64[i1.cpu()], 2, -1), torch.nn.functional.normalize(e64[i2.cpu()], 2, -1)
    ref = ref + (1 - (a_ * b_).sum(1)).mean()
    ref.backward()
    assert abs(float(first[0]) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    assert_close(first[1], e64.grad, 1e-5, 1e-9, "d ent")
    assert_close(first[2], r64.grad, 1e-5, 1e-9, "d rel")


@pytest.mark.parametrize("kind", ["ja", "pair"])
def test_layer0_gradients_are_taken_over_in_place_and_equal_the_added_form(kind):
    """Round 5: the encoder nodes hand comp_att / rel_comp back as layer 0 of the completion layers, so the layer-0 loss gradient
    arrives at the node's backward and the node adds its own input gradient onto that buffer (jmac_amd.encoder._take_grad) instead
    of autograd adding two [N, d] tensors.  The step must (a) actually take that path -- INPLACE_COUNT moves by two per step (the
    entity and the relation table) -- and (b) leave the gradients of the form where nothing is taken over (INPLACE_GRADS = False:
    autograd's adds), to rounding: the association of the three contributions differs, nothing else."""
    sys.path.insert(0, ROOT)
    import bench
    from jmac_amd import encoder
    a = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
    w = bench.JaWorkload(a, torch.device("cuda"), data="real") if kind == "ja" else bench.PairWorkload(a, torch.device("cuda"))
    w.model.completion_dropout.p = 0.0
    res = {}
    for flag in (True, False):
        encoder.INPLACE_GRADS = flag
        try:
            before = encoder.INPLACE_COUNT
            _, grads, _, _ = _run(w, 1)
            res[flag] = (grads, encoder.INPLACE_COUNT - before)
        finally:
            encoder.INPLACE_GRADS = True
    assert res[True][1] == 2 and res[False][1] == 0, (res[True][1], res[False][1])
    for name, g in res[True][0].items():
        ref = res[False][0][name]
        assert float((g - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-30), name
