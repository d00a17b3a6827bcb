"""GPU, BASELINE.json full sizes: size-independent properties of the aggregation op (the oracle cannot run at
these sizes in seconds).  configs[1] = DBP-5L ja shape (N=11 805, E=17 979 / 35 958 bidirectional, d=300);
configs[3] = 1M entities / 20M triples / 1k relations (power-law, hubs of 250k in-edges split into ~490 chunks).

  * softmax normalisation: with Z = 1 and Rz = 0 the output must be sqrt(deg_i) exactly (weights sum to one),
    through every chunk merge;
  * linearity in the message half: out(Z1 + Z2) = out(Z1) + out(Z2) for fixed attention inputs;
  * conservation in the backward: sum_j dZ[j] = sum_i sqrt(deg_i) g_i, sum_t dRz[t] = -sum_j dZ[j],
    sum_i dP[i] = sum_j dQ[j] = -sum_t dRq[t]  (every edge contributes the same vector to one row of each table);
  * deterministic backward: two runs are bitwise identical;
  * graph ingest: CSR slots are a permutation of the edges, rowptr matches bincount.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tables(n, nrel, d, seed, dev):
    gen = torch.Generator(device=dev).manual_seed(seed)
    PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
    RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
    a = torch.randn(d, device=dev, generator=gen) * 0.1
    return PQZ, RR, a


def _graph(kind):
    from jmac_amd import synth
    if kind == "ja":
        return synth.dbp5l_like("ja", 1234)
    if kind == "ja-bidir":
        return synth.dbp5l_like("ja", 1234, bidirectional=True)
    return synth.power_law_graph(1_000_000, 20_000_000, 1000, seed=1234)


@pytest.mark.parametrize("kind", ["ja", "ja-bidir", "config4"])
def test_fullsize_properties(kind):
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    dev = torch.device("cuda")
    d = 300
    ei, et, n, nrel = _graph(kind)
    e = ei.shape[1]
    g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
    # ---- graph ingest
    deg = torch.from_numpy(np.bincount(ei[0], minlength=n)).to(dev)
    assert torch.equal((g.rowptr[1:] - g.rowptr[:-1]).long(), deg)
    perm = g.perm[:e].long()
    assert torch.equal(torch.sort(perm)[0], torch.arange(e, device=dev))
    assert torch.equal(g.col[:e].long(), torch.from_numpy(ei[1]).to(dev)[perm])
    cnt = g.by_dst.counts.tolist()
    if kind == "config4":
        assert cnt[1] > 1000 and int(deg.max()) > 100_000          # hubs really are split
    PQZ, RR, a = _tables(n, nrel, d, 1, dev)

    # ---- softmax normalisation through all chunk merges
    P1 = PQZ.clone()
    P1[:, 2 * d:] = 1.0
    R1 = RR.clone()
    R1[:, d:] = 0.0
    with torch.no_grad():
        nb = ops.rel_attn_aggregate(P1, R1, a, g, 0.05, -1, 1.0)
    want = deg.float().sqrt().view(-1, 1).expand(n, d)
    assert (nb - want).abs().max().item() <= 2e-4 * max(1.0, float(want.max()))
    del P1, R1, nb, want

    # ---- linearity in the message half
    with torch.no_grad():
        Za, Zb = PQZ.clone(), PQZ.clone()
        Zb[:, 2 * d:] = torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(9)) * 0.3
        Ra, Rb = RR.clone(), RR.clone()
        Rb[:, d:] = 0.5 * RR[:, d:] + 0.1
        Zs, Rs = Za.clone(), Ra.clone()
        Zs[:, 2 * d:] = Za[:, 2 * d:] + Zb[:, 2 * d:]
        Rs[:, d:] = Ra[:, d:] + Rb[:, d:]
        oa = ops.rel_attn_aggregate(Za, Ra, a, g, 0.05, nrel - 1, 0.5)
        ob = ops.rel_attn_aggregate(Zb, Rb, a, g, 0.05, nrel - 1, 0.5)
        os_ = ops.rel_attn_aggregate(Zs, Rs, a, g, 0.05, nrel - 1, 0.5)
        scale = float(os_.abs().max())
        assert (os_ - (oa + ob)).abs().max().item() <= 2e-5 * scale
    del Za, Zb, Zs, oa, ob, os_

    # ---- backward conservation laws + bitwise reproducibility
    G = torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    grads = []
    for _ in range(2):
        Pq = PQZ.clone().requires_grad_(True)
        Rq = RR.clone().requires_grad_(True)
        aq = a.clone().requires_grad_(True)
        out = ops.rel_attn_aggregate(Pq, Rq, aq, g, 0.05, -1, 1.0)       # no self term: pure edge op
        out.backward(G)
        grads.append((Pq.grad, Rq.grad, aq.grad))
    for x, y in zip(*grads):
        assert torch.equal(x, y)
    dPQZ, dRR, _ = grads[0]
    dP, dQ, dZ = dPQZ[:, :d].double().sum(0), dPQZ[:, d:2 * d].double().sum(0), dPQZ[:, 2 * d:].double().sum(0)
    dRq, dRz = dRR[:, :d].double().sum(0), dRR[:, d:].double().sum(0)
    want_dz = (deg.double().sqrt().view(-1, 1) * G.double()).sum(0)

    def close(x, y, what):
        tol = 2e-4 * max(float(y.abs().max()), float(x.abs().max()), 1e-30) + 1e-6 * (e ** 0.5)
        assert (x - y).abs().max().item() <= tol, (what, (x - y).abs().max().item(), tol)
    close(dZ, want_dz, "sum dZ = sum sqrt(deg) g")
    close(dRz, -dZ, "sum dRz = -sum dZ")
    close(dQ, dP, "sum dQ = sum dP")
    close(dRq, -dP, "sum dRq = -sum dP")


def test_fullsize_scoring_properties():
    """Scoring at configs[2] size (B=1000 x N=56 589 x d=300): L1 symmetry/identity and rank bounds."""
    from jmac_amd import scoring
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    N, B, d = 56589, 1000, 300
    tab = torch.randn(N, d, device=dev, generator=gen)
    qi = torch.randint(0, N, (B,), device=dev, generator=gen)
    dist = scoring.l1_scores(tab[qi], tab)
    assert dist.shape == (B, N) and float(dist.min()) >= 0.0
    assert float(dist[torch.arange(B, device=dev), qi].abs().max()) == 0.0          # d(x, x) = 0 exactly
    rank = scoring.filtered_rank(dist, qi)                                          # the query itself is the nearest
    assert int(rank.max()) == 1
    d2 = scoring.l1_scores(tab[:64], tab[qi[:32]])
    d3 = scoring.l1_scores(tab[qi[:32]], tab[:64])
    assert torch.allclose(d2, d3.t(), rtol=1e-6, atol=1e-4)                         # symmetry
    a = torch.nn.functional.normalize(tab[:3000])
    idx, val = scoring.sim_topk(a, a, 25, return_values=True)
    assert bool((idx[:, 0] == torch.arange(3000, device=dev)).all())                # self is the most similar
    assert bool((val[:, :-1] >= val[:, 1:]).all())                                  # sorted descending
