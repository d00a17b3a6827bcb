"""GPU: the destination-sharded layer with the HIP kernels as the rank-local op.  Two ranks share the one
GPU of the test box and talk over gloo (RCCL refuses two ranks on one device); the collectives' RCCL path is
exercised by the driver's multi-GPU bench.  Result must equal the single-process float64 oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import oracle.jmac_oracle as orc
from util import assert_close, make_args, random_graph


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(seed=5, n=700, nr=11, d=300, e=9000):
    rng = np.random.default_rng(seed)
    ei, et = random_graph(rng, n, nr, e, hub=900)
    gen = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=gen) * (4 / np.sqrt(d))
    R = torch.randn(nr, d, generator=gen) * (4 / np.sqrt(d))
    G = torch.randn(n, d, generator=gen)
    return ei, et, X, R, G, n, nr, d


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jmac_amd.dist import ShardedGraph, ShardedRelationAwareLayer, allreduce_grads, partition_rows
        from jmac_amd.layer import RelationAwareLayer
        dev = torch.device("cuda", 0)
        ei, et, X, R, G, n, nr, d = _case()
        bounds = partition_rows(np.bincount(ei[0], minlength=n), world)
        if os.environ.get("JMAC_TEST_BOUNDS"):          # hand-made ranges: uneven, and a rank that owns no row at all
            bounds = np.array([int(x) for x in os.environ["JMAC_TEST_BOUNDS"].split(",")], dtype=np.int64)
        sg = ShardedGraph(ei, et, bounds, rank, chunks=int(os.environ.get("JMAC_TEST_CHUNKS", "1")))
        torch.manual_seed(11)
        base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args()).to(dev)
        lay = ShardedRelationAwareLayer(base).train()                   # default local op = HIP kernels
        x = X[sg.lo:sg.hi].to(dev).requires_grad_(True)
        r = R.to(dev).requires_grad_(True)
        out = lay(x, r, sg)
        (out * G[sg.lo:sg.hi].to(dev)).sum().backward()
        allreduce_grads(list(base.parameters()) + [r])
        torch.cuda.synchronize()
        import jmac_amd.dist as jd
        ret[rank] = dict(lo=sg.lo, hi=sg.hi, out=out.detach().cpu(), gx=x.grad.cpu(), gr=r.grad.cpu(),
                         grads={k: v.grad.cpu() for k, v in base.named_parameters()},
                         overlapped=min(jd.OVERLAP_COUNT, jd.HANDOFF_COUNT))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,bounds,chunks", [(2, None, 1), (3, "0,250,250,700", 1), (2, None, 4), (3, "0,250,250,700", 3)],
                         ids=["two-ranks", "three-ranks-one-empty", "two-ranks-pipelined", "three-ranks-one-empty-pipelined"])
def test_sharded_hip_layer_two_ranks(world, bounds, chunks, monkeypatch):
    """``three-ranks-one-empty``: rank 1 owns no row -- its kernels see N = 0, its table slab is padding, and it must still
    take part in every collective of the forward and the backward.  ``pipelined``: the slab-pipelined exchange (chunk-major
    table, per-chunk partial aggregations on the HIP forward kernel, jmac_softmax_parts_merge_f32, whole-graph HIP backward,
    one reduce-scatter per chunk)."""
    monkeypatch.setenv("JMAC_TEST_CHUNKS", str(chunks))
    if bounds:
        monkeypatch.setenv("JMAC_TEST_BOUNDS", bounds)
    else:
        monkeypatch.delenv("JMAC_TEST_BOUNDS", raising=False)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    ei, et, X, R, G, n, nr, d = _case()
    from jmac_amd.layer import RelationAwareLayer
    torch.manual_seed(11)
    base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    f64 = torch.float64
    p = {k: v.detach().clone().to(f64).requires_grad_(True) for k, v in base.named_parameters()}
    Xc, Rc = X.to(f64).requires_grad_(True), R.to(f64).requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, torch.from_numpy(ei), torch.from_numpy(et), 0.05, "sub", "leaky_relu", True,
                            torch.zeros(d, dtype=f64), torch.ones(d, dtype=f64))
    (ref * G.to(f64)).sum().backward()
    gscale = Rc.grad.abs().max().item()
    for r in range(world):
        o = ret[r]
        lo, hi = o["lo"], o["hi"]
        # pipelined: the backward ran pass B slab by slab (jmac_rel_attn_aggregate_bwd_phases_f32) with each slab's reduce-scatter
        # queued as it completed, and the exchange's own backward took the reduced gradient as it was
        assert o["overlapped"] == (1 if chunks > 1 else 0), (r, o["overlapped"])
        assert_close(o["out"], ref[lo:hi], 1e-4, 1e-6, "out")
        assert_close(o["gx"], Xc.grad[lo:hi], 1e-4, 1e-6, "grad_X")
        assert_close(o["gr"], Rc.grad, 1e-4, 1e-6, "grad_R")
        for k, g in o["grads"].items():
            atol = 1e-4 * gscale + 1e-6 if k == "loop_rel" else 1e-6
            assert_close(g, p[k].grad, 1e-4, atol, "grad " + k)


def _rccl_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import jmac_amd.dist as jd
        from jmac_amd.layer import RelationAwareLayer
        jd.FORCE_COLLECTIVES = True                                      # real RCCL calls although world == 1
        dev = torch.device("cuda", 0)
        ei, et, X, R, G, n, nr, d = _case()
        sg = jd.ShardedGraph(ei, et, jd.partition_rows(np.bincount(ei[0], minlength=n), 1), 0,
                             chunks=int(os.environ.get("JMAC_TEST_CHUNKS", "1")))
        torch.manual_seed(11)
        base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args()).to(dev)
        lay = jd.ShardedRelationAwareLayer(base).train()
        x = X.to(dev).requires_grad_(True)
        r = R.to(dev).requires_grad_(True)
        out = lay(x, r, sg)
        (out * G.to(dev)).sum().backward()
        jd.allreduce_grads(list(base.parameters()) + [r])
        torch.cuda.synchronize()
        ret[0] = dict(out=out.detach().cpu(), gx=x.grad.cpu(), gw=base.w_att.grad.cpu())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chunks", [1, 5], ids=["one-piece", "pipelined"])
def test_rccl_entry_points_world1(chunks, monkeypatch):
    """all_gather_into_tensor / reduce_scatter_tensor / all_reduce through RCCL on device tensors; ``pipelined``: the chunked
    exchange's async all-gathers (queued at once, waited for one by one) and per-chunk reduce-scatters."""
    monkeypatch.setenv("JMAC_TEST_CHUNKS", str(chunks))
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    ei, et, X, R, G, n, nr, d = _case()
    from jmac_amd.layer import RelationAwareLayer
    torch.manual_seed(11)
    base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    f64 = torch.float64
    p = {k: v.detach().clone().to(f64).requires_grad_(True) for k, v in base.named_parameters()}
    Xc, Rc = X.to(f64).requires_grad_(True), R.to(f64).requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, torch.from_numpy(ei), torch.from_numpy(et), 0.05, "sub", "leaky_relu", True,
                            torch.zeros(d, dtype=f64), torch.ones(d, dtype=f64))
    (ref * G.to(f64)).sum().backward()
    assert_close(ret[0]["out"], ref, 1e-4, 1e-6, "out")
    assert_close(ret[0]["gx"], Xc.grad, 1e-4, 1e-6, "grad_X")
    assert_close(ret[0]["gw"], p["w_att"].grad, 1e-4, 1e-6, "grad w_att")


def test_parts_merge_equals_whole_graph_forward():
    """jmac_softmax_parts_merge_f32: the forward kernel on the sub-graphs of a 5-way split of the SOURCES, merged, equals the
    forward kernel on the whole graph (output and the (max, denominator) the backward reads) -- hubs of 900 in-edges,
    destinations with no edge in some or all parts, an empty part."""
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    dev = torch.device("cuda", 0)
    ei, et, X, R, G, n, nr, d = _case()
    gen = torch.Generator().manual_seed(2)
    P, QZ = (torch.randn(n, d, generator=gen) * 0.3).to(dev), (torch.randn(n, 2 * d, generator=gen) * 0.3).to(dev)
    RR, a = (torch.randn(nr + 1, 2 * d, generator=gen) * 0.3).to(dev), (torch.randn(d, generator=gen) * 0.3).to(dev)
    eit, ett = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
    whole = RelGraph(eit, ett, n, nr + 1, None)
    want, wm, wl = ops.rel_attn_split_fwd_raw(P, QZ, RR, a, whole, 0.05, 1.0, -1, 0)
    cut = [0, 90, 90, 300, 301, n]                                    # source ranges; the second one is empty
    parts = []
    for c in range(5):
        sel = (eit[1] >= cut[c]) & (eit[1] < cut[c + 1])
        if int(sel.sum()) == 0:
            continue
        g = RelGraph(eit[:, sel].contiguous(), ett[sel].contiguous(), n, nr + 1, None)
        o, m, l = ops.rel_attn_split_fwd_raw(P, QZ, RR, a, g, 0.05, 1.0, -1, 0)
        parts.append((o, m, l, g.rowptr))
    assert len(parts) == 4
    nb, M, L = ops.softmax_parts_merge(parts, n, d, dev)
    deg = np.bincount(ei[0], minlength=n)
    has = torch.from_numpy(deg > 0).to(dev)
    assert_close(nb.cpu(), want.cpu().double(), 1e-5, 1e-6, "merged output")
    assert torch.equal(M[has], wm[has]) and bool((M[~has] == float("-inf")).all())
    assert_close(L[has].cpu(), wl[has].cpu().double(), 1e-5, 1e-7, "merged denominator")
    # with the self table: the layer's fused epilogue out_scale * (nb + Z[i] - Rz[loop]) = the forward kernel's own fused form
    fused, _, _ = ops.rel_attn_split_fwd_raw(P, QZ, RR, a, whole, 0.05, 0.5, nr, 0)
    pre, M2, L2 = ops.softmax_parts_merge(parts, n, d, dev, QZ[:, d:], RR[-1, d:].contiguous(), 0.5)
    assert_close(pre.cpu(), fused.cpu().double(), 1e-5, 1e-6, "merged fused output")
    assert torch.equal(M2, M) and torch.equal(L2, L)
    only_self, _, _ = ops.softmax_parts_merge([], n, d, dev, QZ[:, d:], RR[-1, d:].contiguous(), 0.5)
    assert_close(only_self.cpu(), (0.5 * (QZ[:, d:] - RR[-1, d:])).cpu().double(), 1e-6, 1e-7, "self term alone")
    nb0, M0, L0 = ops.softmax_parts_merge([], n, d, dev)              # no part at all: zeros / -inf / 0
    assert float(nb0.abs().max()) == 0.0 and bool((M0 == float("-inf")).all()) and float(L0.abs().max()) == 0.0


# ---- config 5 (OpenEA 15K shape, alignment only, 2 ranks): query-sharded scoring on the HIP kernels ------------
def _align_case():
    gen = torch.Generator().manual_seed(21)
    n, d = 10500, 300                                   # the 70 % test split of 15K links; N = 30 000 entity table
    base = torch.randn(n, d, generator=gen)
    e1 = base + 0.9 * torch.randn(n, d, generator=gen)
    e2 = base + 0.9 * torch.randn(n, d, generator=gen)
    table = torch.nn.functional.normalize(torch.randn(30000, d, generator=gen), 2, -1)
    ill = torch.randperm(30000, generator=gen)[:3000]
    return e1, e2, table, ill


def _score_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jmac_amd.dist import sharded_alignment_test, sharded_get_neg
        dev = torch.device("cuda", 0)
        e1, e2, table, ill = _align_case()
        neg = sharded_get_neg(ill.tolist(), table.to(dev), table.to(dev), 25)      # DBPv1 get_neg: one table, k = 25
        res = sharded_alignment_test(e1.to(dev), e2.to(dev), (1, 5, 10), csls_k=10)
        torch.cuda.synchronize()
        ret[rank] = (neg.cpu(), res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_scoring_two_ranks_config5_shape():
    from jmac_amd import scoring
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_score_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    e1, e2, table, ill = _align_case()
    dev = torch.device("cuda", 0)
    want_neg = scoring.get_neg(ill.tolist(), table.to(dev), table.to(dev), 25).cpu()
    want = scoring.alignment_test(e1.to(dev), e2.to(dev), (1, 5, 10), csls_k=10)
    assert bool((want_neg.view(-1, 25)[:, 0] == ill).all())          # a row's nearest neighbour in its own table is itself
    for r in range(world):
        neg, res = ret[r]
        assert torch.equal(neg, want_neg)                            # index work: bit-exact against the one-rank path
        assert res[0] == want[0] and res[1] == want[1]
        assert abs(res[2] - want[2]) < 1e-9 and abs(res[3] - want[3]) < 1e-12
    assert want[1][0] > 50.0                                         # the planted alignment is recovered


@pytest.mark.parametrize("n,e,slabs", [(700, 9000, [0, 100, 100, 333, 700]), (70000, 200000, [0, 1, 35000, 69999, 70000])],
                         ids=["small-form", "persistent-form"])
def test_phased_backward_equals_one_call_bitwise(n, e, slabs):
    """jmac_rel_attn_aggregate_bwd_phases_f32 -- pass A + pass C + their merges in one call, then pass B slab by slab over uneven
    (and one empty) ranges of the source rows -- leaves the bits of the one-call backward in every output: the same kernels on
    the same items, each output row summed in the same order."""
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    dev = torch.device("cuda")
    nr, d = 11, 300
    rng = np.random.default_rng(n)
    ei, et = random_graph(rng, n, nr, e, hub=900)
    ei[1, : e // 20] = 5                                               # a hub SOURCE: its d[Q|Z] row is a merge of partial rows
    gen = torch.Generator().manual_seed(n)
    P, QZ = (torch.randn(n, d, generator=gen) * 0.3).to(dev), (torch.randn(n, 2 * d, generator=gen) * 0.3).to(dev)
    RR, a = (torch.randn(nr + 1, 2 * d, generator=gen) * 0.3).to(dev), (torch.randn(d, generator=gen) * 0.1).to(dev)
    G = torch.randn(n, d, generator=gen).to(dev)
    g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nr + 1)
    out, m, l = ops.rel_attn_split_fwd_raw(P, QZ, RR, a, g, 0.05, 0.5, nr, 0)
    want = ops.rel_attn_split_bwd_raw(P, QZ, RR, a, g, 0.05, 0.5, nr, 0, out, m, l, G)
    bp = ops.SplitBackwardPhases(P, QZ, RR, a, g, 0.05, 0.5, nr, 0, out, m, l, G, slabs)
    bp.dQZ.fill_(float("nan"))
    bp.begin()
    torch.cuda.synchronize()
    assert torch.equal(bp.dP, want[0]) and torch.equal(bp.dRR, want[2]) and torch.equal(bp.da, want[3])
    for c in range(len(slabs) - 1):
        piece = bp.slab(c)
        torch.cuda.synchronize()
        assert torch.equal(piece, want[1][slabs[c]:slabs[c + 1]]), c
        assert bool(torch.isnan(bp.dQZ[slabs[c + 1]:]).all())          # a slab call writes its own rows only
    assert torch.equal(bp.dQZ, want[1])


def test_world8_rehearsal_on_one_gpu_small_scale():
    """bench_dist.rehearse_world (round 6): rank 0's share of a weak-scaled graph at world 8 in ONE process -- sources uniform over a
    table eight times the rank's rows, filled locally, no collective -- at 2 % of config 4's size.  The checks bench.py runs on the
    19 GB table before it reports `sharded.rehearsal_world8`: softmax normalisation through every split-segment merge, the
    backward's conservation laws, slab-pipelined forward == one-piece forward, phased backward == one-call backward bit for bit."""
    import argparse
    import bench_dist
    a = argparse.Namespace(dim=300, synth_scale=0.02)
    r = bench_dist.rehearse_world(a, torch.device("cuda"), world=8, chunks=4, check=True)
    assert r["world"] == 8 and r["table_rows"] == 8 * r["local_rows"] and r["local_edges"] == 400_000
    c = r["checks"]
    assert c["ok"], c
    assert c["phased_equals_one_call_bitwise"] and c["pipelined_ok"] and c["softmax_normalisation_ok"] and c["conservation_ok"]
    assert r["pipelined"]["chunks"] == 4 and len(r["phased_bwd"]["slab_ms"]) == 4
    assert 0.0 < r["phased_bwd"]["pass_b_share"] < 1.0
