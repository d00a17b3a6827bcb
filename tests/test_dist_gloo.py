"""CPU, world_size 2, gloo: the destination-sharded layer (jmac_amd/dist.py) -- partitioning, padded
all-gather layout, reduce-scatter adjoint, synchronised BN statistics, gradient all-reduce -- reproduces the
single-process oracle.  The rank-local aggregation is a torch stand-in INJECTED by this test (the product's
default is the HIP op; there is no CPU fallback in jmac_amd)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle.jmac_oracle as orc
from util import make_args, random_graph


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _standin_aggregate(P, QZ, RR, a, sg, slope):
    """Test double for the HIP kernel: the factorised edge formula with the oracle's scatter ops."""
    d = P.shape[1]
    dst, src, typ = torch.from_numpy(sg.dst_local), torch.from_numpy(sg.src_padded), torch.from_numpy(sg.edge_type)
    diff = QZ[src] - RR[typ]
    h = P[dst] + diff[:, :d]
    s = torch.nn.functional.leaky_relu(h, slope) @ a.view(-1, 1)
    alpha = orc.scatter_softmax(s, dst, sg.n_local)
    deg = orc.scatter_sum(torch.ones(dst.shape[0]), dst, sg.n_local)
    nb = orc.scatter_sum(alpha * diff[:, d:], dst, sg.n_local) * deg.sqrt().view(-1, 1)
    # like the product's op (an autograd node on every rank, edges or not), the result always hangs on its inputs: a rank
    # without rows must still run the backward collectives its peers wait in (the oracle's scatter helpers return a
    # constant for empty inputs, which would let autograd prune them on that rank)
    return nb + 0.0 * (QZ.sum() + P.sum() + RR.sum() + a.sum())


class _StandinChunked:
    """Test double for the pipelined path's three kernels (jmac_amd.dist._HipChunked): the partial aggregation of one source
    chunk's edges, the per-destination merge of the partials, the whole-graph backward -- torch on the CPU."""

    @staticmethod
    def partial(P, table, RR, a, sg, chunk, slope):
        d = P.shape[1]
        sel = sg.src_chunk == chunk
        dst, src, typ = (torch.from_numpy(x[sel]) for x in (sg.dst_local, sg.src_padded, sg.edge_type))
        # only rows of THIS chunk's slice of the chunk-major table may be read: the later chunks have not arrived yet
        lo, hi = sg.world * int(sg.chunk_bounds[chunk]), sg.world * int(sg.chunk_bounds[chunk + 1])
        assert ((src >= lo) & (src < hi)).all()
        diff = table[src] - RR[typ]
        s = (torch.nn.functional.leaky_relu(P[dst] + diff[:, :d], slope) @ a.view(-1, 1)).view(-1)
        n = sg.n_local
        m = torch.full((n,), float("-inf")).scatter_reduce(0, dst, s, "amax", include_self=True)
        w = torch.exp(s - m[dst])
        l = torch.zeros(n).index_add_(0, dst, w)
        deg = torch.bincount(dst, minlength=n)
        o = torch.zeros(n, d).index_add_(0, dst, w.view(-1, 1) * diff[:, d:])
        out = o * (deg.float().sqrt() / l.clamp_min(1e-30)).view(-1, 1)
        rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).to(torch.int32)
        return out, m, l, rowptr

    @staticmethod
    def merge(parts, n, d, device, zself, rz_loop, out_scale):
        self_term = (zself - rz_loop) if zself is not None else torch.zeros(n, d)
        if not parts or n == 0:
            return out_scale * self_term, torch.full((max(n, 1),), float("-inf")), torch.zeros(max(n, 1))
        deg = torch.stack([(rp[1:] - rp[:-1]).float() for _, _, _, rp in parts])                 # [C, n]
        m = torch.stack([p[1] for p in parts])
        l = torch.stack([p[2] for p in parts])
        M = torch.where(deg > 0, m, torch.full_like(m, float("-inf"))).max(0).values
        f = torch.where(deg > 0, torch.exp(m - M) * l, torch.zeros_like(l))
        L = f.sum(0)
        w = torch.where((deg > 0) & (L > 0), f / L.clamp_min(1e-30) * (deg.sum(0).sqrt() / deg.clamp_min(1).sqrt()), torch.zeros_like(f))
        nb = sum(w[c].view(-1, 1) * parts[c][0] for c in range(len(parts)))
        return out_scale * (nb + self_term), M, L

    @staticmethod
    def backward(P, table, RR, a, sg, slope, pre, seg_max, seg_den, G):
        d = P.shape[1]
        with torch.enable_grad():
            xs = [t.detach().clone().requires_grad_(True) for t in (P, table, RR, a)]
            nb = _standin_aggregate(*xs, sg, slope)
            out = (nb + xs[1][sg.self_off:sg.self_off + sg.n_local, d:] - xs[2][-1, d:]) * 0.5
            # the merged forward result must BE the whole-graph layer pre-activation (what the HIP backward is handed as ``out``)
            assert torch.allclose(out.detach(), pre, atol=2e-5), float((out.detach() - pre).abs().max())
            return torch.autograd.grad(out, xs, G)


    @staticmethod
    def backward_phased(P, table, RR, a, sg, slope, pre, seg_max, seg_den, G, slab_bounds):
        """The phased form (dist._HipChunked.backward_phased): begin() = everything but d table, slab(c) = rows
        [slab_bounds[c], slab_bounds[c+1]) of d table.  A slab may be asked for only after begin(), in order, once."""
        import types
        dP, dT, dRR, da = _StandinChunked.backward(P, table, RR, a, sg, slope, pre, seg_max, seg_den, G)
        st = types.SimpleNamespace(dP=None, dQZ=torch.full_like(dT, float("nan")), dRR=None, da=None, next=0)

        def begin():
            st.dP, st.dRR, st.da = dP, dRR, da

        def slab(c):
            assert st.dP is not None and c == st.next
            st.next += 1
            r0, r1 = slab_bounds[c], slab_bounds[c + 1]
            st.dQZ[r0:r1] = dT[r0:r1]                    # rows of later slabs are still NaN: a reduce-scatter that read ahead would show
            return st.dQZ[r0:r1]
        st.begin, st.slab = begin, slab
        return st


class _StandinBN:
    """Test double for the phased BN + tanh kernels (include/jmac_hip.h): same contract, torch on the CPU."""

    @staticmethod
    def moments(x):
        m = x.mean(0)
        return m, ((x - m) ** 2).sum(0)

    @staticmethod
    def apply(x, weight, bias, mean, invstd):
        return torch.tanh((x - mean) * invstd * weight + bias)

    @staticmethod
    def bwd_sums(x, y, gy, mean, invstd):
        gz = gy * (1 - y * y)
        return torch.cat([gz.sum(0), (gz * (x - mean) * invstd).sum(0)])

    @staticmethod
    def bwd_apply(x, y, gy, weight, mean, invstd, sums, n_total):
        d = x.shape[1]
        gz = gy * (1 - y * y)
        xh = (x - mean) * invstd
        return weight * invstd * (gz - (sums[:d] + xh * sums[d:]) / n_total)


def _case(seed=3, n=90, nr=7, d=16, e=700):
    rng = np.random.default_rng(seed)
    ei, et = random_graph(rng, n, nr, e, hub=120)
    gen = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=gen) * 0.8
    R = torch.randn(nr, d, generator=gen) * 0.8
    G = torch.randn(n, d, generator=gen)
    return ei, et, X, R, G, n, nr, d


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jmac_amd.dist import ShardedGraph, ShardedRelationAwareLayer, allreduce_grads, partition_rows
        from jmac_amd.layer import RelationAwareLayer
        ei, et, X, R, G, n, nr, d = _case()
        bounds = partition_rows(np.bincount(ei[0], minlength=n), world)
        if os.environ.get("JMAC_TEST_BOUNDS"):          # hand-made ranges: uneven, and a rank that owns no row at all
            bounds = np.array([int(x) for x in os.environ["JMAC_TEST_BOUNDS"].split(",")], dtype=np.int64)
        chunks = int(os.environ.get("JMAC_TEST_CHUNKS", "1"))
        sg = ShardedGraph(ei, et, bounds, rank, chunks=chunks)
        torch.manual_seed(11)
        base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
        lay = ShardedRelationAwareLayer(base, local_aggregate=_standin_aggregate, chunk_kernels=_StandinChunked,
                                        bn_kernels=_StandinBN if os.environ.get("JMAC_TEST_FUSED_BN") else None,
                                        wire_dtype=torch.bfloat16 if os.environ.get("JMAC_TEST_WIRE_BF16") else None).train()
        x = X[sg.lo:sg.hi].clone().requires_grad_(True)
        r = R.clone().requires_grad_(True)
        out = lay(x, r, sg)
        (out * G[sg.lo:sg.hi]).sum().backward()
        params = list(base.parameters()) + [r]
        allreduce_grads(params)
        import jmac_amd.dist as jd
        ret[rank] = dict(overlapped=min(jd.OVERLAP_COUNT, jd.HANDOFF_COUNT), lo=sg.lo, hi=sg.hi, out=out.detach(), gx=x.grad, gr=r.grad,
                         grads={k: v.grad.clone() for k, v in base.named_parameters()},
                         rm=base.bn.running_mean.clone(), rv=base.bn.running_var.clone(),
                         e_local=sg.E_local, n_max=sg.n_max)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fused_bn,world,bounds,chunks", [(False, 2, None, 1), (True, 2, None, 1), (True, 4, "0,25,25,60,90", 1),
                                                          (False, 4, "0,1,40,41,90", 1), (True, 2, None, 3),
                                                          (True, 4, "0,25,25,60,90", 4), (False, 4, "0,1,40,41,90", 2)])
def test_sharded_layer_equals_single_process_oracle(fused_bn, world, bounds, chunks, monkeypatch):
    """fused_bn: BatchNorm + tanh through sync_bn_tanh (per-rank moments, all-gather, Chan combination, all-reduced
    backward sums) with the kernels' torch stand-in; otherwise the plain torch formulation.  world 4 with hand-made row
    ranges: uneven shards, single-row shards and a rank that owns NO row (its table slab is all padding, its BN moments
    carry weight zero in the Chan combination, its reduce-scatter slice is empty).  chunks > 1: the same through the
    slab-pipelined exchange (uneven chunk heights, chunks without edges, a rank without rows)."""
    if bounds:
        monkeypatch.setenv("JMAC_TEST_BOUNDS", bounds)
    else:
        monkeypatch.delenv("JMAC_TEST_BOUNDS", raising=False)
    # chunks > 1: the slab-pipelined exchange -- chunk-major table, one all-gather per row chunk, per-chunk partial
    # aggregations merged per destination, one reduce-scatter per chunk slice in the backward
    monkeypatch.setenv("JMAC_TEST_CHUNKS", str(chunks))
    if fused_bn:
        monkeypatch.setenv("JMAC_TEST_FUSED_BN", "1")
    else:
        monkeypatch.delenv("JMAC_TEST_FUSED_BN", raising=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    ei, et, X, R, G, n, nr, d = _case()
    from jmac_amd.layer import RelationAwareLayer
    torch.manual_seed(11)
    base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    p = {k: v.detach().clone().requires_grad_(True) for k, v in base.named_parameters()}
    Xc, Rc = X.clone().requires_grad_(True), R.clone().requires_grad_(True)
    rm, rv = torch.zeros(d), torch.ones(d)
    ref = orc.layer_forward(p, Xc, Rc, torch.from_numpy(ei), torch.from_numpy(et), 0.05, "sub", "leaky_relu", True, rm, rv)
    (ref * G).sum().backward()
    assert sum(ret[r]["e_local"] for r in range(world)) == ei.shape[1]
    assert ret[0]["lo"] == 0 and ret[world - 1]["hi"] == n and all(ret[r]["hi"] == ret[r + 1]["lo"] for r in range(world - 1))
    if bounds:
        assert [ret[r]["lo"] for r in range(world)] + [n] == [int(x) for x in bounds.split(",")]
    for r in range(world):
        o = ret[r]
        lo, hi = o["lo"], o["hi"]
        # chunks > 1: the backward took the overlapped form (pass B slab by slab, each slab reduce-scattered as it completes)
        assert o["overlapped"] == (1 if chunks > 1 else 0), (r, o["overlapped"])
        assert torch.allclose(o["out"], ref[lo:hi].detach(), atol=2e-5), r
        assert torch.allclose(o["gx"], Xc.grad[lo:hi], atol=2e-4, rtol=1e-3), r
        assert torch.allclose(o["gr"], Rc.grad, atol=2e-4, rtol=1e-3)
        for k, g in o["grads"].items():
            assert torch.allclose(g, p[k].grad, atol=5e-4, rtol=1e-3), k
        assert torch.allclose(o["rm"], rm, atol=1e-6) and torch.allclose(o["rv"], rv, atol=1e-6)


def test_partition_rows_balances_edges():
    from jmac_amd.dist import ShardedGraph, partition_rows
    rng = np.random.default_rng(0)
    deg = rng.zipf(2.0, 5000).clip(max=800)
    for world in (1, 2, 4, 8):
        b = partition_rows(deg, world)
        assert b[0] == 0 and b[-1] == 5000 and (np.diff(b) >= 0).all()
        work = np.array([deg[b[i]:b[i + 1]].sum() + (b[i + 1] - b[i]) for i in range(world)])
        assert work.max() <= work.mean() + deg.max() + 1
    # padded source index space
    ei = np.stack([rng.integers(0, 100, 400), rng.integers(0, 100, 400)])
    b = np.array([0, 30, 100])
    sg = ShardedGraph(ei, rng.integers(0, 5, 400), b, 1)
    assert sg.n_max == 70 and sg.n_local == 70
    owner = (ei[1][(ei[0] >= 30)] >= 30).astype(int)
    src = ei[1][ei[0] >= 30]
    assert (sg.src_padded == owner * 70 + (src - b[owner])).all()
    # chunk-major layout of the pipelined exchange: a bijection of (owner, local row) onto [0, world * n_max), chunk c of every
    # rank inside the slice [world * cb[c], world * cb[c+1]) -- the slice ONE all-gather of that row chunk fills
    with pytest.raises(ValueError):                      # more chunks than one merge call takes (JMAC_MERGE_MAX_PARTS): refused
        ShardedGraph(ei, rng.integers(0, 5, 400), b, 1, chunks=17)    # at construction, before any collective is queued
    for chunks in (2, 3, 7, 16):
        sgc = ShardedGraph(ei, rng.integers(0, 5, 400), b, 1, chunks=chunks)
        cb = sgc.chunk_bounds
        assert cb[0] == 0 and cb[-1] == sgc.n_max and (np.diff(cb) >= 0).all() and sgc.chunks == min(chunks, sgc.n_max)
        pos = {}
        for ow in range(2):
            for loc in range(int(b[ow + 1] - b[ow])):
                c = int(np.searchsorted(cb, loc, side="right") - 1)
                q = 2 * cb[c] + ow * (cb[c + 1] - cb[c]) + (loc - cb[c])
                assert 2 * cb[c] <= q < 2 * cb[c + 1]
                pos[(ow, loc)] = int(q)
        assert len(set(pos.values())) == len(pos) and max(pos.values()) < 2 * sgc.n_max
        assert all(pos[(int(o), int(s_ - b[o]))] == int(q) for o, s_, q in zip(owner, src, sgc.src_padded))
        assert sum(sgc.chunk_edges(c) for c in range(sgc.chunks)) == sgc.E_local


@pytest.mark.timeout(300)
def test_bf16_wire_format_rounds_only_the_gathered_table(monkeypatch):
    """wire_dtype=bfloat16: the [Q|Z] table crosses the all-gather as bf16 and is widened on arrival.  The sharded
    result must equal the single-process layer evaluated with Q and Z rounded to bf16 -- the ONLY change -- at fp32
    tolerance (the rank's own P, the relation tables, logits, sums and BN stay fp32), and the distance to the fp32
    layer is the stated price of the flag: ~2^-9 relative on the gathered rows, a few 1e-3 on the tanh-bounded output."""
    world = 2
    monkeypatch.setenv("JMAC_TEST_WIRE_BF16", "1")
    monkeypatch.setenv("JMAC_TEST_FUSED_BN", "1")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    ei, et, X, R, G, n, nr, d = _case()
    from jmac_amd.layer import RelationAwareLayer
    torch.manual_seed(11)
    base = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    p = {k: v.detach().clone().requires_grad_(True) for k, v in base.named_parameters()}
    eit, ett = torch.from_numpy(ei), torch.from_numpy(et)
    Xg, Rg = X.clone().requires_grad_(True), R.clone().requires_grad_(True)
    # single-process restatement with the wire rounding applied to Q | Z and nothing else; the rounding is a straight-through
    # step in the backward (the product reduce-scatters the fp32 gradient of the widened table as it stands)
    rel = orc.transform_relations(p, Rg, 0.05, "leaky_relu")
    wt, wb, wg = p["w_att"][:d], p["w_att"][d:], p["gcn_weight"]
    P = Xg @ wt
    QZ = torch.cat([Xg @ wb, Xg @ wg], 1)
    QZ = QZ + (QZ.detach().to(torch.bfloat16).float() - QZ.detach())
    RR = torch.cat([rel @ wb, rel @ wg], 1)
    dst, src = eit[0], eit[1]
    diff = QZ[src] - RR[ett]
    sc = torch.nn.functional.leaky_relu(P[dst] + diff[:, :d], 0.05) @ p["a_att"]
    alpha = orc.scatter_softmax(sc, dst, n)
    deg = orc.scatter_sum(torch.ones(dst.shape[0]), dst, n)
    nb = orc.scatter_sum(alpha * diff[:, d:], dst, n) * deg.sqrt().view(-1, 1)
    # self term: this test's torch stand-in adds the rank's OWN fp32 Z rows; the product's fused kernel reads them from
    # the gathered table instead (jmac_amd/dist.py hip_local_layer), i.e. bf16-rounded like every other gathered row
    pre = (nb + Xg @ wg - RR[-1, d:]) * 0.5
    want = torch.tanh(torch.nn.functional.batch_norm(pre, None, None, p["bn.weight"], p["bn.bias"], True, 0.0, 1e-5))
    (want * G).sum().backward()
    with torch.no_grad():
        fp32 = orc.layer_forward({k: v.detach() for k, v in p.items()}, X, R, eit, ett, 0.05, "sub", "leaky_relu", True,
                                 torch.zeros(d), torch.ones(d))
    got = torch.cat([ret[r]["out"] for r in range(world)])
    assert torch.allclose(got, want.detach(), atol=3e-5), float((got - want).abs().max())
    err = float((got - fp32).abs().max())
    assert 1e-5 < err < 3e-2, err                     # the flag is not free -- and not wild either
    # the gradient THROUGH the exchange (round 2 shipped a wire path whose backward received zeros: forward-only checks
    # cannot see that) -- same tolerances as test_sharded_layer_equals_single_process_oracle
    for r in range(world):
        o = ret[r]
        lo, hi = o["lo"], o["hi"]
        assert torch.allclose(o["gx"], Xg.grad[lo:hi], atol=2e-4, rtol=1e-3), (r, float((o["gx"] - Xg.grad[lo:hi]).abs().max()))
        assert torch.allclose(o["gr"], Rg.grad, atol=2e-4, rtol=1e-3), float((o["gr"] - Rg.grad).abs().max())
        for k, g in o["grads"].items():
            assert torch.allclose(g, p[k].grad, atol=5e-4, rtol=1e-3), (k, float((g - p[k].grad).abs().max()))


def _two_consumer_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        import jmac_amd.dist as jd
        from jmac_amd.dist import ShardedGraph, chunked_aggregate, chunked_all_gather, partition_rows
        ei, et, X, R, G, n, nr, d = _case()
        sg = ShardedGraph(ei, et, partition_rows(np.bincount(ei[0], minlength=n), world), rank, chunks=3)
        gen = torch.Generator().manual_seed(5)
        P = torch.randn(sg.n_local, d, generator=gen)
        RR = torch.randn(nr + 1, 2 * d, generator=gen)
        a = torch.randn(d, generator=gen)
        W = torch.randn(sg.table_rows, 2 * d, generator=torch.Generator().manual_seed(50 + rank))
        res = {}
        for order in ("aggregate_first", "second_first"):
            for overlap in (True, False):
                jd.OVERLAP_BACKWARD, jd.HANDOFF_COUNT, jd.OVERLAP_COUNT = overlap, 0, 0
                x = torch.randn(sg.n_local, 2 * d, generator=torch.Generator().manual_seed(9 + rank)).requires_grad_(True)
                table = chunked_all_gather(F.pad(x, (0, 0, 0, sg.n_max - sg.n_local)), sg, None)
                if order == "aggregate_first":
                    pre = chunked_aggregate(P, table, RR, a, sg, 0.05, kernels=_StandinChunked)
                    second = (table * W).sum()
                else:                                       # the node order of the graph decides whose gradient reaches the
                    second = (table * W).sum()              # table's input buffer first (and which one is accumulated in place)
                    pre = chunked_aggregate(P, table, RR, a, sg, 0.05, kernels=_StandinChunked)
                ((pre * G[sg.lo:sg.hi]).sum() + second).backward()
                res[(order, overlap)] = (x.grad.clone(), jd.OVERLAP_COUNT, jd.HANDOFF_COUNT)
        jd.OVERLAP_BACKWARD = True
        # the hand-off itself, without relying on which buffer the autograd engine accumulates into: an untouched d table takes
        # the shortcut, one that was added to in place afterwards is reduced in full
        ctx = types.SimpleNamespace(sg=sg, group=None)
        g = torch.randn(sg.table_rows, 2 * d, generator=torch.Generator().manual_seed(70 + rank))
        full = jd._ChunkedAllGather.backward(ctx, g.clone())[0]
        marker = torch.full((sg.n_max, 2 * d), 7.0)
        g1 = g.clone()
        g1._jmac_reduced = (marker, g1._version, g1.data_ptr())
        took = jd._ChunkedAllGather.backward(ctx, g1)[0]
        g2 = g.clone()
        g2._jmac_reduced = (marker, g2._version, g2.data_ptr())
        g2.add_(1.0)                                        # what AccumulateGrad-style in-place summation does
        full2 = jd._ChunkedAllGather.backward(ctx, g2)[0]
        res["handoff"] = (took is marker, torch.equal(full2, jd._ChunkedAllGather.backward(ctx, g + 1.0)[0]), bool((full2 != 7.0).any()),
                          getattr(g2, "_jmac_reduced", None) is None, torch.isfinite(full).all().item())
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_table_with_a_second_consumer_keeps_both_gradients():
    """the overlapped backward hands the already reduce-scattered gradient to _ChunkedAllGather.backward through
    an attribute on d table.  With a second differentiable consumer of the public chunked_all_gather table, autograd sums the two
    gradients -- in place onto d table when it arrives first -- and the stale hand-off would drop the second consumer's share.
    The hand-off is taken only for an untouched d table (version + storage); otherwise the complete sum is reduced in full."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_two_consumer_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        for order in ("aggregate_first", "second_first"):
            g_over, n_over, n_hand = ret[r][(order, True)]
            g_plain, n_over0, _ = ret[r][(order, False)]
            assert n_over == 1 and n_over0 == 0 and n_hand <= 1
            assert torch.isfinite(g_over).all()
            assert torch.allclose(g_over, g_plain, atol=1e-5, rtol=1e-5), (r, order, float((g_over - g_plain).abs().max()))
        assert ret[r]["handoff"] == (True, True, True, True, True), ret[r]["handoff"]
