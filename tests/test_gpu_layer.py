"""GPU parity: the HIP RelationAwareLayer (through the C ABI) against the golden vectors captured from
the reference and against the CPU oracle on seeded random graphs.  Tolerance: 1e-4 relative
(BASELINE.json north_star), fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.jmac_oracle as orc
from util import expand_rel_act, rel_rows, LAYER_CASES, assert_close, layer_grads, layer_params, load_golden, make_args, random_graph, rel_err, t

RTOL = 1e-4


def _make_layer(g, cls_name="RelationAwareLayer", mode=1, chunk=None):
    from jmac_amd import layer as jl
    d = int(g["d"])
    args = make_args(float(g["slope"]))
    if cls_name == "RelationalAwareLayer":
        lay = jl.RelationalAwareLayer(d, d, int(g["nr"]), rel_dim=d, act=torch.tanh, args=args)
    else:
        lay = jl.RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=args)
    sd = {k: v for k, v in layer_params(g).items()}
    missing = lay.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"bn.running_mean", "bn.running_var", "bn.num_batches_tracked"}
    lay.bwd_mode = mode
    if chunk:
        lay.chunk = chunk
    return lay.cuda()


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("case", LAYER_CASES + ["layer_dbpv1"])
def test_layer_matches_reference_golden(case, mode):
    g = load_golden(case)
    lay = _make_layer(g, "RelationalAwareLayer" if case == "layer_dbpv1" else "RelationAwareLayer", mode,
                      chunk=4 if case == "layer_tiny" else None)
    X = t(g["X"], "cuda").requires_grad_(True)
    R = t(g["R"], "cuda").requires_grad_(True)
    ei, et = t(g["edge_index"], "cuda"), t(g["edge_type"], "cuda")
    lay.train()
    with torch.no_grad():
        pre = lay.pre_bn(X, R, ei, et)
    out = lay(X, R, ei, et)
    assert_close(out, g["out_train"], RTOL, 1e-6, "out_train")
    assert_close(lay.bn.running_mean, g["running_mean_after"], RTOL, 1e-7, "running_mean")
    assert_close(lay.bn.running_var, g["running_var_after"], RTOL, 1e-7, "running_var")
    (out * t(g["G"], "cuda")).sum().backward()
    gscale = float(np.abs(g["grad_R"]).max()) if g["grad_R"].size else 0.0
    assert_close(X.grad, g["grad_X"], RTOL, 1e-6, "grad_X")
    assert_close(R.grad, g["grad_R"], RTOL, 1e-6, "grad_R")
    for name, ref in layer_grads(g).items():
        got = dict(lay.named_parameters())[name].grad
        got = got if got is not None else torch.zeros_like(ref)
        atol = 1e-4 * gscale + 1e-6 if name == "loop_rel" else 1e-6
        assert_close(got, ref, RTOL, atol, "grad " + name)
    lay.eval()
    with torch.no_grad():
        assert_close(lay(X, R, ei, et), g["out_eval"], RTOL, 1e-6, "out_eval")
    # the fused kernel output before BN equals the oracle's (nb + self)/2
    _, _, pre_ref = orc.layer_pre_bn(layer_params(g), t(g["X"]), t(g["R"]), t(g["edge_index"]), t(g["edge_type"]),
                                     float(g["slope"]), "sub", "relu" if case == "layer_dbpv1" else "leaky_relu")
    assert_close(pre, pre_ref, RTOL, 1e-7, "pre_bn")


def _oracle_case(n, nr, d, e, seed, hub=None, slope=0.05):
    rng = np.random.default_rng(seed)
    ei, et = random_graph(rng, n, nr, e, hub=hub)
    gen = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=gen) * (4 / np.sqrt(d))
    R = torch.randn(nr, d, generator=gen) * (4 / np.sqrt(d))
    G = torch.randn(n, d, generator=gen)
    return torch.from_numpy(ei), torch.from_numpy(et), X, R, G


def _kink_flips(lay, p64, X64, R64, ei, et, Xg, Rg, captured=None):
    """Number of attention pre-activations h_e[k] whose SIGN differs between the fp32 tables on the GPU and the
    float64 oracle.  The forward is continuous across the LeakyReLU kink, the gradient is not (slope 1 vs 0.05): with
    E*d ~ 10^6 values of O(1) and fp32 rounding ~1e-6, an element landing on the other side is a matter of chance,
    and where it happens the two gradients legitimately differ (measured: fp32-CPU vs f64 differ by 2e-2 in one
    gradient at d=512).  Such cases are compared at a looser gradient tolerance; the forward tolerance never changes."""
    d = X64.shape[1]
    with torch.no_grad():
        rel_flips = 0
        if captured and "layer.tables" in captured:      # the fused node: the tables IT gathered (a recomputation through
            PQZ, RR = captured["layer.tables"]            # the op-by-op products may round differently), and the relation
            dp = d                                        # transform's own LeakyReLU (src/jmac_model.py:41), same kind of kink
            pre64 = torch.cat([R64, p64["loop_rel"]], 0) @ p64["rel_transform_weight1"]
            nr = R64.shape[0]                             # (the node reports the rows its edges name: compact relation side)
            rows = rel_rows(captured.get("layer.rel_used"), nr)
            rel_flips = int((expand_rel_act(captured["layer.rel_act"], captured.get("layer.rel_used"), nr)[rows] != (pre64 > 0)[rows]).sum())
        else:
            rel32 = lay.transform_relations(Rg)
            PQZ, RR, _, dp = lay._tables(Xg, rel32)
        dst, src = ei[0].cuda(), ei[1].cuda()
        h32 = (PQZ[dst, :d] + PQZ[src, dp:dp + d] - RR[et.cuda(), :d]).cpu()
        rel64 = orc.transform_relations(p64, R64, 0.05, "leaky_relu")
        wt, wb = p64["w_att"][:d], p64["w_att"][d:]
        h64 = (X64 @ wt)[ei[0]] + (X64 @ wb)[ei[1]] - (rel64 @ wb)[et]
    return int(((h32 > 0) != (h64 > 0)).sum()) + rel_flips


def _check_random_layer(n, nr, d, e, hub, chunk, mode, seed, lay_seed, strict):
    """Oracle evaluated in float64: an fp32 CPU evaluation can land on the other side of a LeakyReLU kink
    (observed at d=512: fp32-CPU vs f64 differ by 2e-2 in grad w_att while HIP-fp32 vs f64 agree to 1e-6)."""
    from jmac_amd.layer import RelationAwareLayer
    ei, et, X, R, G = _oracle_case(n, nr, d, e, seed=seed, hub=hub)
    torch.manual_seed(lay_seed)
    lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    with torch.no_grad():
        lay.bn.weight.uniform_(0.5, 1.5)
        lay.bn.bias.uniform_(-0.2, 0.2)
    f64 = torch.float64
    p = {k: v.detach().clone().to(f64).requires_grad_(True) for k, v in lay.named_parameters()}
    Xc, Rc = X.clone().to(f64).requires_grad_(True), R.clone().to(f64).requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, ei, et, 0.05, "sub", "leaky_relu", True,
                            torch.zeros(d, dtype=f64), torch.ones(d, dtype=f64))
    (ref * G.to(f64)).sum().backward()
    lay = lay.cuda()
    lay.bwd_mode, lay.chunk = mode, chunk
    Xg, Rg = X.cuda().requires_grad_(True), R.cuda().requires_grad_(True)
    from jmac_amd import encoder
    captured = {}
    encoder.CAPTURE = captured
    try:
        out = lay(Xg, Rg, ei.cuda(), et.cuda())
    finally:
        encoder.CAPTURE = None
    assert_close(out, ref, RTOL, 1e-6, "out")
    (out * G.cuda()).sum().backward()
    flips = _kink_flips(lay, {k: v.detach() for k, v in p.items()}, Xc.detach(), Rc.detach(), ei, et, Xg.detach(), Rg.detach(),
                        captured)
    if strict:
        assert flips == 0, flips                                     # seed chosen flip-free (tools/closed/flip_probe.py)
    assert flips <= 3, flips                                         # a handful at most out of e*d pre-activations
    grtol = RTOL if flips == 0 else 5e-2                             # see _kink_flips (one flip moved grad w_att by 1.4 %)
    assert_close(Xg.grad, Xc.grad, grtol, 1e-6, "grad_X")
    assert_close(Rg.grad, Rc.grad, grtol, 1e-6, "grad_R")
    gscale = Rc.grad.abs().max().item()
    for name, prm in lay.named_parameters():
        atol = 1e-4 * gscale + 1e-6 if name == "loop_rel" else 1e-6
        assert_close(prm.grad, p[name].grad, grtol, atol, "grad " + name)


@pytest.mark.parametrize("n,nr,d,e,hub,chunk", [
    (600, 25, 300, 5000, 700, 64),      # d=300 (BASELINE dim), a hub split into 11 chunks
    (500, 17, 256, 4000, 300, 128),     # d=256 (reference default)
    (300, 9, 128, 2500, None, 256),
    (257, 6, 20, 1500, 200, 32),        # NCH=1, tiny d
    (200, 8, 512, 900, None, 256),      # max d
    (150, 5, 30, 700, 100, 16),         # d % 4 != 0 -> host pads to 32
])
@pytest.mark.parametrize("mode", [1, 0])
def test_layer_matches_oracle_random(n, nr, d, e, hub, chunk, mode):
    _check_random_layer(n, nr, d, e, hub, chunk, mode, seed=n + d, lay_seed=d, strict=False)


@pytest.mark.parametrize("seed", [101, 102, 103])
@pytest.mark.parametrize("n,nr,d,e,hub,chunk", [(600, 25, 300, 5000, 700, 64), (500, 17, 256, 4000, 300, 128)])
def test_layer_gradients_at_1e4_on_flip_free_seeds(n, nr, d, e, hub, chunk, seed):
    """The BASELINE dims (d=300, and the reference's default d=256) with NO tolerance escape: these seeds have no attention
    pre-activation (and no pre-activation of the relation transform's own LeakyReLU) whose sign differs between the fp32
    tables the layer node gathered and the float64 oracle (tools/closed/flip_probe.py scanned seeds 100-139 on the round-3 node:
    0 flips for 100-104 at d=300 and for 101-106 at d=256), so forward and every gradient must meet 1e-4 outright."""
    _check_random_layer(n, nr, d, e, hub, chunk, 1, seed=seed, lay_seed=seed, strict=True)


def test_deterministic_backward_is_bitwise_reproducible():
    from jmac_amd.layer import RelationAwareLayer
    ei, et, X, R, G = _oracle_case(400, 12, 300, 6000, seed=5, hub=500)
    torch.manual_seed(0)
    lay = RelationAwareLayer(300, 300, rel_dim=300, act=torch.tanh, args=make_args()).cuda()
    lay.chunk = 64
    grads = []
    for _ in range(2):
        Xg, Rg = X.cuda().requires_grad_(True), R.cuda().requires_grad_(True)
        lay.zero_grad()
        (lay(Xg, Rg, ei.cuda(), et.cuda()) * G.cuda()).sum().backward()
        grads.append([Xg.grad.clone(), Rg.grad.clone(), lay.w_att.grad.clone(), lay.a_att.grad.clone()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_edge_permutation_invariance_and_chunking():
    """Same multiset of edges in a different COO order / different chunk size -> same result (to rounding)."""
    from jmac_amd.layer import RelationAwareLayer
    ei, et, X, R, _ = _oracle_case(300, 7, 64, 3000, seed=9, hub=400)
    torch.manual_seed(1)
    lay = RelationAwareLayer(64, 64, rel_dim=64, act=torch.tanh, args=make_args()).cuda().eval()
    outs = []
    for chunk, perm in ((256, None), (16, torch.randperm(3000)), (1024, torch.randperm(3000))):
        lay.chunk = chunk
        e2, t2 = (ei, et) if perm is None else (ei[:, perm], et[perm])
        with torch.no_grad():
            outs.append(lay.pre_bn(X.cuda(), R.cuda(), e2.cuda().contiguous(), t2.cuda().contiguous()))
    assert_close(outs[1], outs[0], 1e-5, 1e-7)
    assert_close(outs[2], outs[0], 1e-5, 1e-7)


@pytest.mark.parametrize("n,nr,d,e,hub", [(120, 5, 16, 500, None), (300, 7, 30, 2500, 900), (64, 3, 8, 0, None)])
def test_comp_op_mult(n, nr, d, e, hub):
    """comp_op='mult' (src/jmac_model.py:61-64): the fused kernel on per-edge rows (layer._pre_bn_mult) against the oracle,
    forward and every gradient; and against the second implementation on the torch_scatter-compatible kernels."""
    from jmac_amd.layer import RelationAwareLayer
    ei, et, X, R, G = _oracle_case(n, nr, d, e, seed=3, hub=hub)
    torch.manual_seed(2)
    lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args(comp_op="mult"))
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in lay.named_parameters()}
    Xc, Rc = X.double().requires_grad_(True), R.double().requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, ei, et, 0.05, "mult", "leaky_relu", True)
    (ref * G.double()).sum().backward()
    lay = lay.cuda()
    Xg, Rg = X.cuda().requires_grad_(True), R.cuda().requires_grad_(True)
    out = lay(Xg, Rg, ei.cuda(), et.cuda())
    assert_close(out, ref, RTOL, 1e-6)
    (out * G.cuda()).sum().backward()
    flips = 0
    with torch.no_grad():                                            # LeakyReLU kinks (see _kink_flips)
        if e:
            rel = lay.transform_relations(Rg)
            h32 = torch.mm(torch.cat((Xg[ei[0].cuda()], Xg[ei[1].cuda()] * rel[et.cuda()]), dim=1), lay.w_att).cpu()
            rel64 = orc.transform_relations(p, Rc, 0.05, "leaky_relu")
            h64 = torch.cat((Xc[ei[0]], Xc[ei[1]] * rel64[et]), dim=1) @ p["w_att"]
            flips = int(((h32 > 0) != (h64 > 0)).sum())
    gtol = RTOL if flips == 0 else 5e-2
    assert flips <= 3
    assert_close(Xg.grad, Xc.grad, gtol, 1e-6, "grad_X")
    assert_close(Rg.grad, Rc.grad, gtol, 1e-6, "grad_R")
    for k, v in lay.named_parameters():
        if p[k].grad is not None and v.grad is not None:
            assert_close(v.grad, p[k].grad, gtol, 1e-6, "grad_" + k)
    with torch.no_grad():
        rel = lay.transform_relations(Rg)
        alt = lay._pre_bn_unfactorised(Xg, rel, ei.cuda(), et.cuda())
        assert_close(lay._pre_bn_mult(Xg, rel, ei.cuda(), et.cuda()), alt, RTOL, 1e-6, "fused vs scatter form")


@pytest.mark.parametrize("n_dst,n_src,e", [(500, 100, 4000), (70000, 64, 30000)], ids=["small", "persistent-form"])
def test_split_tables_fewer_source_rows_than_destinations_no_loop(n_dst, n_src, e):
    """Bipartite / sharded callers of the split-table entry point (include/jmac_hip.h: P [N_dst,d], QZ [N_src,2d]) with
    FEWER source rows than destinations and no loop relation (loop_rel = -1): the fused self row QZ[self_off + i] must not be
    read at all (the forward kernels used to load it unconditionally -- out of bounds for i >= N_src).  Result
    against the oracle on one [N_dst, 3d] table whose Q|Z rows past N_src no edge names."""
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    d, nr = 300, 9
    rng = np.random.default_rng(n_dst + e)
    ei = np.stack([rng.integers(0, n_dst, e), rng.integers(0, n_src, e)]).astype(np.int64)
    ei[0, : e // 10] = 7                                              # a hub destination (split items)
    et = rng.integers(0, nr, e).astype(np.int64)
    gen = torch.Generator().manual_seed(e)
    P, QZ = torch.randn(n_dst, d, generator=gen) * 0.3, torch.randn(n_src, 2 * d, generator=gen) * 0.3
    RR, a = torch.randn(nr, 2 * d, generator=gen) * 0.3, torch.randn(d, generator=gen) * 0.1
    G = torch.randn(n_dst, d, generator=gen)
    dev = torch.device("cuda")
    g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n_dst, nr, num_src=n_src)
    Pg, QZg, Rg, ag = (t.to(dev).requires_grad_(True) for t in (P, QZ, RR, a))
    out = ops.rel_attn_aggregate_split(Pg, QZg, Rg, ag, g, 0.05, 1.0, -1, 0)
    out.backward(G.to(dev))
    PQZ = torch.zeros(n_dst, 3 * d, dtype=torch.float64)
    PQZ[:, :d] = P.double()
    PQZ[:n_src, d:] = QZ.double()
    PQZ.requires_grad_(True)
    R64, a64 = RR.double().requires_grad_(True), a.double().requires_grad_(True)
    ref = orc.aggregate_from_tables(PQZ, R64, a64, torch.from_numpy(ei), torch.from_numpy(et), 0.05, -1, 1.0)
    (ref * G.double()).sum().backward()
    assert_close(out, ref, RTOL, 1e-7, "out")
    with torch.no_grad():
        eig, etg = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
        h32 = (Pg[eig[0]] + (QZg[eig[1], :d] - Rg[etg, :d])).cpu()
        h64 = PQZ[ei[0], :d] + PQZ[ei[1], d:2 * d] - R64[et, :d]
        flips = int(((h32 > 0) != (h64 > 0)).sum())
    gtol = RTOL if flips == 0 else 5e-2
    assert flips <= 3, flips
    assert_close(Pg.grad, PQZ.grad[:, :d], gtol, 1e-9, "dP")
    assert_close(QZg.grad, PQZ.grad[:n_src, d:], gtol, 1e-9, "dQZ")
    assert_close(Rg.grad, R64.grad, gtol, 1e-9, "dRR")
    assert_close(ag.grad, a64.grad, gtol, 1e-9, "da")


@pytest.mark.parametrize("kind", ["no_edges", "star", "duplicates_and_self_edges", "two_nodes_one_edge", "one_relation_everywhere"])
@pytest.mark.parametrize("d", [300, 64])
def test_layer_on_degenerate_graphs(kind, d):
    """Graphs at the edge of what the schedules see (empty segments, single-edge segments, duplicate edges):
    no edge at all (every segment empty: the layer is tanh(BN((X - r_loop'') W_gcn / 2))), every edge into ONE destination from
    every other node (one split segment, all others empty), the same edge repeated and edges from a node to itself (the reference
    counts each occurrence: message_passing.py:24-28 index on positions, not on pairs), two nodes joined by one edge, and one
    relation id on every edge (one hot row of [Rq|Rz]).  Forward and every gradient against the float64 oracle."""
    from jmac_amd.layer import RelationAwareLayer
    rng = np.random.default_rng(len(kind) + d)
    n, nr = (2, 3) if kind == "two_nodes_one_edge" else (97, 6)
    if kind == "no_edges":
        ei, et = np.zeros((2, 0), np.int64), np.zeros(0, np.int64)
    elif kind == "star":
        src = np.arange(1, n)
        ei, et = np.stack([np.zeros_like(src), src]), rng.integers(0, nr, n - 1)
    elif kind == "duplicates_and_self_edges":
        base = np.stack([rng.integers(0, n, 40), rng.integers(0, n, 40)])
        loops = np.stack([np.arange(10), np.arange(10)])
        ei = np.concatenate([base, base, base[:, :7], loops, loops[:, :3]], axis=1)
        et = np.concatenate([rng.integers(0, nr, 40)] * 2 + [rng.integers(0, nr, 7), rng.integers(0, nr, 10), rng.integers(0, nr, 3)])
    elif kind == "two_nodes_one_edge":
        ei, et = np.array([[1], [0]]), np.array([2])
    else:
        ei, et = np.stack([rng.integers(0, n, 500), rng.integers(0, n, 500)]), np.full(500, 4)
    ei, et = torch.from_numpy(ei.astype(np.int64)), torch.from_numpy(et.astype(np.int64))
    gen = torch.Generator().manual_seed(d + n)
    X = torch.randn(n, d, generator=gen) * (4 / np.sqrt(d))
    R = torch.randn(nr, d, generator=gen) * (4 / np.sqrt(d))
    G = torch.randn(n, d, generator=gen)
    torch.manual_seed(d)
    lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    f64 = torch.float64
    p = {k: v.detach().clone().to(f64).requires_grad_(True) for k, v in lay.named_parameters()}
    Xc, Rc = X.clone().to(f64).requires_grad_(True), R.clone().to(f64).requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, ei, et, 0.05, "sub", "leaky_relu", True, torch.zeros(d, dtype=f64), torch.ones(d, dtype=f64))
    (ref * G.to(f64)).sum().backward()
    lay = lay.cuda()
    Xg, Rg = X.cuda().requires_grad_(True), R.cuda().requires_grad_(True)
    out = lay(Xg, Rg, ei.cuda(), et.cuda())
    (out * G.cuda()).sum().backward()
    assert torch.isfinite(out).all()
    # two rows under train-mode BatchNorm: the normalised values are +-1 whatever the inputs, the gradients through the statistics
    # are differences of nearly equal numbers -- forward only there
    assert_close(out, ref, RTOL, 1e-6, "out")
    if kind != "two_nodes_one_edge":
        assert_close(Xg.grad, Xc.grad, 5e-2, 1e-6, "grad_X")           # (tolerance of the random-graph test when a kink flips)
        assert_close(Rg.grad, Rc.grad, 5e-2, 1e-6, "grad_R")
        # gradients that are mathematically zero (no edge: nothing reaches the relation transforms but the loop row, whose constant
        # shift per column train-mode BatchNorm removes) come out as rounding noise on both sides: absolute floor from the scale
        gscale = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None)
        for name, prm in lay.named_parameters():
            if name != "loop_rel":
                assert_close(prm.grad, p[name].grad, 5e-2, 1e-5 * gscale + 1e-6, "grad " + name)
