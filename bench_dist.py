"""Destination-sharded synthetic workload for bench.py (--gpus N > 1, or --workload synth-1m at N = 1).

BASELINE config 4 scaled weakly: every rank owns ``n_loc`` = 1M x scale entities and the ``e_loc`` = 20M x scale
triples that point INTO them (power-law in-degree inside the rank's range, sources uniform over the whole
graph, relation types Zipf over 1 000), so per-GPU work is fixed as N grows ("weak") and the 8-GPU run is
the 8M-entity / 160M-triple graph.  One step = two stacked RelationAwareLayers forward + backward over the
whole graph (jmac_amd.dist.ShardedRelationAwareLayer: per layer one all-gather of the [Q|Z] table over xGMI,
one [2,d] all-reduce for the BN statistics; backward one reduce-scatter per layer) + gradient all-reduce +
fused Adam.  value = 2 layers x E_total / step time.
"""
import os
import time
import types

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0


def _local_graph(rank, world, n_loc, e_loc, nr, seed=1234):
    """Edges whose destination is owned by `rank`; ids are global."""
    from jmac_amd import synth
    ei, et, _, _ = synth.power_law_graph(n_loc, e_loc, nr, seed=seed + 17 * rank)
    rng = np.random.default_rng(seed + 1000 + rank)
    ei[0] += rank * n_loc                                             # own destination range
    ei[1] = rng.integers(0, n_loc * world, size=e_loc, dtype=np.int64)  # sources: anywhere
    return ei, et


STRONG_BLOCKS = 8          # the strong-scaled graph is cut into this many fixed row blocks; a rank owns 8 / world of them


def _strong_graph(rank, world, n_glob, e_glob, nr, seed=4321):
    """north_star's 8-GPU configuration ("a 10x synthetic graph", strong-scaled against one GPU): ONE global graph, the same for
    every world size -- STRONG_BLOCKS fixed row blocks of n_glob / 8 entities and e_glob / 8 triples each (power-law in-degree
    inside the block, sources uniform over the whole graph, types Zipf), generated from the block's own seed; rank r of a world
    of W owns blocks [r 8/W, (r+1) 8/W), i.e. a contiguous destination range with an equal share of the edges (what
    dist.partition_rows gives on this graph).  Only the rank's own edges are generated."""
    from jmac_amd import synth
    if STRONG_BLOCKS % world:
        raise SystemExit("--scaling strong: world size must divide %d" % STRONG_BLOCKS)
    nb, eb = n_glob // STRONG_BLOCKS, e_glob // STRONG_BLOCKS
    per = STRONG_BLOCKS // world
    eis, ets = [], []
    for b in range(rank * per, (rank + 1) * per):
        ei, et, _, _ = synth.power_law_graph(nb, eb, nr, seed=seed + 17 * b)
        rng = np.random.default_rng(seed + 1000 + b)
        ei[0] += b * nb
        ei[1] = rng.integers(0, nb * STRONG_BLOCKS, size=eb, dtype=np.int64)
        eis.append(ei)
        ets.append(et)
    return np.concatenate(eis, axis=1), np.concatenate(ets), nb * per, eb * per


def run_sharded(a, rank, world, device):
    from jmac_amd import ops, synth
    from jmac_amd.dist import ShardedGraph, ShardedRelationAwareLayer, allreduce_grads
    from jmac_amd.layer import RelationAwareLayer
    d, nr = a.dim, 1000
    strong = getattr(a, "scaling", "weak") == "strong"
    if strong:
        # 10x config 4 in triples; 2M entities (not 10M): the [P|Q|Z] tables, their adjoints and two layers' saved activations of
        # 10M x 300 fp32 rows do not fit one GPU's 288 GB beside the per-edge records ("else 2 M / 200 M -- state it")
        n_glob, e_glob = int(200_000 * a.synth_scale), int(20_000_000 * a.synth_scale)
        ei, et, n_loc, e_loc = _strong_graph(rank, world, n_glob, e_glob, nr)
    else:
        n_loc, e_loc = int(1_000_000 * a.synth_scale), int(20_000_000 * a.synth_scale)
        ei, et = _local_graph(rank, world, n_loc, e_loc, nr)
    bounds = np.arange(world + 1, dtype=np.int64) * n_loc
    chunks = max(int(getattr(a, "pipeline_chunks", 0) or 0), 1)      # > 1: the slab-pipelined exchange (jmac_amd.dist)
    sg = ShardedGraph(ei, et, bounds, rank, already_local=True, chunks=chunks)
    del ei, et
    torch.manual_seed(7)                                              # identical replicated parameters on every rank
    largs = types.SimpleNamespace(leaky_relu_w=0.05, comp_op="sub")
    wire = torch.bfloat16 if getattr(a, "wire_bf16", False) else None
    layers = [ShardedRelationAwareLayer(RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=largs).to(device), wire_dtype=wire)
              for _ in range(2)]
    rel = torch.nn.Parameter(torch.randn(nr, d, device=device) * (2.0 / (nr + d)) ** 0.5)
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    x = torch.nn.Parameter(torch.randn(n_loc, d, device=device, generator=gen) * (2.0 / (n_loc * world + d)) ** 0.5)
    target = torch.randn(n_loc, d, device=device, generator=gen)
    shared = [p for l in layers for p in l.parameters()] + [rel]
    opt = torch.optim.Adam(shared + [x], lr=1e-3, fused=True)
    for l in layers:
        l.train()
    sg.rel_graph(device, nr + 1).ensure_backward_views()

    def step():
        opt.zero_grad(set_to_none=True)
        h = x
        for l in layers:
            h = l(h, rel, sg)
        loss = (h * target).mean()
        loss.backward()
        allreduce_grads(shared)
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = tt.item()
    e_total = e_loc * world
    value = 2 * e_total * a.steps / el

    # per-kernel timing of the rank-local aggregation and of the data-path collectives (HIP events on the launch
    # stream, which waits for RCCL's stream), rank 0
    import jmac_amd.dist as jdist
    ops.PROFILE, jdist.COMM_PROFILE = [], []
    nprof = 3
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    crec, jdist.COMM_PROFILE = jdist.COMM_PROFILE, None
    fwd = [e0.elapsed_time(e1) for n, e0, e1 in rec if n == "rel_attn_fwd"]
    # (the overlapped backward of the pipelined exchange records its phases separately: "rel_attn_bwd_phase13", "..._phase2" per slab)
    bwd = [e0.elapsed_time(e1) for n, e0, e1 in rec if n.startswith("rel_attn_bwd")]
    fb = synth.fwd_algorithmic_bytes(n_loc, e_loc, d)
    bb = synth.bwd_algorithmic_bytes(n_loc, e_loc, d)
    # per LAYER: the pipelined exchange runs the forward kernel once per chunk (partial passes over disjoint edge sets)
    fms, bms = float(np.sum(fwd)) / (nprof * len(layers)), float(np.sum(bwd)) / (nprof * len(layers))
    # a record that proves itself: which backend carried the collectives and how many ranks it saw (an all-reduce of ones, not
    # the launcher's WORLD_SIZE); a figure the backend cannot time is null, not 0.0
    backend = dist.get_backend() if (world > 1 and dist.is_initialized()) else None
    seen = None
    if world > 1:
        ones = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        seen = int(round(float(ones.item())))
    comm = {"world": world, "backend": ("rccl (torch 'nccl' on ROCm)" if backend == "nccl" else backend) if world > 1 else "none (one rank: no collective runs)",
            "world_seen_by_all_reduce": seen, "layers": len(layers)}
    per_layer = nprof * len(layers)
    for cname in ("all_gather_qz", "reduce_scatter_dqz"):
        ts = [e0.elapsed_time(e1) for n, e0, e1, _ in crec if n == cname]
        by = [b for n, _, _, b in crec if n == cname]
        if sg.chunks > 1:                                            # pipelined: one record per chunk -> per-layer sums
            tc = [e0.elapsed_time(e1) for n, e0, e1, _ in crec if n == cname + "_chunk"]
            bc = [b for n, _, _, b in crec if n == cname + "_chunk"]
            ts, by = ([sum(tc) / per_layer] if tc else []), ([sum(bc) / per_layer] if bc else [])
        # the all-gather is started before the P-side GEMM and the relation transforms and waited for after them: its
        # figure is launch-to-arrival on the compute stream, i.e. includes the work it overlaps
        comm[cname + "_ms_per_layer"] = float(np.mean(ts)) if ts else None          # None: not timed on this backend / at this world
        comm[cname + "_bytes_per_layer"] = int(np.mean(by)) if by else None
    timed = [comm[c + "_ms_per_layer"] for c in ("all_gather_qz", "reduce_scatter_dqz")]
    comm["collective_ms_per_step"] = (len(layers) * sum(timed)) if all(v is not None for v in timed) else None
    what = ("config 4 x%g STRONG-scaled: ONE global graph of %d entities / %d triples / %d relations (8 fixed row blocks, power-law "
            "in-degree inside a block), the same for every world size; this rank owns %d entities / %d triples"
            % (a.synth_scale, n_loc * world, e_total, nr, n_loc, e_loc)) if strong else \
           ("config 4 weak-scaled: per GPU %d entities / %d triples / %d relations (power-law in-degree)" % (n_loc, e_loc, nr))
    line = {"metric": "gnn_layer_edges_per_s", "value": value, "unit": "edges/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what + ", d=%d; 2 stacked RelationAwareLayers fwd+bwd + grad all-reduce + Adam; "
                                          "destination-sharded, all-gather of [Q|Z] per layer" % d,
                       "global_entities": n_loc * world, "global_triples": e_total, "parallelism": "dst-shard x%d" % world,
                       "wire": "bf16 [Q|Z] all-gather (flag)" if wire is not None else "fp32",
                       "exchange": ("slab-pipelined: %d row chunks, chunk c aggregated while c+1 is on the links" % sg.chunks)
                                   if sg.chunks > 1 else "one-piece all-gather (started before the P projection / relation transforms)",
                       "edges_counted_per_step": 2 * e_total},
            "roofline": {"bound": "hbm", "kernel": "rel_attn_fwd_hw_kernel (half-wave lane map, persistent form; rank 0, local rows)", "achieved": fb / (fms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fb / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": fb, "avg_launch_ms": fms, "launches": len(fwd),
                         "launches_per_layer": sg.chunks,
                         "note": ("avg_launch_ms is the SUM of the %d partial launches of one layer (disjoint edge sets, whole-graph bytes)"
                                  % sg.chunks) if sg.chunks > 1 else None},
            "roofline_bwd": {"bound": "hbm", "achieved": bb / (bms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bb / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": bms,
                             "bytes": "SURVEY 8d backward formula"},
            "comm": comm, "cpu_baseline": None}
    if rank == 0:
        line["scaling_model"] = scaling_model(el / a.steps * 1e3, fms, bms, len(layers), n_loc * world, e_total, d, world, strong,
                                              wire_bytes=2 if wire is not None else 4, chunks=sg.chunks,
                                              rehearsal=getattr(a, "rehearsal", None) if world == 1 else None)
        if not strong:
            # north_star's 8-GPU configuration (a 10x graph, strong-scaled: 2M entities / 200M triples) from THIS run's per-row and
            # per-edge rates -- the graph itself is run by `--scaling strong --synth-scale 10`
            m = scaling_model(el / a.steps * 1e3, fms, bms, len(layers), 2_000_000, 200_000_000, d, 1, True,
                              wire_bytes=2 if wire is not None else 4, chunks=sg.chunks, run_rows=n_loc, run_edges=e_loc,
                              rehearsal=committed_strong_rehearsal(d))
            m["assumptions"]["kind"] = ("strong, 2M entities / 200M triples; extrapolated from the measured per-row / per-edge rates of "
                                        "this run's %d-entity / %d-triple rank" % (n_loc, e_loc))
            del m["measured_here"]
            m["memory_GB_one_gpu"] = step_memory_gb(2_000_000, 200_000_000, d, len(layers))
            if "carried_over" in m:
                m["assumptions"]["aggregation_at_W"] = ("rank 0's share of THIS graph at world 1 / 2 / 4 / 8 measured on one GPU (bench_dist.py "
                                                        "--rehearse-strong -> %s, committed; not collected by this run); the rest of the "
                                                        "step carried over from this run per row" % STRONG_REHEARSAL_FILE)
            line["scaling_model_strong_10x"] = m
            # the other reading of "a 10x graph": 10M entities / 200M triples.  It does not fit ONE GPU (the base of a strong-scaling
            # ratio), which is why the 2M-entity graph is the one that is run; the model for it is printed beside it
            m10 = scaling_model(el / a.steps * 1e3, fms, bms, len(layers), 10_000_000, 200_000_000, d, 1, True,
                                wire_bytes=2 if wire is not None else 4, chunks=sg.chunks, run_rows=n_loc, run_edges=e_loc)
            m10["assumptions"]["kind"] = ("strong, 10M entities / 200M triples (NOT runnable at N = 1: see memory_GB_one_gpu); "
                                          "extrapolated from this run's per-row / per-edge rates")
            del m10["measured_here"]
            m10["memory_GB_one_gpu"] = step_memory_gb(10_000_000, 200_000_000, d, len(layers))
            line["scaling_model_strong_10x_10M_entities"] = m10
    return line


def _ms(fn, n=3, warm=1):
    """Median HIP-event milliseconds of fn() on the current stream."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def rehearse_world(a, device, world=8, chunks=4, check=True):
    """ONE process, no collective: rank 0's share of weak-scaled config 4 at ``world`` ranks -- n_loc local destinations, e_loc
    edges whose sources are uniform over the whole world * n_loc-row [Q|Z] table (19.2 GB fp32 at world 8, d = 300), the table
    filled locally with what the all-gather would have delivered.  Timed, with HIP events on the launch stream:
      (a) the one-piece forward (jmac_rel_attn_aggregate_fwd_f32, fused self loop) and the one-call deterministic backward;
      (b) the slab-pipelined forward: ``chunks`` partial passes over the chunk-major table + jmac_softmax_parts_merge_f32;
      (c) the phased backward: begin() = passes A, C and their merges, then pass B slab by slab (what the reduce-scatters hide behind).
    What scaling_model carried over from the world-1 run (a 1M-row, 2.4 GB table) is measured here at the table the rank would
    really gather from.  check: the size-independent properties of tests/test_gpu_fullsize.py on THIS graph (softmax
    normalisation through every merge, backward conservation laws), pipelined == one-piece, phased == one-call -- the kernels
    index a table of more than 2^32 bytes."""
    from jmac_amd import ops, synth
    from jmac_amd.dist import ShardedGraph, _HipChunked
    d, nr, slope = a.dim, 1000, 0.05
    n_loc, e_loc = int(1_000_000 * a.synth_scale), int(20_000_000 * a.synth_scale)
    ei, et = _local_graph(0, world, n_loc, e_loc, nr)
    bounds = np.arange(world + 1, dtype=np.int64) * n_loc
    sg = ShardedGraph(ei, et, bounds, 0, already_local=True, chunks=1)
    gen = torch.Generator(device=device).manual_seed(77)
    rows = world * n_loc
    QZ = torch.empty(rows, 2 * d, device=device)
    for r0 in range(0, rows, 1 << 20):                                   # filled slab by slab: no second 19 GB temporary
        QZ[r0:r0 + (1 << 20)].normal_(0.0, 0.3, generator=gen)
    P = torch.randn(n_loc, d, device=device, generator=gen) * 0.3
    RR = torch.randn(nr + 1, 2 * d, device=device, generator=gen) * 0.3
    av = torch.randn(d, device=device, generator=gen) * 0.1
    G = torch.randn(n_loc, d, device=device, generator=gen)
    g = sg.rel_graph(device, nr + 1)
    g.ensure_backward_views()
    deg = (g.rowptr[1:] - g.rowptr[:-1]).double()
    res = {"world": world, "local_rows": n_loc, "local_edges": e_loc, "table_rows": rows, "table_GB": rows * 2 * d * 4 / 1e9,
           "what": "rank 0's share of weak-scaled config 4 at world %d in one process: sources uniform over the %d-row [Q|Z] table, "
                   "table filled locally, no collective" % (world, rows)}
    checks = {}
    if check:
        # softmax normalisation: Z = 1, Rz = 0, no self term -> out = sqrt(deg) exactly, through every split-segment merge
        Zsave = QZ[:, d:].clone() if rows * d * 4 < 40e9 else None
        QZ[:, d:] = 1.0
        R1 = RR.clone()
        R1[:, d:] = 0.0
        nb, _, _ = ops.rel_attn_split_fwd_raw(P, QZ, R1, av, g, slope, 1.0, -1, 0)
        want = deg.float().sqrt().view(-1, 1)
        checks["softmax_normalisation_max_err"] = float((nb - want).abs().max())
        checks["softmax_normalisation_ok"] = bool(checks["softmax_normalisation_max_err"] <= 2e-4 * max(1.0, float(want.max())))
        del nb, R1
        QZ[:, d:] = Zsave
        del Zsave
    out, smax, sden = ops.rel_attn_split_fwd_raw(P, QZ, RR, av, g, slope, 0.5, nr, sg.self_off)
    res["one_piece_fwd_ms"] = _ms(lambda: ops.rel_attn_split_fwd_raw(P, QZ, RR, av, g, slope, 0.5, nr, sg.self_off), n=5)
    bw = lambda: ops.rel_attn_split_bwd_raw(P, QZ, RR, av, g, slope, 0.5, nr, sg.self_off, out, smax, sden, G)
    res["one_call_bwd_ms"] = _ms(bw, n=3)
    fb, bb = synth.fwd_algorithmic_bytes(n_loc, e_loc, d), synth.bwd_algorithmic_bytes(n_loc, e_loc, d)
    res["fwd_frac_hbm"] = fb / (res["one_piece_fwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    res["bwd_frac_hbm"] = bb / (res["one_call_bwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    dQZ1_sum = None
    if check:
        dQZ1_sum = bw()[1].sum(0, dtype=torch.float64)                   # column sums of d[Q|Z] of the one-call backward (the 19 GB table is not kept)
    if check:
        # conservation (pure edge op: no self term): sum_j dZ[j] = sum_i sqrt(deg_i) g_i, sum dRz = -sum dZ, sum dQ = sum dP = -sum dRq
        o2, m2, l2 = ops.rel_attn_split_fwd_raw(P, QZ, RR, av, g, slope, 1.0, -1, 0)
        dP, dQZ, dRR, _ = ops.rel_attn_split_bwd_raw(P, QZ, RR, av, g, slope, 1.0, -1, 0, o2, m2, l2, G)
        sdP, sdQ, sdZ = dP.sum(0, dtype=torch.float64), dQZ[:, :d].sum(0, dtype=torch.float64), dQZ[:, d:].sum(0, dtype=torch.float64)
        sdRq, sdRz = dRR[:, :d].sum(0, dtype=torch.float64), dRR[:, d:].sum(0, dtype=torch.float64)
        want_dz = (deg.sqrt().view(-1, 1) * G).sum(0, dtype=torch.float64)

        def rel(x, y):
            return float((x - y).abs().max() / max(float(y.abs().max()), float(x.abs().max()), 1e-30))
        checks["conservation_rel_err"] = {"dZ_vs_sqrtdeg_g": rel(sdZ, want_dz), "dRz_vs_minus_dZ": rel(sdRz, -sdZ),
                                          "dQ_vs_dP": rel(sdQ, sdP), "dRq_vs_minus_dP": rel(sdRq, -sdP)}
        checks["conservation_ok"] = bool(max(checks["conservation_rel_err"].values()) <= 2e-4 + 1e-6 * e_loc ** 0.5)
        del o2, m2, l2, dP, dQZ, dRR
    # ---- (b) the slab-pipelined forward on the chunk-major table (the same values, re-laid: chunk c of every rank in one slice)
    sgc = ShardedGraph(ei, et, bounds, 0, already_local=True, chunks=chunks)
    del ei, et
    cb, C = sgc.chunk_bounds, sgc.chunks
    table = torch.empty(sgc.table_rows, 2 * d, device=device)
    for c in range(C):
        rc = int(cb[c + 1] - cb[c])
        for ow in range(world):
            table[world * int(cb[c]) + ow * rc: world * int(cb[c]) + (ow + 1) * rc] = QZ[ow * sgc.n_max + int(cb[c]): ow * sgc.n_max + int(cb[c + 1])]
    table[world * sgc.n_max:] = QZ[:sgc.n_max]                          # rank 0's own rows once more (fused self loop of the merge)
    del QZ
    for c in range(C):
        sgc.chunk_graph(c, device, nr + 1)
    part_ms = [_ms(lambda c=c: _HipChunked.partial(P, table, RR, av, sgc, c, slope), n=3) for c in range(C)]
    parts = [_HipChunked.partial(P, table, RR, av, sgc, c, slope) for c in range(C)]
    zself = table[sgc.self_off:sgc.self_off + n_loc, d:]
    rz = RR[-1, d:].contiguous()
    merge = lambda: _HipChunked.merge(parts, n_loc, d, device, zself, rz, 0.5)
    res["pipelined"] = {"chunks": C, "partial_ms": part_ms, "merge_ms": _ms(merge, n=3),
                        "table_rows": sgc.table_rows, "table_GB": sgc.table_rows * 2 * d * 4 / 1e9}
    res["pipelined"]["fwd_ms"] = float(sum(part_ms)) + res["pipelined"]["merge_ms"]
    res["pipelined"]["extra_over_one_piece_ms"] = res["pipelined"]["fwd_ms"] - res["one_piece_fwd_ms"]
    pre, pmax, pden = merge()
    if check:
        checks["pipelined_vs_one_piece_max_abs"] = float((pre - out).abs().max())
        checks["pipelined_vs_one_piece_scale"] = float(out.abs().max())
        checks["pipelined_ok"] = bool(checks["pipelined_vs_one_piece_max_abs"] <= 2e-5 * max(1.0, checks["pipelined_vs_one_piece_scale"]))
    del parts
    # ---- (c) the phased backward on the same table: begin() then pass B per slab (last slab = the own-rows copy)
    slabs = [world * int(b) for b in cb] + [int(table.shape[0])]
    gfull = sgc.rel_graph(device, nr + 1)
    gfull.ensure_backward_views()

    def phased():
        bp = _HipChunked.backward_phased(P, table, RR, av, sgc, slope, pre, pmax, pden, G, slabs)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(C + 3)]
        evs[0].record()
        bp.begin()
        evs[1].record()
        for c in range(C + 1):
            bp.slab(c)
            evs[c + 2].record()
        torch.cuda.synchronize()
        return bp, [evs[i].elapsed_time(evs[i + 1]) for i in range(C + 2)]
    phased()
    times, bp = [], None
    for _ in range(3):
        bp = None                                              # one d[Q|Z] table (21.6 GB at world 8) alive at a time
        bp, t = phased()
        times.append(t)
    tms = np.median(np.array(times), axis=0)
    res["phased_bwd"] = {"begin_ms": float(tms[0]), "slab_ms": [float(x) for x in tms[1:C + 1]], "own_rows_slab_ms": float(tms[C + 1]),
                         "total_ms": float(tms.sum()), "pass_b_share": float(tms[1:].sum() / tms.sum()),
                         "extra_over_one_call_ms": float(tms.sum()) - res["one_call_bwd_ms"]}
    if check:
        # the phased form against the one-call backward on the SAME table and forward statistics: bit for bit (same kernels on the
        # same items in the same order per output row); against the one-piece layout (another edge order per source): column sums
        dPc, dTc, dRRc, dac = _HipChunked.backward(P, table, RR, av, sgc, slope, pre, pmax, pden, G)
        checks["phased_equals_one_call_bitwise"] = bool(torch.equal(bp.dP, dPc) and torch.equal(bp.dQZ, dTc) and torch.equal(bp.dRR, dRRc)
                                                        and torch.equal(bp.da, dac))
        tot = bp.dQZ.sum(0, dtype=torch.float64)
        checks["phased_dQZ_column_sums_rel_err_vs_one_piece_layout"] = float((tot - dQZ1_sum).abs().max() / dQZ1_sum.abs().max())
        checks["phased_ok"] = bool(checks["phased_equals_one_call_bitwise"]
                                   and checks["phased_dQZ_column_sums_rel_err_vs_one_piece_layout"] <= 1e-5)
        checks["ok"] = bool(all(v for k, v in checks.items() if k.endswith("_ok")))
        res["checks"] = checks
        del dPc, dTc, dRRc, dac
    res["max_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
    del table, bp, pre, out, P, G
    torch.cuda.empty_cache()
    return res


def rehearse_strong(a, device, worlds=(8, 4, 2, 1), scale=10.0):
    """ONE process, no collective: rank 0's share of the STRONG-scaled graph -- north_star's "10x synthetic graph", 2M entities /
    200M triples at scale 10 (the graph `bench.py --gpus N --scaling strong --synth-scale 10` partitions) -- at each world size:
    n_glob / W local destinations, e_glob / W edges whose sources are uniform over the WHOLE n_glob-row [Q|Z] table (the same
    4.8 GB table at every W, filled locally).  Times the one-piece forward and the one-call backward of the aggregation (HIP
    events, median of 3): what `scaling_model_strong_10x` used to extrapolate from the weak run's per-edge rate (1M rows / 20M
    edges gathering from a 1M-row table) is measured on the shapes a rank really has, the N = 1 base included."""
    from jmac_amd import ops, synth
    from jmac_amd.dist import ShardedGraph
    d, nr, slope = a.dim, 1000, 0.05
    n_glob, e_glob = int(200_000 * scale), int(20_000_000 * scale)
    gen = torch.Generator(device=device).manual_seed(78)
    QZ = torch.empty(n_glob, 2 * d, device=device)
    for r0 in range(0, n_glob, 1 << 20):
        QZ[r0:r0 + (1 << 20)].normal_(0.0, 0.3, generator=gen)
    RR = torch.randn(nr + 1, 2 * d, device=device, generator=gen) * 0.3
    av = torch.randn(d, device=device, generator=gen) * 0.1
    out = {"what": "rank 0's share of the strong-scaled %d-entity / %d-triple graph at each world size in one process: sources uniform "
                   "over the whole %d-row [Q|Z] table (%.1f GB), table filled locally, no collective"
                   % (n_glob, e_glob, n_glob, n_glob * 2 * d * 4 / 1e9), "global_entities": n_glob, "global_triples": e_glob, "worlds": {}}
    for W in worlds:
        t0 = time.perf_counter()
        ei, et, n_loc, e_loc = _strong_graph(0, W, n_glob, e_glob, nr)
        bounds = np.arange(W + 1, dtype=np.int64) * n_loc
        sg = ShardedGraph(ei, et, bounds, 0, already_local=True, chunks=1)
        del ei, et
        g = sg.rel_graph(device, nr + 1)
        g.ensure_backward_views()
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        P = torch.randn(n_loc, d, device=device, generator=gen) * 0.3
        G = torch.randn(n_loc, d, device=device, generator=gen)
        fw = lambda: ops.rel_attn_split_fwd_raw(P, QZ, RR, av, g, slope, 0.5, nr, sg.self_off)
        o, m, l = fw()
        fwd_ms = _ms(fw, n=3)
        bwd_ms = _ms(lambda: ops.rel_attn_split_bwd_raw(P, QZ, RR, av, g, slope, 0.5, nr, sg.self_off, o, m, l, G), n=3)
        fb, bb = synth.fwd_algorithmic_bytes(n_loc, e_loc, d), synth.bwd_algorithmic_bytes(n_loc, e_loc, d)
        out["worlds"][str(W)] = {"local_rows": n_loc, "local_edges": e_loc, "table_rows": n_glob, "fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
                                 "fwd_frac_hbm": fb / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bwd_frac_hbm": bb / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "max_in_degree": int((g.rowptr[1:] - g.rowptr[:-1]).max()), "graph_build_s": build_s}
        del sg, g, P, G, o, m, l
        torch.cuda.empty_cache()
    out["max_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
    w = out["worlds"]
    if "1" in w:
        for W in w:
            w[W]["aggregation_speedup_vs_1"] = (w["1"]["fwd_ms"] + w["1"]["bwd_ms"]) / (w[W]["fwd_ms"] + w[W]["bwd_ms"])
    return out


XGMI_LINK_GBS, XGMI_EFF = 153.0, 0.8     # one xGMI link per peer pair (MI355X: 7 links x ~153 GB/s per GPU), sustained fraction assumed
PASS_B_SHARE = 0.4                       # assumed share of the aggregation backward that is pass B (by source: what a reduce-scatter of
                                         # d[Q|Z] slabs can hide behind); ja-size kernel times: A 28 us, B || C 27 us, merges 11 us


def step_memory_gb(n, e, d, n_layers, hbm_gb=288.0):
    """fp32 bytes one GPU holds for the sharded step at world 1 on an n-entity / e-triple graph (what makes the 10M-entity reading
    of north_star's "10x graph" unrunnable as the N = 1 base): per layer the [Q|Z] table, P, the pre-BN rows and the output kept
    for the backward, plus -- live during one layer's backward -- d[Q|Z], dP, the incoming gradient and the 72-byte per-edge
    records; the input rows with their gradient and two Adam moments; the CSR and its three groupings (int32)."""
    row = d * 4
    per_layer_saved = n * row * (2 + 1 + 1 + 1)            # [Q|Z], P, pre, out
    bwd_live = n * row * (2 + 1 + 1) + e * 72              # d[Q|Z], dP, incoming gradient, per-edge records
    inputs = n * row * 4                                   # x, its gradient, Adam m and v
    graph = e * 4 * 8 + n * 4 * 6                          # col / type / perm / dst-of-slot + two orders + two entry_dst; pointers
    loss_side = n * row * 3                                # this step's regression target, the product h * target, its gradient
    total = n_layers * per_layer_saved + bwd_live + inputs + graph + loss_side
    return {"total": total / 1e9, "saved_activations": n_layers * per_layer_saved / 1e9, "backward_live": bwd_live / 1e9,
            "inputs_and_adam": inputs / 1e9, "graph": graph / 1e9, "loss_side": loss_side / 1e9, "hbm": hbm_gb,
            "fits_in_90_percent_of_hbm": bool(total / 1e9 < 0.9 * hbm_gb),
            "note": "resident fp32 / int32 bytes of the sharded two-layer step at world 1; allocator reserve, GEMM workspaces and the "
                    "partial-row buffers of the split segments come on top"}


STRONG_REHEARSAL_FILE = "profiles/r6_rehearse_strong.json"


def committed_strong_rehearsal(d):
    """rehearse_strong()'s per-world figures for the 2M-entity / 200M-triple graph from the committed run, or None."""
    import json
    import os
    try:
        doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), STRONG_REHEARSAL_FILE)))
        if doc.get("global_entities") != 2_000_000 or doc.get("global_triples") != 200_000_000 or d != 300:
            return None
        return {int(W): v for W, v in doc["worlds"].items()}
    except (OSError, KeyError, ValueError):
        return None


def scaling_model(step_ms, agg_fwd_ms, agg_bwd_ms, n_layers, n_glob, e_glob, d, world, strong, wire_bytes=4, chunks=1,
                  run_rows=None, run_edges=None, rehearsal=None):
    """What this run's own measurements predict for 2 / 4 / 8 GPUs -- an explicit model, NOT a measurement (no multi-GPU node has
    run this code).  Per layer and GPU at world W (destination sharding):
      aggregation   = the measured kernel time x (edges per GPU at W / edges per GPU in this run)       (HBM-bound in E)
      other compute = (step - aggregation) of this run x (rows per GPU at W / rows per GPU in this run)  (N-row GEMMs, BN, Adam)
      all-gather    = every peer sends its (N / W) x 2d x wire-bytes slab over ITS OWN link: (N / W) 2d s / (153 GB/s x 0.8)
      reduce-scatter= the same bytes the other way (fp32)
    no overlap of exchange and compute is assumed beyond what the single-rank step already contains.
    ``pipelined_step_ms``: the same with the slab-pipelined exchange (--pipeline-chunks C, default shown for C = 4 when the run
    itself was one-piece): the forward aggregation hides behind the all-gather except for the first chunk's arrival,
    min(agg_fwd, all-gather x (C-1)/C) per layer, and pays the partial passes' extra traffic, the merge pass and the own-rows
    copy: 7.5 ms per layer and 1M rows / 20M edges per GPU, MEASURED at one rank (profiles/r4_pipeline.txt: 104.4 -> 119.4 ms per
    two-layer step); a run that was itself pipelined already contains that cost.

    ``rehearsal`` (round 6; weak scaling only): {W: rehearse_world(W)} -- the aggregation kernels timed on ONE GPU on the shapes a
    rank has at world W (its 1M rows / 20M edges gathering from the W x 1M-row table).  Where given, the aggregation forward /
    backward, the pipelined form's extra cost and pass B's share of the backward at W are THOSE measurements instead of the
    world-1 figures carried over; the figures carried over stay beside them (``carried_over``) so that the change is visible."""
    n_run, e_run = (run_rows or n_glob / world), (run_edges or e_glob / world)
    # {W: rehearse_world(W)} (weak) or rehearse_strong()["worlds"] (strong, the graph it was measured on only)
    reh = {int(k): v for k, v in (rehearsal or {}).items()}
    if reh and strong != ("fwd_ms" in next(iter(reh.values()))):
        reh = {}                                                # a weak rehearsal says nothing about a strong model and vice versa
    out = {"assumptions": {"xgmi_link_GBps": XGMI_LINK_GBS, "sustained_fraction": XGMI_EFF, "overlap": "none beyond the measured step",
                           "wire_bytes_per_element": wire_bytes, "kind": "strong" if strong else "weak"},
           "measured_here": {"world": world, "step_ms": step_ms, "aggregation_ms": n_layers * (agg_fwd_ms + agg_bwd_ms),
                             "other_ms": max(step_ms - n_layers * (agg_fwd_ms + agg_bwd_ms), 0.0)}}
    carried = _predict(step_ms, agg_fwd_ms, agg_bwd_ms, n_layers, n_glob, e_glob, d, strong, wire_bytes, chunks, n_run, e_run, {})
    if reh:
        used = _predict(step_ms, agg_fwd_ms, agg_bwd_ms, n_layers, n_glob, e_glob, d, strong, wire_bytes, chunks, n_run, e_run, reh)
        out["carried_over"] = {"what": "round 5's model: the world-1 per-edge aggregation rate, the 7.5 ms pipelined overhead and a 0.4 pass-B "
                                       "share carried over to every world size", "predicted": carried[0], "band": carried[1]}
        out["assumptions"]["aggregation_at_W"] = ("measured on one GPU on rank 0's world-W shapes (rehearse_world: W x 1M-row table, no "
                                                  "collective) for W in %s; carried over from world 1 elsewhere" % sorted(reh))
        out["carried_over_vs_rehearsed_speedup"] = {str(W): {"carried_over": carried[0][str(W)]["speedup_vs_1"],
                                                             "rehearsed": used[0][str(W)]["speedup_vs_1"],
                                                             "carried_over_pipelined_bwd_overlap_eff0.8": carried[1]["eff0.8/forward+backward"][str(W)]["speedup_vs_1"],
                                                             "rehearsed_pipelined_bwd_overlap_eff0.8": used[1]["eff0.8/forward+backward"][str(W)]["speedup_vs_1"]}
                                                    for W in (2, 4, 8)}
    else:
        used = carried
    out["predicted"], out["band"] = used
    out["assumptions"]["band"] = ("link efficiency 0.5 / 0.8 of 153 GB/s per link, all 7 links at once, no interference between RCCL and "
                                  "the kernels; 'forward' = the slab-pipelined all-gather (built: --pipeline-chunks), 'forward+backward' = "
                                  "slab-wise reduce-scatters behind pass B with pass B = %.1f of the aggregation backward (rehearsed worlds: "
                                  "the measured share)" % PASS_B_SHARE)
    return out


def _predict(step_ms, agg_fwd_ms, agg_bwd_ms, n_layers, n_glob, e_glob, d, strong, wire_bytes, chunks, n_run, e_run, reh):
    """(predicted, band) of scaling_model; ``reh``: {W: rehearse_world result} overriding the carried-over aggregation figures."""
    agg = n_layers * (agg_fwd_ms + agg_bwd_ms)
    other = max(step_ms - agg, 0.0)

    def at(W, eff):
        """Per GPU at world W: (compute ms, all-gather ms, reduce-scatter ms, forward aggregation ms per layer, pipelined overhead
        per layer, pass-B ms per layer, edges in total)."""
        n_tot, e_tot = (n_glob, e_glob) if strong else (n_run * W, e_run * W)
        n_w, e_w = n_tot / W, e_tot / W
        r = reh.get(W)
        if r is not None and "fwd_ms" in r:                     # rehearse_strong: the aggregation alone, measured on this rank's shapes
            fwd_w, bwd_w = r["fwd_ms"], r["bwd_ms"]
            over = 7.5 * (0.5 * n_w / 1e6 + 0.5 * e_w / 2e7) if chunks == 1 else 0.0
            pass_b, bwd_extra = PASS_B_SHARE * bwd_w, 0.0
        elif r is not None:                                     # rehearse_world: measured on this rank's world-W shapes
            fwd_w, bwd_w = r["one_piece_fwd_ms"], r["one_call_bwd_ms"]
            over = r["pipelined"]["extra_over_one_piece_ms"] if chunks == 1 else 0.0
            pass_b = r["phased_bwd"]["pass_b_share"] * r["phased_bwd"]["total_ms"]
            bwd_extra = max(r["phased_bwd"]["extra_over_one_call_ms"], 0.0)
        else:
            fwd_w, bwd_w = agg_fwd_ms * (e_w / e_run), agg_bwd_ms * (e_w / e_run)
            over = 7.5 * (0.5 * n_w / 1e6 + 0.5 * e_w / 2e7) if chunks == 1 else 0.0        # a pipelined run already paid it
            pass_b, bwd_extra = PASS_B_SHARE * bwd_w, 0.0
        comp = n_layers * (fwd_w + bwd_w) + other * (n_w / n_run)
        slab = n_w * 2 * d
        ag = (slab * wire_bytes) / (XGMI_LINK_GBS * eff * 1e9) * 1e3 if W > 1 else 0.0
        rs = (slab * 4) / (XGMI_LINK_GBS * eff * 1e9) * 1e3 if W > 1 else 0.0
        return comp, ag, rs, fwd_w, over, pass_b, bwd_extra, e_tot

    C = chunks if chunks > 1 else 4
    predicted = {}
    for W in (1, 2, 4, 8):
        comp, ag, rs, fwd_w, over, _, _, e_tot = at(W, XGMI_EFF)
        t = comp + n_layers * (ag + rs)
        hidden = min(fwd_w, ag * (C - 1) / C) if W > 1 else 0.0
        tp = t - n_layers * (hidden - (over if W > 1 else 0.0))
        predicted[str(W)] = {"step_ms": t, "edges_per_s": n_layers * e_tot / (t * 1e-3), "compute_ms": comp,
                             "exchange_ms": n_layers * (ag + rs), "pipelined_step_ms": tp, "pipeline_chunks": C}
    base = predicted["1"]["edges_per_s"]
    for W in ("2", "4", "8"):
        predicted[W]["speedup_vs_1"] = predicted[W]["edges_per_s"] / base
    # the band the single figures above sit in: sustained link efficiency 0.5 / 0.8 x what hides behind the exchange (nothing; the
    # forward aggregation behind the slab-pipelined all-gather; additionally pass B behind slab-wise reduce-scatters)
    band = {}
    for eff in (0.5, 0.8):
        for overlap in ("none", "forward", "forward+backward"):
            row = {}
            for W in (2, 4, 8):
                comp, ag, rs, fwd_w, over, pass_b, bwd_extra, e_tot = at(W, eff)
                t = comp + n_layers * (ag + rs)
                if overlap != "none":
                    t -= n_layers * (min(fwd_w, ag * (C - 1) / C) - over)
                if overlap == "forward+backward":
                    t -= n_layers * (min(pass_b, rs * (C - 1) / C) - bwd_extra)
                row[str(W)] = {"step_ms": t, "speedup_vs_1": (n_layers * e_tot / (t * 1e-3)) / base}
            band["eff%.1f/%s" % (eff, overlap)] = row
    return predicted, band


def main():
    """python bench_dist.py --rehearse-world 8 [--pipeline-chunks 4] [--synth-scale 1.0] [--dim 300]: the one-GPU rehearsal of a
    rank's world-W share alone (one JSON line; bench.py's default run carries the same object as sharded.rehearsal_world8)."""
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--rehearse-world", type=int, default=8)
    ap.add_argument("--pipeline-chunks", type=int, default=4)
    ap.add_argument("--synth-scale", type=float, default=1.0)
    ap.add_argument("--dim", type=int, default=300)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--rehearse-strong", action="store_true",
                    help="instead: rank 0's share of the strong-scaled 2M / 200M graph at world 8, 4, 2, 1 (aggregation forward / backward)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_dist.py needs an MI355X (no CPU fallback for the product path)")
    if a.rehearse_strong:
        res = rehearse_strong(a, torch.device("cuda", 0), scale=10.0 * a.synth_scale)
    else:
        res = rehearse_world(a, torch.device("cuda", 0), a.rehearse_world, a.pipeline_chunks, check=not a.no_check)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
