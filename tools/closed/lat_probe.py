"""Forward-kernel duration on small graphs, CPU overhead removed: 20 launches captured in one hipGraph.
Shapes: all rows empty, every row degree k, the ja profile.  Usage: lat_probe.py [d]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from jmac_amd import synth, ops
from jmac_amd.graph import RelGraph
d = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
nrel = 961
gen = torch.Generator(device=dev).manual_seed(0)


def run(name, ei, et, n, bwd=False):
    e = ei.shape[1]
    g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
    PQZ = (torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3).requires_grad_(bwd)
    RR = (torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3).requires_grad_(bwd)
    a = (torch.randn(d, device=dev, generator=gen) * 0.1).requires_grad_(bwd)
    G = torch.randn(n, d, device=dev, generator=gen)
    if bwd:
        g.ensure_backward_views()
        out = ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
        fn = lambda: torch.autograd.grad(out, [PQZ, RR, a], G, retain_graph=True)
        iters = 100
    else:
        from jmac_amd._lib import lib, ptr, stream, check
        L = lib(); sc = g.by_dst
        out = torch.empty((n, d), device=dev); smax = torch.empty(n, device=dev); sden = torch.empty(n, device=dev)
        wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(sc.n_parts_max, d)); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        st = stream()
        args = (ptr(PQZ), 3 * d, PQZ.data_ptr() + d * 4, 3 * d, ptr(RR), 2 * d, ptr(a), ptr(g.col), ptr(g.etype), C.byref(sc.view()), n, d, 0.05,
                nrel - 1, 0, 0.5, ptr(out), d, ptr(smax), ptr(sden), ptr(ws), wsb, st)
        fn = lambda: L.jmac_rel_attn_aggregate_fwd_f32(*args)
        iters = 300
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fb = synth.fwd_algorithmic_bytes(n, e, d)
    print("%-28s N=%6d E=%7d items=%6d %s %.1f us  (%.0f GB/s alg)" % (name, n, e, g.by_dst.n_items_max, "bwd" if bwd else "fwd", us, fb / us / 1e3))


rng = np.random.default_rng(0)
which = os.environ.get("PROBE", "all")
for n in (2048, 11805, 47220):
    if which in ("all", "shape"):
        et1 = rng.integers(0, nrel - 1, 1)
        run("empty rows", np.array([[0], [1]], dtype=np.int64), et1.astype(np.int64), n)
        for k in (1, 2, 4, 8):
            dst = np.repeat(np.arange(n), k); src = rng.integers(0, n, n * k)
            run("degree %d" % k, np.stack([dst, src]).astype(np.int64), rng.integers(0, nrel - 1, n * k).astype(np.int64), n)
for lang in ("el", "ja", "en"):
    ei, et, n, _ = synth.dbp5l_like(lang, 1234)
    run(lang, ei, et, n)
    if which in ("all", "bwd"):
        run(lang, ei, et, n, bwd=True)
ei, et, n, _ = synth.dbp5l_like("ja", 1234, bidirectional=True)
run("ja bidir", ei, et, n)
