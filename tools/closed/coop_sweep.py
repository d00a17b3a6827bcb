"""ja forward kernel under the cooperative-split thresholds (COOP_MIN) -- schedule-level knob, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from jmac_amd import synth, graph as G
from jmac_amd.graph import RelGraph
from jmac_amd._lib import lib, ptr, stream
d = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda"); nrel = 961
gen = torch.Generator(device=dev).manual_seed(0)
def run(name, ei, et, n):
    e = ei.shape[1]
    g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
    PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
    RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
    a = torch.randn(d, device=dev, generator=gen) * 0.1
    L = lib(); sc = g.by_dst
    out = torch.empty((n, d), device=dev); smax = torch.empty(n, device=dev); sden = torch.empty(n, device=dev)
    wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(sc.n_parts_max, d)); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    args = (ptr(PQZ), 3 * d, PQZ.data_ptr() + d * 4, 3 * d, ptr(RR), 2 * d, ptr(a), ptr(g.col), ptr(g.etype), C.byref(sc.view()), n, d, 0.05, nrel - 1, 0, 0.5, ptr(out), d, ptr(smax), ptr(sden), ptr(ws), wsb, stream())
    fn = lambda: L.jmac_rel_attn_aggregate_fwd_f32(*args)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
    fb = synth.fwd_algorithmic_bytes(n, e, d)
    print("%-10s coop_min=%2d N=%6d E=%6d items=%6d coop=%5d empty=%6d  %.2f us  %.3f of 8 TB/s" % (name, G.COOP_MIN, n, e, sc.n_items_max, sc.n_coop, sc.n_empty, best, fb / best / 1e3 / 8000))
graphs = [("ja",) + synth.dbp5l_like("ja", 1234)[:3], ("ja-bidir",) + synth.dbp5l_like("ja", 1234, True)[:3], ("en",) + synth.dbp5l_like("en", 1234)[:3]]
for cm in (3, 4, 6, 8, 12, 16):
    G.COOP_MIN = cm
    for name, ei, et, n in graphs:
        run(name, ei, et, n)
