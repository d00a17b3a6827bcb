"""config 4: forward / backward aggregation time against the schedule chunk, interleaved repetitions in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth, ops
from jmac_amd.graph import RelGraph
d = 300
n, e, nr = 1_000_000, 20_000_000, 1000
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=1234)
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = (torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
RR = (torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
a = (torch.randn(d, device=dev, generator=gen) * 0.1).requires_grad_(True)
G = torch.randn(n, d, device=dev, generator=gen)
fb, bb = synth.fwd_algorithmic_bytes(n, e, d), synth.bwd_algorithmic_bytes(n, e, d)
eit, ett = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
chunks = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,128,256,512".split(","))]
do_bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
graphs = {}
for c in chunks:
    g = RelGraph(eit, ett, n, nrel, c)
    if do_bwd: g.ensure_backward_views()
    graphs[c] = g
def ev(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
tf = {c: [] for c in chunks}; tb = {c: [] for c in chunks}
for rep in range(7):
    for c in chunks:
        g = graphs[c]
        if do_bwd:
            ms, out = ev(lambda: ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1))
            msb, _ = ev(lambda: torch.autograd.grad(out, [PQZ, RR, a], G))
            tb[c].append(msb)
        else:
            with torch.no_grad():
                ms, _ = ev(lambda: ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1))
        tf[c].append(ms)
for c in chunks:
    f = np.median(tf[c][1:])
    line = "chunk %4d items=%8d splits=%6d  fwd median %.3f ms (min %.3f) %.3f of 8TB/s" % (c, graphs[c].by_dst.n_items_max, graphs[c].by_dst.n_splits_max, f, min(tf[c]), fb / f / 1e6 / 8000)
    if do_bwd:
        b = np.median(tb[c][1:])
        line += "  | bwd median %.3f ms (min %.3f) %.3f (8d bytes)" % (b, min(tb[c]), bb / b / 1e6 / 8000)
    print(line)
