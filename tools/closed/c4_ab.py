"""config-4 forward aggregation: A/B of schedule-level options in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth, ops, graph as G
from jmac_amd.graph import RelGraph
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
d = 300
n, e, nr = int(1_000_000 * scale), int(20_000_000 * scale), 1000
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=1234)
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
a = torch.randn(d, device=dev, generator=gen) * 0.1
fb = synth.fwd_algorithmic_bytes(n, e, d)
eit, ett = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
def t(g, reps=8):
    with torch.no_grad():
        for _ in range(2): ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))
res = {}
for rnd in range(2):
    for srt in (False, True):
        G.SORT_ROWS_BY_TYPE = srt
        g = RelGraph(eit, ett, n, nrel)
        med, mn = t(g)
        print("round %d sort_by_type=%-5s chunk=%d  median %.3f ms (min %.3f)  %.3f of 8 TB/s" % (rnd, srt, g.chunk, med, mn, fb / med / 1e6 / 8000))
        del g
