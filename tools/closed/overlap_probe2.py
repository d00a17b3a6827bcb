"""Do the aggregation backward's three launches (latency-bound at DBP-5L size, small footprint) overlap with N-row library
GEMMs when they sit on two streams / two branches of a hipGraph?  (tools/closed/overlap_probe.py asked the same of the grouped
relation-side GEMM, whose blocks fill a CU's LDS: no.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from jmac_amd import ops
from jmac_amd.graph import RelGraph
from jmac_amd.data import edges_from_triples, load_dbp5l_arrays
dev = torch.device("cuda")
z = load_dbp5l_arrays(bench.REAL_DATA)
ei, et = edges_from_triples(z["ja.train"], False)
N, nr, d = int(z["ja.num_entity"]), int(z["n_relation_lines"]) + 2, 300
ei_t, et_t = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
g = RelGraph(ei_t, et_t, N, nr)
g.ensure_backward_views()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=gen)
from jmac_amd import encoder
PQZ, RR, a, G = r(N, 3 * d) * 0.3, r(nr, 2 * d) * 0.3, r(d) * 0.1, r(N, d)
out, smax, sden = encoder._agg_fwd(PQZ, RR, a, g, 0.05)
X, Wc, P, dP, dX = r(N, d), r(d, 3 * d), r(N, 3 * d), r(N, 3 * d), r(N, d)
side = torch.cuda.Stream()

def agg_bwd():
    encoder._agg_bwd(PQZ, RR, a, g, 0.05, out, smax, sden, G)

def gemms():
    torch.mm(dP, Wc.t(), out=dX)
    torch.mm(X, Wc, out=P)

def serial():
    agg_bwd(); gemms()

def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        agg_bwd()
    gemms()
    main.wait_stream(side)

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def graphed(fn, reps=4):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    return lambda: gr.replay()

ga, gg = graphed(agg_bwd), graphed(gemms)
print("graph: aggregation backward alone %6.1f us   two GEMMs alone %6.1f us" % (timeit(ga) / 4, timeit(gg) / 4))
gs, gf = graphed(serial), graphed(forked)
print("graph: serial %6.1f us   forked %6.1f us  (per iteration)" % (timeit(gs) / 4, timeit(gf) / 4))
