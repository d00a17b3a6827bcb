"""What would running two independent layers' aggregation (conv1_alignment's and conv1_completion's: same graph, different
tables) as ONE launch on the block-diagonal union of two copies of the graph buy at DBP-5L size?  Timing only (both halves
share a_att and the loop relation here; a product form would need them per half)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from jmac_amd import encoder
from jmac_amd.graph import RelGraph
from jmac_amd.data import edges_from_triples, load_dbp5l_arrays
dev = torch.device("cuda")
z = load_dbp5l_arrays(bench.REAL_DATA)
ei, et = edges_from_triples(z["ja.train"], False)
N, nr, d = int(z["ja.num_entity"]), int(z["n_relation_lines"]) + 2, 300
Np = (N + 3) // 4 * 4
ei2 = np.concatenate([ei, ei + Np], 1)
et2 = np.concatenate([et, et + nr], 0)
g1 = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), N, nr)
g2 = RelGraph(torch.from_numpy(ei2).to(dev), torch.from_numpy(et2).to(dev), 2 * Np, 2 * nr)
for g in (g1, g2):
    g.ensure_backward_views()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=gen)

def setup(g, n, nrel):
    PQZ, RR, a, G = r(n, 3 * d) * 0.3, r(nrel, 2 * d) * 0.3, r(d) * 0.1, r(n, d)
    out, smax, sden = encoder._agg_fwd(PQZ, RR, a, g, 0.05)
    return PQZ, RR, a, out, smax, sden, G
s1a, s1b, s2 = setup(g1, N, nr), setup(g1, N, nr), setup(g2, 2 * Np, 2 * nr)

def fwd(g, s):
    encoder._agg_fwd(s[0], s[1], s[2], g, 0.05)
def bwd(g, s):
    encoder._agg_bwd(s[0], s[1], s[2], g, 0.05, s[3], s[4], s[5], s[6])

def graphed(fn, reps=4):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    return lambda: gr.replay()

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
    return e0.elapsed_time(e1) / n * 1e3 / 4

for name, f in (("forward", fwd), ("backward", bwd)):
    two = timeit(graphed(lambda: (f(g1, s1a), f(g1, s1b))))
    one = timeit(graphed(lambda: f(g2, s2)))
    print("aggregation %-8s  two launches on the graph %6.1f us   one launch on two stacked copies %6.1f us" % (name, two, one))
