"""Two INDEPENDENT aggregation backwards (conv1_alignment's and conv1_completion's in a step: same graph, different tables)
as two branches of a hipGraph / on two streams: does a latency-bound launch overlap with another one like it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from jmac_amd import encoder
from jmac_amd.graph import RelGraph
from jmac_amd.data import edges_from_triples, load_dbp5l_arrays
dev = torch.device("cuda")
z = load_dbp5l_arrays(bench.REAL_DATA)
ei, et = edges_from_triples(z["ja.train"], False)
N, nr, d = int(z["ja.num_entity"]), int(z["n_relation_lines"]) + 2, 300
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), N, nr)
g.ensure_backward_views()
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=gen)
sets = []
for _ in range(2):
    PQZ, RR, a, G = r(N, 3 * d) * 0.3, r(nr, 2 * d) * 0.3, r(d) * 0.1, r(N, d)
    out, smax, sden = encoder._agg_fwd(PQZ, RR, a, g, 0.05)
    sets.append((PQZ, RR, a, out, smax, sden, G))
side = torch.cuda.Stream()

def bwd(i):
    PQZ, RR, a, out, smax, sden, G = sets[i]
    encoder._agg_bwd(PQZ, RR, a, g, 0.05, out, smax, sden, G)

def fwd(i):
    PQZ, RR, a, out, smax, sden, G = sets[i]
    encoder._agg_fwd(PQZ, RR, a, g, 0.05)

def serial(f):
    return lambda: (f(0), f(1))

def forked(f):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f(1)
        f(0)
        main.wait_stream(side)
    return run

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

for name, f in (("aggregation backward (3 launches)", bwd), ("aggregation forward (1 launch)", fwd)):
    one = timeit(graphed(lambda: f(0))) / 4
    s_, f_ = timeit(graphed(serial(f))) / 4, timeit(graphed(forked(f))) / 4
    print("%-36s one %6.1f us   two serial %6.1f us   two as branches %6.1f us" % (name, one, s_, f_))
