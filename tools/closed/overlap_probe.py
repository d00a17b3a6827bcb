"""Do the grouped relation-side launches overlap with an N-row library GEMM when they sit on two streams (eager and as
two branches of a hipGraph)?  Per iteration: three grouped levels of five 962x300x300 products + two 11805x300x900 GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd.encoder import gemm_task, grouped_gemm
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
A, W = r(962, 300), r(300, 300)
outs = [r(962, 300) for _ in range(5)]
tasks = [gemm_task(A, W, o) for o in outs]
X, Wc, P = r(11805, 300), r(300, 900), r(11805, 900)
side = torch.cuda.Stream()

def rel():
    for _ in range(3):
        grouped_gemm(tasks)

def node():
    torch.mm(X, Wc, out=P)
    torch.mm(X, Wc, out=P)

def serial():
    rel(); node()

def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rel()
    node()
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

def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            fn()
    return lambda: gr.replay()

print("rel only   %7.1f us" % timeit(rel))
print("node only  %7.1f us" % timeit(node))
print("eager serial %7.1f us   forked %7.1f us" % (timeit(serial), timeit(forked)))
gs, gf = graphed(serial), graphed(forked)
print("graph (x4) serial %7.1f us   forked %7.1f us" % (timeit(gs), timeit(gf)))
