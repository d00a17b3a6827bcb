import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth, ops
from jmac_amd.graph import RelGraph
d, n, nrel = 300, int(sys.argv[1]) if len(sys.argv) > 1 else 11805, 961
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
a = torch.randn(d, device=dev, generator=gen) * 0.1
rng = np.random.default_rng(0)
def timeit(fn, it=300):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
empty = torch.empty(0, device=dev)
t_null = timeit(lambda: torch.empty(1, device=dev))
for deg in (0, 1, 2, 4, 8, 16):
    e = n * deg
    dst = np.repeat(np.arange(n), deg); src = rng.integers(0, n, e); typ = rng.integers(0, nrel - 1, e)
    g = RelGraph(torch.from_numpy(np.stack([dst, src]).astype(np.int64)).to(dev), torch.from_numpy(typ.astype(np.int64)).to(dev), n, nrel)
    with torch.no_grad():
        us = timeit(lambda: ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1))
    fb = synth.fwd_algorithmic_bytes(n, e, d)
    print("N=%d deg=%2d E=%7d: %.1f us  (%.0f GB/s)" % (n, deg, e, us, fb / us / 1e3))
