import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth, ops
from jmac_amd.graph import RelGraph
lang = sys.argv[1] if len(sys.argv) > 1 else "ja"
bid = len(sys.argv) > 2 and sys.argv[2] == "bidir"
d = 300
if lang == "ja-real":      # the REAL DBP-5L ja KG (the bench's default workload): committed integer arrays
    from jmac_amd.data import edges_from_triples, load_dbp5l_arrays
    z = load_dbp5l_arrays(os.path.join(ROOT, "tests", "golden", "dbp5l_ja_el_data.npz"))
    ei, et = edges_from_triples(z["ja.train"], bid)
    n, nrel = int(z["ja.num_entity"]), int(z["n_relation_lines"]) + 2
else:
    ei, et, n, nrel = synth.dbp5l_like(lang, 1234, bidirectional=bid)
dev = torch.device("cuda")
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
g.ensure_backward_views()
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = (torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
RR = (torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
a = (torch.randn(d, device=dev, generator=gen) * 0.1).requires_grad_(True)
G = torch.randn(n, d, device=dev, generator=gen)
e = ei.shape[1]
fb = synth.fwd_algorithmic_bytes(n, e, d)
def timeit(fn, it=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
with torch.no_grad():
    us = timeit(lambda: ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1))
out = ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
ub = timeit(lambda: torch.autograd.grad(out, [PQZ, RR, a], G, retain_graph=True), 100)
print("%s bidir=%s N=%d E=%d GRID=%s U=%s chunk=%d: fwd %.1f us (%.0f GB/s, %.1f%%)  bwd %.1f us" % (lang, bid, n, e, os.environ.get("JMAC_GRID"), os.environ.get("JMAC_FWD_U"), g.chunk, us, fb / us / 1e3, fb / us / 1e3 / 80, ub))
