"""Subprocess body of tests/test_gpu_hw_variants.py: the forward aggregation of one seeded large-form graph (persistent grid: more
than 65 536 items) with whatever JMAC_FWD_* knobs the environment carries -- the knobs are read once per process -- saved as
fp32 / bf16-table outputs + softmax statistics.   usage: python hw_variant_worker.py <out.npz>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from jmac_amd import ops, synth
from jmac_amd.graph import RelGraph

dev = torch.device("cuda")
n, e, nr, d = 80000, 1000000, 120, 300
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=77)
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
assert g.by_dst.n_items_max > 65536 and g.by_dst.item_edges is None           # the persistent form (graph.INLINE_EDGES_MAX_ITEMS)
gen = torch.Generator(device=dev).manual_seed(5)
PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
av = torch.randn(d, device=dev, generator=gen) * 0.1
with torch.no_grad():
    o32 = ops.rel_attn_aggregate(PQZ, RR, av, g, 0.05, nrel - 1, 0.5)
    P16, R16 = ops.pad_table(PQZ.to(torch.bfloat16), d, 3), ops.pad_table(RR.to(torch.bfloat16), d, 2)
    o16 = ops.rel_attn_aggregate(P16, R16, av, g, 0.05, nrel - 1, 0.5)
torch.cuda.synchronize()
np.savez(sys.argv[1], o32=o32.cpu().numpy(), o16=o16.cpu().numpy())
