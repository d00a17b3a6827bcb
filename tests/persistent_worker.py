"""Subprocess body of tests/test_gpu_persistent_oracle.py::test_forward_forms_vs_oracle: the forward aggregation of the seeded
persistent-form case (tests/persistent_case.py) with whatever JMAC_FWD_* knobs the environment carries -- aggregate.hip reads
them once per process -- on fp32 tables and on (padded) bf16 tables.   usage: python persistent_worker.py <d> <out.npz>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import persistent_case as pc
from jmac_amd import ops
from jmac_amd.graph import RelGraph

d, out_path = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda")
ei, et, n, nrel = pc.graph()
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
assert g.by_dst.n_items_max > 65536 and g.by_dst.item_edges is None        # the persistent form
PQZ, RR, a, _ = pc.tables(n, nrel, d)
PQZ, RR, a = PQZ.to(dev), RR.to(dev), a.to(dev)
with torch.no_grad():
    o32 = ops.rel_attn_aggregate(PQZ, RR, a, g, pc.SLOPE, nrel - 1, pc.OUT_SCALE)
    P16, R16 = ops.pad_table(PQZ.to(torch.bfloat16), d, 3), ops.pad_table(RR.to(torch.bfloat16), d, 2)
    o16 = ops.rel_attn_aggregate(P16, R16, a, g, pc.SLOPE, nrel - 1, pc.OUT_SCALE)
torch.cuda.synchronize()
np.savez(out_path, o32=o32.cpu().numpy(), o16=o16.cpu().numpy())
