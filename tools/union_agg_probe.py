#!/usr/bin/env python3
"""Aggregation kernel timing on the DBP-5L-shaped union graph (config 3) and on pair-sized graphs: fwd fp32 / bf16, bwd; env knobs
JMAC_SMALL_ITEMS / JMAC_FWD_U."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import ops, synth
from jmac_amd.graph import RelGraph
import _knobs
_knobs.apply()       # JMAC_SMALL_ITEMS & co. -> jmac_amd.graph attributes

dev = torch.device("cuda")
if os.environ.get("UNION_SYNTH"):
    ei, et, n, nr, eb, rb = synth.dbp5l_union(1234, target="ja")
else:                      # the REAL union of the five DBP-5L KGs (committed integer arrays)
    from jmac_amd import data as jdata
    kgs, _, _, _ = jdata.kgs_from_arrays(jdata.load_dbp5l_arrays(os.path.join(ROOT, "tests", "golden", "dbp5l_all_data.npz")), "ja")
    ei, et, n, nr, eb, rb = jdata.union_edges(kgs)
ei_t, et_t = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
d = 300
g = RelGraph(ei_t, et_t, n, nr + 1)
g.ensure_backward_views()
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = (torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3)
RR = (torch.randn(nr + 1, 2 * d, device=dev, generator=gen) * 0.3)
av = torch.randn(d, device=dev, generator=gen) * 0.1
G = torch.randn(n, d, device=dev, generator=gen)
def t(fn, k=50):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("JMAC_")}, "N": n, "E": int(ei.shape[1]),
       "items": g.by_dst.n_items_max, "coop": g.by_dst.n_coop, "inline": g.by_dst.item_edges is not None}
with torch.no_grad():
    out["fwd_f32_us"] = t(lambda: ops.rel_attn_aggregate(PQZ, RR, av, g, 0.05, nr, 0.5))
    pad = os.environ.get("JMAC_FWD_HW", "1") != "0"
    P16, R16 = (ops.pad_table(PQZ.to(torch.bfloat16), d, 3), ops.pad_table(RR.to(torch.bfloat16), d, 2)) if pad else (PQZ.to(torch.bfloat16), RR.to(torch.bfloat16))
    out["fwd_bf16_us"] = t(lambda: ops.rel_attn_aggregate(P16, R16, av, g, 0.05, nr, 0.5))
Pq = PQZ.clone().requires_grad_(True); Rq = RR.clone().requires_grad_(True); aq = av.clone().requires_grad_(True)
o = ops.rel_attn_aggregate(Pq, Rq, aq, g, 0.05, nr, 0.5, 1)
out["bwd_f32_us"] = t(lambda: torch.autograd.grad(o, [Pq, Rq, aq], G, retain_graph=True), 20)
fb = synth.fwd_algorithmic_bytes(n, int(ei.shape[1]), d); fb16 = synth.fwd_algorithmic_bytes(n, int(ei.shape[1]), d, 2)
out["fwd_f32_frac"] = fb / (out["fwd_f32_us"] * 1e-6) / 8e12
out["fwd_bf16_frac"] = fb16 / (out["fwd_bf16_us"] * 1e-6) / 8e12
print(json.dumps(out), flush=True)
