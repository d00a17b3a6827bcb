#!/usr/bin/env python3
"""Config-4 (1M entities / 20M triples / 1k relations, d=300) forward aggregation: fp32 and bf16 tables; env knobs JMAC_FWD_HW*."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import ops, synth
from jmac_amd.graph import RelGraph
dev = torch.device("cuda")
scale = float(os.environ.get("C4_SCALE", "1.0"))
n, e, nr, d = int(1_000_000 * scale), int(20_000_000 * scale), 1000, 300
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=1234)
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
av = torch.randn(d, device=dev, generator=gen) * 0.1
def t(fn, k=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("JMAC_")}, "N": n, "E": e}
fb, fb16 = synth.fwd_algorithmic_bytes(n, e, d), synth.fwd_algorithmic_bytes(n, e, d, 2)
with torch.no_grad():
    ms = t(lambda: ops.rel_attn_aggregate(PQZ, RR, av, g, 0.05, nrel - 1, 0.5))
    out["fwd_f32_ms"], out["fwd_f32_frac"] = ms, fb / (ms * 1e-3) / 8e12
    o = ops.rel_attn_aggregate(PQZ, RR, av, g, 0.05, nrel - 1, 0.5)
    out["f32_checksum"] = [float(o.double().sum()), float(o.double().abs().sum()), float(o[12345].double().sum())]
    del o
    pad = os.environ.get("JMAC_FWD_HW", "1") != "0"
    P16, R16 = PQZ.to(torch.bfloat16), RR.to(torch.bfloat16)
    if pad:
        P16, R16 = ops.pad_table(P16, d, 3), ops.pad_table(R16, d, 2)
    del PQZ
    ms = t(lambda: ops.rel_attn_aggregate(P16, R16, av, g, 0.05, nrel - 1, 0.5))
    out["fwd_bf16_ms"], out["fwd_bf16_frac"] = ms, fb16 / (ms * 1e-3) / 8e12
    o = ops.rel_attn_aggregate(P16, R16, av, g, 0.05, nrel - 1, 0.5)
    out["bf16_checksum"] = [float(o.double().sum()), float(o.double().abs().sum()), float(o[12345].double().sum())]
print(json.dumps(out), flush=True)
