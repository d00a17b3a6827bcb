#!/usr/bin/env python3
"""Round 5, VERDICT item 7b: does a 128-byte aligned [Q|Z] row pitch pay on config 4 (1M / 20M / 1k, d = 300, fp32)?
  base  : the product's layout -- one [N, 900] table, [Q|Z] = 2 400 B of a 3 600 B row (rows start at every multiple of 16 B inside
          a 128-B line: 19.6 lines per gathered row on average)
  pitch : [Q|Z] rows at a 2 432-B pitch (608 floats, 128-B aligned: exactly 19 lines), P rows at a 1 280-B pitch, in the [N, 928]
          table a padded projection weight would write ([Q|Z|pad8|P|pad20])
Same kernel (jmac_rel_attn_aggregate_fwd_f32 takes P / QZ pointers and pitches), same graph, interleaved repetitions.
Also the backward with the same pitches."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth
from jmac_amd._lib import check, lib, ptr, stream
from jmac_amd.graph import RelGraph
dev = torch.device("cuda")
scale = float(os.environ.get("C4_SCALE", "1.0"))
n, e, nr, d = int(1_000_000 * scale), int(20_000_000 * scale), 1000, 300
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=1234)
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel)
gen = torch.Generator(device=dev).manual_seed(0)
base = torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3
RR = torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3
av = torch.randn(d, device=dev, generator=gen) * 0.1
padded = torch.zeros(n, 928, device=dev)
padded[:, :600] = base[:, d:]
padded[:, 608:908] = base[:, :d]
assert padded.data_ptr() % 128 == 0
L = lib()
s = g.by_dst
out, smax, sden = (torch.empty(n, d, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev))
wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d))
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)

def fwd(P, ldp, QZ, ldqz):
    check(L.jmac_rel_attn_aggregate_fwd_f32(P, ldp, QZ, ldqz, ptr(RR), 2 * d, ptr(av), ptr(g.col), ptr(g.etype), C.byref(s.view()), n, d,
                                            0.05, nrel - 1, 0, 0.5, ptr(out), d, ptr(smax), ptr(sden), ptr(ws), wsb, stream()), "fwd")
variants = {"base [N,900]": lambda: fwd(base.data_ptr(), 900, base.data_ptr() + d * 4, 900),
            "pitch 2432 B ([N,928])": lambda: fwd(padded.data_ptr() + 608 * 4, 928, padded.data_ptr(), 928)}
res = {k: [] for k in variants}
sums = {}
for k, fn in variants.items():
    fn(); torch.cuda.synchronize()
    sums[k] = float(out.double().sum())
for rep in range(4):
    for k, fn in variants.items():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 4)
fb = synth.fwd_algorithmic_bytes(n, e, d)
print(json.dumps({"N": n, "E": e, "checksums": sums,
                  "fwd_ms": {k: {"runs": v, "median": float(np.median(v)), "frac_hbm": fb / (float(np.median(v)) * 1e-3) / 8e12} for k, v in res.items()}}))
