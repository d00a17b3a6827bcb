"""fwd/bwd aggregation timing on the config-4 graph under tuning knobs (env JMAC_GRID / JMAC_FWD_U, arg chunk)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd import synth, ops
from jmac_amd.graph import RelGraph
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != 'auto' else None
d = int(sys.argv[3]) if len(sys.argv) > 3 else 300
do_bwd = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n, e, nr = int(1_000_000 * scale), int(20_000_000 * scale), 1000
ei, et, n, nrel = synth.power_law_graph(n, e, nr, seed=1234)
dbg = os.environ.get("DBG", "")
if "samerel" in dbg: et[:] = 0
if "samesrc" in dbg: ei[1][:] = 7
if "regular" in dbg:
    ei[0] = np.repeat(np.arange(n), e // n)[:e]
if "uniform" in dbg:
    ei[0] = np.random.default_rng(1).integers(0, n, e)
dev = torch.device("cuda")
g = RelGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev), n, nrel, chunk)
gen = torch.Generator(device=dev).manual_seed(0)
PQZ = (torch.randn(n, 3 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
RR = (torch.randn(nrel, 2 * d, device=dev, generator=gen) * 0.3).requires_grad_(True)
a = (torch.randn(d, device=dev, generator=gen) * 0.1).requires_grad_(True)
G = torch.randn(n, d, device=dev, generator=gen)
bf16 = bool(os.environ.get("BF16"))
if bf16:      # the layout the layers produce: halves padded to a multiple of 8 elements (300 -> 304)
    PQZ, RR = ops.pad_table(PQZ.detach().to(torch.bfloat16), d, 3), ops.pad_table(RR.detach().to(torch.bfloat16), d, 2)
    do_bwd = 0
fb = synth.fwd_algorithmic_bytes(n, e, d, 2 if bf16 else 4)
with torch.no_grad():
    for _ in range(2): ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
print("bf16" if bf16 else "f32", dbg, "GRID=%s U=%s chunk=%s items=%d splits=%d  fwd %.3f ms  %.0f GB/s (%.1f%% of 8TB/s)" % (os.environ.get("JMAC_GRID"), os.environ.get("JMAC_FWD_U"), g.chunk, g.by_dst.n_items_max, g.by_dst.n_splits_max, ms, fb / ms / 1e6, fb / ms / 1e6 / 80))
if do_bwd:
    g.ensure_backward_views()
    out = ops.rel_attn_aggregate(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, 1)
    for mode in (1, 0):
        out = ops._RelAttnAggregate.apply(PQZ, RR, a, g, 0.05, nrel - 1, 0.5, mode)
        torch.autograd.grad(out, [PQZ, RR, a], G, retain_graph=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): torch.autograd.grad(out, [PQZ, RR, a], G, retain_graph=True)
        e1.record(); torch.cuda.synchronize()
        print("  bwd mode %d: %.3f ms" % (mode, e0.elapsed_time(e1) / 3))
