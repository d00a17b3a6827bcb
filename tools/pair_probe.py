#!/usr/bin/env python3
"""Pair step (bench.PairWorkload) timing probe: batched vs separate calls as hipGraphs; env knobs JMAC_SMALL_ITEMS / JMAC_FWD_U."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import _knobs
_knobs.apply()       # JMAC_SMALL_ITEMS & co. -> jmac_amd.graph attributes

ap = argparse.ArgumentParser()
ap.add_argument("--batched", type=int, default=1)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--single", action="store_true", help="also time the single-KG ja step")
ap.add_argument("--ja", action="store_true", help="ONLY the single-KG ja step")
ap.add_argument("--union", action="store_true", help="ONLY the five-KG forward_stacked training step (bench.union_train_setup)")
x = ap.parse_args()
a = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
dev = torch.device("cuda")
bench.enable_gemm_tuning(0)
out = {"JMAC_SMALL_ITEMS": os.environ.get("JMAC_SMALL_ITEMS"), "JMAC_FWD_U": os.environ.get("JMAC_FWD_U"), "batched": x.batched}
if x.union:
    a.data, a.torch_adam = "real", False
    m, kgs = bench.union_real_model(a, dev)
    make, E, N = bench.union_train_setup(a, dev, m, kgs)
    step = make(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    bench.freeze_gemm_tuning()
    g1 = bench.try_capture(bench.types_ns(step=step))
    el = bench.time_steps(g1.replay, x.steps, 5, False)
    out.update({"union_train_ms": el / x.steps * 1e3, "E": E, "N": N})
    print(json.dumps(out), flush=True)
    sys.exit(0)
if x.ja:
    w1 = bench.JaWorkload(a, dev, data="real")
    for _ in range(2):
        w1.step()
    torch.cuda.synchronize()
    bench.freeze_gemm_tuning()
    g1 = bench.try_capture(w1)
    el = bench.time_steps(g1.replay, x.steps, 5, False)
    out["single_ja_ms"] = el / x.steps * 1e3
    print(json.dumps(out), flush=True)
    sys.exit(0)
w = bench.PairWorkload(a, dev, batched=bool(x.batched))
for _ in range(2):
    w.step()
torch.cuda.synchronize()
bench.freeze_gemm_tuning()
g = bench.try_capture(w)
el = bench.time_steps(g.replay, x.steps, 5, False)
out["pair_ms"] = el / x.steps * 1e3
st = w.model.forward_stacked(w.blocks()) if x.batched else None
if st is not None:
    from jmac_amd.graph import union_cache
    gr = next(iter(union_cache._d.values()))
    out["graph"] = {"N": gr.N, "E": gr.E, "items": gr.by_dst.n_items_max, "coop": gr.by_dst.n_coop, "inline": gr.by_dst.item_edges is not None}
if x.single:
    a2 = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
    bench.enable_gemm_tuning(0)
    w1 = bench.JaWorkload(a2, dev, data="real")
    for _ in range(2):
        w1.step()
    torch.cuda.synchronize()
    bench.freeze_gemm_tuning()
    g1 = bench.try_capture(w1)
    el = bench.time_steps(g1.replay, x.steps, 5, False)
    out["single_ja_ms"] = el / x.steps * 1e3
print(json.dumps(out), flush=True)
