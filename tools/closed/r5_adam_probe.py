#!/usr/bin/env python3
"""Round 5: the optimizer step alone on the parameter tensors of the headline model (bench.JaWorkload, real ja KG, d = 300):
torch.optim.Adam(fused, capturable) against jmac_amd.optim.Adam (jmac_adam_step_f32; JMAC_ADAM_V = float4s per thread and array),
each as a hipGraph of 20 steps timed with HIP events."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from jmac_amd import optim

dev = torch.device("cuda")
a = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
w = bench.JaWorkload(a, dev, data="real")
shapes = [tuple(p.shape) for p in w.model.parameters() if p.requires_grad]
n = sum(int(torch.tensor(s).prod()) for s in shapes)
out = {"tensors": len(shapes), "parameters": n, "bytes_moved": 7 * 4 * n, "JMAC_ADAM_V": os.environ.get("JMAC_ADAM_V")}


def timed(make):
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev) * 0.1) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p) * 0.01
    opt = make(ps)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                opt.step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3


out["torch_fused_capturable_us"] = timed(lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True, capturable=True))
out["jmac_us"] = timed(lambda ps: optim.Adam(ps, lr=1e-3))
out["jmac_TBps"] = out["bytes_moved"] / out["jmac_us"] / 1e6
out["torch_TBps"] = out["bytes_moved"] / out["torch_fused_capturable_us"] / 1e6
print(json.dumps(out))
