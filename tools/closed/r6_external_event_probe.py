"""round 6 probe: can HIP events recorded INSIDE a captured hipGraph (torch.cuda.Event(enable_timing=True, external=True)) time a
kernel of the replayed step?  (Would make roofline.frac_in_step a live measurement instead of a committed rocprofv3 trace.)"""
import torch
x = torch.randn(4096, 4096, device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        torch.mm(x, x, out=y)
torch.cuda.synchronize()
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external events in this torch:", ex)
    raise SystemExit(0)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y.add_(1.0)
        e0.record()
        torch.mm(x, x, out=y)
        e1.record()
        y.mul_(0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured; elapsed inside the replayed graph: %.3f ms" % e0.elapsed_time(e1))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.mm(x, x, out=y); b.record(); torch.cuda.synchronize()
    print("eager events around the same product: %.3f ms" % a.elapsed_time(b))
except Exception as ex:
    print("external events in a capture failed:", type(ex).__name__, str(ex)[:300])
