"""Which Python lines of the headline step still launch ATen (non-jmac, non-GEMM) kernels?  torch.profiler with stacks
over a few eager steps; prints per (op, innermost repo frame) the launches per step and device time."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
sys.argv = [sys.argv[0], "--no-synth", "--no-cpu-baseline"]
a = bench.parse()
w = bench.JaWorkload(a, "cuda", data="real")
for _ in range(3):
    w.step()
torch.cuda.synchronize()
K = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(K):
        w.step()
    torch.cuda.synchronize()
main_thread = min(e.thread for e in prof.events())
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.self_device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    frame = "?"
    for f in ev.stack or []:
        if ROOT in f and "tools/step_ops" not in f:
            frame = f.replace(ROOT + "/", "")
            break
    if frame == "?":
        frame = str(ev.input_shapes)[:90] + (" bwd" if ev.thread != main_thread else "")
    k = (ev.name, frame)
    acc[k][0] += 1
    acc[k][1] += ev.self_device_time_total
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = 0.0
for (name, frame), (n, t) in rows:
    if "mm" in name:
        continue
    tot += t / K
    print("%7.1f us x %4.1f  %-28s %s" % (t / K, n / K, name, frame))
print("total non-GEMM aten per step: %.1f us" % tot)
