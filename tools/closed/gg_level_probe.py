"""One backward level of the step's relation side in isolation (hipGraph replay, so no host time): which of its products
hold the launch?  Level B0 = two chains' (dT = (dRR W2g^T) act', dW2g = T^T dRR) + the completion MLP's first product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd.encoder import gemm_task, grouped_gemm, DACT_LEAKY
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
nr, d = 962, 300
def chain():
    dRR, W2g, T = r(nr, 2 * d), r(d, 2 * d), r(nr, d)
    return [gemm_task(dRR, W2g, r(nr, d), tb=True, act=DACT_LEAKY, act_src=T, slope=0.05), gemm_task(T, dRR, r(d, 2 * d), ta=True)]
c1, c2 = chain(), chain()
mlp = [gemm_task(r(nr - 1, d), r(d, d), r(nr - 1, d), tb=True, act=DACT_LEAKY, act_src=r(nr - 1, d), slope=0.05)]
sets = {"level B0 (5 tasks)": c1 + c2 + mlp, "NT K=600 x2": [c1[0], c2[0]], "TN 300x600x962 x2": [c1[1], c2[1]], "NT K=600 x1": [c1[0]],
        "TN x1": [c1[1]], "one chain (NT + TN)": c1, "MLP NT K=300": mlp}
def graph_time(tasks, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        grouped_gemm(tasks)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            grouped_gemm(tasks)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
for name, tasks in sets.items():
    print("%-24s %6.1f us per launch" % (name, graph_time(tasks)))
