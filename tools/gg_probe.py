"""Timing of jmac_gemm_grouped_f32 by operand form at the relation-side sizes (HIP events around back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmac_amd.encoder import gemm_task, grouped_gemm, ACT_LEAKY, DACT_LEAKY
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
nr, d = 962, 300
A, W, W2 = r(nr, d), r(d, d), r(d, 2 * d)
G2 = r(nr, 2 * d)
out, out2, outw, outw2 = r(nr, d), r(nr, 2 * d), r(d, d), r(d, 2 * d)
cases = {
    "NN 962x300x300": lambda: [gemm_task(A, W, out)],
    "NN x4": lambda: [gemm_task(A, W, r_) for r_ in (out, out.clone(), out.clone(), out.clone())],
    "NN 962x600x300": lambda: [gemm_task(A, W2, out2)],
    "NT 962x300x600 (dR2)": lambda: [gemm_task(G2, W2, out, tb=True)],
    "NT 962x300x300": lambda: [gemm_task(A, W, out, tb=True)],
    "TN 300x300x962 (dW)": lambda: [gemm_task(A, out, outw, ta=True)],
    "TN 300x600x962 (dwc)": lambda: [gemm_task(A, G2, outw2, ta=True)],
    "torch.mm NN": None,
}
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, mk in cases.items():
    if only is not None and not name.startswith(only):
        continue
    if mk is None:
        fn = lambda: torch.mm(A, W)
    else:
        tasks = mk()
        fn = lambda: grouped_gemm(tasks)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-26s %7.2f us" % (name, e0.elapsed_time(e1) / 50 * 1e3))
