"""split-bf16 GEMM (jmac_gemm_nt_x3_f32) against torch.mm (library fp32) at the step's shapes: accuracy vs float64, time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(11805, 900, 300), (11805, 300, 900), (11805, 300, 600), (11805, 300, 300), (11805, 600, 300), (257, 100, 52), (1000000, 900, 300)]:
    A = torch.randn(M, K, device=dev, generator=g) * (torch.rand(M, 1, device=dev, generator=g) * 3).exp()     # rows of mixed scale
    B = torch.randn(N, K, device=dev, generator=g) * 0.05
    Bt = B.t().contiguous()
    C = torch.empty(M, N, device=dev)
    ops.gemm_nt_x3(A, B, out=C)
    ref32 = torch.mm(A, Bt)
    if M <= 20000:
        ref = (A.double() @ B.double().t())
        scale = (A.double().abs() @ B.double().abs().t())
        e3 = ((C.double() - ref).abs() / scale).max().item()
        e32 = ((ref32.double() - ref).abs() / scale).max().item()
    else:
        e3 = e32 = float("nan")
    us3 = t(lambda: ops.gemm_nt_x3(A, B, out=C))
    usl = t(lambda: torch.mm(A, Bt, out=ref32))
    fl = 2.0 * M * N * K
    print("M=%7d N=%4d K=%4d  x3 %8.1f us (%6.1f TF)   torch.mm %8.1f us (%6.1f TF)   err/sum|ab|: x3 %.2e  fp32 lib %.2e" % (M, N, K, us3, fl / us3 / 1e6, usl, fl / usl / 1e6, e3, e32))
