"""Does the library's fp32 GEMM like K = 304 / 320 better than K = 300 (tail handling of its 16-deep K loop)?  And N = 912 / 960
instead of 900?  TunableOp on, HIP events around back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.enable_gemm_tuning(0)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
def t(fn, n=40):
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = 11805
for K, Nc in ((300, 900), (304, 900), (320, 900), (300, 912), (304, 912), (320, 960)):
    X, W, G = r(N, K), r(K, Nc), r(N, Nc)
    P, dX, dW = torch.empty(N, Nc, device=dev), torch.empty(N, K, device=dev), torch.empty(K, Nc, device=dev)
    f = t(lambda: torch.mm(X, W, out=P))
    b1 = t(lambda: torch.mm(G, W.t(), out=dX))
    b2 = t(lambda: torch.mm(X.t(), G, out=dW))
    fl = 2 * N * K * Nc / 1e6
    print("K=%d N=%d: X W %6.1f us (%5.1f TF)   G W^T %6.1f us (%5.1f TF)   X^T G %6.1f us (%5.1f TF)   sum %6.1f us" %
          (K, Nc, f, fl / f, b1, fl / b1, b2, fl / b2, f + b1 + b2))
