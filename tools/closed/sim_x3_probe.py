"""similarity GEMM A B^T at the config-5 shapes: the exact-fp32 MFMA kernel (jmac_sim_matrix_f32) against the split-bf16 GEMM
(jmac_gemm_nt_x3_f32) -- time, error vs float64, and whether the top-25 index lists agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from jmac_amd import ops, scoring
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(12000, 12000, 300), (3000, 30000, 300), (10500, 10500, 300)]:
    A = F.normalize(torch.randn(M, K, device=dev, generator=g), dim=1)
    B = F.normalize(torch.randn(N, K, device=dev, generator=g), dim=1)
    C3 = torch.empty(M, N, device=dev)
    ops.gemm_nt_x3(A, B, out=C3)
    C1 = scoring.sim_matrix(A, B)
    rows = slice(0, 512)
    ref = A[rows].double() @ B.double().t()
    e3 = (C3[rows].double() - ref).abs().max().item()
    e1 = (C1[rows].double() - ref).abs().max().item()
    k = 25
    i3 = C3.topk(k, dim=1).indices
    i1 = C1.topk(k, dim=1).indices
    same = (i3 == i1).all(dim=1).float().mean().item()
    us3 = t(lambda: ops.gemm_nt_x3(A, B, out=C3))
    us1 = t(lambda: scoring.sim_matrix(A, B))
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%4d  x3 %8.1f us (%6.1f TF)   sim_gemm %8.1f us (%6.1f TF)   max|err| x3 %.2e  fp32-mfma %.2e   rows with identical top-%d lists %.4f"
          % (M, N, K, us3, fl / us3 / 1e6, us1, fl / us1 / 1e6, e3, e1, k, same))
