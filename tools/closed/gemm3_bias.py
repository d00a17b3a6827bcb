import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd import ops
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
for K in (304, 912):
    A = torch.rand(4096, K, device=dev, generator=g) + 0.5      # all positive
    B = torch.rand(512, K, device=dev, generator=g) + 0.5
    ref = A.double() @ B.double().t()
    c3 = ops.gemm_nt_x3(A, B).double()
    c32 = torch.mm(A, B.t()).double()
    for name, c in (("x3", c3), ("fp32 lib", c32)):
        rel = (c - ref) / ref
        print("K=%d %-8s mean signed rel err %+.3e   rms %.3e   max %.3e" % (K, name, rel.mean().item(), rel.pow(2).mean().sqrt().item(), rel.abs().max().item()))
    # random-sign data
    A2 = torch.randn(4096, K, device=dev, generator=g); B2 = torch.randn(512, K, device=dev, generator=g)
    ref = A2.double() @ B2.double().t(); sc = A2.double().abs() @ B2.double().abs().t()
    for name, c in (("x3", ops.gemm_nt_x3(A2, B2).double()), ("fp32 lib", torch.mm(A2, B2.t()).double())):
        rel = (c - ref) / sc
        print("K=%d %-8s randn: mean %+.3e rms %.3e max %.3e" % (K, name, rel.mean().item(), rel.pow(2).mean().sqrt().item(), rel.abs().max().item()))
