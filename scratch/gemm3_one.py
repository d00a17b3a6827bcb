import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmac_amd import ops
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (11805, 900, 300))]
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g); B = torch.randn(N, K, device=dev, generator=g) * 0.05
C = torch.empty(M, N, device=dev)
for _ in range(30): ops.gemm_nt_x3(A, B, out=C)
torch.cuda.synchronize()
