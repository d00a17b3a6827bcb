import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd import scoring
n = 12000
A = torch.nn.functional.normalize(torch.randn(n, 300, device="cuda")); B = torch.nn.functional.normalize(torch.randn(n, 300, device="cuda"))
C = torch.empty(n, n, device="cuda")
for _ in range(12): scoring.sim_matrix(A, B, out=C)
torch.cuda.synchronize()
