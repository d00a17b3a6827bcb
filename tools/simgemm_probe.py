"""config-5 quality GEMM (12000 x 12000 x 300 fp32) for rocprofv3: 12 launches of sim_gemm_kernel, result preallocated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmac_amd import scoring
gen = torch.Generator(device="cuda").manual_seed(0)
tab = torch.nn.functional.normalize(torch.randn(24000, 300, device="cuda", generator=gen))
out = torch.empty((12000, 12000), device="cuda")
for _ in range(12):
    scoring.sim_matrix(tab[:12000], tab[12000:], out=out)
torch.cuda.synchronize()
