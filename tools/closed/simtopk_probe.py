"""get_neg at the config-5 shape (3000 x 30000 x 300, k = 25): the kernels of jmac_sim_topk_f32 for rocprofv3 --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd import scoring
gen = torch.Generator(device="cuda").manual_seed(0)
tab = torch.nn.functional.normalize(torch.randn(30000, 300, device="cuda", generator=gen))
q = tab[torch.randperm(30000, device="cuda", generator=gen)[:3000]]
for _ in range(20):
    scoring.sim_topk(q, tab, 25)
torch.cuda.synchronize()
