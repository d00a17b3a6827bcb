"""l1_score_kernel alone at the bench shape (B=1000, N=11805, d=300): 60 back-to-back launches (rocprofv3 target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd import scoring
g = torch.Generator(device="cuda").manual_seed(0)
tab = torch.randn(11805, 300, device="cuda", generator=g) * 0.3
er = torch.randn(1000, 300, device="cuda", generator=g) * 0.3
out = torch.empty(1000, 11805, device="cuda")
for _ in range(5): scoring.l1_scores(er, tab, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(60): scoring.l1_scores(er, tab, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 60
print("l1 1000x11805x300: %.1f us  -> %.3f of the 78.6 T lane-inst/s VALU issue peak (2 inst / element)" % (ms * 1e3, 2 * 1000 * 11805 * 300 / (ms * 1e-3) / 78.6432e12))
