"""Candidate-list lengths of the fused sim+top-k at the config-5 shape (reads the cnt array out of the workspace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jmac_amd._lib import lib, ptr, stream, check
gen = torch.Generator(device="cuda").manual_seed(0)
N, L, d, k = 30000, 3000, 300, 25
tab = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=gen))
q = tab[torch.randperm(N, device="cuda", generator=gen)[:L]].contiguous()
Lb = lib()
wsb = int(Lb.jmac_sim_topk_workspace_bytes(L, N, k))
ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
idx = torch.empty((L, k), dtype=torch.int32, device="cuda")
check(Lb.jmac_sim_topk_f32(ptr(q), d, ptr(tab), d, L, N, d, k, None, ptr(idx), ptr(ws), wsb, stream()))
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
Ns = max(2048, N // 12); Ns = (Ns + 127) // 128 * 128
off = al(L * Ns * 4) + al(L * k * 4) + al(L * k * 4)
cnt = ws[off:off + L * 4].view(torch.int32)
print("Ns", Ns, "cnt min/mean/max", int(cnt.min()), float(cnt.float().mean()), int(cnt.max()), "rows over cap-k:", int((cnt + k > 1024).sum()))
