import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle.jmac_oracle as orc
from util import make_args, random_graph
from jmac_amd.layer import RelationAwareLayer
def case(seed, n, nr, d, e):
    rng = np.random.default_rng(seed); ei, et = random_graph(rng, n, nr, e, hub=900)
    gen = torch.Generator().manual_seed(seed)
    return ei, et, torch.randn(n, d, generator=gen) * (4/np.sqrt(d)), torch.randn(nr, d, generator=gen) * (4/np.sqrt(d)), torch.randn(n, d, generator=gen)
ei, et, X, R, G = case(6, 500, 11, 300, 6000)
torch.manual_seed(11)
base = RelationAwareLayer(300, 300, rel_dim=300, act=torch.tanh, args=make_args())
res = {}
for dt in (torch.float32, torch.float64):
    p = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in base.named_parameters()}
    Xc, Rc = X.clone().to(dt).requires_grad_(True), R.clone().to(dt).requires_grad_(True)
    ref = orc.layer_forward(p, Xc, Rc, torch.from_numpy(ei), torch.from_numpy(et), 0.05, "sub", "leaky_relu", True, torch.zeros(300, dtype=dt), torch.ones(300, dtype=dt))
    (ref * G.to(dt)).sum().backward(); res[dt] = (ref.detach(), Xc.grad)
lay = base.cuda()
Xg = X.cuda().requires_grad_(True)
out = lay(Xg, R.cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda()); (out * G.cuda()).sum().backward()
r64 = res[torch.float64]
print("gpu vs f64: out %.2e gx %.2e" % ((out.cpu().double()-r64[0]).abs().max()/r64[0].abs().max(), (Xg.grad.cpu().double()-r64[1]).abs().max()/r64[1].abs().max()))
print("cpu32 vs f64: out %.2e gx %.2e" % ((res[torch.float32][0].double()-r64[0]).abs().max()/r64[0].abs().max(), (res[torch.float32][1].double()-r64[1]).abs().max()/r64[1].abs().max()))
err = (Xg.grad.cpu().double() - r64[1]).abs()
rowmax = err.max(1)[0]
top = torch.topk(rowmax, 6)
print("rows with largest grad_X error:", top.indices.tolist(), ["%.2e" % v for v in top.values.tolist()], "median row err %.2e" % rowmax.median().item())
# locate near-zero pre-activations in f64 for edges touching those rows
p = {k: v.detach().clone().to(torch.float64) for k, v in base.cpu().named_parameters()}
d = 300
rel = orc.transform_relations(p, R.double(), 0.05, "leaky_relu")
Pm, Qm = X.double() @ p["w_att"][:d], X.double() @ p["w_att"][d:]
Rq = rel @ p["w_att"][d:]
h = Pm[ei[0]] + Qm[ei[1]] - Rq[et]
amin = h.abs().min(1)
k = torch.topk(-amin[0], 3)
for e in k.indices.tolist():
    print("edge %d (dst %d <- src %d): min |h| = %.3e at k=%d" % (e, ei[0][e], ei[1][e], amin[0][e].item(), amin[1][e].item()))
