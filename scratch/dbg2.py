import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle.jmac_oracle as orc
from util import make_args, random_graph
from jmac_amd.layer import RelationAwareLayer

def run(n, nr, d, e, hub, chunk, mode, seed):
    rng = np.random.default_rng(seed)
    ei, et = random_graph(rng, n, nr, e, hub=hub)
    ei, et = torch.from_numpy(ei), torch.from_numpy(et)
    gen = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=gen) * (4 / np.sqrt(d)); R = torch.randn(nr, d, generator=gen) * (4 / np.sqrt(d)); G = torch.randn(n, d, generator=gen)
    torch.manual_seed(d)
    lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
    res = {}
    for dt in (torch.float32, torch.float64):
        p = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in lay.named_parameters()}
        Xc, Rc = X.detach().clone().to(dt).requires_grad_(True), R.detach().clone().to(dt).requires_grad_(True)
        ref = orc.layer_forward(p, Xc, Rc, ei, et, 0.05, "sub", "leaky_relu", True, torch.zeros(d, dtype=dt), torch.ones(d, dtype=dt))
        (ref * G.to(dt)).sum().backward()
        res[dt] = dict(out=ref.detach(), gX=Xc.grad, gR=Rc.grad, **{"g_" + k: v.grad for k, v in p.items()})
    lay = lay.cuda(); lay.bwd_mode, lay.chunk = mode, chunk
    Xg, Rg = X.detach().clone().cuda().requires_grad_(True), R.detach().clone().cuda().requires_grad_(True)
    out = lay(Xg, Rg, ei.cuda(), et.cuda())
    (out * G.cuda()).sum().backward()
    gpu = dict(out=out.detach().cpu(), gX=Xg.grad.cpu(), gR=Rg.grad.cpu(), **{"g_" + k: v.grad.cpu() for k, v in lay.named_parameters()})
    print("case n=%d d=%d e=%d mode=%d" % (n, d, e, mode))
    for k in gpu:
        r64 = res[torch.float64][k]; sc = r64.abs().max().item() + 1e-30
        print("  %-26s gpu-vs-f64 %.2e   cpu32-vs-f64 %.2e   scale %.2e" % (k, (gpu[k].double() - r64).abs().max().item() / sc, (res[torch.float32][k].double() - r64).abs().max().item() / sc, sc))

run(200, 8, 512, 900, None, 256, 1, 712)
run(600, 25, 300, 5000, 700, 64, 1, 900)
