import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_encoder as T
import jmac_amd.encoder as enc
if os.environ.get('NANFILL'):
    enc._empty = lambda dev, *shape: torch.full(shape, float('nan'), dtype=torch.float32, device=dev)
from util import random_graph
n, nr, d, di = 700, 37, 32, 20
rng = np.random.default_rng(5)
ei, et = random_graph(rng, n, nr, 2600, hub=300)
ei, et = torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda()
m = T._model(d, n, nr, di, False, 11)
gen = torch.Generator(device="cuda").manual_seed(3)
G = {k: torch.randn(s, device="cuda", generator=gen) for k, s in (("align", (n, d)), ("c1", (n, d)), ("c0", (n, d)), ("r1", (nr, d)), ("r0", (nr, d)))}
m.train()
for rep in range(1):
    for fused in (True,):
        junk = torch.full((4000, 4000), 1e6, device="cuda"); del junk     # dirty the allocator's free blocks
        out, g, bn = T._run(m, fused, ei, et, n, nr, ("align",), G)
        print(rep, fused, {k: float(v.abs().max()) for k, v in g.items() if v is not None and k.endswith("loop_rel")}, "NaN grads:", [k for k, v in g.items() if v is not None and bool(torch.isnan(v).any())], "nan out:", [bool(torch.isnan(o).any()) for o in out])
