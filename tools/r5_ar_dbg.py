import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import test_gpu_encoder as T
from jmac_amd import encoder
DEV = "cuda"
n, nr, d, di = 6000, 37, 64, 20
rng = np.random.default_rng(7); e = 9000
dst = rng.choice(n // 2, size=e); src = rng.choice(np.arange(n // 4, n * 3 // 4), size=e); dst[:300] = 11
ei = torch.from_numpy(np.stack([dst, src]).astype(np.int64)).to(DEV); et = torch.from_numpy(rng.integers(0, nr, e).astype(np.int64)).to(DEV)
m = T._model(d, n, nr, di, False, 13)
gen = torch.Generator(device=DEV).manual_seed(4)
G = {k: torch.randn(s, device=DEV, generator=gen) for k, s in (("align", (n, d)), ("c1", (n, d)), ("c0", (n, d)), ("r1", (nr, d)), ("r0", (nr, d)))}
m.train()
real_empty = encoder._empty
use = ("align", "comp", "rel")
for poison in (False, True):
    encoder._empty = (lambda dev, *shape: real_empty(dev, *shape).fill_(float("nan"))) if poison else real_empty
    for flag in (False, True):
        encoder.ACTIVE_ROWS = flag
        cap = {}
        encoder.CAPTURE = cap
        out, g, bn = T._run(m, True, ei, et, n, nr, use, G)
        encoder.CAPTURE = None
        fin = [bool(torch.isfinite(o).all()) for o in out]
        gfin = {k: bool(torch.isfinite(v).all()) for k, v in g.items() if v is not None}
        print("poison", poison, "active", flag, "outputs finite", fin, "bad grads", [k for k, v in gfin.items() if not v])
        if poison and flag and not all(fin):
            for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
                x, r = cap[name]
                print(name, "input finite", bool(torch.isfinite(x).all()))
