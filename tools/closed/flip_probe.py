"""Which seeds of tests/test_gpu_layer.py's random-layer cases are free of LeakyReLU kink flips (fp32 GPU vs float64
oracle) at d=300 / d=256?  Prints the flip count per seed; flip-free seeds are pinned in the test."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_layer as T
from jmac_amd.layer import RelationAwareLayer
from util import make_args
for (n, nr, d, e, hub, chunk) in [(600, 25, 300, 5000, 700, 64), (500, 17, 256, 4000, 300, 128)]:
    out = []
    for seed in range(100, 140):
        ei, et, X, R, G = T._oracle_case(n, nr, d, e, seed=seed, hub=hub)
        torch.manual_seed(seed)
        lay = RelationAwareLayer(d, d, rel_dim=d, act=torch.tanh, args=make_args())
        p = {k: v.detach().clone().double() for k, v in lay.named_parameters()}
        lay = lay.cuda()
        from jmac_amd import encoder
        cap = {}
        encoder.CAPTURE = cap                 # the fused layer node reports the tables it gathered
        with torch.no_grad():
            lay.train()
            lay(X.cuda(), R.cuda(), ei.cuda(), et.cuda())
        encoder.CAPTURE = None
        fl = T._kink_flips(lay, p, X.double(), R.double(), ei, et, X.cuda(), R.cuda(), cap)
        out.append((seed, fl))
    print("d=%d:" % d, out)
