#!/usr/bin/env python3
"""raw forward call: out / seg_max / seg_den of the half-wave kernel vs a torch evaluation, tiny regular graphs (fp32, d=300)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from jmac_amd._lib import lib, ptr, stream
from jmac_amd.graph import RelGraph
L = lib()
d = 300
for deg in (1, 2, 3, 5):
    n, nr = 64, 4
    dst = np.repeat(np.arange(n), deg); src = (dst * 7 + np.tile(np.arange(deg), n) * 3 + 1) % n; typ = (dst + np.tile(np.arange(deg), n)) % (nr - 1)
    ei = torch.from_numpy(np.stack([dst, src]).astype(np.int64)).cuda(); et = torch.from_numpy(typ.astype(np.int64)).cuda()
    g = RelGraph(ei, et, n, nr)
    gen = torch.Generator().manual_seed(1)
    PQZ = (torch.randn(n, 3 * d, generator=gen) * 0.3).cuda(); RR = (torch.randn(nr, 2 * d, generator=gen) * 0.3).cuda(); a = (torch.randn(d, generator=gen) * 0.1).cuda()
    out = torch.empty(n, d, device="cuda"); smax = torch.empty(n, device="cuda"); sden = torch.empty(n, device="cuda")
    s = g.by_dst
    wsb = int(L.jmac_rel_attn_fwd_workspace_bytes(s.n_parts_max, d)); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    rc = L.jmac_rel_attn_aggregate_fwd_f32(ptr(PQZ), 3 * d, PQZ.data_ptr() + d * 4, 3 * d, ptr(RR), 2 * d, ptr(a), ptr(g.col), ptr(g.etype),
                                           C.byref(s.view()), n, d, 0.05, nr - 1, 0, 0.5, ptr(out), d, ptr(smax), ptr(sden), ptr(ws), wsb, stream())
    torch.cuda.synchronize()
    P, Q, Z = PQZ[:, :d], PQZ[:, d:2 * d], PQZ[:, 2 * d:]
    Rq, Rz = RR[:, :d], RR[:, d:]
    dd, ss, tt = ei[0], ei[1], et
    h = P[dd] + Q[ss] - Rq[tt]
    sc = (torch.nn.functional.leaky_relu(h, 0.05) * a).sum(1)
    sc2 = sc.view(n, deg)
    m = sc2.max(1).values
    w = torch.exp(sc2 - m[:, None]); l = w.sum(1)
    alpha = w / l[:, None]
    v = (Z[ss] - Rz[tt]).view(n, deg, d)
    nb = (alpha[:, :, None] * v).sum(1) * np.sqrt(deg)
    ref = 0.5 * (nb + Z - Rz[nr - 1])
    print("deg", deg, "rc", rc, "max|out-ref| %.3e" % float((out - ref).abs().max()), "max|smax-m| %.3e" % float((smax - m).abs().max()),
          "max|sden-l| %.3e" % float((sden - l).abs().max()), "ref scale %.2f" % float(ref.abs().max()))
    if float((out - ref).abs().max()) > 1e-3:
        e = (out - ref).abs()
        r = int(e.max(1).values.argmax())
        bad = torch.nonzero(e[r] > 1e-3).flatten().cpu().numpy()
        print("   worst row", r, "bad cols n=", len(bad), bad[:50].tolist())
        # is out == something recognisable?
        selfonly = 0.5 * (Z - Rz[nr - 1])
        print("   |out - selfonly| %.3e" % float((out - selfonly).abs().max()))
