#!/usr/bin/env python3
"""half-wave forward kernel vs the 64-lane kernel (JMAC_FWD_HW=0/1 in child processes) on small graphs: max error by row class."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = [(200, 11, 1500, 300, None, None, 0), (200, 11, 1500, 300, None, None, 1), (500, 30, 6000, 256, 900, 32, 0),
         (500, 30, 6000, 256, 900, 32, 1), (3000, 40, 9000, 300, 700, None, 0), (3000, 40, 9000, 300, 700, None, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from jmac_amd import ops
    from jmac_amd.graph import RelGraph
    from util import random_graph
    for (n, nr, e, d, hub, chunk, bf) in CASES:
        rng = np.random.default_rng(n + e)
        ei, et = random_graph(rng, n, nr - 1, e, hub)
        gen = torch.Generator().manual_seed(n)
        PQZ = (torch.randn(n, 3 * d, generator=gen) * 0.3); RR = (torch.randn(nr, 2 * d, generator=gen) * 0.3); a = torch.randn(d, generator=gen) * 0.1
        g = RelGraph(torch.from_numpy(ei).cuda(), torch.from_numpy(et).cuda(), n, nr, chunk)
        if bf:
            P_, R_ = ops.pad_table(PQZ.to(torch.bfloat16), d, 3).cuda(), ops.pad_table(RR.to(torch.bfloat16), d, 2).cuda()
        else:
            P_, R_ = PQZ.cuda(), RR.cuda()
        with torch.no_grad():
            out = ops.rel_attn_aggregate(P_, R_, a.cuda(), g, 0.05, nr - 1, 0.5)
        torch.cuda.synchronize()
        np.save("/tmp/hwdbg_%s_%d_%d_%d_%d.npy" % (os.environ.get("JMAC_FWD_HW", "1"), n, e, d, bf), out.cpu().numpy())
        np.save("/tmp/hwdbg_deg_%d_%d.npy" % (n, e), np.bincount(ei[0], minlength=n))
        print("case", n, e, d, bf, "coop", g.by_dst.n_coop, "splits", g.by_dst.n_splits_max, flush=True)
    print("child done", os.environ.get("JMAC_FWD_HW"))
    sys.exit(0)
import numpy as np
for hw in ("0", "1"):
    env = dict(os.environ, JMAC_FWD_HW=hw)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    print(r.stdout[-600:], r.stderr[-1500:])
for (n, nr, e, d, hub, chunk, bf) in CASES:
    try:
        a0 = np.load("/tmp/hwdbg_0_%d_%d_%d_%d.npy" % (n, e, d, bf)); a1 = np.load("/tmp/hwdbg_1_%d_%d_%d_%d.npy" % (n, e, d, bf))
    except Exception as ex:
        print("missing", n, e, d, bf, ex); continue
    deg = np.load("/tmp/hwdbg_deg_%d_%d.npy" % (n, e))
    err = np.nan_to_num(np.abs(a0 - a1), nan=1e9).max(1)
    bad = np.flatnonzero(err > 1e-4 * np.abs(a0).max())
    print("case n=%d e=%d d=%d bf16=%d: max err %.3e scale %.3e, bad rows %d of %d" % (n, e, d, bf, err.max(), np.abs(a0).max(), len(bad), n))
    if len(bad):
        print("   degrees of bad rows (first 20):", deg[bad][:20].tolist(), "rows", bad[:20].tolist())
        r = bad[0]
        dcol = np.nan_to_num(np.abs(a0[r] - a1[r]), nan=1e9)
        print("   row %d: bad cols %s" % (r, np.flatnonzero(dcol > 1e-4 * np.abs(a0).max())[:60].tolist()))
        print("   nan count", int(np.isnan(a1).sum()), " ok-degree histogram of good rows:", np.bincount(np.minimum(deg[err <= 1e-4 * np.abs(a0).max()], 12)).tolist())
