"""Round 5, active-row projections: what do the library GEMM (TunableOp on) and this library's own fp32-MFMA NT kernel
(sim_gemm_kernel: C = A B^T, 128x128 tiles) take on the projection shapes of the headline step -- the full products and the
products restricted to the rows the real ja train graph needs (5 425 destinations, 4 401 sources of 11 805)?  HIP events,
back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from jmac_amd import scoring
bench.enable_gemm_tuning(0)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
N, d, nD, nS = 11805, 300, 5425, 4401

def t(fn, n=50):
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

X, wc, wcT = r(N, d), r(d, 3 * d), r(3 * d, d)
out = torch.empty(N, 3 * d, device=dev)
for rows, cols, what in ((N, 900, "full [N,300]x[300,900]"), (N, 300, "Z  [N,300]x[300,300]"), (nD, 300, "P  [5425,300]x[300,300]"),
                         (nS, 300, "Q  [4401,300]x[300,300]"), (N, 600, "[N,300]x[300,600]")):
    A, B, BT = X[:rows], wc[:, :cols].contiguous(), wcT[:cols].contiguous()
    o = torch.empty(rows, cols, device=dev)
    fl = 2.0 * rows * cols * d
    tl = t(lambda: torch.mm(A, B, out=o))
    ts = t(lambda: scoring.sim_matrix(A, BT, out=o))
    print("%-28s library %6.1f us (%5.1f TF/s)   sim_gemm NT %6.1f us (%5.1f TF/s)" % (what, tl, fl / tl / 1e6, ts, fl / ts / 1e6))
# dgrad: dX = dPQZ wc^T  (NT with B = wc [300, 900]);  wgrad: dwc = X^T dPQZ
dPQZ, dX, dwc = r(N, 3 * d), torch.empty(N, d, device=dev), torch.empty(d, 3 * d, device=dev)
fl = 2.0 * N * 900 * d
tl = t(lambda: torch.mm(dPQZ, wc.t(), out=dX)); ts = t(lambda: scoring.sim_matrix(dPQZ, wc, out=dX))
print("%-28s library %6.1f us (%5.1f TF/s)   sim_gemm NT %6.1f us (%5.1f TF/s)" % ("dgrad [N,900]x[900,300]", tl, fl / tl / 1e6, ts, fl / ts / 1e6))
tl = t(lambda: torch.mm(X.t(), dPQZ, out=dwc))
print("%-28s library %6.1f us (%5.1f TF/s)" % ("wgrad [300,N]x[N,900]", tl, fl / tl / 1e6))
for rows in (nD, nS):
    tl = t(lambda: torch.mm(X[:rows].t(), dPQZ[:rows, :d], out=dwc[:, :d]))
    print("wgrad segment rows=%d        library %6.1f us (%5.1f TF/s)" % (rows, tl, 2.0 * rows * d * d / tl / 1e6))
