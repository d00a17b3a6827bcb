"""What do the node-side projection GEMMs cost when P / Q / Z are computed only for the rows that need them (real ja train
graph: 4 401 destinations, 5 425 sources, 4 332 isolated of 11 805)?  Full products vs three row ranges with column
slices, in the three forms of a step (TunableOp on, HIP events, back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.enable_gemm_tuning(0)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
N, d = 11805, 300
nP, q1 = int(sys.argv[1]) if len(sys.argv) > 1 else 4401, int(sys.argv[2]) if len(sys.argv) > 2 else 7473
X, wc, PQZ, dPQZ, dX, dwc = r(N, d), r(d, 3 * d), r(N, 3 * d), r(N, 3 * d), r(N, d), r(d, 3 * d)

def t(fn, n=40):
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

def fwd_full():
    torch.mm(X, wc, out=PQZ)
def fwd_split():
    torch.mm(X[:nP], wc, out=PQZ[:nP])
    torch.mm(X[nP:q1], wc[:, d:], out=PQZ[nP:q1, d:])
    torch.mm(X[q1:], wc[:, 2 * d:], out=PQZ[q1:, 2 * d:])
def dx_full():
    torch.mm(dPQZ, wc.t(), out=dX)
def dx_split():
    torch.mm(dPQZ[:nP], wc.t(), out=dX[:nP])
    torch.mm(dPQZ[nP:q1, d:], wc[:, d:].t(), out=dX[nP:q1])
    torch.mm(dPQZ[q1:, 2 * d:], wc[:, 2 * d:].t(), out=dX[q1:])
def dw_full():
    torch.mm(X.t(), dPQZ, out=dwc)
def dw_split():
    torch.mm(X[:nP].t(), dPQZ[:nP], out=dwc)
    dwc[:, d:].addmm_(X[nP:q1].t(), dPQZ[nP:q1, d:])
    dwc[:, 2 * d:].addmm_(X[q1:].t(), dPQZ[q1:, 2 * d:])
def dw_split2():
    torch.mm(X[:nP].t(), dPQZ[:nP, :d], out=dwc[:, :d])
    torch.mm(X[:q1].t(), dPQZ[:q1, d:2 * d], out=dwc[:, d:2 * d])
    torch.mm(X.t(), dPQZ[:, 2 * d:], out=dwc[:, 2 * d:])
for name, a, b in (("X wc", fwd_full, fwd_split), ("dPQZ wc^T", dx_full, dx_split), ("X^T dPQZ", dw_full, dw_split), ("X^T dPQZ (by column block)", dw_full, dw_split2)):
    print("%-28s full %6.1f us   row ranges %6.1f us" % (name, t(a), t(b)))
