"""Round 5: the projection products of one training step (3 layers x {forward, input gradient, weight gradient}) as hipGraph
replays -- full [N,300]x[300,900] products against products restricted to row RANGES, as they would be if the entities were
ordered by class [destination only | destination and source | source only | neither] inside the encoder node (real ja train graph:
3 072 / 2 353 / 2 048 / 4 332 rows) with the table columns ordered [Q|Z|P].  TunableOp on; warm operands; what an internal
permutation could buy at most (its gather / scatter copies not included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.enable_gemm_tuning(0)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
N, d = 11805, 300
nDo, nDS, nSo = 3072, 2353, 2048
nD, s0, s1 = nDo + nDS, nDo, nDo + nDS + nSo          # destinations [0, nD), sources [s0, s1)
L = 3
X = [r(N, d) for _ in range(L)]
wc = [r(d, 3 * d) for _ in range(L)]                  # columns [Q|Z|P]
PQZ = [torch.empty(N, 3 * d, device=dev) for _ in range(L)]
dPQZ = [r(N, 3 * d) for _ in range(L)]
dX = [torch.empty(N, d, device=dev) for _ in range(L)]
dwc = [torch.empty(d, 3 * d, device=dev) for _ in range(L)]

def full():
    for l in range(L):
        torch.mm(X[l], wc[l], out=PQZ[l])
    for l in range(L):
        torch.mm(dPQZ[l], wc[l].t(), out=dX[l])
        torch.mm(X[l].t(), dPQZ[l], out=dwc[l])

def ranged():
    for l in range(L):
        torch.mm(X[l], wc[l][:, d:2 * d], out=PQZ[l][:, d:2 * d])                       # Z: all rows
        torch.mm(X[l][s0:s1], wc[l][:, :d], out=PQZ[l][s0:s1, :d])                      # Q: source rows
        torch.mm(X[l][:nD], wc[l][:, 2 * d:], out=PQZ[l][:nD, 2 * d:])                  # P: destination rows
    for l in range(L):
        torch.mm(dPQZ[l][:, d:2 * d], wc[l][:, d:2 * d].t(), out=dX[l])
        dX[l][s0:s1].addmm_(dPQZ[l][s0:s1, :d], wc[l][:, :d].t())
        dX[l][:nD].addmm_(dPQZ[l][:nD, 2 * d:], wc[l][:, 2 * d:].t())
        torch.mm(X[l].t(), dPQZ[l], out=dwc[l])

def ranged_classes():                                  # input gradient by row class (non-overlapping, K = the class's columns)
    for l in range(L):
        torch.mm(X[l], wc[l][:, d:2 * d], out=PQZ[l][:, d:2 * d])
        torch.mm(X[l][s0:s1], wc[l][:, :d], out=PQZ[l][s0:s1, :d])
        torch.mm(X[l][:nD], wc[l][:, 2 * d:], out=PQZ[l][:nD, 2 * d:])
    for l in range(L):
        torch.mm(dPQZ[l][:s0, d:], wc[l][:, d:].t(), out=dX[l][:s0])                    # destination only: Z, P
        torch.mm(dPQZ[l][s0:nD], wc[l].t(), out=dX[l][s0:nD])                           # both: Q, Z, P
        torch.mm(dPQZ[l][nD:s1, :2 * d], wc[l][:, :2 * d].t(), out=dX[l][nD:s1])        # source only: Q, Z
        torch.mm(dPQZ[l][s1:, d:2 * d], wc[l][:, d:2 * d].t(), out=dX[l][s1:])          # neither: Z
        torch.mm(X[l].t(), dPQZ[l], out=dwc[l])

def ranged_w():                                        # ... and the weight gradient's K range per column block
    for l in range(L):
        torch.mm(X[l], wc[l][:, d:2 * d], out=PQZ[l][:, d:2 * d])
        torch.mm(X[l][s0:s1], wc[l][:, :d], out=PQZ[l][s0:s1, :d])
        torch.mm(X[l][:nD], wc[l][:, 2 * d:], out=PQZ[l][:nD, 2 * d:])
    for l in range(L):
        torch.mm(dPQZ[l][:, d:2 * d], wc[l][:, d:2 * d].t(), out=dX[l])
        dX[l][s0:s1].addmm_(dPQZ[l][s0:s1, :d], wc[l][:, :d].t())
        dX[l][:nD].addmm_(dPQZ[l][:nD, 2 * d:], wc[l][:, 2 * d:].t())
        torch.mm(X[l].t(), dPQZ[l][:, d:2 * d], out=dwc[l][:, d:2 * d])
        torch.mm(X[l][s0:s1].t(), dPQZ[l][s0:s1, :d], out=dwc[l][:, :d])
        torch.mm(X[l][:nD].t(), dPQZ[l][:nD, 2 * d:], out=dwc[l][:, 2 * d:])


def graph_time(fn, reps=30):
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    bench.freeze_gemm_tuning()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    bench.enable_gemm_tuning(0)
    return e0.elapsed_time(e1) / reps * 1e3

for name, fn in (("full products", full), ("row ranges (accumulating input gradient)", ranged), ("row ranges (input gradient by class)", ranged_classes),
                 ("row ranges, weight gradient by K range too", ranged_w)):
    print("%-44s %7.1f us per step's 9 projection products" % (name, graph_time(fn)))
