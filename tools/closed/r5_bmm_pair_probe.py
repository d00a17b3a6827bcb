"""Round 5: would the two independent first layers of forward_name (conv1_alignment || conv1_completion, src/jmac_model.py:183,190:
same graph, same shapes, different inputs and weights) gain from running their projection products as ONE strided-batched product
(torch.bmm, batch 2) instead of two library GEMMs?  Row ranges of the class order as in the product (real ja train graph), hipGraph
replays, TunableOp on, warm operands.  Upper bound: the stacked [2, N, d] input / [2, N, 3d] table buffers it needs are assumed free."""
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
nD, s0, s1 = nDo + nDS, nDo, nDo + nDS + nSo
X = r(2, N, d)
wc = r(2, d, 3 * d)                                   # columns [Q|Z|P]
PQZ = torch.empty(2, N, 3 * d, device=dev)
dPQZ = r(2, N, 3 * d)
dX = torch.empty(2, N, d, device=dev)
dwc = torch.empty(2, d, 3 * d, device=dev)


def two_fwd():
    for l in range(2):
        torch.mm(X[l], wc[l][:, d:2 * d], out=PQZ[l][:, d:2 * d])
        torch.mm(X[l][s0:s1], wc[l][:, :d], out=PQZ[l][s0:s1, :d])
        torch.mm(X[l][:nD], wc[l][:, 2 * d:], out=PQZ[l][:nD, 2 * d:])


Zo, Qo, Po = torch.empty(2, N, d, device=dev), torch.empty(2, s1 - s0, d, device=dev), torch.empty(2, nD, d, device=dev)
wz, wq, wp = wc[:, :, d:2 * d].contiguous(), wc[:, :, :d].contiguous(), wc[:, :, 2 * d:].contiguous()
dZ, dQ, dP = dPQZ[:, :, d:2 * d].contiguous(), dPQZ[:, s0:s1, :d].contiguous(), dPQZ[:, :nD, 2 * d:].contiguous()
dXq, dXp = torch.empty(2, s1 - s0, d, device=dev), torch.empty(2, nD, d, device=dev)


def bmm_fwd():                  # contiguous outputs (torch.bmm refuses strided out= views here): an upper bound for the batched form
    torch.bmm(X, wz, out=Zo)
    torch.bmm(X[:, s0:s1], wq, out=Qo)
    torch.bmm(X[:, :nD], wp, out=Po)


def two_dgrad():
    for l in range(2):
        torch.mm(dPQZ[l][:, d:2 * d], wc[l][:, d:2 * d].t(), out=dX[l])
        dX[l][s0:s1].addmm_(dPQZ[l][s0:s1, :d], wc[l][:, :d].t())
        dX[l][:nD].addmm_(dPQZ[l][:nD, 2 * d:], wc[l][:, 2 * d:].t())


def bmm_dgrad():                # the two accumulating range products as plain batched products into their own buffers (upper bound)
    torch.bmm(dZ, wz.transpose(1, 2), out=dX)
    torch.bmm(dQ, wq.transpose(1, 2), out=dXq)
    torch.bmm(dP, wp.transpose(1, 2), out=dXp)


def two_wgrad():
    for l in range(2):
        torch.mm(X[l].t(), dPQZ[l], out=dwc[l])


def bmm_wgrad():
    torch.bmm(X.transpose(1, 2), dPQZ, out=dwc)


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


for name, a, b in (("projection (3 row ranges)", two_fwd, bmm_fwd), ("input gradient (3 row ranges)", two_dgrad, bmm_dgrad),
                   ("weight gradient", two_wgrad, bmm_wgrad)):
    print("%-32s two layers as 2 x mm %7.1f us    as bmm (batch 2) %7.1f us" % (name, graph_time(a), graph_time(b)))
