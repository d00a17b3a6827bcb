"""Round 5: tools/closed/r5_bmm_pair_probe.py product by product, with the outputs where the product needs them (column blocks of the stacked
[2, N, 3d] tables: strided out=), each as a hipGraph replay; a product torch.bmm refuses is reported as such."""
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
Xb = r(2, N, 3 * d)                                   # inputs at pitch 3d (align0 lives inside catA)
X = Xb[:, :, :d]
wc = r(2, d, 3 * d)
PQZ = torch.empty(2, N, 3 * d, device=dev)
dPQZ = r(2, N, 3 * d)
dXb = torch.zeros(2, N, 3 * d, device=dev)
dX = dXb[:, :, :d]
dwc = torch.empty(2, d, 3 * d, device=dev)


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


cases = {
    "Z   fwd": (lambda: [torch.mm(X[l], wc[l][:, d:2 * d], out=PQZ[l][:, d:2 * d]) for l in range(2)],
                lambda: torch.bmm(X, wc[:, :, d:2 * d], out=PQZ[:, :, d:2 * d])),
    "Q   fwd": (lambda: [torch.mm(X[l][s0:s1], wc[l][:, :d], out=PQZ[l][s0:s1, :d]) for l in range(2)],
                lambda: torch.bmm(X[:, s0:s1], wc[:, :, :d], out=PQZ[:, s0:s1, :d])),
    "P   fwd": (lambda: [torch.mm(X[l][:nD], wc[l][:, 2 * d:], out=PQZ[l][:nD, 2 * d:]) for l in range(2)],
                lambda: torch.bmm(X[:, :nD], wc[:, :, 2 * d:], out=PQZ[:, :nD, 2 * d:])),
    "dZ  dgrad (beta 1)": (lambda: [dX[l].addmm_(dPQZ[l][:, d:2 * d], wc[l][:, d:2 * d].t()) for l in range(2)],
                           lambda: dX.baddbmm_(dPQZ[:, :, d:2 * d], wc[:, :, d:2 * d].transpose(1, 2))),
    "dQ  dgrad (beta 1)": (lambda: [dX[l][s0:s1].addmm_(dPQZ[l][s0:s1, :d], wc[l][:, :d].t()) for l in range(2)],
                           lambda: dX[:, s0:s1].baddbmm_(dPQZ[:, s0:s1, :d], wc[:, :, :d].transpose(1, 2))),
    "dP  dgrad (beta 1)": (lambda: [dX[l][:nD].addmm_(dPQZ[l][:nD, 2 * d:], wc[l][:, 2 * d:].t()) for l in range(2)],
                           lambda: dX[:, :nD].baddbmm_(dPQZ[:, :nD, 2 * d:], wc[:, :, 2 * d:].transpose(1, 2))),
    "wgrad": (lambda: [torch.mm(X[l].t(), dPQZ[l], out=dwc[l]) for l in range(2)],
              lambda: torch.bmm(X.transpose(1, 2), dPQZ, out=dwc)),
}
tot = [0.0, 0.0]
for name, (two, one) in cases.items():
    a = graph_time(two)
    try:
        b = graph_time(one)
        torch.cuda.synchronize()
    except Exception as ex:
        b = float("nan")
        print(name, "bmm FAILED:", str(ex).splitlines()[0])
        break
    tot[0] += a
    tot[1] += b
    print("%-20s 2 x mm %7.1f us    bmm %7.1f us" % (name, a, b), flush=True)
print("total 2 x mm %.1f  bmm %.1f" % tuple(tot))
