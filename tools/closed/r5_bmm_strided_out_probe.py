"""Round 5: does torch.bmm write column blocks of a row-major table (out= a strided view), with and without TunableOp?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = "cuda"
N, d = 11805, 300
X = torch.randn(2, N, d, device=dev)
W = torch.randn(2, d, 3 * d, device=dev)
ref = torch.bmm(X, W[:, :, d:2 * d].contiguous())
for mode in sys.argv[1:] or ["off"]:
    if mode == "tune":
        bench.enable_gemm_tuning(0)
    if mode == "tune-rocblas":
        os.environ["PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED"] = "0"
        bench.enable_gemm_tuning(0)
    out = torch.zeros(2, N, 3 * d, device=dev)
    try:
        for _ in range(12):
            torch.bmm(X, W[:, :, d:2 * d], out=out[:, :, d:2 * d])
        torch.cuda.synchronize()
        ok = torch.allclose(out[:, :, d:2 * d], ref, rtol=1e-4, atol=1e-3) and float(out[:, :, :d].abs().max()) == 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            torch.bmm(X, W[:, :, d:2 * d], out=out[:, :, d:2 * d])
        e1.record()
        torch.cuda.synchronize()
        print(mode, "strided out ok:", ok, "%.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
    except Exception as ex:
        print(mode, "FAILED:", str(ex).splitlines()[0])
