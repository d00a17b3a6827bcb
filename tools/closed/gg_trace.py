"""Phase timeline of the grouped GEMM's blocks (debug build with -DJMAC_GG_TRACE; see tools/closed/gg_trace.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jmac_amd import _lib
from jmac_amd.encoder import gemm_task, grouped_gemm
L = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
nr, d = 962, 300
A, W = r(nr, d), r(d, d)
ntask = int(sys.argv[1]) if len(sys.argv) > 1 else 1
form = sys.argv[2] if len(sys.argv) > 2 else "nn"
if form == "tn":                                  # weight-gradient form: [300, 600] = T^T dRR, K = 962
    G = r(nr, 2 * d)
    outs = [r(d, 2 * d) for _ in range(ntask)]
    tasks = [gemm_task(A, G, o, ta=True) for o in outs]
    per = 190
else:
    outs = [r(nr, d) for _ in range(ntask)]
    tasks = [gemm_task(A, W, o) for o in outs]
    per = 310
nblk = per * ntask
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
L.jmac_gemm_trace_buffer.argtypes = [ctypes.c_void_p]
assert L.jmac_gemm_trace_buffer(buf.data_ptr()) == 0
for _ in range(5):
    grouped_gemm(tasks)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
grouped_gemm(tasks)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
raw = raw[raw[:, 4] > 0]
rt = (raw[:, 5:7] - raw[:, 5].min()) / 100.0          # s_memrealtime: 100 MHz, one clock for the whole chip -> us
print("wall clock: block starts  p10 %.2f  median %.2f  p90 %.2f  max %.2f us;  block ends  median %.2f  p90 %.2f  max %.2f us;  block life median %.2f us"
      % (np.percentile(rt[:, 0], 10), np.median(rt[:, 0]), np.percentile(rt[:, 0], 90), rt[:, 0].max(), np.median(rt[:, 1]),
         np.percentile(rt[:, 1], 90), rt[:, 1].max(), np.median(rt[:, 1] - rt[:, 0])))
t = raw[:, :5]
t0 = t[:, 0].min()
t = (t - t0) / 100.0      # units of 100 shader-clock cycles (s_memtime counts the shader clock here)
names = ["start", "task loaded", "panels in LDS", "mfma done", "end"]
for i, n in enumerate(names):
    print("%-14s min %6.2f  median %6.2f  max %6.2f us" % (n, t[:, i].min(), np.median(t[:, i]), t[:, i].max()))
d_ = np.diff(t, axis=1)
for i, n in enumerate(["task load", "global loads + LDS store", "frag + mfma", "epilogue"]):
    print("phase %-26s median %6.2f  p90 %6.2f us" % (n, np.median(d_[:, i]), np.percentile(d_[:, i], 90)))
