"""What the library offers below strict fp32 for the node-side GEMMs on gfx950 (not used by the product: the headline keeps
exact fp32 MFMA products): time and error of [11805,300] x [300,900] under torch's float32 matmul precision switches."""
import os, sys, torch
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
X, W = torch.randn(11805, 300, device=dev, generator=g), torch.randn(300, 900, device=dev, generator=g) * 0.05
ref = (X.double() @ W.double())
def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def report(tag):
    out = torch.mm(X, W)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print("%-34s %7.1f us   max|err|/max|ref| %.2e" % (tag, t(lambda: torch.mm(X, W)), err))
report("default (fp32)")
for prec in ("high", "medium"):
    torch.set_float32_matmul_precision(prec)
    report("float32_matmul_precision=" + prec)
torch.set_float32_matmul_precision("highest")
torch.backends.cuda.matmul.allow_tf32 = True
report("allow_tf32")
torch.backends.cuda.matmul.allow_tf32 = False
Xb, Wb = X.bfloat16(), W.bfloat16()
print("%-34s %7.1f us" % ("bf16 x bf16 (for scale)", t(lambda: torch.mm(Xb, Wb))))
