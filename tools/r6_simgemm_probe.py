"""round 6: sim_gemm_kernel (JMAC_LIB_PATH selects the build) or the library's fp32 NT GEMM on the config-5 shapes: HIP-event ms
per launch and TFLOP/s.   usage: r6_simgemm_probe.py --name <tag> | --lib-bar"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--name", default="base")
ap.add_argument("--lib-bar", action="store_true")
x = ap.parse_args()
SHAPES = [("quality", 12000, 12000), ("get_neg", 3000, 30000), ("csls", 10500, 10500)]
d = 300
gen = torch.Generator(device="cuda").manual_seed(0)
tab = torch.nn.functional.normalize(torch.randn(42000, d, device="cuda", generator=gen))


def ms(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if x.lib_bar:
    for tuned in (False, True):
        if tuned:
            import torch.cuda.tunable as tun
            tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(50); tun.set_max_tuning_iterations(20)
            tun.set_filename("/tmp/r6_sg_tunable.csv")
        out = {"build": "library fp32 NT (torch.mm(a, b.T, out=)), TunableOp %s" % ("on" if tuned else "off")}
        for nm, M, N in SHAPES:
            a, b = tab[:M], tab[M:M + N]
            c = torch.empty(M, N, device="cuda")
            t = ms(lambda: torch.mm(a, b.t(), out=c))
            out[nm] = {"ms": round(t, 4), "tflops": round(2.0 * M * N * d / (t * 1e-3) / 1e12, 2)}
        print(json.dumps(out), flush=True)
    sys.exit(0)
from jmac_amd import scoring
out = {"build": x.name}
for nm, M, N in SHAPES:
    a, b = tab[:M], tab[M:M + N]
    c = torch.empty(M, N, device="cuda")
    t = ms(lambda: scoring.sim_matrix(a, b, out=c))
    out[nm] = {"ms": round(t, 4), "tflops": round(2.0 * M * N * d / (t * 1e-3) / 1e12, 2)}
    if not any(t in x.name for t in ("nostore", "noload", "mfma_only")):     # result-preserving builds: check it
        ref = torch.mm(a[:256].double(), b[:512].double().t())
        out[nm]["max_abs_err_vs_f64"] = float((c[:256, :512].double() - ref).abs().max())
print(json.dumps(out), flush=True)
