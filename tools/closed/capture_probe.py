"""Which part of the step breaks hipGraph capture under TunableOp?  Each variant runs in its own process (a failed capture can
take the process down): python tools/closed/capture_probe.py [variant]   (no argument: run them all as children)."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = ["fwd", "fwdbwd", "step", "step_tuned", "step_unfused"]
if len(sys.argv) < 2:
    for v in VARIANTS:
        env = dict(os.environ, JMAC_ENC_DEBUG=v.split(":")[1] if ":" in v else "0")
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v.split(":")[0]], capture_output=True, text=True, env=env)
        tail = (r.stderr.strip().split("\n") or [""])[-1][:200]
        print("%-14s rc=%d %s | %s" % (v, r.returncode, r.stdout.strip().split("\n")[-1][:120], tail), flush=True)
        if r.returncode:
            print("\n".join(l[:220] for l in r.stderr.split("\n")[-60:]))
    sys.exit(0)
v = sys.argv[1]
import torch
import bench
a = argparse.Namespace(dim=300, batch=1000, negatives=25, bwd_mode=1)
dev = torch.device("cuda")
if v == "step_tuned":
    bench.enable_gemm_tuning(0)
    import torch.cuda.tunable as tun
    tun.set_filename(os.path.join(ROOT, "gpurun_out", "tunable_%s.csv" % v))
    import torch.cuda.tunable as tun
    tun.set_filename(os.path.join(ROOT, "gpurun_out", "tunable_%s.csv" % v))
w = bench.JaWorkload(a, dev, data="real")
if v == "step_unfused":
    w.model.fused_encoder = False
for _ in range(2):
    w.step()
torch.cuda.synchronize()
if v == "step_tuned":
    bench.freeze_gemm_tuning()
fn = {"fwd": lambda: w.forward_loss()[0], "fwdbwd": lambda: (w.opt.zero_grad(set_to_none=True), w.forward_loss()[0].backward())}.get(v, w.step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("captured + replayed", v)
