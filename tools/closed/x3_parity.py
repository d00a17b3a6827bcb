"""Full-size ja step gradients vs the float64 oracle (GPU kink masks) with and without the split-bf16 GEMM."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_ja_oracle as T
from jmac_amd import ops
for thr in (2048, 10**9):
    ops.X3_MIN_ROWS = thr
    w = T._workload(300, False)
    captured, hooks = {}, []
    for name in ("conv1_alignment", "conv1_completion", "conv2_alignment"):
        def pre(mod, args, name=name):
            captured[name] = (args[0].detach(), args[1].detach())
        hooks.append(getattr(w.model, name).register_forward_pre_hook(pre))
    w.opt.zero_grad(set_to_none=True)
    loss, align_out, comp, _ = w.forward_loss()
    loss.backward(); torch.cuda.synchronize()
    for h in hooks: h.remove()
    masks = T._gpu_kink_masks(w, captured)
    o_loss, _, _, grads = w.oracle_pass(torch.float64, kink_masks=masks, backward=True)
    errs = []
    for name, prm in w.model.named_parameters():
        ref = grads.get(name)
        if ref is None or prm.grad is None: continue
        e = (prm.grad.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        errs.append((e, name))
    errs.sort(reverse=True)
    print("X3_MIN_ROWS=%d  loss rel err %.2e  worst grads: %s" % (thr, abs(float(loss) - float(o_loss)) / float(o_loss), ", ".join("%s %.2e" % (n, e) for e, n in errs[:6])))
