#!/usr/bin/env python3
"""Cross-timing of bench.py's cpu_baseline (the oracle, kind "port") against the REFERENCE itself, in the build
container ("the same restatement must be cross-timed against the imported reference once").

Same step as bench.py's JaWorkload: JMAC.forward_base (num_gcn_layer=2 -> 3 RelationAwareLayer calls) on the DBP-5L
``ja``-shaped synthetic graph + the completion-style loss on a 26 000-triple batch + the alignment-style loss +
backward, dropout off, CPU, N threads.  Prints seconds per step for the reference (src/jmac_model.py, torch_scatter
stand-in of oracle/_shim) and for oracle/jmac_oracle.py, and their ratio.  Not a test: tools/r6_crosstime.sh runs it five times
and commits the lines (profiles/r6_crosstime.txt) and the median ratio (profiles/r6_crosstime.json, read by bench.py as
cpu_baseline.port_vs_reference_cost_ratio; quoted in DESIGN.md section 2).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/crosstime_reference.py [dim] [threads]
"""
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("JMAC_REFERENCE", "/root/reference")
np.int, np.float = int, float        # noqa  (numpy >= 1.24)
sys.path[:0] = [REPO, os.path.join(REPO, "oracle", "_shim"), REF]

import torch  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 300
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(threads)

from src.jmac_model import JMAC  # noqa: E402  (the reference)
import oracle.jmac_oracle as orc  # noqa: E402
from jmac_amd import synth  # noqa: E402

ei, et, n, nr = synth.dbp5l_like("ja", 1234)
rng = np.random.default_rng(1235)
args = types.SimpleNamespace(dim=dim, dropout=0.0, leaky_relu_w=0.05, comp_op="sub", num_gcn_layer=2, num_negative=25,
                             margin_align=1.0, margin_completion=5.0, batch_size=1000, no_name_info=False,
                             device=torch.device("cpu"))
torch.manual_seed(1234)
name_emb = rng.standard_normal((n, 300)).astype(np.float32)
model = JMAC(args, name_emb, nr, n)
model.train()
B, K = 1000, 25
trip = rng.integers(0, ei.shape[1], B)
h = torch.from_numpy(np.tile(ei[0][trip], K + 1))
r = torch.from_numpy(np.tile(et[trip], K + 1))
t = torch.from_numpy(np.concatenate([ei[1][trip], rng.integers(0, n, B * K)]))
pairs = torch.from_numpy(rng.integers(0, n, (2264, 2)))
eit, ett = torch.from_numpy(ei), torch.from_numpy(et)


def loss_fn(align_out, comp, rel, margin):
    loss = 0
    for ent, rl in zip(comp, rel):
        score = torch.norm(ent[h] + rl[r] - ent[t], 1, -1)
        pos = score[:B].view(-1, B).permute(1, 0)
        neg = score[B:].view(-1, B).permute(1, 0)
        loss = loss + torch.max(pos - neg, -margin).mean() + margin
    a = torch.nn.functional.normalize(align_out[pairs[:, 0]], 2, -1)
    b = torch.nn.functional.normalize(align_out[pairs[:, 1]], 2, -1)
    return loss + (1 - (a * b).sum(1)).mean()


def ref_step():
    model.zero_grad()
    out = model.forward_base(eit, ett, [0, n], [0, nr])
    loss_fn(out[0], out[1], out[2], model.margin_completion).backward()


st = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k
                                           and k != "margin_completion") for k, v in model.state_dict().items()}
bn = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
leaves = [v for v in st.values() if v.requires_grad]
nm = torch.from_numpy(name_emb)


def orc_step():
    for v in leaves:
        v.grad = None
    out = orc.forward_name(st, nm, eit, ett, [0, n], [0, nr], 2, 0.05, "sub", True, bn)
    loss_fn(out[0], out[1], out[2], st["margin_completion"].detach()).backward()


def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


tr, to = timeit(ref_step), timeit(orc_step)
print("ja shape N=%d E=%d d=%d, %d threads: reference %.2f s/step, oracle %.2f s/step, oracle/reference = %.2f"
      % (n, ei.shape[1], dim, threads, tr, to, to / tr))
