"""round 6: the L1 scoring kernels (JMAC_LIB_PATH selects the build): l1_score_kernel<float> at the bench shape and the fused
link-prediction rank (jmac_linkpred_rank_f32 / _bf16, 2 layers) on the ja-size and the union-size candidate table; HIP-event
ms per call, fraction of the 78.6 T lane-instr/s VALU issue peak (2 instructions per (b, n, k)), and a checksum of the ranks
(the arithmetic is the same two instructions per element in every build: the ranks must not move)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmac_amd import scoring
PEAK = 78.6432e12
name = sys.argv[1] if len(sys.argv) > 1 else "base"
g = torch.Generator(device="cuda").manual_seed(0)
out = {"build": name}


def ms(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, d = 1000, 300
for tag, N in (("ja", 11805), ("union", 56589)):
    tab = [torch.randn(N, d, device="cuda", generator=g) * 0.3 for _ in range(2)]
    rel = [torch.randn(961, d, device="cuda", generator=g) * 0.3 for _ in range(2)]
    h = torch.randint(0, N, (B,), device="cuda", generator=g)
    r = torch.randint(0, 961, (B,), device="cuda", generator=g)
    gold = torch.randint(0, N, (B,), device="cuda", generator=g)
    fptr = torch.arange(0, 3 * B + 1, 3, dtype=torch.int32, device="cuda")
    fidx = torch.randint(0, N, (3 * B,), device="cuda", generator=g).to(torch.int32)
    if tag == "ja":
        er = tab[0][h] + rel[0][r]
        o = torch.empty(B, N, device="cuda")
        t = ms(lambda: scoring.l1_scores(er, tab[0], out=o))
        out["l1_score_f32_ja"] = {"ms": round(t, 4), "valu_issue_frac": round(2 * B * N * d / (t * 1e-3) / PEAK, 4), "checksum": float(o.double().sum())}
    for dt, nm in ((None, "f32"), (torch.bfloat16, "bf16")):
        fn = lambda: scoring.linkpred_ranks(tab, rel, h, r, gold, fptr, fidx, **({"table_dtype": dt} if dt is not None else {}))
        t = ms(fn, n=15)
        rk = fn()
        out["fused_rank_%s_%s" % (nm, tag)] = {"ms": round(t, 4), "valu_issue_frac": round(2 * 2 * B * N * d / (t * 1e-3) / PEAK, 4),
                                              "scored_triples_per_s": round(B / (t * 1e-3)), "rank_sum": int(rk.sum())}
print(json.dumps(out), flush=True)
