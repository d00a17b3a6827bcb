import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd import scoring
def T(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/it*1e3
g=torch.Generator(device="cuda").manual_seed(0)
tab=torch.nn.functional.normalize(torch.randn(30000,300,device="cuda",generator=g))
q=tab[torch.randperm(30000,device="cuda")[:3000]]
print("config5 get_neg L=3000 N=30000 k=25: %.3f ms" % T(lambda: scoring.sim_topk(q, tab, 25)))
S=scoring.sim_matrix(q, tab)
print("  sim_matrix only: %.3f ms ; row_topk only: %.3f ms" % (T(lambda: scoring.sim_matrix(q, tab)), T(lambda: scoring.row_topk(S, 25))))
e1=tab[:10500]; e2=tab[10500:21000]
print("config5 alignment_test 10500^2 csls10: %.3f ms" % T(lambda: scoring.alignment_test(e1, e2, (1,5,10), csls_k=10), 3))
S2=scoring.sim_matrix(e1,e2)
print("  sim %.3f ms; row_topk(k=10) %.3f ms; transpose %.3f ms; rank %.3f ms" % (T(lambda: scoring.sim_matrix(e1,e2)), T(lambda: scoring.row_topk(S2,10)), T(lambda: S2.t().contiguous()), T(lambda: scoring.filtered_rank(-S2, torch.arange(10500,device="cuda",dtype=torch.int32)))))
print("config5 quality [12000,12000] entropy: %.3f ms" % T(lambda: scoring.align_entropy(tab[:12000], tab[12000:24000]), 3))
ja=tab[:11805]; 
print("ja get_neg L=2264 N=11805 k=25: %.3f ms" % T(lambda: scoring.sim_topk(ja[:2264], ja, 25)))
er=torch.randn(1000,300,device="cuda"); 
print("l1 B=1000 N=11805: %.3f ms ; N=56589: %.3f ms" % (T(lambda: scoring.l1_scores(er, ja)), T(lambda: scoring.l1_scores(er, torch.cat([tab,tab])[:56589]))))
