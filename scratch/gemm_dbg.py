import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd import scoring
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
A=torch.randn(n,300,device="cuda"); B=torch.randn(n,300,device="cuda")
for _ in range(3): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("dbg=%s n=%d %7.1f us %6.1f TF" % (os.environ.get("JMAC_GEMM_DBG"), n, dt*1e6, 2*n*300*n/dt/1e12))
