import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = [(11805,300,900),(11805,900,300),(300,11805,900),(962,300,600),(11805,600,300),(11805,300,300),(300,11805,300)]
def bench(lib):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); return
    for (m,k,n) in shapes:
        a=torch.randn(m,k,device="cuda"); b=torch.randn(k,n,device="cuda")
        for _ in range(5): a@b
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(50): a@b
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
        print("%-10s %6dx%5dx%5d  %7.1f us  %6.1f TF" % (lib,m,k,n,dt*1e6,2*m*k*n/dt/1e12), flush=True)
for lib in ("cublaslt",):
    bench(lib)
# transposed-operand forms as autograd produces them (dW = X^T G, dX = G W^T)
torch.backends.cuda.preferred_blas_library("cublaslt")
for lib in ("cublaslt","cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    X=torch.randn(11805,300,device="cuda"); G=torch.randn(11805,900,device="cuda"); W=torch.randn(300,900,device="cuda")
    for name,fn in (("dW=X^T G", lambda: X.t()@G), ("dX=G W^T", lambda: G@W.t())):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
        print("%-10s %-10s %7.1f us %6.1f TF" % (lib,name,dt*1e6,2*11805*300*900/dt/1e12), flush=True)
# this library's fp32 MFMA sim GEMM (A B^T)
from jmac_amd import scoring
A=torch.randn(11805,300,device="cuda"); B=torch.randn(900,300,device="cuda")
for _ in range(3): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print("jmac sim_gemm 11805x300x900 %7.1f us %6.1f TF" % (dt*1e6, 2*11805*300*900/dt/1e12))
A=torch.randn(12000,300,device="cuda"); B=torch.randn(12000,300,device="cuda")
for _ in range(3): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): scoring.sim_matrix(A,B)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("jmac sim_gemm 12000x300x12000 %7.1f us %6.1f TF" % (dt*1e6, 2*12000*300*12000/dt/1e12))
for _ in range(3): A@B.t()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): A@B.t()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("torch A@B.T 12000x300x12000 %7.1f us %6.1f TF" % (dt*1e6, 2*12000*300*12000/dt/1e12))
