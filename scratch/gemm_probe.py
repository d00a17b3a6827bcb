import torch, time, sys
def bench(lib):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); return
    for (m,k,n) in [(11805,300,900),(11805,900,300),(300,11805,900),(962,300,300),(962,300,600),(11805,600,300),(11805,900,300)]:
        a=torch.randn(m,k,device="cuda"); b=torch.randn(k,n,device="cuda")
        for _ in range(5): a@b
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(50): a@b
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
        print("%-10s %6dx%5dx%5d  %7.1f us  %6.1f TF" % (lib,m,k,n,dt*1e6,2*m*k*n/dt/1e12))
for lib in ("default","hipblaslt"):
    bench(lib)
