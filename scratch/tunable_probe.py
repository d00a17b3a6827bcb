import torch, time, os, sys
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True)
tun.set_max_tuning_duration(50); tun.set_max_tuning_iterations(20)
tun.set_filename("/tmp/tunableop.csv")
shapes=[(11805,300,900),(11805,900,300),(300,11805,900),(962,300,600),(11805,600,300),(11805,300,300),(300,11805,300),(962,300,300)]
t0=time.time()
for (m,k,n) in shapes:
    a=torch.randn(m,k,device="cuda"); b=torch.randn(k,n,device="cuda")
    for _ in range(3): a@b
torch.cuda.synchronize(); print("tuning took %.1f s" % (time.time()-t0))
for (m,k,n) in shapes:
    a=torch.randn(m,k,device="cuda"); b=torch.randn(k,n,device="cuda")
    for _ in range(5): a@b
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): a@b
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
    print("tuned %6dx%5dx%5d  %7.1f us  %6.1f TF" % (m,k,n,dt*1e6,2*m*k*n/dt/1e12), flush=True)
X=torch.randn(11805,300,device="cuda"); G=torch.randn(11805,900,device="cuda"); W=torch.randn(300,900,device="cuda")
for name,fn in (("dW=X^T G", lambda: X.t()@G), ("dX=G W^T", lambda: G@W.t())):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
    print("tuned %-10s %7.1f us %6.1f TF" % (name,dt*1e6,2*11805*300*900/dt/1e12), flush=True)
