import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd._lib import lib, ptr, stream
L=lib(); st=stream()
def T(fn, it=300):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
for (M,N,K) in [(962,300,8),(962,300,64),(962,300,300),(962,300,1200),(32,32,300),(256,256,300),(2048,300,300),(4096,600,300)]:
    A=torch.randn(M,K,device="cuda"); B=torch.randn(K,N,device="cuda"); C=torch.empty(M,N,device="cuda")
    t=T(lambda: L.jmac_gemm_f32(ptr(A),K,0,ptr(B),N,0,M,N,K,ptr(C),N,st))
    print("NN M=%d N=%d K=%d: %.1f us (%.1f TF) tiles=%d" % (M,N,K,t,2*M*N*K/t/1e6, ((M+31)//32)*((N+31)//32)))
